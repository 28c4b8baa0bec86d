set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd $R
timeout -k 10 600 python -u -m pytest tests/test_gpu_k1_chain.py tests/test_gpu_flash_attention.py tests/test_gpu_paged_prefill.py tests/test_gpu_paged_fuzz.py tests/test_gpu_baseline_configs.py tests/test_zz_no_alloc.py -x -q --timeout 120 --timeout-method thread > $O/ring4_tests.txt 2>&1 || { tail -30 $O/ring4_tests.txt; exit 1; }
tail -3 $O/ring4_tests.txt
L=./scripts/k1_lab/k1_lab; : > $O/lab_ring4.txt
for cfg in "16:B=256 S=2048" "32:B=256 S=2048" "64:B=256 S=2048" "128:B=64 S=2048" "64:B=16 S=4096"; do
  T=${cfg%%:*}; args=${cfg#*:}
  LAB_PAGED=$T timeout -k 10 60 $L $args Hq=32 Hkv=8 rounds=3 >> $O/lab_ring4.txt 2>&1 || { echo FAILED; tail -5 $O/lab_ring4.txt; exit 1; }
done
grep -E "shape|median|cycles per wave" $O/lab_ring4.txt
SWEEP_T=12,13,16,24,32,48,64,96,128 timeout -k 10 200 python scripts/diag/verify_T_sweep.py 2>&1 | grep "^T "
