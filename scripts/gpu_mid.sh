set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd $R
timeout -k 10 300 python -u -m pytest tests/test_gpu_paged_mid.py -x -q --timeout 120 --timeout-method thread > $O/mid_tests.txt 2>&1 || { tail -40 $O/mid_tests.txt; exit 1; }
tail -3 $O/mid_tests.txt
for m in 0 1; do
  echo "== NVLLM_PAGED_MID=$m"
  NVLLM_PAGED_MID=$m SWEEP_T=12,13,16,20,24,28,32 timeout -k 10 200 python scripts/diag/verify_T_sweep.py 2>&1 | grep "^T " || exit 1
  NVLLM_PAGED_MID=$m SWEEP_N=64 SWEEP_CTX=2048 SWEEP_T=16,32 timeout -k 10 200 python scripts/diag/verify_T_sweep.py 2>&1 | grep "^T \|^N " || exit 1
  NVLLM_PAGED_MID=$m SWEEP_N=32 SWEEP_CTX=4096 SWEEP_T=16,32 timeout -k 10 200 python scripts/diag/verify_T_sweep.py 2>&1 | grep "^T \|^N " || exit 1
  NVLLM_PAGED_MID=$m SWEEP_N=1024 SWEEP_CTX=512 SWEEP_T=16,32 timeout -k 10 200 python scripts/diag/verify_T_sweep.py 2>&1 | grep "^T \|^N " || exit 1
done
