# end of round 3: full GPU suite, default bench line, refreshed sweeps (mid kernel + NT hint in)
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd $R
timeout -k 10 1000 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > $O/gpu_tests_r03.txt 2>&1 || { tail -30 $O/gpu_tests_r03.txt; exit 1; }
tail -2 $O/gpu_tests_r03.txt
timeout -k 10 400 python bench.py > $O/r03_bench.json 2> $O/r03_bench.err && echo bench-ok || { tail -5 $O/r03_bench.err; exit 1; }
: > $O/r03_paged_mid.txt
for m in 0 -1 1; do
  echo "== NVLLM_PAGED_MID=$m (0 = prefill kernel over the cache, -1 = dispatch rule, 1 = mid kernel wherever it applies, from 33 rows)" >> $O/r03_paged_mid.txt
  NVLLM_PAGED_MID=$m SWEEP_T=1,8,12,13,16,20,24,28,32 timeout -k 10 200 python scripts/diag/verify_T_sweep.py 2>&1 | grep "^T \|^N " >> $O/r03_paged_mid.txt || exit 1
  NVLLM_PAGED_MID=$m SWEEP_N=1024 SWEEP_CTX=512 SWEEP_T=13,16,24,32 timeout -k 10 200 python scripts/diag/verify_T_sweep.py 2>&1 | grep "^T \|^N " >> $O/r03_paged_mid.txt || exit 1
  NVLLM_PAGED_MID=$m SWEEP_N=512 SWEEP_CTX=1024 SWEEP_T=16,32 timeout -k 10 200 python scripts/diag/verify_T_sweep.py 2>&1 | grep "^T \|^N " >> $O/r03_paged_mid.txt || exit 1
  NVLLM_PAGED_MID=$m SWEEP_N=128 SWEEP_CTX=4096 SWEEP_T=16,24,32 timeout -k 10 200 python scripts/diag/verify_T_sweep.py 2>&1 | grep "^T \|^N " >> $O/r03_paged_mid.txt || exit 1
  NVLLM_PAGED_MID=$m SWEEP_N=64 SWEEP_CTX=2048 SWEEP_T=16 timeout -k 10 200 python scripts/diag/verify_T_sweep.py 2>&1 | grep "^T \|^N " >> $O/r03_paged_mid.txt || exit 1
done
V=$O/r03_verify_T_sweep.txt; : > $V
SWEEP_T=1,4,8,12,13,16,24,32,48,64,96,128,256 timeout -k 10 200 python scripts/diag/verify_T_sweep.py >> $V 2>&1 || { tail $V; exit 1; }
SWEEP_N=16 SWEEP_CTX=4096 SWEEP_T=1,4,8,12,13,16,32,64,128 timeout -k 10 200 python scripts/diag/verify_T_sweep.py >> $V 2>&1 || { tail $V; exit 1; }
SWEEP_N=32 SWEEP_CTX=4096 SWEEP_T=16,32,64,128 timeout -k 10 200 python scripts/diag/verify_T_sweep.py >> $V 2>&1 || { tail $V; exit 1; }
grep -v amdgpu $V | head -16
