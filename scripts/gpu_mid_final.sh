# mid kernel (lagged P.V, rows by SIMD): its tests, the paged tests around it, then the sweeps
set -o pipefail
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 600 python -u -m pytest tests/test_gpu_paged_mid.py tests/test_gpu_paged_prefill.py tests/test_gpu_paged_attention.py tests/test_gpu_baseline_configs.py -x -q -m gpu --timeout 180 --timeout-method thread 2>&1 | tail -5 || exit 1
for m in 0 ""; do
  echo "== NVLLM_PAGED_MID=$m"
  NVLLM_PAGED_MID=$m SWEEP_T=1,4,8,12,13,16,20,24,28,32,48,64 timeout -k 10 200 python scripts/diag/verify_T_sweep.py 2>&1 | grep "^T " || exit 1
  NVLLM_PAGED_MID=$m SWEEP_N=1024 SWEEP_CTX=512 SWEEP_T=16,13,16,24,32 timeout -k 10 200 python scripts/diag/verify_T_sweep.py 2>&1 | grep "^T \|^N " || exit 1
  NVLLM_PAGED_MID=$m SWEEP_N=128 SWEEP_CTX=4096 SWEEP_T=16,16,32 timeout -k 10 200 python scripts/diag/verify_T_sweep.py 2>&1 | grep "^T \|^N " || exit 1
  NVLLM_PAGED_MID=$m SWEEP_N=512 SWEEP_CTX=1024 SWEEP_T=16,16,32 timeout -k 10 200 python scripts/diag/verify_T_sweep.py 2>&1 | grep "^T \|^N " || exit 1
  NVLLM_PAGED_MID=$m SWEEP_N=64 SWEEP_CTX=2048 SWEEP_T=16,16,32 timeout -k 10 200 python scripts/diag/verify_T_sweep.py 2>&1 | grep "^T \|^N " || exit 1
done
