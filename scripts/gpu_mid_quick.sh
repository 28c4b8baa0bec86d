set -o pipefail
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 300 python -u -m pytest tests/test_gpu_paged_mid.py -x -q --timeout 120 --timeout-method thread 2>&1 | tail -2 || exit 1
for hot in "" 32; do
  echo "== hot_blocks=$hot"
  SWEEP_HOT_BLOCKS=$hot SWEEP_T=13,16,20,24,28,32 timeout -k 10 200 python scripts/diag/verify_T_sweep.py 2>&1 | grep "^T " || exit 1
done
NVLLM_PAGED_MID=1 SWEEP_N=1024 SWEEP_CTX=512 SWEEP_T=13,16,24,32 timeout -k 10 200 python scripts/diag/verify_T_sweep.py 2>&1 | grep "^T \|^N " || exit 1
NVLLM_PAGED_MID=1 SWEEP_N=128 SWEEP_CTX=4096 SWEEP_T=16,24,32 timeout -k 10 200 python scripts/diag/verify_T_sweep.py 2>&1 | grep "^T \|^N " || exit 1
NVLLM_PAGED_MID=1 SWEEP_N=64 SWEEP_CTX=2048 SWEEP_T=16,24,32 timeout -k 10 200 python scripts/diag/verify_T_sweep.py 2>&1 | grep "^T \|^N " || exit 1
