set -o pipefail
R=$GRAFT_REPO_ROOT
cd $R
for m in 0 1 0 1; do
  echo "== NVLLM_PAGED_MID=$m"
  NVLLM_PAGED_MID=$m SWEEP_T=1,12,13,16,24,32 timeout -k 10 200 python scripts/diag/verify_T_sweep.py 2>&1 | grep "^T " || exit 1
done
for m in 0 1; do
  echo "== NVLLM_PAGED_MID=$m"
  NVLLM_PAGED_MID=$m SWEEP_N=1024 SWEEP_CTX=512 SWEEP_T=16,32 timeout -k 10 200 python scripts/diag/verify_T_sweep.py 2>&1 | grep "^T \|^N " || exit 1
  NVLLM_PAGED_MID=$m SWEEP_N=128 SWEEP_CTX=4096 SWEEP_T=16,32 timeout -k 10 200 python scripts/diag/verify_T_sweep.py 2>&1 | grep "^T \|^N " || exit 1
done
