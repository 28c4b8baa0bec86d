# as gpu_mid_hot.sh with 32 hot blocks (2 MiB of K+V: inside every XCD's 4 MiB L2, not only the Infinity Cache)
set -o pipefail
R=$GRAFT_REPO_ROOT
cd $R
for m in 0 1 2; do
  echo "== NVLLM_PAGED_MID=$m hot_blocks=32"
  SWEEP_HOT_BLOCKS=32 NVLLM_PAGED_MID=$m SWEEP_T=1,8,13,16,24,32,64 timeout -k 10 200 python scripts/diag/verify_T_sweep.py 2>&1 | grep "^T " || exit 1
done
