# where does the mid kernel start to win on small launches?  units = N x 8 kv heads on 256 CUs
set -o pipefail
R=$GRAFT_REPO_ROOT
cd $R
for n in 16 32 48 64 96 128; do
for m in 0 1; do
  echo "== N=$n ctx 2048 NVLLM_PAGED_MID=$m"
  NVLLM_PAGED_MID=$m SWEEP_N=$n SWEEP_CTX=2048 SWEEP_T=13,16,24,32 timeout -k 10 200 python scripts/diag/verify_T_sweep.py 2>&1 | grep "^T " || exit 1
done
done
for n in 16 32 64; do
for m in 0 1; do
  echo "== N=$n ctx 8192 NVLLM_PAGED_MID=$m"
  NVLLM_PAGED_MID=$m SWEEP_N=$n SWEEP_CTX=8192 SWEEP_T=16,32 timeout -k 10 200 python scripts/diag/verify_T_sweep.py 2>&1 | grep "^T " || exit 1
done
done
