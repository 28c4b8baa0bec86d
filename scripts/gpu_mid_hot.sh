# Where does the time of the short-chunk kernels go?  The same launches with every table entry among
# the first 256 blocks (K/V served by L2): what remains is compute + synchronisation, not HBM.
set -o pipefail
R=$GRAFT_REPO_ROOT
cd $R
for hot in "" 256; do
for m in 0 1; do
  echo "== NVLLM_PAGED_MID=$m hot_blocks=$hot"
  SWEEP_HOT_BLOCKS=$hot NVLLM_PAGED_MID=$m SWEEP_T=1,8,16,24,32,64 timeout -k 10 200 python scripts/diag/verify_T_sweep.py 2>&1 | grep "^T " || exit 1
done
done
