# Does the 256-byte-rows-at-2-KiB-stride pattern (one kv head of eight per workgroup) cost HBM efficiency?
# Same bytes, same number of workgroups, but one kv head in the cache: every block is 4 KiB contiguous.
set -o pipefail
R=$GRAFT_REPO_ROOT
cd $R
for m in 0 2; do
  echo "== NVLLM_PAGED_MID=$m  32Q/8KV, 256 seqs"
  NVLLM_PAGED_MID=$m SWEEP_T=8,13,16,24,32 timeout -k 10 200 python scripts/diag/verify_T_sweep.py 2>&1 | grep "^T " || exit 1
  echo "== NVLLM_PAGED_MID=$m  4Q/1KV, 2048 seqs"
  NVLLM_PAGED_MID=$m SWEEP_HQ=4 SWEEP_HKV=1 SWEEP_N=2048 SWEEP_T=8,13,16,24,32 timeout -k 10 200 python scripts/diag/verify_T_sweep.py 2>&1 | grep "^T " || exit 1
done
