# random block tables against sequential ones (DRAM locality), 8 kv heads and 1 kv head (contiguous blocks)
set -o pipefail
R=$GRAFT_REPO_ROOT
cd $R
for seqt in "" 1; do
  echo "== sequential=$seqt 32Q/8KV 256 seqs"
  SWEEP_SEQUENTIAL=$seqt SWEEP_T=1,8,16,32 timeout -k 10 200 python scripts/diag/verify_T_sweep.py 2>&1 | grep "^T " || exit 1
  echo "== sequential=$seqt 4Q/1KV 2048 seqs"
  SWEEP_SEQUENTIAL=$seqt SWEEP_HQ=4 SWEEP_HKV=1 SWEEP_N=2048 SWEEP_T=1,8,16,32 timeout -k 10 200 python scripts/diag/verify_T_sweep.py 2>&1 | grep "^T " || exit 1
done
