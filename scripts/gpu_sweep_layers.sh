# one cache re-read by every launch (the sweeps so far) against caches used in turn and a pool with slack (bench.py)
set -o pipefail
R=$GRAFT_REPO_ROOT
cd $R
for cfg in "1 1.0" "2 1.0" "1 1.25" "2 1.25" "4 1.25"; do
  set -- $cfg
  echo "== layers=$1 slack=$2"
  SWEEP_LAYERS=$1 SWEEP_SLACK=$2 SWEEP_T=1,8,13,16,24,32,64 timeout -k 10 200 python scripts/diag/verify_T_sweep.py 2>&1 | grep "^T " || exit 1
done
