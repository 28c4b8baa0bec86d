# K1-focused GPU tests, then the lab (paged and dense shapes)
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd $R
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread -k "flash or prefill or chunk or verify or scheduler or split or config or plumbing or continuous" > $O/k1_tests.log 2>&1 || { tail -30 $O/k1_tests.log; exit 1; }
tail -2 $O/k1_tests.log
bash scripts/gpu_lab_paged.sh
