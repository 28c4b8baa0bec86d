# the whole GPU suite, as the driver runs it at round end
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd $R
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > $O/gpu_tests.log 2>&1 || { tail -40 $O/gpu_tests.log; exit 1; }
tail -4 $O/gpu_tests.log
