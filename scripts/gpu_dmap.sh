set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd $R
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread -k "flash or prefill or chunk or verify or scheduler or split or config or plumbing or continuous" > $O/k1_tests.log 2>&1 || { tail -30 $O/k1_tests.log; exit 1; }
tail -2 $O/k1_tests.log
SWEEP_T=13,16,32,48,64,128 timeout -k 10 200 python scripts/diag/verify_T_sweep.py 2>&1 | grep -v amdgpu.ids || exit 1
bash scripts/gpu_lab_paged.sh
