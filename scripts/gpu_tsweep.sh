set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02_verify_T_sweep.txt; : > $O
cd $R
SWEEP_T=1,4,8,12,13,16,24,32,48,64,96,128,256 timeout -k 10 200 python scripts/diag/verify_T_sweep.py >> $O 2>&1 || { tail $O; exit 1; }
SWEEP_N=16 SWEEP_CTX=4096 SWEEP_T=1,4,8,12,13,16,32,64,128 timeout -k 10 200 python scripts/diag/verify_T_sweep.py >> $O 2>&1 || { tail $O; exit 1; }
SWEEP_N=32 SWEEP_CTX=4096 SWEEP_T=16,32,64,128 timeout -k 10 200 python scripts/diag/verify_T_sweep.py >> $O 2>&1 || { tail $O; exit 1; }
cat $O
