# K1 lab: the four pieces of a request behind one write of M0 (K1_DEV_M0SHARE) against the product kernel:
# checked against the f64 reference, dense config 2 / B=8 and the paged shapes
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/lab_m0share.txt; : > $O
cd $R
L=./scripts/k1_lab/k1_lab
timeout -k 10 100 $L B=1 S=4096 rounds=5 >> $O 2>&1 || { tail -5 $O; exit 1; }
timeout -k 10 100 $L B=8 S=4096 rounds=5 >> $O 2>&1 || { tail -5 $O; exit 1; }
timeout -k 10 100 $L B=3 S=777 rounds=2 >> $O 2>&1 || { tail -5 $O; exit 1; }
for T in 32 64; do
  NVLLM_PAGED_MID=0 LAB_PAGED=$T timeout -k 10 100 $L B=256 S=2048 Hq=32 Hkv=8 rounds=3 >> $O 2>&1 || { tail -5 $O; exit 1; }
done
grep -E "shape|median|check" $O
