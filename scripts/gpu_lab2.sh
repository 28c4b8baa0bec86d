set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/lab_r02b.txt; : > $O
cd $R
L=./scripts/k1_lab/k1_lab
for args in "B=8 S=4096" "B=1 S=4096"; do
  timeout -k 10 100 $L $args rounds=3 check=0 >> $O 2>&1 || { echo "FAILED: $args" >> $O; tail -5 $O; exit 1; }
done
grep -E "shape|median|cycles per wave|stamps over" $O
