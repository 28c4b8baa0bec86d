# K1 lab, round 3: variants built by scripts/k1_lab/build.sh; args = lab shapes.  Output gpurun_out/lab_r03.txt
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${LAB_OUT:-lab_r03.txt}; : > $O
cd $R
L=./scripts/k1_lab/k1_lab
IFS=';' read -ra SHAPES <<< "${LAB_SHAPES:-B=1 S=4096;B=8 S=4096}"
for args in "${SHAPES[@]}"; do
  timeout -k 10 120 $L $args rounds=${LAB_ROUNDS:-5} check=${LAB_CHECK:-1} >> $O 2>&1 || { echo "FAILED: $args" >> $O; tail -5 $O; exit 1; }
done
grep -E "shape|check|median|cycles per wave|stamps over|longest" $O
