# K1 lab A/B: correctness vs the generic f32 kernel + timing + stamps, over shapes that exercise
# every phase combination.  Output: gpurun_out/lab_r02.txt
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/lab_r02.txt; : > $O
cd $R
L=./scripts/k1_lab/k1_lab
for args in "B=1 S=4096" "B=8 S=4096" "B=2 S=1024" "B=1 S=4096 causal=0" "B=1 S=1000" "B=3 S=520 causal=0" "B=1 S=128" "B=2 S=200" "B=1 S=4096 spike=1" "B=4 S=2048 Hq=32 Hkv=8"; do
  timeout -k 10 100 $L $args rounds=3 >> $O 2>&1 || { echo "FAILED: $args" >> $O; tail -5 $O; exit 1; }
done
grep -E "shape|check|median|cycles per wave" $O
