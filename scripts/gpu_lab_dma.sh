set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/lab_dma.txt; : > $O
cd $R
L=./scripts/k1_lab/k1_lab
for args in "B=1 S=128" "B=2 S=200" "B=1 S=1000" "B=3 S=520 causal=0" "B=2 S=1024" "B=1 S=4096" "B=8 S=4096" "B=1 S=4096 causal=0" "B=4 S=2048 Hq=32 Hkv=8"; do
  timeout -k 10 60 $L $args rounds=3 >> $O 2>&1 || { echo "FAILED: $args" >> $O; tail -5 $O; exit 1; }
done
grep -E "shape|median|check|cycles per wave" $O
