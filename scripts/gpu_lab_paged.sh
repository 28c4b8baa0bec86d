# K1 lab, paged mode: stamped segments of the paged variant next to the dense kernel
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/lab_paged.txt; : > $O
cd $R
L=./scripts/k1_lab/k1_lab
for cfg in "64:B=32 S=2048" "64:B=256 S=2048" "32:B=256 S=2048" "48:B=256 S=2048" "128:B=64 S=2048"; do
  T=${cfg%%:*}; args=${cfg#*:}
  LAB_PAGED=$T timeout -k 10 60 $L $args Hq=32 Hkv=8 rounds=3 >> $O 2>&1 || { echo "FAILED: $cfg" >> $O; tail -5 $O; exit 1; }
done
timeout -k 10 60 $L B=1 S=4096 rounds=3 >> $O 2>&1 || exit 1
timeout -k 10 60 $L B=8 S=4096 rounds=3 >> $O 2>&1 || exit 1
grep -E "shape|median|cycles per wave|stamps over|wall:|check" $O
