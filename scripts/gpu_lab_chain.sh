# K1 lab: the chained kernel's stamped segments (paged T=32 / T=64 at 256 x 2048, dense B=8), chain off / on
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/lab_chain.txt; : > $O
cd $R
L=./scripts/k1_lab/k1_lab
for c in 0 4 8; do
  echo "=== NVLLM_K1_CHAIN=$c" >> $O
  for T in 32 64; do
    NVLLM_K1_CHAIN=$c LAB_PAGED=$T timeout -k 10 60 $L B=256 S=2048 Hq=32 Hkv=8 rounds=3 >> $O 2>&1 || { echo "FAILED" >> $O; tail -5 $O; exit 1; }
  done
  NVLLM_K1_CHAIN=$c timeout -k 10 60 $L B=8 S=4096 rounds=3 check=0 >> $O 2>&1 || exit 1
done
grep -E "===|shape|median|cycles per wave|stamps over" $O
