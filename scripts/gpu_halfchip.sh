# how fast can fewer workgroups stream? decode, one split, 64 / 128 / 256 sequences of 2048 tokens
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/halfchip.txt; : > $O
cd $R
for n in 64 128 192 256; do
  NVLLM_PAGED_SPLITS=1 timeout -k 10 100 python bench.py --num-seqs $n --no-cpu-baseline --no-also --steps 100 --warmup 10 > $O.json 2>/dev/null || exit 1
  python3 -c "import json; d=json.load(open('$O.json')); print('decode splits=1 num_seqs=$n', round(d['value'],1), d['unit'], round(d['ms_per_step']*1000,1), 'us')" >> $O
done
cat $O
