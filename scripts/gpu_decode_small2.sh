cd $GRAFT_REPO_ROOT
run() { python3 -c "import json,sys; d=json.load(sys.stdin); r=d['roofline']; print('$1', round(d['value'],1), d['unit'], round(d['ms_per_step']*1000,1), 'us')"; }
for n in 1 2 4 8; do for c in 2048 8192 32768; do for hpb in 0; do
  NVLLM_PAGED_HPB=$hpb timeout -k 10 100 python bench.py --num-seqs $n --ctx $c --steps 50 --warmup 5 --no-cpu-baseline --no-also | run "decode num_seqs=$n ctx=$c hpb=$hpb" || exit 1
done; done; done
