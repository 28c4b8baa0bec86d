# decode at small batch / long context: where split-K and the merge pass carry the launch
cd $GRAFT_REPO_ROOT
run() { python3 -c "import json,sys; d=json.load(sys.stdin); r=d['roofline']; print('$1', round(d['value'],1), d['unit'], round(d['ms_per_step']*1000,1), 'us')"; }
for n in 1 2 4 8 16 32; do for c in 2048 8192 32768; do
  timeout -k 10 100 python bench.py --num-seqs $n --ctx $c --steps 50 --warmup 5 --no-cpu-baseline --no-also | run "decode num_seqs=$n ctx=$c" || exit 1
done; done
