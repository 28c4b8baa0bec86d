# decode: adaptive cache policy and wave target, configs 3 / 4 / 5 and small batches
cd $GRAFT_REPO_ROOT
run() { python3 -c "import json,sys; d=json.load(sys.stdin); r=d['roofline']; print('$1', round(d['value'],1), d['unit'], round(d['ms_per_step']*1000,1), 'us', 'eff_logical', round(r.get('effective_logical_GBps',0),1))"; }
timeout -k 10 300 python -u -m pytest tests/test_gpu_paged_attention.py tests/test_golden.py -m gpu -x -q --timeout 120 --timeout-method thread 2>&1 | tail -2
timeout -k 10 100 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-also | run "config3 default" || exit 1
for w in shared long; do
  st=30; [ $w = long ] && st=6
  timeout -k 10 300 python bench.py --workload $w --steps $st --warmup 2 --no-cpu-baseline --no-also | run "$w default" || exit 1
done
for n in 16 64 128; do for wt in 2048 4096 8192; do
  NVLLM_PAGED_WAVE_TARGET=$wt timeout -k 10 100 python bench.py --num-seqs $n --steps 50 --warmup 5 --no-cpu-baseline --no-also | run "decode num_seqs=$n wave_target=$wt" || exit 1
done; done
