# flash parity tests, then prefill throughput at B = 1, 2, 8
set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -u -m pytest tests/test_gpu_flash_attention.py tests/test_golden.py -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/tf.log 2>&1 || { tail -30 gpurun_out/tf.log | cut -c1-300; exit 1; }
tail -1 gpurun_out/tf.log
for b in 1 2 8; do
  timeout -k 10 100 python bench.py --workload prefill --batch $b --steps 100 --warmup 10 --no-cpu-baseline | python3 -c "import json,sys; d=json.load(sys.stdin); print(d['config']['workload'], round(d['value'],1), 'TF', round(d['ms_per_step']*1000,1), 'us')" || exit 1
done
