# Chain: correctness (bit-identity vs unchained, oracle parity suites), then timing with the chain off / on.
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd $R
timeout -k 10 500 python -u -m pytest tests/test_gpu_k1_chain.py tests/test_gpu_flash_attention.py tests/test_gpu_paged_prefill.py tests/test_gpu_paged_fuzz.py -x -q --timeout 120 --timeout-method thread > $O/chain_tests.txt 2>&1 || { tail -30 $O/chain_tests.txt; exit 1; }
tail -3 $O/chain_tests.txt
for c in 0 2 4 8; do
  echo "== NVLLM_K1_CHAIN=$c" >> $O/chain_timing.txt
  for b in 2 4 8; do
    NVLLM_K1_CHAIN=$c timeout -k 10 100 python bench.py --workload prefill --batch $b --steps 60 --warmup 10 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('prefill B=$b', round(d['ms_per_step']*1000,1), 'us', round(d['roofline']['frac'],4))" >> $O/chain_timing.txt || exit 1
  done
  NVLLM_K1_CHAIN=$c SWEEP_T=12,13,16,24,32,48,64,96,128 timeout -k 10 200 python scripts/diag/verify_T_sweep.py 2>&1 | grep "^T " >> $O/chain_timing.txt || exit 1
done
cat $O/chain_timing.txt
