# Round-1 evidence run: default bench line, kernel-trace stats of the same command, HBM traffic
# counters (separate passes), prefill batch sweep.  Outputs under gpurun_out/r01c_*.
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd $R
timeout -k 10 300 python bench.py > $O/r01c_bench.json 2> $O/r01c_bench.err && echo bench-ok || { tail -5 $O/r01c_bench.err; exit 1; }
cd /tmp && export TMPDIR=/tmp
rm -rf $O/r01c_trace $O/r01c_fetch $O/r01c_write
timeout -s KILL 200 rocprofv3 --kernel-trace --stats -d $O/r01c_trace --output-format csv -- python3 $R/bench.py --no-cpu-baseline > $O/r01c_trace.json 2> $O/r01c_trace.err && echo trace-ok
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/r01c_fetch --output-format csv -- python3 $R/bench.py --steps 20 --warmup 2 --no-cpu-baseline > $O/r01c_fetch.json 2> $O/r01c_fetch.err && echo fetch-ok
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/r01c_write --output-format csv -- python3 $R/bench.py --steps 20 --warmup 2 --no-cpu-baseline > $O/r01c_write.json 2> $O/r01c_write.err && echo write-ok
cd $R
for b in 1 2 4 8; do
  timeout -k 10 100 python bench.py --workload prefill --batch $b --steps 100 --warmup 10 --no-cpu-baseline > $O/r01c_prefill_b$b.json 2>/dev/null && python3 -c "import json; d=json.load(open('$O/r01c_prefill_b$b.json')); print(d['config']['workload'], round(d['value'],1), 'TF', round(d['ms_per_step']*1000,1), 'us')"
done
ls $O/r01c_trace/*/ | head
for w in shared long producer; do
  timeout -k 10 280 python bench.py --workload $w --no-cpu-baseline > $O/r01c_bench_$w.json 2>$O/r01c_bench_$w.err && python3 -c "import json; d=json.load(open('$O/r01c_bench_$w.json')); print('$w', round(d['value'],1), d['unit'], d['roofline']['frac'])"
done
find $O/r01c_trace -name "*kernel_stats.csv" | head -2
