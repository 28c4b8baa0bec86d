# Round-2 evidence run: default bench line, kernel-trace stats of the same command, HBM traffic
# counters (separate passes), prefill batch sweep, configs 4/5.  Outputs under gpurun_out/r02_*.
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd $R
timeout -k 10 300 python bench.py > $O/r02_bench.json 2> $O/r02_bench.err && echo bench-ok || { tail -5 $O/r02_bench.err; exit 1; }
cd /tmp && export TMPDIR=/tmp
rm -rf $O/r02_trace $O/r02_fetch $O/r02_write
timeout -s KILL 200 rocprofv3 --kernel-trace --stats -d $O/r02_trace --output-format csv -- python3 $R/bench.py --no-cpu-baseline > $O/r02_trace.json 2> $O/r02_trace.err && echo trace-ok || exit 1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/r02_fetch --output-format csv -- python3 $R/bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-also > $O/r02_fetch.json 2> $O/r02_fetch.err && echo fetch-ok || exit 1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/r02_write --output-format csv -- python3 $R/bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-also > $O/r02_write.json 2> $O/r02_write.err && echo write-ok || exit 1
cd $R
for b in 1 2 4 8; do
  timeout -k 10 100 python bench.py --workload prefill --batch $b --steps 100 --warmup 10 --no-cpu-baseline > $O/r02_prefill_b$b.json 2>/dev/null && python3 -c "import json; d=json.load(open('$O/r02_prefill_b$b.json')); print(d['config']['workload'], round(d['value'],1), 'TF', round(d['ms_per_step']*1000,1), 'us')" || exit 1
done
for w in shared long producer; do
  timeout -k 10 280 python bench.py --workload $w --cpu-seconds 4 > $O/r02_bench_$w.json 2>$O/r02_bench_$w.err && python3 -c "import json; d=json.load(open('$O/r02_bench_$w.json')); print('$w', round(d['value'],1), d['unit'], d['roofline']['frac'])" || exit 1
done
find $O/r02_trace -name "*kernel_stats.csv" | head -2
python3 scripts/pmc_summary.py $O/r02_fetch $O/r02_write | grep -i "paged_attn_kernel\|duration" | head
