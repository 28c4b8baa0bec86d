# Round-1 evidence run: default bench line, kernel-trace stats of the same command, HBM traffic
# counters (separate passes), prefill batch sweep.  Outputs under gpurun_out/r01b_*.
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd $R
timeout -k 10 300 python bench.py > $O/r01b_bench.json 2> $O/r01b_bench.err && echo bench-ok || { tail -5 $O/r01b_bench.err; exit 1; }
cd /tmp && export TMPDIR=/tmp
rm -rf $O/r01b_trace $O/r01b_fetch $O/r01b_write
timeout -s KILL 200 rocprofv3 --kernel-trace --stats -d $O/r01b_trace --output-format csv -- python3 $R/bench.py --no-cpu-baseline > $O/r01b_trace.json 2> $O/r01b_trace.err && echo trace-ok
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/r01b_fetch --output-format csv -- python3 $R/bench.py --steps 20 --warmup 2 --no-cpu-baseline > $O/r01b_fetch.json 2> $O/r01b_fetch.err && echo fetch-ok
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/r01b_write --output-format csv -- python3 $R/bench.py --steps 20 --warmup 2 --no-cpu-baseline > $O/r01b_write.json 2> $O/r01b_write.err && echo write-ok
cd $R
for b in 1 2 4 8; do
  timeout -k 10 100 python bench.py --workload prefill --batch $b --steps 100 --warmup 10 --no-cpu-baseline > $O/r01b_prefill_b$b.json 2>/dev/null && python3 -c "import json; d=json.load(open('$O/r01b_prefill_b$b.json')); print(d['config']['workload'], round(d['value'],1), 'TF', round(d['ms_per_step']*1000,1), 'us')"
done
ls $O/r01b_trace/*/ | head
