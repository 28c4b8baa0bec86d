# Round-3 evidence run: default bench line, kernel-trace stats of the same command, HBM traffic
# counters (separate passes) for config 3 (decode) and config 2 (prefill), K1's matrix-pipe and
# instruction-fetch counters, prefill batch sweep, configs 4/5, producer, verify T sweep, head-sharded
# mode at one rank.  Outputs under gpurun_out/r03_*.
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd $R
timeout -k 10 400 python bench.py > $O/r03_bench.json 2> $O/r03_bench.err && echo bench-ok || { tail -5 $O/r03_bench.err; exit 1; }
cd /tmp && export TMPDIR=/tmp
rm -rf $O/r03_trace $O/r03_fetch $O/r03_write $O/r03_pfetch $O/r03_pwrite
timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $O/r03_trace --output-format csv -- python3 $R/bench.py --no-cpu-baseline > $O/r03_trace.json 2> $O/r03_trace.err && echo trace-ok || exit 1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/r03_fetch --output-format csv -- python3 $R/bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-also > $O/r03_fetch.json 2> $O/r03_fetch.err && echo fetch-ok || exit 1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/r03_write --output-format csv -- python3 $R/bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-also > $O/r03_write.json 2> $O/r03_write.err && echo write-ok || exit 1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/r03_pfetch --output-format csv -- python3 $R/bench.py --workload prefill --steps 20 --warmup 2 --no-cpu-baseline > $O/r03_pfetch.json 2> $O/r03_pfetch.err && echo pfetch-ok || exit 1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/r03_pwrite --output-format csv -- python3 $R/bench.py --workload prefill --steps 20 --warmup 2 --no-cpu-baseline > $O/r03_pwrite.json 2> $O/r03_pwrite.err && echo pwrite-ok || exit 1
timeout -s KILL 60 rocprofv3 -L > $O/r03_counters_avail.txt 2>&1; grep -o "SQC_ICACHE[A-Z_]*\|SQ_IFETCH[A-Z_]*\|SQ_WAIT_IFETCH[A-Z_]*\|SQ_INST_CYCLES_VMEM[A-Z_]*" $O/r03_counters_avail.txt | sort -u | tr '\n' ' '; echo
for shape in "B=1 S=4096" "B=8 S=4096"; do
  tag=$(echo $shape | tr -d ' =' )
  ARGS="$shape check=0 rounds=1 iters=3"
  rm -rf $O/r03_k1pmc_a_$tag $O/r03_k1pmc_b_$tag $O/r03_k1pmc_c_$tag
  timeout -s KILL 120 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE --kernel-trace -d $O/r03_k1pmc_a_$tag --output-format csv -- $R/scripts/k1_lab/k1_lab $ARGS > $O/r03_k1pmc_a_$tag.log 2>&1 && echo passA-$tag-ok || { tail -5 $O/r03_k1pmc_a_$tag.log; exit 1; }
  timeout -s KILL 120 rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_MFMA --kernel-trace -d $O/r03_k1pmc_b_$tag --output-format csv -- $R/scripts/k1_lab/k1_lab $ARGS > $O/r03_k1pmc_b_$tag.log 2>&1 && echo passB-$tag-ok || { tail -5 $O/r03_k1pmc_b_$tag.log; exit 1; }
  timeout -s KILL 120 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH GRBM_GUI_ACTIVE --kernel-trace -d $O/r03_k1pmc_c_$tag --output-format csv -- $R/scripts/k1_lab/k1_lab $ARGS > $O/r03_k1pmc_c_$tag.log 2>&1 && echo passC-$tag-ok || { echo passC-$tag-FAILED; tail -3 $O/r03_k1pmc_c_$tag.log; }
  echo "#### $shape causal 16Q/8KV hd128"
  python3 $R/scripts/pmc_k1_compare.py 6 cur_r03,stamped $O/r03_k1pmc_a_$tag $O/r03_k1pmc_b_$tag $O/r03_k1pmc_c_$tag
done > $O/r03_pmc_k1.txt
cd $R
for b in 1 2 4 8; do
  timeout -k 10 100 python bench.py --workload prefill --batch $b --steps 100 --warmup 10 --no-cpu-baseline > $O/r03_prefill_b$b.json 2>/dev/null && python3 -c "import json; d=json.load(open('$O/r03_prefill_b$b.json')); print(d['config']['workload'], round(d['value'],1), 'TF', round(d['ms_per_step']*1000,1), 'us', d['roofline'].get('launch_ms_median'))" || exit 1
done
for w in shared long producer; do
  timeout -k 10 280 python bench.py --workload $w --cpu-seconds 4 > $O/r03_bench_$w.json 2>$O/r03_bench_$w.err && python3 -c "import json; d=json.load(open('$O/r03_bench_$w.json')); print('$w', round(d['value'],1), d['unit'], d['roofline']['frac'])" || exit 1
done
timeout -k 10 200 python bench.py --parallelism heads --gpus 1 --no-cpu-baseline > $O/r03_bench_heads_1rank.json 2>/dev/null && echo heads-ok || exit 1
V=$O/r03_verify_T_sweep.txt; : > $V
SWEEP_T=1,4,8,12,13,16,24,32,48,64,96,128,256 timeout -k 10 200 python scripts/diag/verify_T_sweep.py >> $V 2>&1 || { tail $V; exit 1; }
SWEEP_N=16 SWEEP_CTX=4096 SWEEP_T=1,4,8,12,13,16,32,64,128 timeout -k 10 200 python scripts/diag/verify_T_sweep.py >> $V 2>&1 || { tail $V; exit 1; }
SWEEP_N=32 SWEEP_CTX=4096 SWEEP_T=16,32,64,128 timeout -k 10 200 python scripts/diag/verify_T_sweep.py >> $V 2>&1 || { tail $V; exit 1; }
find $O/r03_trace -name "*kernel_stats.csv" | head -2
python3 scripts/pmc_summary.py $O/r03_fetch $O/r03_write $O/r03_pfetch $O/r03_pwrite | grep -i "paged_attn_kernel\|flash_attn_mfma\|duration" | head -20
cat $O/r03_bench.json | head -c 3000
