set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $O/c4_fetch $O/c4_tcc
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/c4_fetch --output-format csv -- python3 $R/bench.py --workload shared --steps 20 --warmup 2 --no-cpu-baseline --no-also > $O/c4_fetch.json 2> $O/c4_fetch.err && echo fetch-ok || exit 1
timeout -s KILL 120 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace -d $O/c4_tcc --output-format csv -- python3 $R/bench.py --workload shared --steps 20 --warmup 2 --no-cpu-baseline --no-also > $O/c4_tcc.json 2> $O/c4_tcc.err && echo tcc-ok || exit 1
cd $R
python3 scripts/pmc_summary.py $O/c4_fetch $O/c4_tcc | grep -i "paged_attn_kernel" | head
