set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
timeout -k 10 400 python -u -m pytest $R/tests -m gpu -x -q --timeout 120 --timeout-method thread > $O/t4.log 2>&1 || { tail -30 $O/t4.log; exit 1; }
tail -2 $O/t4.log
cd $R
timeout -k 10 200 python bench.py --workload shared --steps 100 --warmup 10 --cpu-seconds 4 > $O/bench_shared.json 2> $O/bench_shared.err && echo shared-ok
timeout -k 10 400 python bench.py --workload long --steps 10 --warmup 2 --cpu-seconds 4 > $O/bench_long.json 2> $O/bench_long.err && echo long-ok
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pmc_fetch --output-format csv -- python3 $R/bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-also > $O/pmc_fetch.json 2> $O/pmc_fetch.err && echo fetch-ok
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/pmc_write --output-format csv -- python3 $R/bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-also > $O/pmc_write.json 2> $O/pmc_write.err && echo write-ok
ls $O/pmc_fetch/*/ | head
