# HBM traffic of the mid kernel (verify, T = 32 and T = 16 on config 3's batch): FETCH_SIZE and WRITE_SIZE in separate passes
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for T in 32 16; do
  rm -rf $O/r03_mid_fetch_T$T $O/r03_mid_write_T$T
  timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/r03_mid_fetch_T$T --output-format csv -- python3 $R/bench.py --workload verify --tokens $T --steps 20 --warmup 2 --no-cpu-baseline > $O/r03_mid_fetch_T$T.json 2> $O/r03_mid_fetch_T$T.err && echo fetch-$T-ok || exit 1
  timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/r03_mid_write_T$T --output-format csv -- python3 $R/bench.py --workload verify --tokens $T --steps 20 --warmup 2 --no-cpu-baseline > $O/r03_mid_write_T$T.json 2> $O/r03_mid_write_T$T.err && echo write-$T-ok || exit 1
  python3 $R/scripts/pmc_summary.py $O/r03_mid_fetch_T$T $O/r03_mid_write_T$T | grep -i "paged_attn_mid\|duration"
done
