# K4 lab: variants timed, then FETCH_SIZE / WRITE_SIZE passes (separate runs) of the same program.
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd $R
timeout -k 10 150 scripts/k4_lab/k4_variants > $O/k4_lab.txt 2> $O/k4_lab.err && echo lab-ok || { tail -5 $O/k4_lab.err; exit 1; }
cd /tmp && export TMPDIR=/tmp
rm -rf $O/k4_fetch $O/k4_write
timeout -s KILL 150 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/k4_fetch --output-format csv -- $R/scripts/k4_lab/k4_variants > /dev/null 2> $O/k4_fetch.err && echo fetch-ok || exit 1
timeout -s KILL 150 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/k4_write --output-format csv -- $R/scripts/k4_lab/k4_variants > /dev/null 2> $O/k4_write.err && echo write-ok || exit 1
cd $R
python3 scripts/k4_lab/pmc_by_grid.py $O/k4_fetch $O/k4_write > $O/k4_pmc.txt 2>&1; tail -5 $O/k4_pmc.txt; cat $O/k4_lab.txt
