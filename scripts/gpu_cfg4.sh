set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd $R
timeout -k 10 300 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread -k "paged or config or shared or prefix or decode or golden or no_alloc" > $O/cfg4_tests.log 2>&1 || { tail -30 $O/cfg4_tests.log; exit 1; }
tail -2 $O/cfg4_tests.log
timeout -k 10 120 python scripts/diag/decode_shared_phases.py || exit 1
timeout -k 10 200 python bench.py --workload shared --no-cpu-baseline --steps 200 --warmup 20 > $O/cfg4.json 2> $O/cfg4.err || { tail -5 $O/cfg4.err; exit 1; }
python3 -c "import json; d=json.load(open('$O/cfg4.json')); print('shared', round(d['value'],1), d['unit'], d['roofline']['frac'], d['ms_per_step'])"
timeout -k 10 200 python bench.py --no-cpu-baseline --steps 100 --warmup 10 > $O/cfg3.json 2> $O/cfg3.err || { tail -5 $O/cfg3.err; exit 1; }
python3 -c "
import json; d=json.load(open('$O/cfg3.json')); print('decode', round(d['value'],1), d['unit'], d['roofline']['frac'], d['ms_per_step'])
for k,v in d.get('also',{}).items(): print(k, v)"
timeout -k 10 200 python bench.py --workload long --no-cpu-baseline --steps 50 --warmup 5 > $O/cfg5.json 2> $O/cfg5.err || { tail -5 $O/cfg5.err; exit 1; }
python3 -c "import json; d=json.load(open('$O/cfg5.json')); print('long', round(d['value'],1), d['unit'], d['roofline']['frac'], d['ms_per_step'])"
