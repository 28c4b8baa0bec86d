# end-of-round check: smoke, the whole GPU suite, the default bench line
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd $R
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 || exit 1
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > $O/gpu_tests.log 2>&1 || { tail -40 $O/gpu_tests.log; exit 1; }
tail -3 $O/gpu_tests.log
timeout -k 10 300 python bench.py > $O/final_bench.json 2> $O/final_bench.err || { tail -5 $O/final_bench.err; exit 1; }
python3 -c "
import json; d=json.load(open('$O/final_bench.json')); print('decode', round(d['value'],1), d['roofline']['frac'], d['ms_per_step'])
for k,v in d['also'].items(): print(k, round(v['value'],1), v.get('frac', v.get('roofline',{}).get('frac')), v['ms_per_step'])"
