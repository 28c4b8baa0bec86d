set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd $R
timeout -k 10 200 python -u -m pytest tests/wip_no_alloc.py -x -q --timeout 120 --timeout-method thread > $O/r02a_noalloc.log 2>&1 || { tail -40 $O/r02a_noalloc.log; exit 1; }
tail -3 $O/r02a_noalloc.log
timeout -k 10 300 python bench.py > $O/r02a_bench.json 2> $O/r02a_bench.err && echo bench-ok || { tail -5 $O/r02a_bench.err; exit 1; }
python3 -c "
import json; d=json.load(open('$O/r02a_bench.json'))
print('decode', round(d['value'],1), d['unit'], 'frac', round(d['roofline']['frac'],3))
for k,v in d['also'].items(): print(k, round(v['value'],1), v.get('unit'), round(v.get('frac', v.get('roofline',{}).get('frac',0)),3), round(v['ms_per_step']*1000,1),'us')
"
