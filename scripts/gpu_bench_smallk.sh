# the bench line at the driver's short settings and at the defaults (does the settle phase make them agree?)
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd $R
for kw in "20 5" "5 1" "200 20"; do
  set -- $kw
  timeout -k 10 400 python bench.py --steps $1 --warmup $2 --cpu-seconds 3 > $O/bench_k$1.json 2> $O/bench_k$1.err || { tail -5 $O/bench_k$1.err; exit 1; }
  python3 -c "
import json
d=json.load(open('$O/bench_k$1.json'))
print('K=$1 W=$2', 'decode', round(d['ms_per_step'],4), round(d['roofline']['frac'],4), '| prefill', round(d['also']['prefill']['ms_per_step'],4), round(d['also']['prefill']['roofline']['frac'],4), d['also']['prefill']['roofline'].get('launch_ms_median'), '| T8', round(d['also']['verify_T8']['frac'],4), 'T32', round(d['also']['verify_T32']['frac'],4), '| kv32k', round(d['also']['kv_write_32768']['frac'],4), 'prod', round(d['also']['qk_norm_rope_cache_32768']['frac'],4))
"
done
