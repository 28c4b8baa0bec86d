import math, os, sys
import numpy as np, torch
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import nano_vllm_rs_amd as nv, oracle
from helpers import *
dev = torch.device('cuda:0')
def _dev(a, dtype=torch.bfloat16): return torch.from_numpy(np.ascontiguousarray(a)).to(device=dev, dtype=dtype)
def run(q, kc, vc, tables, ctx, scale):
    bt = torch.from_numpy(pad_tables(tables)).to(dev); cl = torch.tensor(ctx, dtype=torch.int32, device=dev)
    o = nv.paged_attention_batched(_dev(q), _dev(kc), _dev(vc), bt, cl, scale); torch.cuda.synchronize()
    return o.float().cpu().numpy()
Hq, Hkv, D, bs, T = 32, 8, 128, 16, 64
rng = np.random.default_rng(11)
ctx_lens = [100, 333]
kc, vc, tables = make_paged_case(rng, ctx_lens, 64, bs, Hkv, D)
kc = oracle.bf16_round(kc * 10)
q = bf16_randn(rng, (2, T, Hq, D), sigma=10.0)
scale = 1 / math.sqrt(D)
got = run(q, kc, vc, tables, ctx_lens, scale)
os.environ['NVLLM_PAGED_FLASH'] = '0'
gen = run(q, kc, vc, tables, ctx_lens, scale)
want = oracle_paged_batch(q, kc, vc, tables, ctx_lens, bs, scale)
for name, a in (('flash', got), ('generic', gen)):
    d = np.abs(a - want).reshape(2, T, Hq, D).max(-1)
    print(name, 'max', d.max(), 'mean', np.abs(a - want).mean(), 'rows>0.05:', np.argwhere(d > 0.05)[:20].tolist())
# gap between top-2 scores for the worst row
d = np.abs(got - want).reshape(2, T, Hq, D).max(-1)
s, t, h = np.unravel_index(d.argmax(), d.shape)
ids = tables[s]; ctx = ctx_lens[s]
K = np.concatenate([kc[b] for b in ids])[:ctx][:, h // 4]
sc = (K @ q[s, t, h]) * scale
lim = ctx - T + t
sc = np.sort(sc[:lim + 1])[::-1]
print('worst', s, t, h, 'top scores', sc[:4])
