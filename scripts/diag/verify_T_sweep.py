"""Multi-token verify on the paged cache: time vs number of new tokens T (256 seqs x 2048 ctx, 32Q/8KV).
T*G <= 32 rows per kv head runs in the decode kernel, above that in the flash kernel over the cache."""
import math, os, sys
import torch
sys.path.insert(0, '.')
import nano_vllm_rs_amd as nv
dev = torch.device('cuda:0')
Hq, Hkv, D, bs = int(os.environ.get('SWEEP_HQ', 32)), int(os.environ.get('SWEEP_HKV', 8)), 128, 16
N, ctx = int(os.environ.get('SWEEP_N', 256)), int(os.environ.get('SWEEP_CTX', 2048))
nblk = ctx // bs
nb = N * nblk
# SWEEP_LAYERS caches used in turn (a model's layers: nothing of a launch's K/V is left in the Infinity
# Cache for the next), SWEEP_SLACK = pool blocks per used block (bench.py: 2 caches, 1.25)
LAYERS, SLACK = int(os.environ.get('SWEEP_LAYERS', 1)), float(os.environ.get('SWEEP_SLACK', 1.0))
pool = int(nb * SLACK)
caches = [(torch.randn((pool, bs, Hkv, D), device=dev, dtype=torch.bfloat16),
           torch.randn((pool, bs, Hkv, D), device=dev, dtype=torch.bfloat16)) for _ in range(LAYERS)]
kc, vc = caches[0]
bt = torch.randperm(pool)[:nb].to(torch.int32).reshape(N, nblk).to(dev)
if os.environ.get('SWEEP_SEQUENTIAL'):    # every sequence's blocks in ascending order, one sequence after the other
    bt = torch.arange(nb, dtype=torch.int32).reshape(N, nblk).to(dev)
if os.environ.get('SWEEP_HOT_BLOCKS'):   # every table entry among the first few blocks: K/V come from L2 (what is left is not HBM time)
    bt = bt % int(os.environ['SWEEP_HOT_BLOCKS'])
cl = torch.full((N,), ctx, dtype=torch.int32, device=dev)
def timeit(fn, iters=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3
Ts = [int(x) for x in os.environ.get('SWEEP_T', '1,4,8,9,12,13,16,24,32,36,48,64,96').split(',')]
print('N', N, 'ctx', ctx, 'Hq', Hq, 'Hkv', Hkv, 'layers', LAYERS, 'slack', SLACK, 'max_chunks', os.environ.get('NVLLM_PAGED_MAX_CHUNKS'), flush=True)
first = True
for T in Ts:
    q = torch.randn((N, T, Hq, D), device=dev, dtype=torch.bfloat16)
    if first:   # clocks and caches of a fresh process: the first entry used to read 5-10 % slow
        timeit(lambda: nv.paged_attention_batched(q, kc, vc, bt, cl, 1 / math.sqrt(D)), iters=30)
        first = False
    turn = [0]
    def launch():
        k, v = caches[turn[0] % LAYERS]
        turn[0] += 1
        return nv.paged_attention_batched(q, k, v, bt, cl, 1 / math.sqrt(D))
    t = timeit(launch)
    by = N * ctx * Hkv * D * 2 * 2
    print('T', T, 'rows/kvhead', T * Hq // Hkv, 'us', round(t, 1), 'GB/s of KV', round(by / t / 1e3, 1), flush=True)
