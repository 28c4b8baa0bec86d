"""K1 over the cache at 256 seqs x 2048 ctx: time with every sequence on its own blocks (K/V from
HBM) against every sequence on the SAME blocks (K/V from L2): what the memory side costs."""
import math, sys, torch
sys.path.insert(0, '.')
import nano_vllm_rs_amd as nv
dev = torch.device('cuda:0')
Hq, Hkv, D, bs = 32, 8, 128, 16
N, ctx = 256, 2048
nblk = ctx // bs
nb = N * nblk
kc = torch.randn((nb, bs, Hkv, D), device=dev, dtype=torch.bfloat16)
vc = torch.randn((nb, bs, Hkv, D), device=dev, dtype=torch.bfloat16)
own = torch.randperm(nb).to(torch.int32).reshape(N, nblk).to(dev)
same = own[:1].repeat(N, 1).contiguous()
cl = torch.full((N,), ctx, dtype=torch.int32, device=dev)
def timeit(fn, iters=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3
for T in (16, 32, 48, 64, 128):
    q = torch.randn((N, T, Hq, D), device=dev, dtype=torch.bfloat16)
    t_own = timeit(lambda: nv.paged_attention_batched(q, kc, vc, own, cl, 1 / math.sqrt(D)))
    t_same = timeit(lambda: nv.paged_attention_batched(q, kc, vc, same, cl, 1 / math.sqrt(D)))
    fl = 4.0 * N * T * Hq * ctx * D
    print('T', T, 'own tables %.1f us (%.0f TF, %.2f TB/s)' % (t_own, fl / t_own / 1e6, N * ctx * 4096 / t_own / 1e6),
          '| same table %.1f us (%.0f TF)' % (t_same, fl / t_same / 1e6), flush=True)
