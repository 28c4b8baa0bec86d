"""Decode kernel at 256 seqs, T = 4: what the shared-prefix half and the private half of config 4
cost on their own -- all sequences on the SAME 1024 tokens (every read from L2), every sequence on
its own 1024 tokens (HBM), and config 4 itself (1024 shared + 1024 private)."""
import math, sys, torch
sys.path.insert(0, '.')
import nano_vllm_rs_amd as nv
dev = torch.device('cuda:0')
Hq, Hkv, D, bs = 32, 8, 128, 16
N, T = 256, 4
half = 1024 // bs
nb = N * half + half + 64
kc = torch.randn((nb, bs, Hkv, D), device=dev, dtype=torch.bfloat16)
vc = torch.randn((nb, bs, Hkv, D), device=dev, dtype=torch.bfloat16)
perm = torch.randperm(nb).to(torch.int32)
shared = perm[:half]
priv = perm[half:half + N * half].reshape(N, half)
t_same = shared[None].repeat(N, 1).contiguous().to(dev)
t_own = priv.contiguous().to(dev)
t_cfg4 = torch.cat([shared[None].repeat(N, 1), priv], dim=1).contiguous().to(dev)
q = torch.randn((N, T, Hq, D), device=dev, dtype=torch.bfloat16)
def timeit(fn, iters=50, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3
sc = 1 / math.sqrt(D)
# half of the sequences (alternating groups of 8 / 4 / 1) on the same table, the others on their own
def mixed(shift):
    t = t_own.clone()
    sel = ((torch.arange(N) >> shift) & 1).bool().to(dev)
    t[sel] = t_same[sel]
    return t.contiguous()
for name, tab, ctx in (("same 1024", t_same, 1024), ("own 1024", t_own, 1024), ("config 4 (1024 + 1024)", t_cfg4, 2048),
                       ("half same / half own, groups of 8", mixed(3), 1024), ("... groups of 4", mixed(2), 1024),
                       ("... alternating", mixed(0), 1024), ("... halves of the batch", mixed(7), 1024)):
    cl = torch.full((N,), ctx, dtype=torch.int32, device=dev)
    for Tq in (1, 4):
        qq = q[:, :Tq].contiguous()
        t = timeit(lambda: nv.paged_attention_batched(qq, kc, vc, tab, cl, sc))
        print('%-24s T=%d %.1f us  logical %.2f TB/s' % (name, Tq, t, N * ctx * 4096 / t / 1e6), flush=True)
