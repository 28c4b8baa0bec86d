"""K5 (fused per-head RMSNorm + RoPE + KV write) achieved HBM bandwidth, algorithmic bytes."""
import json, sys
import torch
sys.path.insert(0, '.')
import nano_vllm_rs_amd as nv

dev = torch.device('cuda:0')
Hq, Hkv, D, bs = 32, 8, 128, 16
rot = nv.RotaryEmbedding(D, 40960, 1e6, torch.bfloat16, dev)
for n in (256, 4096, 32768, 262144):
    nb = max(n // bs * 2, 64)
    NBUF = 4 if n < 200000 else 2          # rotate buffers so the caches do not serve re-reads
    bufs = []
    for _ in range(NBUF):
        q = torch.randn((n, Hq, D), device=dev, dtype=torch.bfloat16)
        k = torch.randn((n, Hkv, D), device=dev, dtype=torch.bfloat16)
        v = torch.randn((n, Hkv, D), device=dev, dtype=torch.bfloat16)
        kc = torch.zeros((nb, bs, Hkv, D), device=dev, dtype=torch.bfloat16)
        vc = torch.zeros((nb, bs, Hkv, D), device=dev, dtype=torch.bfloat16)
        bufs.append((q, k, v, kc, vc))
    w = torch.ones(D, device=dev, dtype=torch.bfloat16)
    pos = torch.randint(0, 40960, (n,), device=dev, dtype=torch.int32)
    slots = torch.randperm(nb * bs, device=dev)[:n].to(torch.int64)
    def step(i):
        q, k, v, kc, vc = bufs[i % NBUF]
        nv.qk_norm_rope_cache(q, k, v, w, w, 1e-6, rot, pos, kc, vc, slots)
    for i in range(8): step(i)
    torch.cuda.synchronize()
    iters = 40
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(iters): step(i)
    b.record(); torch.cuda.synchronize()
    t = a.elapsed_time(b) / iters * 1e-3
    by = n * ((2 * Hq + 4 * Hkv) * D * 2 + 2 * D * 2 + 12)
    print(json.dumps(dict(tokens=n, us=t * 1e6, bytes=by, GBps=by / t / 1e9, frac_of_8TBps=by / t / 8e12)), flush=True)
