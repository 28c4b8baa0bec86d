"""K4 (reshape_and_cache) achieved HBM bandwidth: random and block-contiguous slots."""
import json, sys
import torch
sys.path.insert(0, '.')
import nano_vllm_rs_amd as nv

dev = torch.device('cuda:0')
Hkv, D, bs = 8, 128, 16
for n in (256, 4096, 32768, 262144):
    nb = max(n // bs * 2, 4096)
    NBUF = 4
    bufs = [(torch.randn((n, Hkv, D), device=dev, dtype=torch.bfloat16),
             torch.randn((n, Hkv, D), device=dev, dtype=torch.bfloat16),
             torch.zeros((nb, bs, Hkv, D), device=dev, dtype=torch.bfloat16),
             torch.zeros((nb, bs, Hkv, D), device=dev, dtype=torch.bfloat16)) for _ in range(NBUF)]
    for kind in ("random", "blocks"):
        if kind == "random":
            slots = torch.randperm(nb * bs, device=dev)[:n].to(torch.int64)
        else:   # whole blocks in random order, tokens consecutive inside a block (a prefill)
            blk = torch.randperm(nb, device=dev)[:(n + bs - 1) // bs].to(torch.int64)
            slots = (blk[:, None] * bs + torch.arange(bs, device=dev)[None, :]).reshape(-1)[:n].contiguous()
        def step(i):
            k, v, kc, vc = bufs[i % NBUF]
            nv.reshape_and_cache(k, v, kc, vc, slots)
        for i in range(8): step(i)
        torch.cuda.synchronize()
        iters = 40
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for i in range(iters): step(i)
        b.record(); torch.cuda.synchronize()
        t = a.elapsed_time(b) / iters * 1e-3
        by = n * (2 * Hkv * D * 2 * 2 + 8)
        print(json.dumps(dict(tokens=n, slots=kind, us=round(t * 1e6, 2), GBps=round(by / t / 1e9, 1),
                              frac_of_8TBps=round(by / t / 8e12, 4))), flush=True)
