"""Chunked-prefill throughput of K1 over the paged cache (TFLOP/s, algorithmic causal FLOPs),
next to the dense kernel on the same shape where the chunk is the whole sequence."""
import json, math, sys
import torch
sys.path.insert(0, '.')
import nano_vllm_rs_amd as nv

dev = torch.device('cuda:0')
Hq, Hkv, D, bs = 32, 8, 128, 16


def timeit(fn, iters=20, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e-3


rows = []
for N, T, ctx in [(1, 4096, 4096), (8, 4096, 4096), (1, 512, 4096), (8, 512, 4096), (8, 512, 4096 + 13),
                  (4, 2048, 8192), (32, 128, 2048), (64, 64, 1024), (16, 1024, 16384)]:
    nblk = -(-ctx // bs)
    nb = N * nblk
    g = torch.Generator(device='cpu').manual_seed(1)
    kc = torch.randn((nb, bs, Hkv, D), device=dev, dtype=torch.bfloat16)
    vc = torch.randn((nb, bs, Hkv, D), device=dev, dtype=torch.bfloat16)
    q = torch.randn((N, T, Hq, D), device=dev, dtype=torch.bfloat16)
    bt = torch.randperm(nb, generator=g).to(torch.int32).reshape(N, nblk).to(dev)
    cl = torch.full((N,), ctx, dtype=torch.int32, device=dev)
    scale = 1 / math.sqrt(D)
    t = timeit(lambda: nv.paged_attention_batched(q, kc, vc, bt, cl, scale))
    flops = 4.0 * N * Hq * D * (T * (ctx - T) + T * (T + 1) / 2)
    row = dict(num_seqs=N, T=T, ctx=ctx, ms=t * 1e3, tflops=flops / t / 1e12)
    if T == ctx:
        k = torch.randn((N, ctx, Hkv, D), device=dev, dtype=torch.bfloat16)
        v = torch.randn((N, ctx, Hkv, D), device=dev, dtype=torch.bfloat16)
        cfg = nv.FlashAttentionConfig(D, True)
        td = timeit(lambda: nv.flash_attention(q, k, v, cfg))
        row['dense_ms'] = td * 1e3
        row['dense_tflops'] = flops / td / 1e12
    rows.append(row)
    print(json.dumps(row), flush=True)
