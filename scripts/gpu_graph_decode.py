"""Decode step of L layers (K5 producer -> K2 paged decode per layer), launched eagerly through the
ctypes harness and replayed from one HIP graph: what the launch path costs at small batch."""
import json, math, sys
import torch
sys.path.insert(0, '.')
import nano_vllm_rs_amd as nv

dev = torch.device('cuda:0')
Hq, Hkv, D, bs, L = 32, 8, 128, 16, 36
scale = 1 / math.sqrt(D)
rot = nv.RotaryEmbedding(D, 40960, 1e6, torch.bfloat16, dev)
w = torch.ones(D, dtype=torch.bfloat16, device=dev)
for N, ctx in ((1, 2048), (1, 8192), (8, 4096), (64, 2048)):
    max_blocks = ctx // bs + 1
    nb = N * max_blocks
    layers = [(torch.randn((nb, bs, Hkv, D), device=dev, dtype=torch.bfloat16),
               torch.randn((nb, bs, Hkv, D), device=dev, dtype=torch.bfloat16)) for _ in range(L)]
    q = [torch.randn((N, Hq, D), device=dev, dtype=torch.bfloat16) for _ in range(L)]
    k = [torch.randn((N, Hkv, D), device=dev, dtype=torch.bfloat16) for _ in range(L)]
    v = [torch.randn((N, Hkv, D), device=dev, dtype=torch.bfloat16) for _ in range(L)]
    bt = torch.randperm(nb).to(torch.int32).reshape(N, max_blocks).to(dev)
    cl = torch.full((N,), ctx, dtype=torch.int32, device=dev)
    pos = cl - 1
    slots = (bt[torch.arange(N, device=dev), (pos // bs).long()].long() * bs + (pos % bs).long())
    need = nv.capi.lib().nvllm_paged_attn_workspace_bytes(N, 1, Hq, Hkv, D, bs, max_blocks)
    ws = torch.empty(max(int(need), 16), dtype=torch.uint8, device=dev)
    qo = [torch.empty_like(x) for x in q]

    def step():
        outs = []
        for (kc, vc), qi, ki, vi, qoi in zip(layers, q, k, v, qo):
            qd = nv.qk_norm_rope_cache(qi, ki, vi, w, w, 1e-6, rot, pos, kc, vc, slots, q_out=qoi)
            outs.append(nv.paged_attention_batched(qd.unsqueeze(1), kc, vc, bt, cl, scale, workspace=ws))
        return outs

    def timeit(fn, iters=20):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(iters): fn()
        b.record(); torch.cuda.synchronize()
        return a.elapsed_time(b) / iters * 1e3
    eager = timeit(step)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        step()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        step()
    graph = timeit(g.replay)
    print(json.dumps(dict(num_seqs=N, ctx=ctx, layers=L, eager_us_per_step=round(eager, 1),
                          graph_us_per_step=round(graph, 1), graph_us_per_layer=round(graph / L, 2))), flush=True)
