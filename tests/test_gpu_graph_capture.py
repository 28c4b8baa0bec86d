"""The decode inner loop under HIP graph capture: every launch decision of the C ABI depends on
host-visible sizes only (never on context_lens or table contents) and the entry points neither
allocate nor synchronise, so a step of L layers (K5 producer -> K2 paged decode) can be captured
once and replayed while the device-side metadata advances.  Replays must reproduce the eager run
bit for bit."""
import math

import pytest
import torch

import nano_vllm_rs_amd as nv

pytestmark = pytest.mark.gpu


def _step(layers, q, k, v, w, rot, pos, slots, bt, cl, scale, ws):
    outs = []
    for (kc, vc), qi, ki, vi in zip(layers, q, k, v):
        qd = nv.qk_norm_rope_cache(qi, ki, vi, w, w, 1e-6, rot, pos, kc, vc, slots, q_out=torch.empty_like(qi))
        outs.append(nv.paged_attention_batched(qd.unsqueeze(1), kc, vc, bt, cl, scale, workspace=ws))
    return outs


@pytest.mark.parametrize("N,ctx0", [(1, 700), (8, 300)])
def test_decode_steps_replayed_from_a_graph_match_eager(dev, N, ctx0):
    Hq, Hkv, D, bs, L, steps = 32, 8, 128, 16, 3, 5
    max_blocks = (ctx0 + steps + bs - 1) // bs + 1
    nb = N * max_blocks
    scale = 1.0 / math.sqrt(D)
    g = torch.Generator(device="cpu").manual_seed(N)
    rnd = lambda *s: torch.randn(s, generator=g).to(dev, torch.bfloat16)
    rot = nv.RotaryEmbedding(D, 2048, 1e6, torch.bfloat16, dev)
    w = torch.ones(D, dtype=torch.bfloat16, device=dev)
    bt = torch.randperm(nb, generator=g).to(torch.int32).reshape(N, max_blocks).to(dev)
    need = nv.capi.lib().nvllm_paged_attn_workspace_bytes(N, 1, Hq, Hkv, D, bs, max_blocks)
    ws = torch.empty(max(int(need), 16), dtype=torch.uint8, device=dev)
    # per-step inputs (a model would produce them inside the graph; here they are staged per step)
    qs = [[rnd(N, Hq, D) for _ in range(L)] for _ in range(steps)]
    ks = [[rnd(N, Hkv, D) for _ in range(L)] for _ in range(steps)]
    vs = [[rnd(N, Hkv, D) for _ in range(L)] for _ in range(steps)]
    base = [(rnd(nb, bs, Hkv, D), rnd(nb, bs, Hkv, D)) for _ in range(L)]

    def metadata(step):
        ctx = torch.full((N,), ctx0 + step + 1, dtype=torch.int32)
        pos = ctx - 1
        blk = bt.cpu()[torch.arange(N), (pos // bs).long()].long()
        return ctx.to(dev), pos.to(dev), (blk * bs + (pos % bs).long()).to(dev)

    # eager reference
    layers = [(a.clone(), b.clone()) for a, b in base]
    want = []
    for s in range(steps):
        cl, pos, slots = metadata(s)
        want.append([o.clone() for o in _step(layers, qs[s], ks[s], vs[s], w, rot, pos, slots, bt, cl, scale, ws)])
    torch.cuda.synchronize()

    # captured once, replayed per step with the static buffers refreshed
    layers = [(a.clone(), b.clone()) for a, b in base]
    sq = [torch.empty_like(x) for x in qs[0]]
    sk = [torch.empty_like(x) for x in ks[0]]
    sv = [torch.empty_like(x) for x in vs[0]]
    cl, pos, slots = metadata(0)
    scl, spos, sslots = cl.clone(), pos.clone(), slots.clone()

    def load(s):
        for dst, src in zip(sq + sk + sv, qs[s] + ks[s] + vs[s]):
            dst.copy_(src)
        c, p, sl = metadata(s)
        scl.copy_(c); spos.copy_(p); sslots.copy_(sl)

    load(0)
    snap = [(a.clone(), b.clone()) for a, b in layers]
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):                      # warm-up outside capture (lazy init, caches)
        _step(layers, sq, sk, sv, w, rot, spos, sslots, bt, scl, scale, ws)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    for (a, b), (a0, b0) in zip(layers, snap):         # undo the warm-up's cache writes
        a.copy_(a0); b.copy_(b0)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        outs = _step(layers, sq, sk, sv, w, rot, spos, sslots, bt, scl, scale, ws)
    for (a, b), (a0, b0) in zip(layers, snap):         # capture does not execute; keep it explicit
        a.copy_(a0); b.copy_(b0)
    for s in range(steps):
        load(s)
        graph.replay()
        torch.cuda.synchronize()
        for got, ref in zip(outs, want[s]):
            assert torch.equal(got, ref), (s,)


def test_verify_launch_of_the_mid_kernel_replayed_from_a_graph(dev):
    """Speculative verify (T = 16 new tokens of 48 sequences: the mid kernel, paged_attn_mid.hip)
    captured once and replayed while context_lens advance on the device: the kernel reads lengths
    and tables only on the device, decides which wave computes which rows from hardware registers,
    and uses no workspace -- replays must equal eager launches bit for bit."""
    Hq, Hkv, D, bs, N, T = 32, 8, 128, 16, 48, 16
    max_blocks = 24
    nb = N * max_blocks
    scale = 1.0 / math.sqrt(D)
    g = torch.Generator(device="cpu").manual_seed(5)
    rnd = lambda *s: torch.randn(s, generator=g).to(dev, torch.bfloat16)
    kc, vc = rnd(nb, bs, Hkv, D), rnd(nb, bs, Hkv, D)
    bt = torch.randperm(nb, generator=g).to(torch.int32).reshape(N, max_blocks).to(dev)
    assert nv.paged_attention_path(N, T, Hq, Hkv, D, bs, max_blocks, nb) == "paged_mid"
    q = rnd(N, T, Hq, D)
    lens = [torch.randint(T, max_blocks * bs, (N,), generator=g, dtype=torch.int32).to(dev) for _ in range(4)]
    want = []
    for cl in lens:
        want.append(nv.paged_attention_batched(q, kc, vc, bt, cl, scale).clone())
    torch.cuda.synchronize()
    scl = lens[0].clone()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):                      # warm-up outside capture
        nv.paged_attention_batched(q, kc, vc, bt, scl, scale)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):                      # (the output lives in the graph's memory pool)
        out = nv.paged_attention_batched(q, kc, vc, bt, scl, scale)
    for cl, w in zip(lens, want):
        scl.copy_(cl)
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, w)
