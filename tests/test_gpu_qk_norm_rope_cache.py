"""K5 parity: fused per-head RMSNorm + RoPE + paged KV write vs the CPU oracle.

Oracle: oracle.qk_norm_rope (restates src/model/norm.rs:83-100 and src/model/rope.rs:107-152 as
Qwen3Attention::forward chains them, src/model/attention.rs:229-234) followed by
oracle.write_kv_to_cache (paged.rs:208-320), on the SAME bf16-rounded inputs (weights and cos/sin
tables included, the reference holds them in the model dtype).  Bars:
  * f32: 1e-5 max-abs (the bound the reference's own norm test uses, model_norm_test.rs:68-71);
  * bf16: the kernel rounds once from f32, so every output must be within one bf16 ulp of the
    oracle's f32 value (|diff| <= 2^-7 |want| + 1e-6);
  * V in the cache, untouched slots and guard blocks: bit-exact.
RoPE numerics are unpinned by the reference's tests (qualitative only); see
tests/test_norm_rope_oracle.py for the pins that do exist.
"""
import math

import numpy as np
import pytest
import torch

import nano_vllm_rs_amd as nv
import oracle
from helpers import assert_attention_close, bf16_randn

pytestmark = pytest.mark.gpu


def _case(rng, dev, dtype, n, Hq, Hkv, D, nb, bs, max_pos, theta=1e6, spread_pos=True):
    rnd = bf16_randn if dtype == torch.bfloat16 else (lambda r, s: r.standard_normal(s, dtype=np.float32))
    q, k, v = rnd(rng, (n, Hq, D)), rnd(rng, (n, Hkv, D)), rnd(rng, (n, Hkv, D))
    qw = 1.0 + 0.25 * rnd(rng, (D,))
    kw = 1.0 + 0.25 * rnd(rng, (D,))
    if dtype == torch.bfloat16:
        qw, kw = oracle.bf16_round(qw), oracle.bf16_round(kw)
    pos = rng.integers(0, max_pos, size=n).astype(np.int32) if spread_pos else np.arange(n, dtype=np.int32)
    slots = rng.permutation(nb * bs)[:n].astype(np.int64)
    rot = nv.RotaryEmbedding(D, max_pos, theta, dtype, dev)
    cos, sin = rot.cos_cache().float().cpu().numpy(), rot.sin_cache().float().cpu().numpy()
    return q, k, v, qw, kw, pos, slots, rot, cos, sin


def _oracle(q, k, v, qw, kw, eps, cos, sin, pos, slots, kc0, vc0, bs):
    n = q.shape[0]
    qo, ko = np.empty_like(q), np.empty_like(k)
    for i in range(n):     # one token at a time = apply(.., start_pos = pos[i]) on a length-1 sequence
        a, b = oracle.qk_norm_rope(q[None, i:i + 1], k[None, i:i + 1], qw, kw, eps, cos, sin, int(pos[i]))
        qo[i], ko[i] = a[0, 0], b[0, 0]
    kc, vc = kc0.copy(), vc0.copy()
    live = [i for i in range(n) if 0 <= slots[i] < kc.shape[0] * bs]
    if live:
        kc, vc = oracle.write_kv_to_cache(ko[None, live], v[None, live], kc, vc, slots[live], bs)
    return qo, ko, kc, vc


def _close_bf16(got, want):
    got, want = np.asarray(got, np.float32), np.asarray(want, np.float32)
    assert np.isfinite(got).all()
    bad = np.abs(got - want) > (2.0 ** -7) * np.abs(want) + 1e-6
    assert not bad.any(), "max |diff| %.3e at %s" % (np.abs(got - want).max(), np.argwhere(bad)[:3].tolist())


def _run(dev, dtype, q, k, v, qw, kw, eps, rot, pos, slots, nb, bs, guard=2, k_out=True, in_place=True):
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device=dev, dtype=dtype)
    Hkv, D = k.shape[1], k.shape[2]
    sent = 768.0
    kc = torch.full((nb + 2 * guard, bs, Hkv, D), sent, dtype=dtype, device=dev)
    vc = torch.full((nb + 2 * guard, bs, Hkv, D), sent, dtype=dtype, device=dev)
    kcv, vcv = kc[guard:guard + nb], vc[guard:guard + nb]
    qd, kd, vd = d(q), d(k), d(v)
    ko = torch.full_like(kd, sent) if k_out else None
    qo = None if in_place else torch.empty_like(qd)
    ret = nv.qk_norm_rope_cache(qd, kd, vd, d(qw), d(kw), eps, rot, torch.from_numpy(pos).to(dev),
                                kcv, vcv, torch.from_numpy(slots).to(dev), k_out=ko, q_out=qo)
    torch.cuda.synchronize()
    for g in (kc[:guard], kc[guard + nb:], vc[:guard], vc[guard + nb:]):
        assert bool((g == sent).all()), "write outside the cache"
    assert torch.equal(kd, d(k)) and torch.equal(vd, d(v)), "K/V inputs must not change"
    if not in_place:
        assert torch.equal(qd, d(q))
    f = lambda t: t.float().cpu().numpy()
    return f(ret), (f(ko) if k_out else None), f(kcv), f(vcv)


@pytest.mark.parametrize("n,Hq,Hkv", [(1, 32, 8), (7, 32, 8), (256, 32, 8), (333, 16, 8), (50, 8, 8), (9, 4, 1)])
def test_bf16_fast_path(dev, n, Hq, Hkv):
    D, nb, bs, max_pos, eps = 128, 40, 16, 5000, 1e-6
    rng = np.random.default_rng(n * 100 + Hq)
    q, k, v, qw, kw, pos, slots, rot, cos, sin = _case(rng, dev, torch.bfloat16, n, Hq, Hkv, D, nb, bs, max_pos)
    sent = np.full((nb, bs, Hkv, D), 768.0, np.float32)
    wq, wk, wkc, wvc = _oracle(q, k, v, qw, kw, eps, cos, sin, pos, slots, sent, sent, bs)
    gq, gk, gkc, gvc = _run(dev, torch.bfloat16, q, k, v, qw, kw, eps, rot, pos, slots, nb, bs)
    _close_bf16(gq, wq)
    _close_bf16(gk, wk)
    _close_bf16(gkc, wkc)
    assert np.array_equal(gvc, wvc)                       # V is a copy: bit-exact
    untouched = np.ones(nb * bs, bool)
    untouched[slots] = False
    assert (gkc.reshape(nb * bs, -1)[untouched] == 768.0).all()
    # the cache row and k_out of a token are the same rounding of the same value
    assert np.array_equal(gkc.reshape(nb * bs, Hkv, D)[slots], gk)


def test_bf16_fast_path_streaming_launch(dev):
    """>= 2048 tokens: the launch whose outputs leave as non-temporal stores; same bars."""
    n, Hq, Hkv, D, nb, bs, max_pos, eps = 2100, 8, 2, 128, 140, 16, 5000, 1e-6
    rng = np.random.default_rng(77)
    q, k, v, qw, kw, pos, slots, rot, cos, sin = _case(rng, dev, torch.bfloat16, n, Hq, Hkv, D, nb, bs, max_pos)
    sent = np.full((nb, bs, Hkv, D), 768.0, np.float32)
    wq, wk, wkc, wvc = _oracle(q, k, v, qw, kw, eps, cos, sin, pos, slots, sent, sent, bs)
    gq, gk, gkc, gvc = _run(dev, torch.bfloat16, q, k, v, qw, kw, eps, rot, pos, slots, nb, bs)
    _close_bf16(gq, wq)
    _close_bf16(gk, wk)
    _close_bf16(gkc, wkc)
    assert np.array_equal(gvc, wvc)
    untouched = np.ones(nb * bs, bool)
    untouched[slots] = False
    assert (gkc.reshape(nb * bs, -1)[untouched] == 768.0).all()
    assert (gvc.reshape(nb * bs, -1)[untouched] == 768.0).all()


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("D", [4, 32, 64, 96, 128, 256])
def test_generic_path(dev, dtype, D):
    if dtype == torch.bfloat16 and D == 128:
        pytest.skip("fast path, covered above")
    n, Hq, Hkv, nb, bs, max_pos, eps = 37, 6, 2, 12, 4, 300, 1e-5
    rng = np.random.default_rng(D)
    q, k, v, qw, kw, pos, slots, rot, cos, sin = _case(rng, dev, dtype, n, Hq, Hkv, D, nb, bs, max_pos,
                                                       theta=10000.0)
    sent = np.full((nb, bs, Hkv, D), 768.0, np.float32)
    wq, wk, wkc, wvc = _oracle(q, k, v, qw, kw, eps, cos, sin, pos, slots, sent, sent, bs)
    gq, gk, gkc, gvc = _run(dev, dtype, q, k, v, qw, kw, eps, rot, pos, slots, nb, bs, in_place=False)
    if dtype == torch.float32:
        for g, w in ((gq, wq), (gk, wk), (gkc, wkc)):
            assert np.abs(g - w).max() < 1e-5
    else:
        _close_bf16(gq, wq); _close_bf16(gk, wk); _close_bf16(gkc, wkc)
    assert np.array_equal(gvc, wvc)


@pytest.mark.parametrize("dtype,D", [(torch.bfloat16, 128), (torch.bfloat16, 64), (torch.float32, 128), (torch.float32, 32)])
def test_in_place_equals_out_of_place(dev, dtype, D):
    """q_out may be q itself (the default): both K5 kernels must give the same bits either way
    (the fast path and the generic one)."""
    n, Hq, Hkv, nb, bs, max_pos, eps = 45, 8, 2, 12, 4, 300, 1e-6
    rng = np.random.default_rng(1000 + D)
    q, k, v, qw, kw, pos, slots, rot, cos, sin = _case(rng, dev, dtype, n, Hq, Hkv, D, nb, bs, max_pos)
    a = _run(dev, dtype, q, k, v, qw, kw, eps, rot, pos, slots, nb, bs, in_place=True)
    b = _run(dev, dtype, q, k, v, qw, kw, eps, rot, pos, slots, nb, bs, in_place=False)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)


def test_reference_known_answer_through_the_kernel(dev):
    # model_norm_test.rs:47-72 ([1,2,3,4] -> x / rms, < 1e-5) at position 0, where RoPE is the
    # identity (cos 1, sin 0); then position 5 must change the values (model_rope_test.rs:72-99)
    D, eps = 4, 1e-6
    rot = nv.RotaryEmbedding(D, 1024, 10000.0, torch.float32, dev)
    x = torch.tensor([[[1.0, 2.0, 3.0, 4.0]]], device=dev)
    w = torch.ones(D, device=dev)
    kc = torch.zeros((2, 4, 1, D), device=dev)
    vc = torch.zeros((2, 4, 1, D), device=dev)
    slot = torch.tensor([5], dtype=torch.int64, device=dev)
    out0 = nv.qk_norm_rope_cache(x.clone(), x.clone(), x.clone(), w, w, eps, rot,
                                 torch.tensor([0], dtype=torch.int32, device=dev), kc, vc, slot)
    rms = math.sqrt(7.5 + 1e-6)
    assert np.abs(out0.cpu().numpy().ravel() - np.array([1, 2, 3, 4]) / rms).max() < 1e-5
    assert np.abs(kc[1, 1, 0].cpu().numpy() - np.array([1, 2, 3, 4]) / rms).max() < 1e-5
    assert torch.equal(vc[1, 1, 0], x[0, 0])
    out5 = nv.qk_norm_rope_cache(x.clone(), x.clone(), x.clone(), w, w, eps, rot,
                                 torch.tensor([5], dtype=torch.int32, device=dev), kc, vc, slot)
    assert not torch.equal(out5, out0)


def test_skipped_slots_and_clamped_positions(dev):
    n, Hq, Hkv, D, nb, bs, max_pos, eps = 20, 32, 8, 128, 4, 16, 64, 1e-6
    rng = np.random.default_rng(2)
    q, k, v, qw, kw, pos, slots, rot, cos, sin = _case(rng, dev, torch.bfloat16, n, Hq, Hkv, D, nb, bs, max_pos)
    slots[3], slots[7], slots[11] = -1, nb * bs, nb * bs + 100          # skip marker / out of pool
    pos[0], pos[1] = -5, max_pos + 9                                      # clamped to the table's ends
    gq, gk, gkc, gvc = _run(dev, torch.bfloat16, q, k, v, qw, kw, eps, rot, pos, slots, nb, bs)
    sent = np.full((nb, bs, Hkv, D), 768.0, np.float32)
    wq, wk, wkc, wvc = _oracle(q, k, v, qw, kw, eps, cos, sin, np.clip(pos, 0, max_pos - 1), slots,
                               sent, sent, bs)
    _close_bf16(gq, wq); _close_bf16(gk, wk); _close_bf16(gkc, wkc)
    assert np.array_equal(gvc, wvc)


def test_error_behaviour(dev):
    rot = nv.RotaryEmbedding(4, 8, 10000.0, torch.float32, dev)
    with pytest.raises(nv.ConfigError):                                   # rope.rs:49
        nv.RotaryEmbedding(5, 8, 10000.0, torch.float32, dev)
    x = torch.zeros((1, 1, 6), device=dev)
    kc = torch.zeros((1, 4, 1, 6), device=dev)
    i32, i64 = torch.zeros(1, dtype=torch.int32, device=dev), torch.zeros(1, dtype=torch.int64, device=dev)
    with pytest.raises(AssertionError):                                   # table width != head_dim
        nv.qk_norm_rope_cache(x, x, x, x[0, 0], x[0, 0], 1e-6, rot, i32, kc, kc, i64)


def test_producer_then_paged_decode_end_to_end(dev):
    """One decode step of a layer as the serving loop runs it: BatchMetadata -> K5 -> K2, against
    the oracle chain norm -> rope -> write_kv_to_cache -> paged_attention per sequence."""
    Hq, Hkv, D, bs, nb, eps, max_pos = 32, 8, 128, 16, 64, 1e-6, 4096
    rng = np.random.default_rng(77)
    mgr = nv.BlockManager(nb, bs)
    ctxs = [5, 16, 17, 100, 33]
    tables = []
    for c in ctxs:
        t = nv.BlockTable(bs)
        for b in mgr.allocate_many(nv.compute_num_blocks(c, bs)):
            t.append_block(b)
        tables.append(t)
    rot = nv.RotaryEmbedding(D, max_pos, 1e6, torch.bfloat16, dev)
    cos, sin = rot.cos_cache().float().cpu().numpy(), rot.sin_cache().float().cpu().numpy()
    qw, kw = oracle.bf16_round(1 + 0.1 * rng.standard_normal(D).astype(np.float32)), np.ones(D, np.float32)
    layer = nv.LayerKVCache(nv.KVCacheConfig(nb, bs, Hkv, D, 1, torch.bfloat16), dev)
    okc = np.zeros((nb, bs, Hkv, D), np.float32)
    ovc = np.zeros((nb, bs, Hkv, D), np.float32)
    d = lambda a, dt=torch.bfloat16: torch.from_numpy(np.ascontiguousarray(a)).to(device=dev, dtype=dt)
    # prefill every sequence's first ctx-1 tokens, then one decode token each: two K5 launches
    for phase in ("prefill", "decode"):
        batch = nv.BatchMetadata(8)
        for t, c in zip(tables, ctxs):
            batch.add_sequence(t, c - 1 if phase == "prefill" else c, c - 1 if phase == "prefill" else 1)
        n = batch.num_tokens()
        q, k, v = bf16_randn(rng, (n, Hq, D)), bf16_randn(rng, (n, Hkv, D)), bf16_randn(rng, (n, Hkv, D))
        bt, cl, sm, qs, ps = batch.upload(dev)
        qd = nv.qk_norm_rope_cache(d(q), d(k), d(v), d(qw), d(kw), eps, rot, ps, layer.key_cache(),
                                   layer.value_cache(), sm)
        wq, wk, okc, ovc = _oracle(q, k, v, qw, kw, eps, cos, sin, batch.positions(),
                                   batch.slot_mapping(), okc, ovc, bs)
        okc = oracle.bf16_round(okc)      # the cache holds bf16
    _close_bf16(layer.key_cache().float().cpu().numpy(), okc)
    scale = 1 / math.sqrt(D)
    out = nv.paged_attention_batched(qd.reshape(len(ctxs), 1, Hq, D), layer.key_cache(),
                                     layer.value_cache(), bt, cl, scale).float().cpu().numpy()
    wqb = oracle.bf16_round(wq)
    for i, (t, c) in enumerate(zip(tables, ctxs)):
        want = oracle.paged_attention(wqb[None, i:i + 1], okc, ovc, t.get_physical_block_ids(), bs, c,
                                      Hq, Hkv, D, scale)
        assert_attention_close(out[i:i + 1], want)
