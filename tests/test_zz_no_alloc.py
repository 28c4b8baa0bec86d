"""nvllm_paged_attn_bf16_noalloc: the paged-attention entry point that allocates nothing (split-K
scratch in a device array built into the library) against nvllm_paged_attn_bf16 with a caller
workspace -- same kernels, so the outputs must be bit-identical -- on two calls in a row, the second
with a smaller input (the scratch keeps the first call's contents), then a shape that needs no
scratch and one that needs more than the array holds."""
import ctypes
import math

import numpy as np
import pytest
import torch

import nano_vllm_rs_amd as nv
from helpers import bf16_randn, make_paged_case, oracle_paged_batch, assert_attention_close, pad_tables

pytestmark = pytest.mark.gpu

Hq, Hkv, D, bs = 32, 8, 128, 16
SCALE = 1.0 / math.sqrt(D)


def _case(dev, seed, ctx_lens, T, pool):
    rng = np.random.default_rng(seed)
    kc, vc, tables = make_paged_case(rng, ctx_lens, pool, bs, Hkv, D)
    q = bf16_randn(rng, (len(ctx_lens), T, Hq, D))
    to = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev, torch.bfloat16)
    return dict(q=to(q), kc=to(kc), vc=to(vc), bt=torch.from_numpy(pad_tables(tables)).to(dev),
                cl=torch.tensor(ctx_lens, dtype=torch.int32, device=dev), host=(q, kc, vc, tables, ctx_lens))


def _noalloc(c, out):
    N, T = c["q"].shape[0], c["q"].shape[1]
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    return nv.capi.lib().nvllm_paged_attn_bf16_noalloc(
        p(c["q"]), p(c["kc"]), p(c["vc"]), p(c["bt"]), p(c["cl"]), p(out), N, T, Hq, Hkv, D, bs,
        c["bt"].shape[1], c["kc"].shape[0], SCALE, st)


def _need(c):
    return nv.capi.lib().nvllm_paged_attn_workspace_bytes(c["q"].shape[0], c["q"].shape[1], Hq, Hkv, D, bs,
                                                          c["bt"].shape[1])


def test_noalloc_equals_the_workspace_entry_on_two_calls_in_a_row(dev):
    big = _case(dev, 1, [2048, 1999, 1500, 2048, 77, 1024, 2040, 1111], 1, 1100)       # 8 seqs: split-K
    small = _case(dev, 2, [700, 512, 33], 4, 140)                                      # fewer, T = 4
    assert _need(big) > 0 and _need(small) > 0
    outs = []
    for c in (big, small):            # back to back on one stream, the scratch is not cleared between
        out = torch.full((c["q"].shape[0], c["q"].shape[1], Hq * D), float("nan"), dtype=torch.bfloat16, device=dev)
        before = torch.cuda.memory_allocated(dev)
        assert _noalloc(c, out) == 0, nv.capi.last_error()
        assert torch.cuda.memory_allocated(dev) == before
        outs.append(out)
    torch.cuda.synchronize()
    for c, out in zip((big, small), outs):
        want = nv.paged_attention_batched(c["q"], c["kc"], c["vc"], c["bt"], c["cl"], SCALE)
        torch.cuda.synchronize()
        assert torch.equal(out, want)
        q, kc, vc, tables, ctx_lens = c["host"]
        assert_attention_close(out.float().cpu().numpy(), oracle_paged_batch(q, kc, vc, tables, ctx_lens, bs, SCALE))


def test_noalloc_through_the_prefill_kernels_key_split(dev):
    """T = 16 new tokens of ONE sequence against 4096 / 1500 cached keys: 64 query rows per kv head go
    to the prefill kernel over the cache, which splits the key walk of so small a launch (parts > 1)
    and leaves its partials -- bf16 O, f32 (m, l), another slot layout than the decode kernel's
    split-K -- in the same built-in scratch.  Two calls in a row, the second smaller (fewer key
    tiles: some parts of some rows carry l = 0 over the first call's leftovers)."""
    T = 16
    big = _case(dev, 11, [4096], T, 300)
    small = _case(dev, 12, [1500], T, 120)
    for c in (big, small):
        assert nv.paged_attention_path(1, T, Hq, Hkv, D, bs, c["bt"].shape[1], c["kc"].shape[0]) == "k1_paged"
        assert _need(c) > 0
    outs = []
    for c in (big, small):
        out = torch.full((1, T, Hq * D), float("nan"), dtype=torch.bfloat16, device=dev)
        before = torch.cuda.memory_allocated(dev)
        assert _noalloc(c, out) == 0, nv.capi.last_error()
        assert torch.cuda.memory_allocated(dev) == before
        outs.append(out)
    torch.cuda.synchronize()
    for c, out in zip((big, small), outs):
        want = nv.paged_attention_batched(c["q"], c["kc"], c["vc"], c["bt"], c["cl"], SCALE)
        torch.cuda.synchronize()
        assert torch.equal(out, want)
        q, kc, vc, tables, ctx_lens = c["host"]
        assert_attention_close(out.float().cpu().numpy(), oracle_paged_batch(q, kc, vc, tables, ctx_lens, bs, SCALE))


def test_noalloc_without_scratch_and_beyond_the_limit(dev):
    L = nv.capi.lib()
    # a shape that needs no workspace: identical too (the scratch is never touched)
    c = _case(dev, 3, [40 + (i % 30) for i in range(256)], 1, 1400)
    assert _need(c) == 0
    out = torch.empty((256, 1, Hq * D), dtype=torch.bfloat16, device=dev)
    assert _noalloc(c, out) == 0, nv.capi.last_error()
    want = nv.paged_attention_batched(c["q"], c["kc"], c["vc"], c["bt"], c["cl"], SCALE)
    torch.cuda.synchronize()
    assert torch.equal(out, want)
    # a launch the mid kernel takes (48 sequences x 8 kv heads = 1.5 units per CU, 64 rows each): no
    # scratch either, same kernel through both entries
    m = _case(dev, 4, [100 + 13 * i for i in range(48)], 16, 2400)
    assert nv.paged_attention_path(48, 16, Hq, Hkv, D, bs, m["bt"].shape[1], m["kc"].shape[0]) == "paged_mid"
    out = torch.full((48, 16, Hq * D), float("nan"), dtype=torch.bfloat16, device=dev)
    before = torch.cuda.memory_allocated(dev)
    assert _noalloc(m, out) == 0, nv.capi.last_error()
    assert torch.cuda.memory_allocated(dev) == before
    want = nv.paged_attention_batched(m["q"], m["kc"], m["vc"], m["bt"], m["cl"], SCALE)
    torch.cuda.synchronize()
    assert torch.equal(out, want)
    # more split-K scratch than the built-in array holds: refused, nothing launched
    limit = L.nvllm_paged_attn_noalloc_max_scratch_bytes()
    assert limit == 32 << 20
    assert L.nvllm_paged_attn_workspace_bytes(16, 12, Hq, Hkv, D, bs, 2048) > limit
    fake = ctypes.c_void_p(0x1000)
    rc = L.nvllm_paged_attn_bf16_noalloc(fake, fake, fake, fake, fake, fake, 16, 12, Hq, Hkv, D, bs, 2048,
                                         1 << 20, SCALE, None)
    assert rc == nv.capi.NVLLM_ERR_UNSUPPORTED and "built-in scratch holds" in nv.capi.last_error()
