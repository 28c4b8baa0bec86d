"""K2/K2b/K3 parity: paged attention vs the CPU oracle.

Oracle: oracle.paged_attention (restates reference src/attention/paged.rs:117-192),
fed the SAME bf16-rounded inputs in f32.  Tolerance (north_star): 2e-2 max-abs and
1e-3 mean-abs.  The reference's own paged tests assert shapes only (all-zero
caches), so numeric parity here is pinned by the oracle, not by reference vectors.
Dead cache slots are poisoned with NaN: any read of them into the result fails.
"""
import math

import numpy as np
import pytest
import torch

import nano_vllm_rs_amd as nv
import oracle
from helpers import (assert_attention_close, bf16_randn, make_paged_case, oracle_paged_batch,
                     pad_tables)

pytestmark = pytest.mark.gpu


def _dev(a, dev, dtype=torch.bfloat16):
    return torch.from_numpy(np.ascontiguousarray(a)).to(device=dev, dtype=dtype)


def _run_batched(dev, q, kc, vc, tables, ctx_lens, scale, dtype=torch.bfloat16, max_blocks=None):
    bt = torch.from_numpy(pad_tables(tables, max_blocks)).to(dev)
    cl = torch.tensor(ctx_lens, dtype=torch.int32, device=dev)
    out = nv.paged_attention_batched(_dev(q, dev, dtype), _dev(kc, dev, dtype), _dev(vc, dev, dtype),
                                     bt, cl, scale)
    torch.cuda.synchronize()
    return out.float().cpu().numpy()


# ---- the reference's own paged tests (shape-only there), now with numeric parity ----------
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("table,ctx,T", [
    ([0, 1], 6, 1),          # tests/paged_attention_test.rs:111-149
    ([0, 1], 8, 2),          # :152-185 (decode causal mask)
    ([0, 2, 5], 10, 1),      # :266-301 (non-contiguous)
    ([0, 1, 2], 12, 4),      # :304-349 (chunked prefill)
])
def test_reference_shapes_generic_kernel(dev, dtype, table, ctx, T):
    nb, bs, Hkv, D, Hq = 8, 4, 2, 8, 4
    rng = np.random.default_rng(ctx * 10 + T)
    kc, vc = bf16_randn(rng, (nb, bs, Hkv, D)), bf16_randn(rng, (nb, bs, Hkv, D))
    q = bf16_randn(rng, (1, T, Hq, D))
    scale = 1.0 / math.sqrt(D)
    layer = nv.LayerKVCache(nv.KVCacheConfig(nb, bs, Hkv, D, 1, dtype), dev)
    layer.key_cache().copy_(_dev(kc, dev, dtype))
    layer.value_cache().copy_(_dev(vc, dev, dtype))
    bt = nv.BlockTable(bs)
    for b in table:
        bt.append_block(b)
    out = nv.paged_attention(_dev(q, dev, dtype), layer, bt, ctx, Hq, Hkv, D, scale)
    assert tuple(out.shape) == (1, T, Hq * D)
    want = oracle.paged_attention(q, kc, vc, table, bs, ctx, Hq, Hkv, D, scale)
    if dtype == torch.float32:
        assert np.abs(out.cpu().numpy() - want).max() < 1e-5
    else:
        assert_attention_close(out.float().cpu().numpy(), want)


def test_zero_cache_gives_zero_output(dev):
    # what the reference tests actually run: an all-zero cache
    layer = nv.LayerKVCache(nv.KVCacheConfig(4, 4, 2, 8, 1, torch.float32), dev)
    bt = nv.BlockTable(4)
    bt.append_block(0); bt.append_block(1)
    q = torch.randn((1, 1, 4, 8), device=dev)
    out = nv.paged_attention(q, layer, bt, 6, 4, 2, 8, 1 / math.sqrt(8))
    assert tuple(out.shape) == (1, 1, 32) and float(out.abs().max()) == 0.0


def test_error_behaviour_matches_reference(dev):
    layer = nv.LayerKVCache(nv.KVCacheConfig(4, 4, 2, 8, 1, torch.float32), dev)
    q = torch.randn((1, 1, 4, 8), device=dev)
    empty = nv.BlockTable(4)
    with pytest.raises(nv.ConfigError):          # paged.rs:133-137
        nv.paged_attention(q, layer, empty, 4, 4, 2, 8, 0.35)
    bt = nv.BlockTable(4)
    bt.append_block(0)
    with pytest.raises(nv.TensorError):          # narrow past the gathered blocks (paged.rs:151)
        nv.paged_attention(q, layer, bt, 5, 4, 2, 8, 0.35)
    bt.append_block(9)
    with pytest.raises(nv.TensorError):          # index_select out of range (kv_cache.rs:130)
        nv.paged_attention(q, layer, bt, 5, 4, 2, 8, 0.35)
    z = np.zeros((4, 4, 2, 8), np.float32)
    with pytest.raises(oracle.OracleConfigError):
        oracle.paged_attention(np.zeros((1, 1, 4, 8), np.float32), z, z, [], 4, 4, 4, 2, 8, 0.35)
    with pytest.raises(oracle.OracleTensorError):
        oracle.paged_attention(np.zeros((1, 1, 4, 8), np.float32), z, z, [0], 4, 5, 4, 2, 8, 0.35)


# ---- fast path (bf16, D 128, block 16) -------------------------------------------------
QWEN4B = dict(Hq=32, Hkv=8)   # BASELINE config 3/4/5 head shape
QWEN06B = dict(Hq=16, Hkv=8)  # BASELINE config 1/2 head shape


@pytest.mark.parametrize("heads", [QWEN4B, QWEN06B, dict(Hq=8, Hkv=8), dict(Hq=4, Hkv=1)])
@pytest.mark.parametrize("T", [1, 4])
def test_fast_path_ragged_batch(dev, heads, T):
    Hq, Hkv, D, bs = heads["Hq"], heads["Hkv"], 128, 16
    rng = np.random.default_rng(Hq * 100 + T)
    # ragged contexts: partial last blocks, odd block counts, a single token, exact multiples
    ctx_lens = [max(T, c) for c in (1, 5, 16, 17, 31, 32, 33, 100, 257, 640, 1000)]
    kc, vc, tables = make_paged_case(rng, ctx_lens, 400, bs, Hkv, D, slack_blocks=1)
    q = bf16_randn(rng, (len(ctx_lens), T, Hq, D))
    scale = 1.0 / math.sqrt(D)
    got = _run_batched(dev, q, kc, vc, tables, ctx_lens, scale)
    want = oracle_paged_batch(q, kc, vc, tables, ctx_lens, bs, scale)
    assert_attention_close(got, want)


@pytest.mark.parametrize("T", [1, 2, 3, 5, 8])
def test_fast_path_multi_token_causal(dev, T):
    # K3: speculative verify / chunked prefill, bottom-right mask (paged.rs:357-381)
    Hq, Hkv, D, bs = 16, 8, 128, 16     # G = 2 -> up to 16 query rows
    rng = np.random.default_rng(50 + T)
    ctx_lens = [T, T + 1, 16, 24, 47, 48, 49, 300]
    ctx_lens = [max(T, c) for c in ctx_lens]
    kc, vc, tables = make_paged_case(rng, ctx_lens, 200, bs, Hkv, D)
    q = bf16_randn(rng, (len(ctx_lens), T, Hq, D))
    scale = 1.0 / math.sqrt(D)
    got = _run_batched(dev, q, kc, vc, tables, ctx_lens, scale)
    want = oracle_paged_batch(q, kc, vc, tables, ctx_lens, bs, scale)
    assert_attention_close(got, want)


@pytest.mark.parametrize("pin", ["48", None])
def test_fast_path_32_query_rows(dev, policy, pin):
    # T=8 with G=4 -> two 16-row query tiles in the decode kernel (pinned: a launch this small
    # would otherwise go to the prefill kernel over the cache, the second case)
    if pin:
        policy.setenv("NVLLM_PAGED_MAX_ROWS", pin)
    Hq, Hkv, D, bs, T = 32, 8, 128, 16, 8
    rng = np.random.default_rng(808)
    ctx_lens = [8, 40, 130, 515]
    kc, vc, tables = make_paged_case(rng, ctx_lens, 128, bs, Hkv, D)
    q = bf16_randn(rng, (len(ctx_lens), T, Hq, D))
    scale = 1.0 / math.sqrt(D)
    got = _run_batched(dev, q, kc, vc, tables, ctx_lens, scale)
    want = oracle_paged_batch(q, kc, vc, tables, ctx_lens, bs, scale)
    assert_attention_close(got, want)


@pytest.mark.parametrize("T,splits", [(9, 0), (10, 2), (12, 0), (12, 3), (13, 0), (16, 0), (24, 0), (30, 0), (47, 0)])
def test_verify_lengths_around_the_kernel_boundary(dev, T, splits, policy):
    """More than 32 query rows per kv head: the decode kernel's third query tile (T = 9..12 at
    G = 4; with and without split-K; NVLLM_PAGED_MAX_ROWS pins the decode kernel, which the
    launcher would only pick for an HBM-bound launch), then the prefill kernel over the cache with
    the head group packed into its rows (T >= 13)."""
    policy.setenv("NVLLM_PAGED_MAX_ROWS", "48")
    if splits:
        policy.setenv("NVLLM_PAGED_SPLITS", str(splits))
    Hq, Hkv, D, bs = 32, 8, 128, 16
    rng = np.random.default_rng(900 + T)
    ctx_lens = [max(T, c) for c in (T, T + 3, T + 1, 16, 47, 48, 49, 130, 300, 515, 1000)]
    kc, vc, tables = make_paged_case(rng, ctx_lens, 300, bs, Hkv, D, slack_blocks=1)
    q = bf16_randn(rng, (len(ctx_lens), T, Hq, D))
    scale = 1.0 / math.sqrt(D)
    got = _run_batched(dev, q, kc, vc, tables, ctx_lens, scale, max_blocks=64)
    want = oracle_paged_batch(q, kc, vc, tables, ctx_lens, bs, scale)
    assert_attention_close(got, want)


@pytest.mark.parametrize("T", [12, 30])
def test_context_shorter_than_the_new_tokens(dev, policy, T):
    policy.setenv("NVLLM_PAGED_MAX_ROWS", "48")
    """Invalid for the reference (ctx >= T there); every kernel gives rows whose position is
    negative a zero output (T = 12: decode kernel, T = 30: packed prefill kernel; against the
    generic kernel)."""
    Hq, Hkv, D, bs = 32, 8, 128, 16
    rng = np.random.default_rng(931)
    ctx_lens = [0, 7, 11, T - 1]
    kc, vc, tables = make_paged_case(rng, [64] * 4, 40, bs, Hkv, D)
    q = bf16_randn(rng, (4, T, Hq, D))
    scale = 1.0 / math.sqrt(D)
    got = _run_batched(dev, q, kc, vc, tables, ctx_lens, scale)
    f32 = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    gen = nv.paged_attention_batched(f32(q).to(dev), f32(kc).to(dev), f32(vc).to(dev),
                                     torch.from_numpy(pad_tables(tables, None)).to(dev),
                                     torch.tensor(ctx_lens, dtype=torch.int32, device=dev), scale)
    want = gen.float().cpu().numpy()      # f32 tensors run the generic kernel
    for i, ctx in enumerate(ctx_lens):
        assert (got[i, :T - ctx] == 0).all()
    assert_attention_close(got, want)


@pytest.mark.parametrize("splits", [1, 2, 3, 8])
def test_split_k_matches_oracle(dev, splits, policy):
    # K2b: forced split counts, contexts that leave some splits empty
    policy.setenv("NVLLM_PAGED_SPLITS", str(splits))
    Hq, Hkv, D, bs, T = 32, 8, 128, 16, 1
    rng = np.random.default_rng(splits)
    ctx_lens = [3, 200, 1024, 2048, 1555]
    kc, vc, tables = make_paged_case(rng, ctx_lens, 400, bs, Hkv, D)
    q = bf16_randn(rng, (len(ctx_lens), T, Hq, D))
    scale = 1.0 / math.sqrt(D)
    got = _run_batched(dev, q, kc, vc, tables, ctx_lens, scale, max_blocks=128)
    want = oracle_paged_batch(q, kc, vc, tables, ctx_lens, bs, scale)
    assert_attention_close(got, want)


def test_shared_prefix_blocks_and_spec_verify(dev):
    # BASELINE config 4 in miniature: every table starts with the same physical blocks
    Hq, Hkv, D, bs, T = 32, 8, 128, 16, 4
    rng = np.random.default_rng(4)
    n = 12
    ctx_lens = [256 + 16 * (i % 5) + (i % 3) for i in range(n)]
    kc, vc, tables = make_paged_case(rng, ctx_lens, 300, bs, Hkv, D, shared_prefix_blocks=8)
    for t in tables:
        assert t[:8] == tables[0][:8]
    q = bf16_randn(rng, (n, T, Hq, D))
    scale = 1.0 / math.sqrt(D)
    got = _run_batched(dev, q, kc, vc, tables, ctx_lens, scale)
    want = oracle_paged_batch(q, kc, vc, tables, ctx_lens, bs, scale)
    assert_attention_close(got, want)


@pytest.mark.parametrize("nt", [0, 1])
@pytest.mark.parametrize("prefix_blocks,splits", [(0, 1), (1, 1), (5, 1), (8, 2), (11, 3), (40, 1)])
def test_adaptive_cache_policy_is_invisible(dev, policy, nt, prefix_blocks, splits):
    # K2 reads a shared leading prefix with the default cache policy and everything after it
    # non-temporally, in two separate tile loops; the switch-over tile (odd prefix lengths fall
    # in the middle of a 2-block tile), split boundaries and an odd sequence count (the last
    # sequence has no neighbour) must all give the oracle's result, with the policy on or off
    policy.setenv("NVLLM_PAGED_NT", str(nt))
    policy.setenv("NVLLM_PAGED_SPLITS", str(splits))
    Hq, Hkv, D, bs, T = 32, 8, 128, 16, 1
    rng = np.random.default_rng(100 + prefix_blocks)
    ctx_lens = [700, 650, 1000, 777, 16 * prefix_blocks + 5 if prefix_blocks else 40, 900, 333]
    kc, vc, tables = make_paged_case(rng, ctx_lens, 500, bs, Hkv, D,
                                     shared_prefix_blocks=prefix_blocks)
    q = bf16_randn(rng, (len(ctx_lens), T, Hq, D))
    scale = 1.0 / math.sqrt(D)
    got = _run_batched(dev, q, kc, vc, tables, ctx_lens, scale, max_blocks=64)
    want = oracle_paged_batch(q, kc, vc, tables, ctx_lens, bs, scale)
    assert_attention_close(got, want)


@pytest.mark.parametrize("T", [1, 4, 8, 12])
@pytest.mark.parametrize("prefix_blocks,splits", [(8, 1), (9, 1), (8, 2), (30, 1)])
def test_private_tiles_first_walk(dev, policy, T, prefix_blocks, splits):
    # sequences 8..15, 24..31 ... walk their private tiles before the shared prefix (config 4's
    # overlap of L2 and HBM phases): same result as the oracle's front-to-back walk, for one to
    # three query tiles, a prefix that ends inside a 2-block tile, splits, contexts that end inside
    # the prefix's last tile or barely past it, and the causal tail of T > 1
    policy.setenv("NVLLM_PAGED_NT", "1")
    policy.setenv("NVLLM_PAGED_SPLITS", str(splits))
    Hq, Hkv, D, bs = 32, 8, 128, 16
    rng = np.random.default_rng(1000 * T + prefix_blocks)
    n = 20
    ctx_lens = [16 * prefix_blocks + (1 + 37 * i) % 400 + T for i in range(n)]
    ctx_lens[9] = 16 * prefix_blocks + T          # nothing but the prefix and the new tokens
    ctx_lens[10] = 16 * prefix_blocks + 17 + T    # one private block and a bit
    kc, vc, tables = make_paged_case(rng, ctx_lens, 700, bs, Hkv, D, shared_prefix_blocks=prefix_blocks)
    q = bf16_randn(rng, (n, T, Hq, D))
    scale = 1.0 / math.sqrt(D)
    got = _run_batched(dev, q, kc, vc, tables, ctx_lens, scale, max_blocks=64)
    want = oracle_paged_batch(q, kc, vc, tables, ctx_lens, bs, scale)
    assert_attention_close(got, want)


def test_zero_context_rows_are_zero(dev):
    Hq, Hkv, D, bs = 32, 8, 128, 16
    rng = np.random.default_rng(0)
    ctx_lens = [0, 40, 0]
    kc, vc, tables = make_paged_case(rng, [16, 40, 16], 32, bs, Hkv, D)
    q = bf16_randn(rng, (3, 1, Hq, D))
    got = _run_batched(dev, q, kc, vc, tables, ctx_lens, 1 / math.sqrt(D))
    assert (got[0] == 0).all() and (got[2] == 0).all()
    want = oracle.paged_attention(q[1:2], kc, vc, tables[1], bs, 40, Hq, Hkv, D, 1 / math.sqrt(D))
    assert_attention_close(got[1:2], want)


def test_large_scores_stay_finite(dev):
    # reference stability check (sigma = 10, tests/flash_attention_test.rs:109-131) on the paged path
    Hq, Hkv, D, bs = 32, 8, 128, 16
    rng = np.random.default_rng(10)
    ctx_lens = [100, 333]
    kc, vc, tables = make_paged_case(rng, ctx_lens, 64, bs, Hkv, D)
    kc = oracle.bf16_round(kc * 10)
    q = bf16_randn(rng, (2, 1, Hq, D), sigma=10.0)
    scale = 1 / math.sqrt(D)
    got = _run_batched(dev, q, kc, vc, tables, ctx_lens, scale)
    want = oracle_paged_batch(q, kc, vc, tables, ctx_lens, bs, scale)
    assert np.isfinite(got).all()
    assert_attention_close(got, want, max_tol=5e-2, mean_tol=5e-3)  # near one-hot softmax of bf16 P


def test_paged_equals_contiguous_sdpa_round_trip(dev):
    """docs/paged_attention.md:328-357 identity at a size the oracle would take long on:
    write K/V through reshape_and_cache, read back through paged attention, compare with
    flash attention over the contiguous K/V (last row = the decode query)."""
    Hq, Hkv, D, bs = 32, 8, 128, 16
    S = 4096
    g = torch.Generator(device="cpu").manual_seed(5)
    k = torch.randn((S, Hkv, D), generator=g).to(dev, torch.bfloat16)
    v = torch.randn((S, Hkv, D), generator=g).to(dev, torch.bfloat16)
    q = torch.randn((1, 1, Hq, D), generator=g).to(dev, torch.bfloat16)
    nb = 600
    ids = torch.randperm(nb, generator=g)[: S // bs].tolist()
    bt = nv.BlockTable(bs)
    for b in ids:
        bt.append_block(b)
    layer = nv.LayerKVCache(nv.KVCacheConfig(nb, bs, Hkv, D, 1, torch.bfloat16), dev)
    nv.write_kv_to_cache(k[None], v[None], layer.key_cache(), layer.value_cache(),
                         bt.get_slot_mapping(S), bs)
    out = nv.paged_attention(q, layer, bt, S, Hq, Hkv, D, 1 / math.sqrt(D)).float()
    kf = k.float().repeat_interleave(Hq // Hkv, dim=1)   # [S, Hq, D]
    vf = v.float().repeat_interleave(Hq // Hkv, dim=1)
    s = torch.einsum("hd,shd->hs", q[0, 0].float(), kf) / math.sqrt(D)
    ref = torch.einsum("hs,shd->hd", torch.softmax(s, dim=-1), vf).reshape(1, 1, Hq * D)
    assert_attention_close(out.cpu().numpy(), ref.cpu().numpy())


def test_debug_validate_tables_counts_what_a_launch_would_clamp(dev):
    # nvllm_debug_validate_tables (no reference counterpart: paged.rs:134-146 index_select fails on
    # a bad id; the kernels here clamp or read zeros).  Only the blocks a context length uses count.
    bs, nb, max_blocks = 16, 100, 8
    bt = torch.tensor([[1, 2, 3, 99, -1, 500, 0, 0],          # ctx 40 -> 3 blocks used: clean
                       [5, 100, -7, 6, 7, 8, 9, 10],          # ctx 128 -> 8 used: 100 and -7 are bad
                       [0, 1, 2, 3, 4, 5, 6, 7],              # ctx too long for the row
                       [-1, -1, -1, -1, -1, -1, -1, -1]],     # ctx 0: nothing used
                      dtype=torch.int32, device=dev)
    cl = torch.tensor([40, 128, 129, 0], dtype=torch.int32, device=dev)
    assert nv.validate_tables(bt, cl, bs, nb) == (2, 1)
    cl_ok = torch.tensor([40, 16, 128, 0], dtype=torch.int32, device=dev)
    assert nv.validate_tables(bt, cl_ok, bs, nb) == (0, 0)
    cl_neg = torch.tensor([-3, 16, 128, 0], dtype=torch.int32, device=dev)
    assert nv.validate_tables(bt, cl_neg, bs, nb) == (0, 1)
