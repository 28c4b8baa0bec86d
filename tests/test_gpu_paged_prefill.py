"""K1 over the paged cache: chunked prefill / long verify (more than 48 query rows per kv head).

Same contract and oracle as the decode tests (oracle.paged_attention restates reference
src/attention/paged.rs:117-192; the bottom-right causal mask is paged.rs:357-381): T new tokens
per sequence attend to context_lens[seq] cached keys, query t at position ctx - T + t.  The
kernel tiles query POSITIONS from a multiple of 32, so the cases walk every residue of
(ctx - T) mod 32 / mod 64, partial last blocks, block tables whose row length is not a multiple
of four, garbage ids behind a sequence's blocks and NaN in every cache slot no sequence owns.
Tolerance (north_star): 2e-2 max-abs, 1e-3 mean-abs.
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


def _run(dev, q, kc, vc, tables, ctx_lens, scale, max_blocks=None, fill=0):
    bt = torch.from_numpy(pad_tables(tables, max_blocks, fill)).to(dev)
    cl = torch.tensor(ctx_lens, dtype=torch.int32, device=dev)
    out = nv.paged_attention_batched(_dev(q, dev), _dev(kc, dev), _dev(vc, dev), bt, cl, scale)
    torch.cuda.synchronize()
    return out.float().cpu().numpy()


@pytest.mark.parametrize("T", [9, 12, 13, 16, 31, 33, 64, 100, 128, 129, 200, 300])
def test_chunk_lengths_and_alignments(dev, T):
    Hq, Hkv, D, bs = 32, 8, 128, 16          # G = 4: T >= 13 leaves the decode kernel's 48 rows (9 and 12: its third query tile)
    rng = np.random.default_rng(7000 + T)
    ctx_lens = [T + c for c in (0, 1, 15, 16, 17, 31, 32, 33, 47, 63, 64, 65, 100, 333)]
    kc, vc, tables = make_paged_case(rng, ctx_lens, 420, bs, Hkv, D, slack_blocks=1)
    q = bf16_randn(rng, (len(ctx_lens), T, Hq, D))
    scale = 1.0 / math.sqrt(D)
    got = _run(dev, q, kc, vc, tables, ctx_lens, scale, fill=-1)
    want = oracle_paged_batch(q, kc, vc, tables, ctx_lens, bs, scale)
    assert_attention_close(got, want)


@pytest.mark.parametrize("heads", [dict(Hq=16, Hkv=8), dict(Hq=8, Hkv=8), dict(Hq=4, Hkv=1),
                                   dict(Hq=64, Hkv=1), dict(Hq=16, Hkv=2), dict(Hq=6, Hkv=2),
                                   dict(Hq=16, Hkv=1)])
def test_head_shapes(dev, heads):
    # group sizes 2, 4, 8 are packed into the rows; 1, 3, 16, 64 are not
    Hq, Hkv, D, bs = heads["Hq"], heads["Hkv"], 128, 16
    T = 1 if Hq // Hkv > 32 else 70           # Hq/Hkv = 64: even T = 1 exceeds the decode kernel
    rng = np.random.default_rng(Hq * 10 + Hkv)
    ctx_lens = [T, T + 5, T + 200, T + 77]
    kc, vc, tables = make_paged_case(rng, ctx_lens, 100, bs, Hkv, D)
    q = bf16_randn(rng, (len(ctx_lens), T, Hq, D))
    scale = 1.0 / math.sqrt(D)
    got = _run(dev, q, kc, vc, tables, ctx_lens, scale)
    want = oracle_paged_batch(q, kc, vc, tables, ctx_lens, bs, scale)
    assert_attention_close(got, want)


@pytest.mark.parametrize("max_blocks", [5, 6, 7, 9])
def test_table_rows_not_a_multiple_of_four(dev, max_blocks):
    # the LAST sequence uses every entry of its row, the row ends the table: the id fetch of a
    # key tile must not lose ids that share a 16-byte read with entries past the table
    Hq, Hkv, D, bs, T = 32, 8, 128, 16, 40
    rng = np.random.default_rng(max_blocks)
    ctx_lens = [40, 16 * max_blocks - 9, 16 * max_blocks]
    kc, vc, tables = make_paged_case(rng, ctx_lens, 40, bs, Hkv, D)
    q = bf16_randn(rng, (len(ctx_lens), T, Hq, D))
    scale = 1.0 / math.sqrt(D)
    got = _run(dev, q, kc, vc, tables, ctx_lens, scale, max_blocks=max_blocks, fill=-1)
    want = oracle_paged_batch(q, kc, vc, tables, ctx_lens, bs, scale)
    assert_attention_close(got, want)


def test_shared_prefix_blocks(dev):
    Hq, Hkv, D, bs, T = 32, 8, 128, 16, 48
    rng = np.random.default_rng(44)
    ctx_lens = [256 + 16 * (i % 5) + (i % 3) for i in range(6)]
    kc, vc, tables = make_paged_case(rng, ctx_lens, 200, bs, Hkv, D, shared_prefix_blocks=8)
    q = bf16_randn(rng, (len(ctx_lens), T, Hq, D))
    scale = 1.0 / math.sqrt(D)
    got = _run(dev, q, kc, vc, tables, ctx_lens, scale)
    want = oracle_paged_batch(q, kc, vc, tables, ctx_lens, bs, scale)
    assert_attention_close(got, want)


def test_context_shorter_than_chunk_matches_generic_kernel(dev, policy):
    # invalid for the reference (a chunk's own K/V are in the cache, so ctx >= T); both kernels
    # give rows whose position is negative a zero output
    Hq, Hkv, D, bs, T = 32, 8, 128, 16, 50
    rng = np.random.default_rng(3)
    ctx_lens = [0, 7, 33, 49]
    kc, vc, tables = make_paged_case(rng, [64] * 4, 40, bs, Hkv, D)
    kc, vc = np.nan_to_num(kc), np.nan_to_num(vc)
    q = bf16_randn(rng, (4, T, Hq, D))
    scale = 1.0 / math.sqrt(D)
    got = _run(dev, q, kc, vc, tables, ctx_lens, scale)
    policy.setenv("NVLLM_PAGED_FLASH", "0")
    want = _run(dev, q, kc, vc, tables, ctx_lens, scale)
    assert (got[0] == 0).all()
    for i, ctx in enumerate(ctx_lens):
        assert (got[i, :T - ctx] == 0).all()
    assert_attention_close(got, want)


def test_large_scores_stay_finite(dev):
    Hq, Hkv, D, bs, T = 32, 8, 128, 16, 64
    rng = np.random.default_rng(11)
    ctx_lens = [100, 333]
    kc, vc, tables = make_paged_case(rng, ctx_lens, 64, bs, Hkv, D)
    kc = oracle.bf16_round(kc * 10)
    q = bf16_randn(rng, (2, T, Hq, D), sigma=10.0)
    scale = 1 / math.sqrt(D)
    got = _run(dev, q, kc, vc, tables, ctx_lens, scale)
    want = oracle_paged_batch(q, kc, vc, tables, ctx_lens, bs, scale)
    assert np.isfinite(got).all()
    assert_attention_close(got, want, max_tol=5e-2, mean_tol=5e-3)


@pytest.mark.parametrize("pack", [None, "0"])
@pytest.mark.parametrize("S,T", [(4096, 4096), (4096, 512), (3000, 1000), (8192, 2048)])
def test_paged_chunk_equals_contiguous_prefill(dev, policy, S, T, pack):
    """Size-independent property at sizes the oracle would take minutes on: K/V written through
    reshape_and_cache and read back through the block table give the last T rows of the
    contiguous causal prefill over the same K/V (the K1 dense path, itself pinned to the oracle
    in test_gpu_flash_attention.py).  With T == S and the head group NOT packed into the rows
    (NVLLM_K1_PACK=0) the two walks are the same arithmetic in the same order: the outputs must be
    bit-identical.  Packed (the default) a wave holds other rows, so its deferred-rescale decisions
    may differ: same bar as everywhere else."""
    if pack is not None:
        policy.setenv("NVLLM_K1_PACK", pack)
    Hq, Hkv, D, bs = 32, 8, 128, 16
    g = torch.Generator(device="cpu").manual_seed(S + T)
    k = torch.randn((1, S, Hkv, D), generator=g).to(dev, torch.bfloat16)
    v = torch.randn((1, S, Hkv, D), generator=g).to(dev, torch.bfloat16)
    q = torch.randn((1, S, Hq, D), generator=g).to(dev, torch.bfloat16)
    nblk = -(-S // bs)
    nb = nblk + 50
    ids = torch.randperm(nb, generator=g)[:nblk].tolist()
    bt = nv.BlockTable(bs)
    for b in ids:
        bt.append_block(b)
    layer = nv.LayerKVCache(nv.KVCacheConfig(nb, bs, Hkv, D, 1, torch.bfloat16), dev)
    layer.key_cache().fill_(float("nan"))
    layer.value_cache().fill_(float("nan"))
    nv.write_kv_to_cache(k, v, layer.key_cache(), layer.value_cache(), bt.get_slot_mapping(S), bs)
    scale = 1 / math.sqrt(D)
    dense = nv.flash_attention(q, k, v, nv.FlashAttentionConfig(D, True))   # [1, S, Hq, D]
    paged = nv.paged_attention(q[:, S - T:].contiguous(), layer, bt, S, Hq, Hkv, D, scale)
    torch.cuda.synchronize()
    want = dense[:, S - T:].reshape(1, T, Hq * D)
    if T == S and pack == "0":
        assert torch.equal(paged, want)
    assert_attention_close(paged.float().cpu().numpy(), want.float().cpu().numpy())


# ---- key split of small launches (KSPLIT parts + merge kernel) -------------------------------
@pytest.mark.parametrize("parts", [2, 3, 4, 8])
@pytest.mark.parametrize("T", [40, 130, 300])
def test_key_split_matches_oracle(dev, policy, parts, T):
    # forced part counts, contexts short enough that some parts are empty, alignments that put
    # dead rows in front, both tile pairings
    policy.setenv("NVLLM_K1_KSPLIT", str(parts))
    Hq, Hkv, D, bs = 32, 8, 128, 16
    rng = np.random.default_rng(900 + parts * 10 + T)
    ctx_lens = [T, T + 1, T + 47, T + 64, T + 333, T + 1000]
    kc, vc, tables = make_paged_case(rng, ctx_lens, 520, bs, Hkv, D, slack_blocks=1)
    q = bf16_randn(rng, (len(ctx_lens), T, Hq, D))
    scale = 1.0 / math.sqrt(D)
    got = _run(dev, q, kc, vc, tables, ctx_lens, scale, fill=-1)
    want = oracle_paged_batch(q, kc, vc, tables, ctx_lens, bs, scale)
    assert_attention_close(got, want)
    policy.setenv("NVLLM_K1_PAIRING", "a")
    got = _run(dev, q, kc, vc, tables, ctx_lens, scale, fill=-1)
    assert_attention_close(got, want)


def test_key_split_is_chosen_for_one_long_chunk_and_matches_the_plain_launch(dev, policy):
    # one 512-token chunk against 4096 cached keys: 64 workgroups unsplit -> 4 parts by default
    Hq, Hkv, D, bs, T, S = 32, 8, 128, 16, 512, 4096
    L = nv.capi.lib()
    assert L.nvllm_paged_attn_workspace_bytes(1, T, Hq, Hkv, D, bs, S // bs) > 0
    assert L.nvllm_paged_attn_workspace_bytes(16, T, Hq, Hkv, D, bs, S // bs) == 0     # enough workgroups
    assert L.nvllm_paged_attn_workspace_bytes(1, T, Hq, Hkv, D, bs, 8) == 0             # too few keys
    g = torch.Generator(device="cpu").manual_seed(3)
    nb = S // bs + 20
    kc = torch.randn((nb, bs, Hkv, D), generator=g).to(dev, torch.bfloat16)
    vc = torch.randn((nb, bs, Hkv, D), generator=g).to(dev, torch.bfloat16)
    q = torch.randn((1, T, Hq, D), generator=g).to(dev, torch.bfloat16)
    bt = torch.randperm(nb, generator=g)[:S // bs].to(torch.int32).reshape(1, -1).to(dev)
    cl = torch.tensor([S - 5], dtype=torch.int32, device=dev)
    scale = 1 / math.sqrt(D)
    split = nv.paged_attention_batched(q, kc, vc, bt, cl, scale).float()
    policy.setenv("NVLLM_K1_KSPLIT", "1")
    plain = nv.paged_attention_batched(q, kc, vc, bt, cl, scale).float()
    torch.cuda.synchronize()
    d = (split - plain).abs()
    assert float(d.max()) <= 2e-2 and float(d.mean()) <= 1e-3
    assert not torch.equal(split, plain)      # the split path really ran
