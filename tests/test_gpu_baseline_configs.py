"""BASELINE.json configs 3, 4 and 5 at FULL size through the C ABI, launch policy untouched (the
launcher's own splits / cache policy / kernel choice), checked against the CPU oracle
(oracle.paged_attention, the restatement of reference src/attention/paged.rs:117-192,357-381) on a
sample of the sequences -- the oracle takes seconds per long sequence -- and, for config 5, against
a float32 evaluation of the same formula on every sequence.  Tolerance: north_star's 2e-2 max-abs /
1e-3 mean-abs.  The workloads are built by bench.py's own builder, so what is benchmarked is what is
checked here."""
import math

import numpy as np
import pytest
import torch

import bench
import nano_vllm_rs_amd as nv
import oracle
from helpers import assert_attention_close

pytestmark = pytest.mark.gpu

Hq, Hkv, D, bs = 32, 8, 128, 16      # Qwen3-4B head shape (BASELINE configs 3-5)
SCALE = 1.0 / math.sqrt(D)


def _oracle_rows(case, q, ctx, T, sample, layer=0):
    """oracle.paged_attention on the sampled sequences; each gets its own blocks gathered to the
    host as f32 (the values are bf16, so the oracle sees exactly what the kernel sees)."""
    kc, vc = case["caches"][layer]
    outs = []
    for i in sample:
        ids = case["tables_np"][i][: -(-ctx // bs)]
        idx = torch.from_numpy(ids.astype(np.int64)).to(kc.device)
        kch, vch = kc[idx].float().cpu().numpy(), vc[idx].float().cpu().numpy()
        outs.append(oracle.paged_attention(q[i:i + 1].float().cpu().numpy(), kch, vch,
                                           list(range(len(ids))), bs, ctx, Hq, Hkv, D, SCALE))
    return np.concatenate(outs, axis=0)


def _launch(case, T, layer=0):
    kc, vc = case["caches"][layer]
    out = nv.paged_attention_batched(case["q"], kc, vc, case["block_tables"], case["context_lens"], SCALE)
    torch.cuda.synchronize()
    return out


def test_config3_full_size_decode(dev):
    """256 sequences x 2048-token contexts, tables a seeded permutation of a 40,960-block pool."""
    N, ctx, T = 256, 2048, 1
    case = bench.build_decode_case(nv, dev, 3, N, ctx, T, Hq, Hkv, D, bs, 1, 0, 40960)
    L = nv.capi.lib()
    assert L.nvllm_paged_attn_workspace_bytes(N, T, Hq, Hkv, D, bs, ctx // bs) == 0   # splits == 1
    assert int(case["tables_np"].max()) > 30000 and len(np.unique(case["tables_np"])) == N * ctx // bs
    got = _launch(case, T).float().cpu().numpy()
    sample = list(range(0, N, 16)) + [N - 1]          # 17 sequences, first and last included
    want = _oracle_rows(case, case["q"], ctx, T, sample)
    assert_attention_close(got[sample], want)
    assert np.isfinite(got).all()


def test_config4_full_size_shared_prefix_verify(dev):
    """256 sequences, T = 4 new tokens, ctx 2048 = 1024 shared (64 blocks through
    BlockManager::allocate_with_prefix, identical ids in every table) + 1024 private."""
    N, ctx, T, shared = 256, 2048, 4, 1024
    case = bench.build_decode_case(nv, dev, 4, N, ctx, T, Hq, Hkv, D, bs, 1, shared, None)
    tabs = case["tables_np"]
    assert (tabs[:, :shared // bs] == tabs[0, :shared // bs]).all()
    assert len(np.unique(tabs[:, shared // bs:])) == N * (ctx - shared) // bs
    assert case["unique_blocks"] == 64 + N * 64
    got = _launch(case, T).float().cpu().numpy()
    sample = list(range(0, N, 16)) + [1, N - 1]       # both members of a neighbour pair included
    want = _oracle_rows(case, case["q"], ctx, T, sample)
    assert_attention_close(got[sample], want)
    assert np.isfinite(got).all()


def _long_case(dev, N, ctx, pool_blocks, seed, T=1):
    """Config 5 shape: N sequences x ctx tokens whose block ids are drawn from the WHOLE
    pool_blocks range (2 Mi blocks = 64 GiB per cache: the pool config 5 needs at 1024 sequences).
    Only the blocks the tables touch are filled; the rest of the pool stays uninitialised."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    nblk = ctx // bs
    ids = torch.randperm(pool_blocks, generator=g)[: N * nblk]
    tables = ids.view(N, nblk).to(torch.int32)
    kc = torch.empty((pool_blocks, bs, Hkv, D), dtype=torch.bfloat16, device=dev)
    vc = torch.empty_like(kc)
    idx = ids.to(dev)
    step = 16384
    for b in range(0, idx.numel(), step):
        sl = idx[b:b + step]
        kc[sl] = torch.randn((sl.numel(), bs, Hkv, D), device=dev, dtype=torch.bfloat16)
        vc[sl] = torch.randn((sl.numel(), bs, Hkv, D), device=dev, dtype=torch.bfloat16)
    q = torch.randn((N, T, Hq, D), device=dev, dtype=torch.bfloat16)
    return dict(q=q, caches=[(kc, vc)], block_tables=tables.to(dev).contiguous(),
                context_lens=torch.full((N,), ctx, dtype=torch.int32, device=dev),
                tables_np=tables.numpy())


def _f32_reference(case, i, ctx):
    """softmax(q k^T * scale + bottom-right causal mask) v in float32 on the device, for one
    sequence with T new tokens (paged.rs:357-381: query t sits at position ctx - T + t)."""
    kc, vc = case["caches"][0]
    idx = torch.from_numpy(case["tables_np"][i].astype(np.int64)).to(kc.device)
    G = Hq // Hkv
    T = case["q"].shape[1]
    k = kc[idx].reshape(-1, Hkv, D)[:ctx].float()                     # [S, Hkv, D]
    v = vc[idx].reshape(-1, Hkv, D)[:ctx].float()
    q = case["q"][i].float().reshape(T, Hkv, G, D)
    s = torch.einsum("tkgd,skd->tkgs", q, k) * SCALE                  # [T, Hkv, G, S]
    if T > 1:
        pos = torch.arange(ctx, device=kc.device)[None, :] > (ctx - T + torch.arange(T, device=kc.device))[:, None]
        s = s.masked_fill(pos[:, None, None, :], float("-inf"))
    return torch.einsum("tkgs,skd->tkgd", torch.softmax(s, dim=-1), v).reshape(1, T, Hq * D)


@pytest.mark.parametrize("N,expect_splits", [(8, 32), (128, 2), (256, 1)])
def test_config5_long_context_decode(dev, N, expect_splits):
    """32,768-token contexts, max_blocks_per_seq 2048, block ids up to 2 Mi: 8 sequences (deep
    split-K: 32 splits + merge) checked on every sequence by the oracle, 128 sequences
    (splits == 2, the merge pass of a full-rate launch) and 256 sequences (splits == 1: the
    launch BASELINE config 5 itself makes at 1024 sequences -- one wave walks all 1024 tiles of
    its sequence, no workspace, no merge) on a sample; every sequence of all three also against
    the float32 formula."""
    ctx, pool = 32768, 2 * 1024 * 1024
    case = _long_case(dev, N, ctx, pool, 5 + N)
    assert int(case["tables_np"].max()) > 2_000_000 and case["tables_np"].shape[1] == 2048
    L = nv.capi.lib()
    need = L.nvllm_paged_attn_workspace_bytes(N, 1, Hq, Hkv, D, bs, 2048)
    assert need == (N * Hq * expect_splits * (D + 2) * 4 if expect_splits > 1 else 0)
    assert nv.paged_attention_path(N, 1, Hq, Hkv, D, bs, 2048, pool) == "decode"
    got = _launch(case, 1).float().cpu().numpy()
    assert np.isfinite(got).all()
    sample = list(range(N)) if N <= 8 else [0, 37, 90, N - 1]
    want = _oracle_rows(case, case["q"], ctx, 1, sample)
    assert_attention_close(got[sample], want)
    ref = torch.cat([_f32_reference(case, i, ctx) for i in range(N)], 0).cpu().numpy()
    assert_attention_close(got.reshape(N, -1), ref.reshape(N, -1))


def _launch_us(case, rounds=15):
    """median HIP-event time of one launch, microseconds (workspace allocated once, outside)"""
    kc, vc = case["caches"][0]
    N, T = case["q"].shape[0], case["q"].shape[1]
    need = nv.capi.lib().nvllm_paged_attn_workspace_bytes(N, T, Hq, Hkv, D, bs, case["block_tables"].shape[1])
    ws = torch.empty(max(int(need), 16), dtype=torch.uint8, device=kc.device) if need else None
    times = []
    for r in range(rounds + 3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        nv.paged_attention_batched(case["q"], kc, vc, case["block_tables"], case["context_lens"], SCALE, workspace=ws)
        e1.record()
        e1.synchronize()
        if r >= 3:
            times.append(e0.elapsed_time(e1) * 1e3)
    return float(np.median(times))


@pytest.mark.parametrize("T", [16, 64])
def test_k1_over_the_cache_on_the_config5_pool(dev, T):
    """A verify / prefill chunk (src/speculative/engine.rs:146-169 verifies K+1 positions against the
    whole cached context; paged.rs:357-381 is its mask) against the 2 Mi-block pool of config 5:
    64 GiB per cache, past what a 32-bit index of 16-byte chunks reaches.  The launch must stay on
    the prefill kernel over the cache (not the scalar generic kernel), match the oracle on sampled
    sequences and the float32 formula on all, and take no longer than 1.2 x the same shape on a
    pool that holds just its own blocks."""
    N, ctx, pool = 32, 4096, 2 * 1024 * 1024
    assert nv.paged_attention_path(N, T, Hq, Hkv, D, bs, ctx // bs, pool) == "k1_paged"
    assert nv.capi.lib().nvllm_kv_cache_layer_bytes(pool, bs, Hkv, D, 0) == 64 << 30
    case = _long_case(dev, N, ctx, pool, 500 + T, T=T)
    case["context_lens"] = torch.tensor([ctx - 13 * (i % 7) for i in range(N)], dtype=torch.int32, device=dev)
    ctxs = case["context_lens"].cpu().tolist()
    assert int(case["tables_np"].max()) > 2_000_000
    got = _launch(case, T).float().cpu().numpy()
    assert np.isfinite(got).all()
    kc, vc = case["caches"][0]
    outs = []
    sample = [0, 13, N - 1]
    for i in sample:
        idx = torch.from_numpy(case["tables_np"][i].astype(np.int64)).to(dev)
        outs.append(oracle.paged_attention(case["q"][i:i + 1].float().cpu().numpy(), kc[idx].float().cpu().numpy(),
                                           vc[idx].float().cpu().numpy(), list(range(ctx // bs)), bs, ctxs[i],
                                           Hq, Hkv, D, SCALE))
    assert_attention_close(got[sample], np.concatenate(outs, 0))
    ref = torch.cat([_f32_reference(case, i, ctxs[i]) for i in range(N)], 0).cpu().numpy()
    assert_attention_close(got, ref)
    t_big = _launch_us(case)
    del case, kc, vc
    torch.cuda.empty_cache()
    small = _long_case(dev, N, ctx, N * (ctx // bs) + 64, 500 + T, T=T)
    small["context_lens"] = torch.tensor(ctxs, dtype=torch.int32, device=dev)
    t_small = _launch_us(small)
    print("K1 over the cache, %d x %d, T = %d: 2 Mi-block pool %.1f us, own-blocks pool %.1f us" % (N, ctx, T, t_big, t_small))
    assert t_big <= 1.2 * t_small, (t_big, t_small)


def test_mid_kernel_on_the_config5_pool(dev, policy):
    """The mid kernel (csrc/paged_attn_mid.hip, 49..128 rows per kv head on large launches) against
    the same 64 GiB caches: block addresses are 64-bit there too.  Forced (32 sequences are fewer than
    its dispatch rule asks for); oracle on sampled sequences, float32 formula on all."""
    N, ctx, pool, T = 32, 4096, 2 * 1024 * 1024, 16
    policy.setenv("NVLLM_PAGED_MID", 1)
    assert nv.paged_attention_path(N, T, Hq, Hkv, D, bs, ctx // bs, pool) == "paged_mid"
    case = _long_case(dev, N, ctx, pool, 700, T=T)
    case["context_lens"] = torch.tensor([ctx - 13 * (i % 7) for i in range(N)], dtype=torch.int32, device=dev)
    ctxs = case["context_lens"].cpu().tolist()
    assert int(case["tables_np"].max()) > 2_000_000
    got = _launch(case, T).float().cpu().numpy()
    assert np.isfinite(got).all()
    kc, vc = case["caches"][0]
    outs = []
    sample = [0, 13, N - 1]
    for i in sample:
        idx = torch.from_numpy(case["tables_np"][i].astype(np.int64)).to(dev)
        outs.append(oracle.paged_attention(case["q"][i:i + 1].float().cpu().numpy(), kc[idx].float().cpu().numpy(),
                                           vc[idx].float().cpu().numpy(), list(range(ctx // bs)), bs, ctxs[i],
                                           Hq, Hkv, D, SCALE))
    assert_attention_close(got[sample], np.concatenate(outs, 0))
    ref = torch.cat([_f32_reference(case, i, ctxs[i]) for i in range(N)], 0).cpu().numpy()
    assert_attention_close(got, ref)


@pytest.mark.parametrize("world", [2, 4, 8])
@pytest.mark.parametrize("T", [1, 4])
def test_kv_head_sharded_slices_through_the_hip_path(dev, policy, world, T):
    """Head-sharded mode on ONE GPU: every rank's slice -- cache [nb, 16, Hkv/n, 128], Hq/n query
    heads -- goes through nvllm_paged_attn_bf16 in turn, the outputs are assembled with the
    permutation all_gather_heads uses, and the result must equal the unsharded launch bit for bit
    (a wave owns one kv head, so its arithmetic does not depend on how many heads the launch has;
    the split count is pinned because it does: it is chosen from num_seqs x Hkv)."""
    sh = nv.sharding
    N, ctx = 24, 1040
    case = bench.build_decode_case(nv, dev, 70 + world, N, ctx, T, Hq, Hkv, D, bs, 1, 0, None)
    case["context_lens"] = torch.tensor([ctx - 7 * (i % 5) for i in range(N)], dtype=torch.int32, device=dev)
    kc, vc = case["caches"][0]
    for splits in (1, 3):
        policy.setenv("NVLLM_PAGED_SPLITS", splits)
        full = _launch(case, T)
        parts = []
        for r in range(world):
            out = nv.paged_attention_batched(sh.shard_query_heads(case["q"], Hkv, world, r),
                                             sh.shard_cache_heads(kc, world, r),
                                             sh.shard_cache_heads(vc, world, r),
                                             case["block_tables"], case["context_lens"], SCALE)
            parts.append(out)
        torch.cuda.synchronize()
        assembled = sh.assemble_heads(torch.cat(parts, 0), world)
        assert torch.equal(assembled, full), (world, T, splits)
    ctxs = case["context_lens"].cpu().tolist()
    sample = [0, 7, N - 1]
    outs = []
    for i in sample:
        ids = case["tables_np"][i]
        idx = torch.from_numpy(ids.astype(np.int64)).to(dev)
        outs.append(oracle.paged_attention(case["q"][i:i + 1].float().cpu().numpy(), kc[idx].float().cpu().numpy(),
                                           vc[idx].float().cpu().numpy(), list(range(len(ids))), bs, ctxs[i],
                                           Hq, Hkv, D, SCALE))
    assert_attention_close(assembled[sample].float().cpu().numpy(), np.concatenate(outs, 0))
