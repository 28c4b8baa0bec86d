"""K3b (csrc/paged_attn_mid.hip): paged attention for 49..128 query rows per kv head with two
workgroups per CU -- speculative verify with T > 12 and short prefill chunks over many sequences.

Same contract and oracle as the other paged kernels (oracle.paged_attention restates reference
src/attention/paged.rs:117-192; bottom-right causal mask :357-381).  The kernel is chosen by launch
size (from 1.5 (sequence, kv head) units per CU for up to 64 rows to 4 for 128 rows on contexts up to
2048 tokens); the tests force it with NVLLM_PAGED_MID=1 on small launches and assert the path.  Cases: every group size the row packing
sees (G = 1, 2, 4, 8 and a non-power-of-two 3), row counts at both ends of the range, every
alignment of the context against the 64-key tiles and the 16-token blocks, contexts shorter than
one tile, NaN in every slot no sequence owns, garbage ids behind a sequence's blocks.
Tolerance (north_star): 2e-2 max-abs, 1e-3 mean-abs.
"""
import math

import numpy as np
import pytest
import torch

import nano_vllm_rs_amd as nv
from helpers import assert_attention_close, bf16_randn, make_paged_case, oracle_paged_batch, pad_tables

pytestmark = pytest.mark.gpu


def _run(dev, q, kc, vc, tables, ctx_lens, scale):
    bt = torch.from_numpy(pad_tables(tables, None, -1)).to(dev)
    cl = torch.tensor(ctx_lens, dtype=torch.int32, device=dev)
    out = nv.paged_attention_batched(torch.from_numpy(q).to(dev, torch.bfloat16),
                                     torch.from_numpy(kc).to(dev, torch.bfloat16),
                                     torch.from_numpy(vc).to(dev, torch.bfloat16), bt, cl, scale)
    torch.cuda.synchronize()
    return out.float().cpu().numpy()


@pytest.mark.parametrize("Hq,Hkv,T", [(32, 8, 13), (32, 8, 16), (32, 8, 21), (32, 8, 32),
                                      (16, 8, 25), (16, 8, 64), (64, 8, 7), (64, 8, 16),
                                      (8, 8, 49), (8, 8, 128), (6, 2, 17), (6, 2, 42)])
def test_mid_kernel_matches_oracle(dev, policy, Hq, Hkv, T):
    D, bs = 128, 16
    policy.setenv("NVLLM_PAGED_MID", 1)
    rng = np.random.default_rng(9500 + Hq * 131 + T)
    ctx_lens = [T + c for c in (0, 1, 15, 16, 17, 31, 47, 48, 63, 64, 65, 100, 333, 1000)]
    kc, vc, tables = make_paged_case(rng, ctx_lens, 520, bs, Hkv, D, slack_blocks=1)
    assert nv.paged_attention_path(len(ctx_lens), T, Hq, Hkv, D, bs, max(len(t) for t in tables),
                                   kc.shape[0]) == "paged_mid"
    q = bf16_randn(rng, (len(ctx_lens), T, Hq, D))
    scale = 1.0 / math.sqrt(D)
    got = _run(dev, q, kc, vc, tables, ctx_lens, scale)
    assert_attention_close(got, oracle_paged_batch(q, kc, vc, tables, ctx_lens, bs, scale))


def test_mid_kernel_agrees_with_the_prefill_kernel_and_is_chosen_by_size(dev, policy):
    Hq, Hkv, D, bs, T = 32, 8, 128, 16, 16
    rng = np.random.default_rng(9600)
    N = 136                                            # 136 x 8 = 1088 units >= 4 x 256 CUs
    ctx_lens = [int(x) for x in rng.integers(T, 700, N)]
    kc, vc, tables = make_paged_case(rng, ctx_lens, 6400, bs, Hkv, D, slack_blocks=1)
    q = bf16_randn(rng, (N, T, Hq, D))
    scale = 1.0 / math.sqrt(D)
    mb = max(len(t) for t in tables)
    path = lambda n, t, blocks=mb: nv.paged_attention_path(n, t, Hq, Hkv, D, bs, blocks, kc.shape[0])
    assert path(N, T) == "paged_mid"                   # 64 rows, enough units
    assert path(8, T) == "k1_paged"                    # too few units: the prefill kernel and its key split
    assert path(48, T) == "paged_mid"                  # 1.5 units per CU: up to 64 rows
    assert path(48, 24) == "k1_paged"                  # ... 96 rows want 3 units per CU
    assert path(96, 24) == "paged_mid"
    assert path(N, 4) == "decode"                      # 16 rows
    assert path(N, 33) == "k1_paged"                   # 132 rows
    assert path(N, 24, 128) == "paged_mid"             # 96 rows
    assert path(N, 32, 128) == "paged_mid"             # 128 rows, the most it takes, contexts up to 2048 tokens
    assert path(N, 28, 129) == "k1_paged"              # 112 rows, longer contexts
    assert nv.capi.lib().nvllm_paged_attn_workspace_bytes(N, T, Hq, Hkv, D, bs, mb) == 0
    mid = _run(dev, q, kc, vc, tables, ctx_lens, scale)
    policy.setenv("NVLLM_PAGED_MID", 0)
    assert path(N, T) == "k1_paged"
    k1 = _run(dev, q, kc, vc, tables, ctx_lens, scale)
    assert np.isfinite(mid).all()
    d = np.abs(mid - k1)
    assert d.max() <= 2e-2 and d.mean() <= 1e-3
    assert_attention_close(mid, oracle_paged_batch(q, kc, vc, tables, ctx_lens, bs, scale))


def test_mid_kernel_context_shorter_than_the_chunk_and_empty(dev, policy):
    # ctx < T: the first T - ctx tokens see no key (zeros, as the other kernels give); ctx = 0: all zeros
    Hq, Hkv, D, bs, T = 32, 8, 128, 16, 20
    policy.setenv("NVLLM_PAGED_MID", 1)
    rng = np.random.default_rng(9700)
    ctx_lens = [0, 1, 5, 19, 20]
    kc, vc, tables = make_paged_case(rng, [max(c, 1) for c in ctx_lens], 64, bs, Hkv, D, slack_blocks=1)
    q = bf16_randn(rng, (len(ctx_lens), T, Hq, D))
    scale = 1.0 / math.sqrt(D)
    got = _run(dev, q, kc, vc, tables, ctx_lens, scale)
    assert np.isfinite(got).all()
    policy.setenv("NVLLM_PAGED_MID", 0)
    ref = _run(dev, q, kc, vc, tables, ctx_lens, scale)
    d = np.abs(got - ref)
    assert d.max() <= 2e-2 and d.mean() <= 1e-3
    assert not got[0].any()


@pytest.mark.parametrize("seed", range(8))
def test_mid_kernel_random_shapes_against_the_prefill_kernel(dev, policy, seed):
    # random group sizes, chunk lengths and contexts (empty, shorter than the chunk, one / two / many
    # tiles, tile and block boundaries): the mid kernel against K1 over the cache on the same launch,
    # and against the oracle on a sample of the sequences
    rng = np.random.default_rng(9800 + seed)
    D, bs = 128, 16
    Hkv = int(rng.choice([1, 2, 4, 8]))
    G = int(rng.choice([1, 2, 3, 4, 6, 8]))
    Hq = Hkv * G
    T = int(rng.integers(-(-33 // G), 128 // G + 1))
    edge = [0, 1, T - 1, T, T + 1, 63, 64, 65, 127, 128, 129, 16 * int(rng.integers(1, 40))]
    ctx_lens = [max(int(c), 0) for c in edge] + [int(x) for x in rng.integers(1, 900, 12)]
    kc, vc, tables = make_paged_case(rng, [max(c, 1) for c in ctx_lens], 1500, bs, Hkv, D, slack_blocks=1)
    q = bf16_randn(rng, (len(ctx_lens), T, Hq, D))
    scale = 1.0 / math.sqrt(D)
    policy.setenv("NVLLM_PAGED_MID", 1)
    assert nv.paged_attention_path(len(ctx_lens), T, Hq, Hkv, D, bs, max(len(t) for t in tables),
                                   kc.shape[0]) == "paged_mid", (Hq, Hkv, T)
    got = _run(dev, q, kc, vc, tables, ctx_lens, scale)
    again = _run(dev, q, kc, vc, tables, ctx_lens, scale)
    assert np.array_equal(got, again)                  # (the row assignment may differ from launch to launch: same bits)
    policy.setenv("NVLLM_PAGED_MID", 0)
    ref = _run(dev, q, kc, vc, tables, ctx_lens, scale)
    assert np.isfinite(got).all()
    d = np.abs(got - ref)
    assert d.max() <= 2e-2 and d.mean() <= 1e-3, (Hq, Hkv, T, float(d.max()))
    pick = [i for i, c in enumerate(ctx_lens) if c >= T][:6]
    want = oracle_paged_batch(q[pick], kc, vc, [tables[i] for i in pick], [ctx_lens[i] for i in pick], bs, scale)
    assert_attention_close(got[pick], want)


@pytest.mark.parametrize("T", [13, 16, 24, 32])
def test_rows_by_simd_and_rows_by_wave_give_the_same_bits(dev, policy, T):
    # which wave computes (and which waves fetch) a kv head's rows is decided by where the hardware put the
    # waves; NVLLM_PAGED_MID=2 takes the kernel's fallback (row block = wave index) -- every row's arithmetic
    # is the same instruction sequence either way, so the outputs must be identical
    Hq, Hkv, D, bs = 32, 8, 128, 16
    rng = np.random.default_rng(9900 + T)
    ctx_lens = [T + int(x) for x in rng.integers(0, 600, 40)]
    kc, vc, tables = make_paged_case(rng, ctx_lens, 1700, bs, Hkv, D, slack_blocks=1)
    q = bf16_randn(rng, (len(ctx_lens), T, Hq, D))
    scale = 1.0 / math.sqrt(D)
    policy.setenv("NVLLM_PAGED_MID", 1)
    a = _run(dev, q, kc, vc, tables, ctx_lens, scale)
    policy.setenv("NVLLM_PAGED_MID", 2)
    assert nv.paged_attention_path(len(ctx_lens), T, Hq, Hkv, D, bs, max(len(t) for t in tables),
                                   kc.shape[0]) == "paged_mid"
    b = _run(dev, q, kc, vc, tables, ctx_lens, scale)
    assert np.array_equal(a, b)
