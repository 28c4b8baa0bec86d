"""K1 parity: flash attention vs the CPU oracle.

Oracle: oracle.flash_attention_cpu (restates reference src/attention/flash.rs:137-275)
on the SAME bf16-rounded inputs in f32.  Tolerance (north_star): 2e-2 max-abs,
1e-3 mean-abs for bf16; the f32 generic kernel is held to the reference's own
1e-4 bound (tests/flash_attention_test.rs:209-212).
The first block of tests ports tests/flash_attention_test.rs one for one.
"""
import math

import numpy as np
import pytest
import torch

import nano_vllm_rs_amd as nv
import oracle
from helpers import assert_attention_close, bf16_randn

pytestmark = pytest.mark.gpu


def _dev(a, dev, dtype):
    return torch.from_numpy(np.ascontiguousarray(a)).to(device=dev, dtype=dtype)


def _flash(dev, q, k, v, causal, dtype, scale=None, return_lse=False):
    cfg = nv.FlashAttentionConfig.new(q.shape[-1], causal)
    if scale is not None:
        cfg = cfg.with_scale(scale)
    r = nv.flash_attention(_dev(q, dev, dtype), _dev(k, dev, dtype), _dev(v, dev, dtype), cfg,
                           return_lse=return_lse)
    torch.cuda.synchronize()
    if return_lse:
        return r[0].float().cpu().numpy(), r[1].cpu().numpy()
    return r.float().cpu().numpy()


# ---- ports of tests/flash_attention_test.rs ---------------------------------------------
def test_flash_attention_basic(dev):                       # :10-26
    rng = np.random.default_rng(1)
    q, k, v = (bf16_randn(rng, (1, 16, 4, 32)) for _ in range(3))
    out = _flash(dev, q, k, v, True, torch.float32)
    assert out.shape == (1, 16, 4, 32)


@pytest.mark.parametrize("shape", [(1, 8, 2, 16), (2, 32, 4, 32), (1, 64, 8, 64), (2, 128, 4, 64)])
def test_flash_attention_various_sizes(dev, shape):        # :28-54
    rng = np.random.default_rng(sum(shape))
    q, k, v = (bf16_randn(rng, shape) for _ in range(3))
    out = _flash(dev, q, k, v, True, torch.float32)
    assert out.shape == shape
    assert np.abs(out - oracle.flash_attention_cpu(q, k, v, True)).max() < 1e-4


def test_flash_attention_non_causal(dev):                  # :56-73
    rng = np.random.default_rng(3)
    q, k, v = (bf16_randn(rng, (1, 16, 2, 32)) for _ in range(3))
    out = _flash(dev, q, k, v, False, torch.float32)
    assert out.shape == (1, 16, 2, 32)
    assert np.abs(out - oracle.flash_attention_cpu(q, k, v, False)).max() < 1e-4


def test_flash_attention_gqa(dev):                         # :75-106
    rng = np.random.default_rng(4)
    q = bf16_randn(rng, (1, 16, 8, 32))
    k, v = bf16_randn(rng, (1, 16, 2, 32)), bf16_randn(rng, (1, 16, 2, 32))
    out = _flash(dev, q, k, v, True, torch.float32)
    assert out.shape == (1, 16, 8, 32)
    assert np.abs(out - oracle.flash_attention_cpu(q, k, v, True)).max() < 1e-4


def test_flash_attention_numerical_stability(dev):         # :108-131, sigma = 10
    rng = np.random.default_rng(5)
    q, k, v = (bf16_randn(rng, (1, 8, 2, 16), sigma=10.0) for _ in range(3))
    out = _flash(dev, q, k, v, True, torch.float32)
    assert np.isfinite(out).all()
    assert np.abs(out - oracle.flash_attention_cpu(q, k, v, True)).max() < 1e-3


def test_flash_attention_block_size_invariance(dev):       # :133-174
    rng = np.random.default_rng(6)
    q, k, v = (bf16_randn(rng, (1, 32, 2, 16)) for _ in range(3))
    outs = []
    for bq, bkv in [(8, 8), (16, 16), (4, 8), (8, 4)]:
        cfg = nv.FlashAttentionConfig.new(16, True).with_block_sizes(bq, bkv)
        o = nv.flash_attention(_dev(q, dev, torch.float32), _dev(k, dev, torch.float32),
                               _dev(v, dev, torch.float32), cfg)
        outs.append(o.cpu().numpy())
        ref = oracle.flash_attention_cpu(q, k, v, True, block_size_q=bq, block_size_kv=bkv)
        assert np.abs(outs[-1] - ref).max() < 1e-4
    for o in outs[1:]:
        assert np.abs(outs[0] - o).max() < 1e-4


def test_flash_attention_vs_naive_sdpa(dev):               # :176-213 (the reference's 1e-4 bound)
    rng = np.random.default_rng(7)
    q, k, v = (bf16_randn(rng, (1, 16, 4, 32)) for _ in range(3))
    out = _flash(dev, q, k, v, True, torch.float32)
    naive = oracle.naive_sdpa(q, k, v, 1.0 / np.sqrt(np.float32(32)), True)
    assert np.abs(out - naive).max() < 1e-4


def test_prefill_attention_matches_oracle(dev):            # tests/paged_attention_test.rs:12-108
    rng = np.random.default_rng(8)
    q = bf16_randn(rng, (1, 4, 4, 16))
    k, v = bf16_randn(rng, (1, 4, 2, 16)), bf16_randn(rng, (1, 4, 2, 16))
    scale = 1.0 / math.sqrt(16)
    out = nv.prefill_attention(_dev(q, dev, torch.float32), _dev(k, dev, torch.float32),
                               _dev(v, dev, torch.float32), 4, 2, scale, True)
    assert tuple(out.shape) == (1, 4, 4 * 16)
    want = oracle.prefill_attention(q, k, v, 4, 2, scale, True)
    assert np.abs(out.cpu().numpy() - want).max() < 1e-4
    # seq_len == 1: no mask even when causal (paged.rs:79)
    out1 = nv.prefill_attention(_dev(q[:, :1], dev, torch.float32), _dev(k[:, :1], dev, torch.float32),
                                _dev(v[:, :1], dev, torch.float32), 4, 2, scale, True)
    want1 = oracle.prefill_attention(q[:, :1], k[:, :1], v[:, :1], 4, 2, scale, True)
    assert np.abs(out1.cpu().numpy() - want1).max() < 1e-5


# ---- bf16 generic kernel (head_dim != 128) -----------------------------------------------
@pytest.mark.parametrize("shape,hkv", [((1, 16, 4, 32), 4), ((1, 16, 8, 32), 2), ((2, 128, 4, 64), 4)])
def test_bf16_generic(dev, shape, hkv):
    rng = np.random.default_rng(11)
    B, S, H, D = shape
    q = bf16_randn(rng, shape)
    k, v = bf16_randn(rng, (B, S, hkv, D)), bf16_randn(rng, (B, S, hkv, D))
    out = _flash(dev, q, k, v, True, torch.bfloat16)
    assert_attention_close(out, oracle.flash_attention_cpu(q, k, v, True))


# ---- MFMA path: bf16, head_dim 128 --------------------------------------------------------
@pytest.mark.parametrize("B,Sq,Hq,Hkv,causal", [
    (1, 256, 16, 8, True),      # Qwen3-0.6B head shape (SURVEY 8c golden shape)
    (1, 256, 16, 8, False),
    (2, 128, 4, 4, True),       # MHA
    (1, 64, 32, 8, True),       # Qwen3-4B head shape, one KV tile
    (1, 1, 16, 8, True),        # single row
    (1, 33, 16, 8, True),       # ragged: one full wave + one row
    (1, 100, 8, 2, True),       # ragged inside a tile
    (3, 129, 8, 8, True),       # one row into the second q tile
    (1, 200, 4, 1, False),      # MQA, non-causal, ragged
    (1, 577, 16, 8, True),      # several q tiles + ragged tail
])
def test_mfma_matches_oracle(dev, B, Sq, Hq, Hkv, causal):
    rng = np.random.default_rng(B * 1000 + Sq)
    q = bf16_randn(rng, (B, Sq, Hq, 128))
    k, v = bf16_randn(rng, (B, Sq, Hkv, 128)), bf16_randn(rng, (B, Sq, Hkv, 128))
    out = _flash(dev, q, k, v, causal, torch.bfloat16)
    assert_attention_close(out, oracle.flash_attention_cpu(q, k, v, causal))


@pytest.mark.parametrize("Sq,Skv,causal", [(64, 200, False), (200, 64, False), (96, 160, True),
                                           (160, 96, True)])
def test_mfma_unequal_lengths_top_left_mask(dev, Sq, Skv, causal):
    # flash.rs:296-306: the mask is top-left aligned whatever the lengths
    rng = np.random.default_rng(Sq + Skv)
    q = bf16_randn(rng, (1, Sq, 8, 128))
    k, v = bf16_randn(rng, (1, Skv, 4, 128)), bf16_randn(rng, (1, Skv, 4, 128))
    out = _flash(dev, q, k, v, causal, torch.bfloat16)
    assert_attention_close(out, oracle.flash_attention_cpu(q, k, v, causal))


def test_mfma_sharp_softmax_forces_rescale(dev):
    """A spiked K row makes the running max jump at a chosen tile (guide rule 26): the
    O/l rescale branch is exercised, not just bounded random data."""
    rng = np.random.default_rng(26)
    S = 384
    q = bf16_randn(rng, (1, S, 4, 128))
    k, v = bf16_randn(rng, (1, S, 2, 128)), bf16_randn(rng, (1, S, 2, 128))
    k[0, 200, :, :] = oracle.bf16_round(3.0 * q[0, 300, 0, :])   # key 200 spikes for query 300
    k[0, 70, :, :] = oracle.bf16_round(2.0 * q[0, 150, 2, :])
    out = _flash(dev, q, k, v, True, torch.bfloat16)
    want = oracle.flash_attention_cpu(q, k, v, True)
    assert_attention_close(out, want, max_tol=3e-2)


def test_mfma_large_sigma_finite(dev):
    rng = np.random.default_rng(27)
    q, k, v = (bf16_randn(rng, (1, 256, 4, 128), sigma=10.0) for _ in range(3))
    out = _flash(dev, q, k, v, True, torch.bfloat16)
    assert np.isfinite(out).all()
    want = oracle.flash_attention_cpu(q, k, v, True)
    assert np.abs(out - want).max() < 0.5          # |V| ~ 30+: bf16 output quantum is 0.125-0.25


def test_mfma_lse(dev):
    rng = np.random.default_rng(28)
    q = bf16_randn(rng, (1, 160, 4, 128))
    k, v = bf16_randn(rng, (1, 160, 2, 128)), bf16_randn(rng, (1, 160, 2, 128))
    out, lse = _flash(dev, q, k, v, True, torch.bfloat16, return_lse=True)
    scale = 1 / math.sqrt(128)
    kk = np.repeat(k, 2, axis=2)
    s = np.einsum("bihd,bjhd->bhij", q.astype(np.float64), kk.astype(np.float64)) * scale
    s = np.where(np.triu(np.ones((160, 160), bool), 1), -np.inf, s)
    ref = np.log(np.exp(s - s.max(-1, keepdims=True)).sum(-1)) + s.max(-1)
    assert np.abs(lse - ref).max() < 2e-3


def test_mfma_config2_full_size(dev):
    """BASELINE config 2 at full size (B=1, S=4096, 16 Q / 8 KV heads, hd128, causal):
    all heads against the oracle (the CPU restatement needs ~1 s for this)."""
    rng = np.random.default_rng(2)
    q = bf16_randn(rng, (1, 4096, 16, 128))
    k, v = bf16_randn(rng, (1, 4096, 8, 128)), bf16_randn(rng, (1, 4096, 8, 128))
    out = _flash(dev, q, k, v, True, torch.bfloat16)
    assert_attention_close(out, oracle.flash_attention_cpu(q, k, v, True))


def test_mfma_linearity_in_v(dev):
    """Size-independent property: attention is linear in V (holds for any S)."""
    g = torch.Generator(device="cpu").manual_seed(9)
    B, S, Hq, Hkv, D = 2, 2048, 16, 8, 128
    q = torch.randn((B, S, Hq, D), generator=g).to(dev, torch.bfloat16)
    k = torch.randn((B, S, Hkv, D), generator=g).to(dev, torch.bfloat16)
    v1 = torch.randn((B, S, Hkv, D), generator=g).to(dev, torch.bfloat16)
    v2 = torch.randn((B, S, Hkv, D), generator=g).to(dev, torch.bfloat16)
    cfg = nv.FlashAttentionConfig.new(D, True)
    o1 = nv.flash_attention(q, k, v1, cfg).float()
    o2 = nv.flash_attention(q, k, v2, cfg).float()
    o12 = nv.flash_attention(q, k, (v1.float() + v2.float()).to(torch.bfloat16), cfg).float()
    err = (o12 - (o1 + o2)).abs()
    assert float(err.max()) < 6e-2 and float(err.mean()) < 4e-3   # three bf16 roundings of |x| < 4
    # first query row attends to key 0 only: output equals V[0] exactly
    assert torch.equal(nv.flash_attention(q, k, v1, cfg)[:, 0, :, :].reshape(B, Hkv, 2, D)[:, :, 0],
                       v1[:, 0])


@pytest.mark.parametrize("seed", [21, 22])
def test_mfma_random_shapes_match_the_generic_kernel(dev, seed):
    """Randomised (B, Sq, Skv, heads, causal) through the MFMA kernel against the generic kernel on
    the same values in f32 (the generic kernel is pinned to the oracle above): every phase
    combination of the tile walk (both blocks, one block, masked last tiles in either, ragged
    rows and keys, balanced and adjacent tile pairs)."""
    rng = np.random.default_rng(seed)
    for case in range(30):
        Hkv = int(rng.choice([1, 2, 4, 8]))
        Hq = Hkv * int(rng.choice([1, 2, 4, 8]))
        B = int(rng.integers(1, 4))
        Sq = int(rng.choice([1, 31, 32, 33, 64, 127, 128, 129, 255, 256, 257, 300, 500, 777, 1030]))
        causal = bool(rng.integers(0, 2))
        Skv = Sq if rng.random() < 0.6 else int(rng.integers(1, 1100))
        g = torch.Generator(device="cpu").manual_seed(1000 * seed + case)
        q = torch.randn((B, Sq, Hq, 128), generator=g).to(dev, torch.bfloat16)
        k = torch.randn((B, Skv, Hkv, 128), generator=g).to(dev, torch.bfloat16)
        v = torch.randn((B, Skv, Hkv, 128), generator=g).to(dev, torch.bfloat16)
        cfg = nv.FlashAttentionConfig.new(128, causal)
        got = nv.flash_attention(q, k, v, cfg).float()
        want = nv.flash_attention(q.float(), k.float(), v.float(), cfg)
        err = (got - want).abs()
        what = dict(case=case, B=B, Sq=Sq, Skv=Skv, Hq=Hq, Hkv=Hkv, causal=causal)
        assert bool(torch.isfinite(got).all()), what
        assert float(err.max()) < 2e-2 and float(err.mean()) < 1e-3, (what, float(err.max()), float(err.mean()))
