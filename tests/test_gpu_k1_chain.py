"""K1's item chain: a workgroup that works through several items must give, bit for bit, what one
workgroup per item gives.

The chain (flash_attn_mfma.hip, CHAIN) only changes WHO computes an item and when its Q rows and
first K/V tiles are requested (before the item in front writes its O); every item's arithmetic is
the same instruction sequence, so the outputs are compared with torch.equal, not a tolerance --
dense (flash.rs:137-275) and over the paged cache (paged.rs:117-192, mask :357-381), both tile
pairings, chains that do not divide the item count, paged items without rows (the grid is sized
for the worst alignment) in the middle of a chain, and contexts so short that an item has no key
tile.  NVLLM_K1_CHAIN = batch elements (sequences) per workgroup, 0 = one workgroup per item, the
reference run.
The unchained kernel itself is held against the oracle in test_gpu_flash_attention.py /
test_gpu_paged_prefill.py; one case here repeats that for a chained launch.
"""
import math

import numpy as np
import pytest
import torch

import nano_vllm_rs_amd as nv
from helpers import assert_attention_close, bf16_randn, make_paged_case, oracle_paged_batch, pad_tables

pytestmark = pytest.mark.gpu


def _dense(dev, q, k, v, lse=False):
    cfg = nv.FlashAttentionConfig.new(128, True)
    r = nv.flash_attention(q, k, v, cfg, return_lse=lse)
    torch.cuda.synchronize()
    return r


@pytest.mark.parametrize("chain", [2, 3, 4, 7])
@pytest.mark.parametrize("pairing", ["b", "a"])
def test_dense_chain_is_bit_identical(dev, policy, chain, pairing):
    B, S, Hq, Hkv = 3, 640, 8, 2           # 5 query tiles -> 3 pairs (one of them a lone tile): 72 items
    g = torch.Generator(device="cpu").manual_seed(11)
    q = torch.randn((B, S, Hq, 128), generator=g).to(dev, torch.bfloat16)
    k = torch.randn((B, S, Hkv, 128), generator=g).to(dev, torch.bfloat16)
    v = torch.randn((B, S, Hkv, 128), generator=g).to(dev, torch.bfloat16)
    policy.setenv("NVLLM_K1_PAIRING", pairing)
    policy.setenv("NVLLM_K1_CHAIN", "0")
    want, want_lse = _dense(dev, q, k, v, lse=True)
    policy.setenv("NVLLM_K1_CHAIN", str(chain))
    got, got_lse = _dense(dev, q, k, v, lse=True)
    assert torch.equal(got, want)
    assert torch.equal(got_lse, want_lse)
    got2 = _dense(dev, q, k, v)            # and again: nothing is left behind in LDS or registers
    assert torch.equal(got2, want)


def test_dense_chain_ragged_lengths(dev, policy):
    # Sq not a multiple of the tile, Skv != Sq (top-left mask): rows past Sq and keys past Skv
    g = torch.Generator(device="cpu").manual_seed(12)
    for Sq, Skv in ((200, 200), (333, 97), (96, 700)):
        q = torch.randn((4, Sq, 16, 128), generator=g).to(dev, torch.bfloat16)
        k = torch.randn((4, Skv, 4, 128), generator=g).to(dev, torch.bfloat16)
        v = torch.randn((4, Skv, 4, 128), generator=g).to(dev, torch.bfloat16)
        policy.setenv("NVLLM_K1_CHAIN", "0")
        want = _dense(dev, q, k, v)
        for chain in (2, 5):
            policy.setenv("NVLLM_K1_CHAIN", str(chain))
            assert torch.equal(_dense(dev, q, k, v), want), (Sq, Skv, chain)


def _paged(dev, q, kc, vc, tables, ctx_lens, scale):
    bt = torch.from_numpy(pad_tables(tables, None, -1)).to(dev)
    cl = torch.tensor(ctx_lens, dtype=torch.int32, device=dev)
    out = nv.paged_attention_batched(q, kc, vc, bt, cl, scale)
    torch.cuda.synchronize()
    return out


@pytest.mark.parametrize("T", [16, 40, 64, 200])
def test_paged_chain_is_bit_identical(dev, policy, T):
    Hq, Hkv, D, bs = 32, 8, 128, 16
    rng = np.random.default_rng(8100 + T)
    # every alignment class, contexts barely longer than the chunk (items without a key tile for
    # some waves), long ones, and enough sequences that a chain crosses sequences and kv heads
    ctx_lens = [T + c for c in (0, 1, 15, 16, 17, 31, 32, 33, 47, 63, 64, 65, 100, 333, 700, 1500)]
    kc, vc, tables = make_paged_case(rng, ctx_lens, 1200, bs, Hkv, D, slack_blocks=1)
    q = bf16_randn(rng, (len(ctx_lens), T, Hq, D))
    scale = 1.0 / math.sqrt(D)
    qd = torch.from_numpy(q).to(dev, torch.bfloat16)
    kcd = torch.from_numpy(kc).to(dev, torch.bfloat16)
    vcd = torch.from_numpy(vc).to(dev, torch.bfloat16)
    policy.setenv("NVLLM_K1_KSPLIT", "1")
    assert nv.paged_attention_path(len(ctx_lens), T, Hq, Hkv, D, bs, max(len(t) for t in tables),
                                   kc.shape[0]) == "k1_paged"
    for pairing in ("b", "a"):
        policy.setenv("NVLLM_K1_PAIRING", pairing)
        policy.setenv("NVLLM_K1_CHAIN", "0")
        want = _paged(dev, qd, kcd, vcd, tables, ctx_lens, scale)
        for chain in (2, 3, 4, 9):
            policy.setenv("NVLLM_K1_CHAIN", str(chain))
            got = _paged(dev, qd, kcd, vcd, tables, ctx_lens, scale)
            assert torch.equal(got, want), (pairing, chain)
    if T in (16, 200):
        policy.setenv("NVLLM_K1_CHAIN", "4")
        got = _paged(dev, qd, kcd, vcd, tables, ctx_lens, scale).float().cpu().numpy()
        assert_attention_close(got, oracle_paged_batch(q, kc, vc, tables, ctx_lens, bs, scale))


def test_dense_chain_is_the_default_once_there_are_four_rounds_of_items(dev, policy):
    # B = 8, S = 2048, 16 heads: 8 pairs x 16 heads x 8 = 1024 items on 256 CUs -> chained by default
    # (chains of 2); the result must not depend on it
    g = torch.Generator(device="cpu").manual_seed(13)
    q = torch.randn((8, 2048, 16, 128), generator=g).to(dev, torch.bfloat16)
    k = torch.randn((8, 2048, 8, 128), generator=g).to(dev, torch.bfloat16)
    v = torch.randn((8, 2048, 8, 128), generator=g).to(dev, torch.bfloat16)
    default = _dense(dev, q, k, v)
    policy.setenv("NVLLM_K1_CHAIN", "0")
    plain = _dense(dev, q, k, v)
    assert torch.equal(default, plain)


@pytest.mark.parametrize("seed", range(6))
def test_dense_chain_random_shapes(dev, policy, seed):
    # random batch / length / head shapes, forced chain lengths: bit-identical to one workgroup per item
    rng = np.random.default_rng(9300 + seed)
    B = int(rng.integers(2, 9))
    Sq = int(rng.integers(1, 900))
    Skv = Sq if rng.random() < 0.7 else int(rng.integers(1, 900))
    Hkv = int(rng.choice([1, 2, 4, 8]))
    Hq = Hkv * int(rng.choice([1, 2, 4]))
    g = torch.Generator(device="cpu").manual_seed(9300 + seed)
    q = torch.randn((B, Sq, Hq, 128), generator=g).to(dev, torch.bfloat16)
    k = torch.randn((B, Skv, Hkv, 128), generator=g).to(dev, torch.bfloat16)
    v = torch.randn((B, Skv, Hkv, 128), generator=g).to(dev, torch.bfloat16)
    policy.setenv("NVLLM_K1_CHAIN", "0")
    want = _dense(dev, q, k, v)
    assert torch.isfinite(want.float()).all()
    for chain in (2, int(rng.integers(3, 9))):
        policy.setenv("NVLLM_K1_CHAIN", str(chain))
        assert torch.equal(_dense(dev, q, k, v), want), (B, Sq, Skv, Hq, Hkv, chain)
