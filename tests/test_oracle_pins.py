"""Pin the CPU oracle against everything the reference's own tests assert for this path
(SURVEY.md 8c).  Runs without a GPU.

Bit-exact integer pins: tests/block_test.rs:75-90,114-123, docs/paged_attention.md:88-95,
tests/paged_attention_test.rs:188-263.  Floating point pins: tests/flash_attention_test.rs
:109-131 (finite at sigma 10), :134-174 (block-size invariance < 1e-4), :177-213 (flash ==
naive SDPA < 1e-4).  paged_attention numerics are NOT pinned by any reference test (shape-only
on zero caches): pinned here by the write->paged == contiguous-SDPA identity
(docs/paged_attention.md:328-357) and an independent torch CPU SDPA cross-check.
"""
import math

import numpy as np
import pytest
import torch

import oracle


def test_siphash24_published_vector():
    # Aumasson & Bernstein, "SipHash: a fast short-input PRF", Appendix A
    k0 = int.from_bytes(bytes(range(8)), "little")
    k1 = int.from_bytes(bytes(range(8, 16)), "little")
    assert oracle.siphash(2, 4, k0, k1, bytes(range(15))) == 0xA129CA6149BE45E5


def test_hash_token_block_properties():          # tests/block_test.rs:93-111
    tokens = [1, 2, 3, 4, 5]
    h1 = oracle.hash_token_block(tokens)
    h2 = oracle.hash_token_block(tokens, 999)
    assert h1 != h2
    assert h2 == oracle.hash_token_block(tokens, 999)
    assert h1 != oracle.hash_token_block([1, 2, 3, 4, 6])
    # doc example block.rs:36-41: same tokens, chained parent -> different hash
    assert oracle.hash_token_block([1, 2, 3, 4], oracle.hash_token_block([1, 2, 3, 4])) != \
        oracle.hash_token_block([1, 2, 3, 4])


def test_compute_num_blocks_table():             # tests/block_test.rs:113-123
    for seq_len, want in [(0, 0), (1, 1), (15, 1), (16, 1), (17, 2), (32, 2), (35, 3), (100, 7)]:
        assert oracle.compute_num_blocks(seq_len, 16) == want


def test_slot_mapping_known_answers():
    s = oracle.get_slot_mapping([5, 12], 16, 20)   # tests/block_test.rs:75-90
    assert len(s) == 20 and (s[0], s[15], s[16], s[19]) == (80, 95, 192, 195)
    s = oracle.get_slot_mapping([5, 12, 3], 16, 35)  # docs/paged_attention.md:88-95
    assert (s[0], s[15], s[16], s[31], s[32], s[34]) == (80, 95, 192, 207, 48, 50)
    # positions past the last allocated block are dropped (block.rs:276)
    assert len(oracle.get_slot_mapping([5, 12], 16, 40)) == 32
    assert len(oracle.get_slot_mapping([], 16, 5)) == 0


def test_write_kv_to_cache_reference_case():     # tests/paged_attention_test.rs:188-263
    nb, bs, Hkv, D = 2, 4, 2, 4
    z = np.zeros((nb, bs, Hkv, D), np.float32)
    key = np.ones((1, 3, Hkv, D), np.float32)
    val = 2 * np.ones((1, 3, Hkv, D), np.float32)
    k, v = oracle.write_kv_to_cache(key, val, z, z, [0, 1, 4], bs)
    assert k.shape == (nb, bs, Hkv, D) and v.shape == (nb, bs, Hkv, D)
    assert (k[0, 0] == 1).all() and (k[1, 0] == 1).all() and (k[0, 2] == 0).all()
    assert (v[0, 1] == 2).all() and (v[1, 0] == 2).all()
    assert k.sum() == 3 * Hkv * D and (z == 0).all()      # functional: inputs untouched
    with pytest.raises(oracle.OracleConfigError):          # paged.rs:218-224
        oracle.write_kv_to_cache(key, val, z, z, [0, 1], bs)


def test_write_kv_last_writer_wins():
    z = np.zeros((2, 4, 1, 2), np.float32)
    key = np.arange(1, 7, dtype=np.float32).reshape(1, 3, 1, 2)
    k, _ = oracle.write_kv_to_cache(key, key, z, z, [5, 2, 5], 4)
    assert (k[1, 1, 0] == [5, 6]).all() and (k[0, 2, 0] == [3, 4]).all()


def test_flash_vs_naive_sdpa():                  # tests/flash_attention_test.rs:177-213
    rng = np.random.default_rng(0)
    q, k, v = (rng.standard_normal((1, 16, 4, 32), dtype=np.float32) for _ in range(3))
    f = oracle.flash_attention_cpu(q, k, v, True)
    n = oracle.naive_sdpa(q, k, v, 1 / np.sqrt(np.float32(32)), True)
    assert np.abs(f - n).max() < 1e-4


def test_flash_block_size_invariance():          # tests/flash_attention_test.rs:134-174
    rng = np.random.default_rng(1)
    q, k, v = (rng.standard_normal((1, 32, 2, 16), dtype=np.float32) for _ in range(3))
    outs = [oracle.flash_attention_cpu(q, k, v, True, block_size_q=a, block_size_kv=b)
            for a, b in [(8, 8), (16, 16), (4, 8), (8, 4)]]
    for o in outs[1:]:
        assert np.abs(outs[0] - o).max() < 1e-4


def test_flash_finite_at_sigma_10():             # tests/flash_attention_test.rs:109-131
    rng = np.random.default_rng(2)
    q, k, v = (10 * rng.standard_normal((1, 8, 2, 16), dtype=np.float32) for _ in range(3))
    assert np.isfinite(oracle.flash_attention_cpu(q, k, v, True)).all()


@pytest.mark.parametrize("shape", [(1, 8, 2, 16), (2, 32, 4, 32), (1, 64, 8, 64), (2, 128, 4, 64)])
def test_flash_shapes(shape):                    # tests/flash_attention_test.rs:28-54
    rng = np.random.default_rng(3)
    q, k, v = (rng.standard_normal(shape, dtype=np.float32) for _ in range(3))
    assert oracle.flash_attention_cpu(q, k, v, True).shape == shape


@pytest.mark.parametrize("causal", [True, False])
@pytest.mark.parametrize("Hq,Hkv", [(4, 4), (8, 2), (16, 8)])
def test_flash_and_prefill_vs_torch_sdpa(causal, Hq, Hkv):
    """Independent second opinion (torch CPU), GQA included."""
    rng = np.random.default_rng(4)
    B, S, D = 2, 70, 32
    q = rng.standard_normal((B, S, Hq, D), dtype=np.float32)
    k = rng.standard_normal((B, S, Hkv, D), dtype=np.float32)
    v = rng.standard_normal((B, S, Hkv, D), dtype=np.float32)
    tq, tk, tv = (torch.from_numpy(x).transpose(1, 2).double() for x in (q, k, v))
    tk = tk.repeat_interleave(Hq // Hkv, dim=1)
    tv = tv.repeat_interleave(Hq // Hkv, dim=1)
    ref = torch.nn.functional.scaled_dot_product_attention(tq, tk, tv, is_causal=causal)
    ref = ref.transpose(1, 2).numpy()
    assert np.abs(oracle.flash_attention_cpu(q, k, v, causal) - ref).max() < 1e-5
    p = oracle.prefill_attention(q, k, v, Hq, Hkv, 1 / math.sqrt(D), causal)
    assert p.shape == (B, S, Hq * D)
    assert np.abs(p.reshape(B, S, Hq, D) - ref).max() < 1e-5


def test_prefill_no_mask_for_single_token():     # paged.rs:79
    rng = np.random.default_rng(5)
    q, k, v = (rng.standard_normal((1, 1, 2, 8), dtype=np.float32) for _ in range(3))
    a = oracle.prefill_attention(q, k, v, 2, 2, 0.3, True)
    b = oracle.prefill_attention(q, k, v, 2, 2, 0.3, False)
    assert np.array_equal(a, b) and np.allclose(a.reshape(1, 1, 2, 8), v, atol=1e-6)


# ---- paged attention: shape tests of the reference + the numeric pins described above -----
@pytest.mark.parametrize("table,ctx,T", [([0, 1], 6, 1), ([0, 1], 8, 2), ([0, 2, 5], 10, 1),
                                         ([0, 1, 2], 12, 4)])
def test_paged_reference_shapes_on_zero_cache(table, ctx, T):   # tests/paged_attention_test.rs
    z = np.zeros((8, 4, 2, 8), np.float32)
    q = np.random.default_rng(6).standard_normal((1, T, 4, 8), dtype=np.float32)
    out = oracle.paged_attention(q, z, z, table, 4, ctx, 4, 2, 8, 1 / math.sqrt(8))
    assert out.shape == (1, T, 32) and (out == 0).all()


def test_paged_errors():
    z = np.zeros((4, 4, 2, 8), np.float32)
    q = np.zeros((1, 1, 4, 8), np.float32)
    with pytest.raises(oracle.OracleConfigError):    # paged.rs:133-137
        oracle.paged_attention(q, z, z, [], 4, 4, 4, 2, 8, 0.3)
    with pytest.raises(oracle.OracleTensorError):    # narrow beyond blocks (paged.rs:151)
        oracle.paged_attention(q, z, z, [0], 4, 5, 4, 2, 8, 0.3)
    with pytest.raises(oracle.OracleTensorError):    # index_select out of range
        oracle.paged_attention(q, z, z, [7], 4, 4, 4, 2, 8, 0.3)


@pytest.mark.parametrize("T", [1, 3, 4])
@pytest.mark.parametrize("Hq,Hkv", [(4, 2), (32, 8)])
def test_paged_round_trip_equals_contiguous_sdpa(T, Hq, Hkv):
    """write_kv_to_cache -> paged_attention == attention of the last T queries over the
    contiguous K/V with the bottom-right mask (paged.rs:357-381), via torch float64."""
    rng = np.random.default_rng(7 + T)
    bs, D, S, nb = 4, 16, 23, 12
    k = rng.standard_normal((1, S, Hkv, D), dtype=np.float32)
    v = rng.standard_normal((1, S, Hkv, D), dtype=np.float32)
    q = rng.standard_normal((1, T, Hq, D), dtype=np.float32)
    table = rng.permutation(nb)[: oracle.compute_num_blocks(S, bs)].tolist()
    slots = oracle.get_slot_mapping(table, bs, S)
    junk = 1e3 * np.ones((nb, bs, Hkv, D), np.float32)    # dead slots must not matter
    kc, vc = oracle.write_kv_to_cache(k, v, junk, junk, slots, bs)
    scale = 1 / math.sqrt(D)
    got = oracle.paged_attention(q, kc, vc, table, bs, S, Hq, Hkv, D, scale).reshape(1, T, Hq, D)
    tq = torch.from_numpy(q).transpose(1, 2).double()
    tk = torch.from_numpy(k).transpose(1, 2).double().repeat_interleave(Hq // Hkv, dim=1)
    tv = torch.from_numpy(v).transpose(1, 2).double().repeat_interleave(Hq // Hkv, dim=1)
    s = tq @ tk.transpose(-1, -2) * scale
    if T > 1:
        i = torch.arange(T)[:, None]
        j = torch.arange(S)[None, :]
        s = s.masked_fill(j > (S - T + i), float("-inf"))
    ref = (torch.softmax(s, -1) @ tv).transpose(1, 2).numpy()
    assert np.abs(got - ref).max() < 1e-5


def test_paged_single_token_has_no_mask_but_chunk_has():
    # T == 1 attends to everything; T > 1 masks the future of earlier new tokens (paged.rs:174-181)
    rng = np.random.default_rng(9)
    kc = rng.standard_normal((2, 4, 1, 8), dtype=np.float32)
    vc = rng.standard_normal((2, 4, 1, 8), dtype=np.float32)
    q = rng.standard_normal((1, 2, 1, 8), dtype=np.float32)
    both = oracle.paged_attention(q, kc, vc, [0, 1], 4, 8, 1, 1, 8, 0.5)
    last = oracle.paged_attention(q[:, 1:], kc, vc, [0, 1], 4, 8, 1, 1, 8, 0.5)
    first = oracle.paged_attention(q[:, :1], kc, vc, [0, 1], 4, 7, 1, 1, 8, 0.5)
    assert np.allclose(both[:, 1], last[:, 0], atol=1e-6)
    assert np.allclose(both[:, 0], first[:, 0], atol=1e-6)


def test_bf16_round_matches_torch():
    x = np.random.default_rng(10).standard_normal(1000, dtype=np.float32) * 100
    assert np.array_equal(oracle.bf16_round(x), torch.from_numpy(x).bfloat16().float().numpy())
