"""K4 parity: reshape_and_cache / write_kv_to_cache vs the CPU oracle, BIT-EXACT.

Oracle: oracle.write_kv_to_cache (restates reference src/attention/paged.rs:208-320).
Reference known-answer case ported from tests/paged_attention_test.rs:188-263.
"""
import numpy as np
import pytest
import torch

import nano_vllm_rs_amd as nv
import oracle
from helpers import bf16_randn

pytestmark = pytest.mark.gpu


def _to_dev(a, dev, dtype):
    return torch.from_numpy(np.ascontiguousarray(a)).to(device=dev, dtype=dtype)


def _bits(t):
    """Exact bit pattern of a tensor as a numpy integer array."""
    if t.dtype == torch.bfloat16:
        return t.view(torch.int16).cpu().numpy()
    return t.view(torch.int32).cpu().numpy()


def _bits_of_f32_as(a, dtype):
    t = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dtype)
    return _bits(t)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_reference_write_kv_to_cache_case(dev, dtype):
    # tests/paged_attention_test.rs:188-263: 2 blocks x 4 slots, Hkv 2, D 4; slots [0,1,4]
    nb, bs, Hkv, D = 2, 4, 2, 4
    kc = torch.zeros((nb, bs, Hkv, D), dtype=dtype, device=dev)
    vc = torch.zeros_like(kc)
    key = torch.ones((1, 3, Hkv, D), dtype=dtype, device=dev)
    value = torch.ones((1, 3, Hkv, D), dtype=dtype, device=dev) * 2
    nk, nvv = nv.write_kv_to_cache(key, value, kc, vc, [0, 1, 4], bs)
    assert tuple(nk.shape) == (nb, bs, Hkv, D) and tuple(nvv.shape) == (nb, bs, Hkv, D)
    k = nk.float().cpu().numpy()
    v = nvv.float().cpu().numpy()
    assert (k[0, 0] == 1).all() and (k[1, 0] == 1).all() and (k[0, 1] == 1).all()
    assert (k[0, 2] == 0).all() and (k[0, 3] == 0).all() and (k[1, 1:] == 0).all()
    assert (v[0, 0] == 2).all() and (v[1, 0] == 2).all() and (v[0, 2] == 0).all()
    ok, ov = oracle.write_kv_to_cache(np.ones((1, 3, Hkv, D), np.float32),
                                      2 * np.ones((1, 3, Hkv, D), np.float32),
                                      np.zeros((nb, bs, Hkv, D), np.float32),
                                      np.zeros((nb, bs, Hkv, D), np.float32), [0, 1, 4], bs)
    assert np.array_equal(k, ok) and np.array_equal(v, ov)


def test_slot_mapping_length_mismatch_is_config_error(dev):
    # paged.rs:218-224
    kc = torch.zeros((2, 4, 2, 4), dtype=torch.float32, device=dev)
    key = torch.ones((1, 3, 2, 4), dtype=torch.float32, device=dev)
    with pytest.raises(nv.ConfigError):
        nv.write_kv_to_cache(key, key, kc, kc.clone(), [0, 1], 4)
    with pytest.raises(oracle.OracleConfigError):
        z = np.zeros((2, 4, 2, 4), np.float32)
        oracle.write_kv_to_cache(np.ones((1, 3, 2, 4), np.float32), np.ones((1, 3, 2, 4), np.float32),
                                 z, z, [0, 1], 4)


def test_slot_outside_cache_is_tensor_error(dev):
    kc = torch.zeros((2, 4, 2, 4), dtype=torch.float32, device=dev)
    key = torch.ones((1, 1, 2, 4), dtype=torch.float32, device=dev)
    with pytest.raises(nv.TensorError):
        nv.write_kv_to_cache(key, key, kc, kc.clone(), [8], 4)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("S,table,bs,Hkv,D", [
    (20, [5, 12], 16, 8, 128),          # tests/block_test.rs:75-90 table, Qwen3 row shape
    (35, [5, 12, 3], 16, 8, 128),       # docs/paged_attention.md:88-95
    (128, list(range(7, -1, -1)), 16, 8, 128),   # config-1 prompt, 8 blocks
    (10, [0, 2, 5], 4, 2, 8),           # tests/paged_attention_test.rs:266-301 geometry
    (7, [3, 1], 4, 3, 5),               # odd row size (not a multiple of 16 bytes)
])
def test_write_kv_matches_oracle_bit_exact(dev, dtype, S, table, bs, Hkv, D):
    rng = np.random.default_rng(1234 + S)
    nb = max(table) + 2
    key = bf16_randn(rng, (1, S, Hkv, D))
    value = bf16_randn(rng, (1, S, Hkv, D))
    kc0 = bf16_randn(rng, (nb, bs, Hkv, D))   # pre-existing contents must survive elsewhere
    vc0 = bf16_randn(rng, (nb, bs, Hkv, D))
    bt = nv.BlockTable(bs)
    for b in table:
        bt.append_block(b)
    slots = bt.get_slot_mapping(S)
    assert slots == oracle.get_slot_mapping(table, bs, S).tolist()
    want_k, want_v = oracle.write_kv_to_cache(key, value, kc0, vc0, slots, bs)
    kc, vc = _to_dev(kc0, dev, dtype), _to_dev(vc0, dev, dtype)
    nv.write_kv_to_cache(_to_dev(key, dev, dtype), _to_dev(value, dev, dtype), kc, vc, slots, bs)
    torch.cuda.synchronize()
    assert np.array_equal(_bits(kc), _bits_of_f32_as(want_k, dtype))
    assert np.array_equal(_bits(vc), _bits_of_f32_as(want_v, dtype))


def test_duplicate_slots_last_writer_wins(dev):
    # sequential semantics of paged.rs:231-262
    rng = np.random.default_rng(7)
    nb, bs, Hkv, D = 4, 4, 2, 8
    S = 9
    slots = [3, 5, 3, 7, 5, 5, 0, 3, 15]
    key, value = bf16_randn(rng, (1, S, Hkv, D)), bf16_randn(rng, (1, S, Hkv, D))
    z = np.zeros((nb, bs, Hkv, D), np.float32)
    want_k, want_v = oracle.write_kv_to_cache(key, value, z, z, slots, bs)
    kc = torch.zeros((nb, bs, Hkv, D), dtype=torch.bfloat16, device=dev)
    vc = torch.zeros_like(kc)
    for _ in range(5):   # a racy scatter would flip between runs
        kc.zero_(); vc.zero_()
        nv.write_kv_to_cache(_to_dev(key, dev, torch.bfloat16), _to_dev(value, dev, torch.bfloat16),
                             kc, vc, slots, bs)
        assert np.array_equal(kc.float().cpu().numpy(), want_k)
        assert np.array_equal(vc.float().cpu().numpy(), want_v)


@pytest.mark.parametrize("num_tokens", [1, 256, 4096])
def test_batched_reshape_and_cache_bf16(dev, num_tokens):
    # SURVEY 8(d) KV-write sizes: 256 (decode batch) and 4096 (prefill) tokens, Hkv 8, D 128
    rng = np.random.default_rng(num_tokens)
    nb, bs, Hkv, D = 512, 16, 8, 128
    slots = rng.permutation(nb * bs)[:num_tokens].astype(np.int64)
    slots_with_skip = slots.copy()
    if num_tokens > 4:
        slots_with_skip[3] = -1                      # skip marker
    key, value = bf16_randn(rng, (num_tokens, Hkv, D)), bf16_randn(rng, (num_tokens, Hkv, D))
    guard = 2    # guard blocks either side, filled with a sentinel that must survive
    kc = torch.full((nb + 2 * guard, bs, Hkv, D), 768.0, dtype=torch.bfloat16, device=dev)
    vc = torch.full_like(kc, 768.0)
    kcv, vcv = kc[guard:guard + nb], vc[guard:guard + nb]
    kcv.zero_(); vcv.zero_()
    nv.reshape_and_cache(_to_dev(key, dev, torch.bfloat16), _to_dev(value, dev, torch.bfloat16),
                         kcv, vcv, torch.from_numpy(slots_with_skip).to(dev))
    torch.cuda.synchronize()
    keep = slots_with_skip >= 0
    z = np.zeros((nb, bs, Hkv, D), np.float32)
    want_k, want_v = oracle.write_kv_to_cache(key[keep][None], value[keep][None], z, z,
                                              slots_with_skip[keep], bs)
    assert np.array_equal(kcv.float().cpu().numpy(), want_k)
    assert np.array_equal(vcv.float().cpu().numpy(), want_v)
    for g in (kc[:guard], kc[guard + nb:], vc[:guard], vc[guard + nb:]):
        assert bool((g.float() == 768.0).all())


def test_round_trip_gather_recovers_tokens(dev):
    # size-independent property at a larger size: write then read back through the table
    rng = np.random.default_rng(99)
    nb, bs, Hkv, D = 4096, 16, 8, 128
    S = 16384
    ids = rng.permutation(nb)[: S // bs]
    bt = nv.BlockTable(bs)
    for b in ids:
        bt.append_block(int(b))
    slots = torch.tensor(bt.get_slot_mapping(S), dtype=torch.int64, device=dev)
    key = torch.randn((S, Hkv, D), device=dev).to(torch.bfloat16)
    value = torch.randn((S, Hkv, D), device=dev).to(torch.bfloat16)
    kc = torch.zeros((nb, bs, Hkv, D), dtype=torch.bfloat16, device=dev)
    vc = torch.zeros_like(kc)
    nv.reshape_and_cache(key, value, kc, vc, slots)
    back_k = kc.view(nb * bs, Hkv, D)[slots]
    back_v = vc.view(nb * bs, Hkv, D)[slots]
    assert torch.equal(back_k, key) and torch.equal(back_v, value)
    assert int((kc.view(nb * bs, -1).abs().sum(dim=1) > 0).sum()) == S
