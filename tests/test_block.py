"""Ports of the reference's tests/block_test.rs against the product host code (C++ behind the
C ABI) and, for the integer results, against the CPU oracle (bit-exact).  No GPU needed."""
import numpy as np
import pytest

import nano_vllm_rs_amd as nv
import oracle


def test_block_table_basic():                 # block_test.rs:43-56
    t = nv.BlockTable(16)
    assert t.is_empty() and t.num_blocks() == 0
    for b in (5, 12, 3):
        t.append_block(b)
    assert not t.is_empty() and t.num_blocks() == 3
    assert t.get_physical_block_ids() == [5, 12, 3]
    assert t.block_size() == 16
    t.clear()
    assert t.is_empty()


def test_block_table_get_block_id():          # block_test.rs:58-72
    t = nv.BlockTable(16)
    for b in (5, 12, 3):
        t.append_block(b)
    assert [t.get_block_id(i) for i in range(3)] == [5, 12, 3]
    with pytest.raises(nv.BlockIndexOutOfBoundsError) as e:
        t.get_block_id(3)
    assert "logical block 3 not allocated (table has 3 blocks)" in str(e.value)   # error.rs:16
    with pytest.raises(nv.BlockIndexOutOfBoundsError):
        t.get_block_id(100)


def test_block_table_slot_mapping():          # block_test.rs:74-90
    t = nv.BlockTable(16)
    t.append_block(5)
    t.append_block(12)
    s = t.get_slot_mapping(20)
    assert len(s) == 20 and (s[0], s[15], s[16], s[19]) == (80, 95, 192, 195)


def test_slot_mapping_doc_example():          # docs/paged_attention.md:88-95
    t = nv.BlockTable(16)
    for b in (5, 12, 3):
        t.append_block(b)
    s = t.get_slot_mapping(35)
    assert (s[0], s[15], s[16], s[31], s[32], s[34]) == (80, 95, 192, 207, 48, 50)


def test_slot_mapping_drops_unallocated_positions():   # block.rs:276
    t = nv.BlockTable(16)
    t.append_block(7)
    assert len(t.get_slot_mapping(40)) == 16
    assert nv.BlockTable(16).get_slot_mapping(5) == []


def test_hash_token_block():                  # block_test.rs:92-111
    tokens = [1, 2, 3, 4, 5]
    h1 = nv.hash_token_block(tokens)
    h2 = nv.hash_token_block(tokens, 999)
    assert h1 != h2 and h2 == nv.hash_token_block(tokens, 999)
    assert h1 != nv.hash_token_block([1, 2, 3, 4, 6])


def test_compute_num_blocks():                # block_test.rs:113-123
    for seq_len, want in [(0, 0), (1, 1), (15, 1), (16, 1), (17, 2), (32, 2), (35, 3), (100, 7)]:
        assert nv.compute_num_blocks(seq_len, 16) == want


def test_integer_paths_bit_exact_vs_oracle():
    rng = np.random.default_rng(0)
    for _ in range(200):
        bs = int(rng.choice([1, 3, 4, 16, 32]))
        n = int(rng.integers(0, 40))
        ids = rng.integers(0, 1 << 21, size=n).tolist()      # config-5 sized block ids
        seq_len = int(rng.integers(0, (n + 2) * bs + 1))
        t = nv.BlockTable(bs)
        for b in ids:
            t.append_block(b)
        assert t.get_slot_mapping(seq_len) == oracle.get_slot_mapping(ids, bs, seq_len).tolist()
        assert nv.compute_num_blocks(seq_len, bs) == oracle.compute_num_blocks(seq_len, bs)
        toks = rng.integers(0, 1 << 32, size=int(rng.integers(0, 40)), dtype=np.uint64).astype(np.uint32)
        parent = None if rng.random() < 0.5 else int(rng.integers(0, 1 << 63))
        assert nv.hash_token_block(toks, parent) == oracle.hash_token_block(toks, parent)


def test_slot_mapping_last_writer_helper():
    import ctypes
    slots = np.array([3, 5, 3, 7, 5, 5, 0, 3, 15, -1], dtype=np.int64)
    out = np.empty_like(slots)
    p = ctypes.POINTER(ctypes.c_int64)
    n = nv.capi.lib().nvllm_slot_mapping_last_writer(slots.ctypes.data_as(p), slots.size,
                                                     out.ctypes.data_as(p))
    assert n == 4
    assert out.tolist() == [-1, -1, -1, 7, -1, 5, 0, 3, 15, -1]
