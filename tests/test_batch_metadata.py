"""batched step metadata.  The flat arrays the batched kernels take must be exactly
what the reference's per-sequence path would use: block ids from BlockTable
(block.rs:183-289), slots from get_slot_mapping (block.rs:269-283) for the new tokens only.
CPU only (host logic through the C ABI)."""
import numpy as np
import pytest

import nano_vllm_rs_amd as nv


def _table(ids, bs=16):
    t = nv.BlockTable(bs)
    for b in ids:
        t.append_block(b)
    return t


def test_matches_per_sequence_slot_mapping():
    rng = np.random.default_rng(0)
    bs = 16
    batch = nv.BatchMetadata(8)
    want_slots, want_ctx, want_tables, starts, want_pos = [], [], [], [0], []
    for _ in range(9):
        nblk = int(rng.integers(1, 9))
        ids = rng.choice(1000, size=nblk, replace=False).tolist()
        t = _table(ids, bs)
        ctx = int(rng.integers(1, nblk * bs + 1))
        new = int(rng.integers(0, min(ctx, 40) + 1))
        batch.add_sequence(t, ctx, new)
        full = t.get_slot_mapping(ctx)          # the reference's mapping of the whole context
        want_slots += full[ctx - new:]
        want_pos += list(range(ctx - new, ctx))   # rope.rs:107-113: start_pos + i
        want_ctx.append(ctx)
        want_tables.append(ids + [0] * (8 - nblk))
        starts.append(starts[-1] + new)
    assert batch.num_seqs() == 9 and batch.num_tokens() == len(want_slots)
    assert batch.slot_mapping().tolist() == want_slots
    assert batch.context_lens().tolist() == want_ctx
    assert batch.block_tables().tolist() == want_tables
    assert batch.query_start().tolist() == starts
    assert batch.positions().tolist() == want_pos and batch.positions().dtype == np.int32
    assert batch.block_tables().dtype == np.int32 and batch.slot_mapping().dtype == np.int64


def test_reference_known_answers():
    # tests/block_test.rs:75-90: table [5,12], bs 16, 20 tokens -> 80, 95, 192, 195
    b = nv.BatchMetadata(2)
    b.add_sequence(_table([5, 12]), 20, 20)
    s = b.slot_mapping()
    assert (s[0], s[15], s[16], s[19]) == (80, 95, 192, 195)
    # decode step of the same sequence: one new token at position 19
    b.clear()
    b.add_sequence(_table([5, 12]), 20, 1)
    assert b.slot_mapping().tolist() == [195] and b.query_start().tolist() == [0, 1]


def test_errors_and_clear():
    b = nv.BatchMetadata(2)
    with pytest.raises(nv.ConfigError):
        b.add_sequence(_table([1, 2]), 33, 1)      # context beyond 2 blocks (paged.rs:150 narrow)
    with pytest.raises(nv.ConfigError):
        b.add_sequence(_table([1, 2]), 10, 11)     # more new tokens than context
    with pytest.raises(nv.ConfigError):
        b.add_sequence(_table([1, 2, 3]), 10, 1)   # table longer than max_blocks_per_seq
    assert b.num_seqs() == 0 and b.num_tokens() == 0  # failed adds leave nothing behind
    b.add_sequence(_table([7]), 3, 3)
    assert b.num_seqs() == 1
    b.clear()
    assert b.num_seqs() == 0 and b.query_start().tolist() == [0]
    with pytest.raises(nv.ConfigError):
        nv.BatchMetadata(0)
