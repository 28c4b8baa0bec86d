"""Sequence + Scheduler (src/core/sequence.rs, src/scheduler/batch.rs).  CPU only.

Every test of the reference's tests/scheduler_test.rs and tests/sequence_test.rs is ported and run
against BOTH the product (C++ behind nvllm_scheduler_*) and the oracle restatement
(oracle/scheduler_oracle.py), which pins the oracle on the reference's own assertions; a seeded
random trace then requires product == oracle decision for decision (ids, chunk sizes, physical
block ids: integer work, bit-exact).  The product's two opt-in extensions have no reference
counterpart and are checked on properties at the end.
"""
import numpy as np
import pytest

import nano_vllm_rs_amd as nv
import oracle
from nano_vllm_rs_amd import scheduler as S


def test_config(**kw):
    base = dict(max_num_seqs=4, max_prefill_tokens=100, enable_chunked_prefill=False, chunk_size=512,
                enable_priority=True, enable_preemption=False)   # scheduler_test.rs:7-16
    base.update(kw)
    return base


test_config.__test__ = False


class Product:
    """The product scheduler behind the interface the ported tests use."""

    def __init__(self, cfg, block_size, num_blocks):
        self.s = nv.Scheduler(nv.SchedulerConfig(**cfg), block_size, num_blocks)

    def add(self, seq_id, prompt, priority=0):
        self.s.add_sequence(seq_id, prompt, priority)

    def schedule(self):
        return self.s.schedule()

    def seq(self, seq_id):
        v = self.s.get_sequence(seq_id)
        if v is None:
            return None
        return dict(status=v.status(), finish_reason=v.finish_reason(), output_len=v.output_len(),
                    output_token_ids=v.output_token_ids(), prefilled=v.num_prefilled_tokens(),
                    complete=v.is_prefill_complete(), blocks=v.block_ids(), total_len=v.total_len())

    def used_blocks(self):
        return self.s.num_used_blocks()

    def __getattr__(self, name):
        return getattr(self.s, name)


class Oracle:
    def __init__(self, cfg, block_size, num_blocks):
        cfg = {k: v for k, v in cfg.items() if k not in ("reschedule_partial_prefill", "enable_prefix_caching")}
        self.s = oracle.SchedulerOracle(block_size=block_size, num_blocks=num_blocks, **cfg)

    def add(self, seq_id, prompt, priority=0):
        self.s.add_sequence(oracle.SequenceOracle(seq_id, prompt, priority))

    def schedule(self):
        return self.s.schedule()

    def seq(self, seq_id):
        q = self.s.sequences.get(seq_id)
        if q is None:
            return None
        return dict(status=q.status, finish_reason=q.finish_reason, output_len=q.output_len(),
                    output_token_ids=list(q.output_token_ids), prefilled=q.num_prefilled_tokens,
                    complete=q.is_prefill_complete(), blocks=list(q.block_ids), total_len=q.total_len())

    def used_blocks(self):
        return self.s.bm.num_used_blocks()

    def remove_sequence(self, seq_id):
        return self.s.remove_sequence(seq_id) is not None

    def __getattr__(self, name):
        return getattr(self.s, name)


@pytest.fixture(params=[Product, Oracle], ids=["product", "oracle"])
def impl(request):
    return request.param


# ---- tests/scheduler_test.rs, in order -----------------------------------------------------
def test_scheduler_creation(impl):
    s = impl(test_config(), 16, 100)
    assert (s.num_waiting(), s.num_running(), s.num_swapped()) == (0, 0, 0)
    assert not s.has_unfinished_sequences()


def test_add_sequence(impl):
    s = impl(test_config(), 16, 100)
    s.add(1, [1, 2, 3, 4])
    assert s.num_waiting() == 1 and s.has_unfinished_sequences()


def test_basic_scheduling(impl):
    s = impl(test_config(), 16, 100)
    s.add(1, [1, 2, 3, 4])
    out = s.schedule()
    assert out.prefill_sequences == [1] and out.decode_sequences == []
    assert s.num_running() == 1 and s.num_waiting() == 0


def test_multiple_sequences(impl):
    s = impl(test_config(), 16, 100)
    for i in (1, 2, 3):
        s.add(i, [1, 2, 3, 4])
    out = s.schedule()
    assert len(out.prefill_sequences) == 3 and s.num_running() == 3


def test_max_num_seqs_limit(impl):
    s = impl(test_config(max_num_seqs=2), 16, 100)
    for i in (1, 2, 3):
        s.add(i, [1, 2, 3, 4])
    out = s.schedule()
    assert len(out.prefill_sequences) == 2
    assert s.num_running() == 2 and s.num_waiting() == 1


def test_max_prefill_tokens_limit(impl):
    s = impl(test_config(max_prefill_tokens=10), 16, 100)
    s.add(1, list(range(1, 9)))
    s.add(2, list(range(1, 9)))
    out = s.schedule()
    assert len(out.prefill_sequences) == 1 and out.num_prefill_tokens == 8
    assert s.num_running() == 1 and s.num_waiting() == 1


def test_priority_scheduling(impl):
    s = impl(test_config(), 16, 100)
    s.add(1, [1, 2, 3, 4], -10)
    s.add(3, [9, 10, 11, 12])
    s.add(2, [5, 6, 7, 8], 10)
    assert s.schedule().prefill_sequences == [2, 3, 1]


def test_decode_scheduling(impl):
    s = impl(test_config(), 16, 100)
    s.add(1, [1, 2, 3, 4])
    s.schedule()
    s.mark_prefilled(1, 4)
    out = s.schedule()
    assert out.prefill_sequences == [] and out.decode_sequences == [1] and out.num_decode_tokens == 1


def test_mixed_prefill_decode(impl):
    s = impl(test_config(), 16, 100)
    s.add(1, [1, 2, 3, 4])
    s.schedule()
    s.mark_prefilled(1, 4)
    s.add(2, [5, 6, 7, 8])
    out = s.schedule()
    assert out.decode_sequences == [1] and out.prefill_sequences == [2]


def test_finish_sequence(impl):
    s = impl(test_config(), 16, 100)
    s.add(1, [1, 2, 3, 4])
    s.schedule()
    s.finish_sequence(1, S.END_OF_SEQUENCE)
    assert s.num_running() == 0
    q = s.seq(1)
    assert q["status"] == S.FINISHED and q["finish_reason"] == S.END_OF_SEQUENCE


def test_remove_sequence(impl):
    s = impl(test_config(), 16, 100)
    s.add(1, [1, 2, 3, 4])
    s.schedule()
    assert s.remove_sequence(1)
    assert s.num_running() == 0 and s.seq(1) is None
    assert not s.remove_sequence(1)


def test_append_token(impl):
    s = impl(test_config(), 16, 100)
    s.add(1, [1, 2, 3, 4])
    s.schedule()
    s.append_token(1, 100)
    q = s.seq(1)
    assert q["output_len"] == 1 and q["output_token_ids"] == [100]


def test_chunked_prefill(impl):
    s = impl(test_config(enable_chunked_prefill=True, chunk_size=4), 16, 100)
    s.add(1, list(range(1, 9)))
    out = s.schedule()
    assert out.prefill_sequences == [1] and out.prefill_chunk_sizes[1] == 4
    s.mark_prefilled(1, 4)
    assert not s.seq(1)["complete"]


def test_preemption(impl):
    s = impl(test_config(enable_preemption=True, max_num_seqs=10), 16, 4)   # only 4 blocks
    s.add(1, list(range(64)), -10)
    s.schedule()
    s.add(2, [1, 2, 3, 4], 10)
    out = s.schedule()
    assert 1 in out.preempted_sequences and s.num_swapped() == 1
    assert s.seq(1)["status"] == S.SWAPPED and s.seq(1)["blocks"] == []


def test_scheduler_reset(impl):
    s = impl(test_config(), 16, 100)
    for i in (1, 2, 3):
        s.add(i, [1, 2, 3, 4])
    s.schedule()
    s.reset()
    assert (s.num_waiting(), s.num_running(), s.num_swapped()) == (0, 0, 0)
    assert not s.has_unfinished_sequences()


def test_scheduler_outputs_helpers():
    out = nv.SchedulerOutputs()
    assert out.is_empty() and out.num_sequences() == 0
    out.prefill_sequences.append(1)
    out.decode_sequences.append(2)
    assert not out.is_empty() and out.num_sequences() == 2
    assert sorted(out.all_sequence_ids()) == [1, 2]


def test_block_allocation(impl):
    s = impl(test_config(), 16, 100)
    s.add(1, [1, 2, 3, 4])
    s.schedule()
    assert len(s.seq(1)["blocks"]) == 1 and s.used_blocks() == 1


def test_out_of_blocks(impl):
    s = impl(test_config(), 16, 2)
    for i, p in ((1, [1, 2, 3, 4]), (2, [5, 6, 7, 8]), (3, [9, 10, 11, 12])):
        s.add(i, p)
    out = s.schedule()
    assert len(out.prefill_sequences) == 2 and s.num_waiting() == 1


# ---- tests/sequence_test.rs (through the scheduler that owns the sequences) -----------------
def test_sequence_creation_priority_tokens_and_prefill_tracking():
    s = nv.Scheduler(nv.SchedulerConfig(), 16, 10)
    s.add_sequence(1, [10, 20, 30, 40])
    q = s.get_sequence(1)
    assert (q.seq_id(), q.prompt_len(), q.output_len(), q.total_len()) == (1, 4, 0, 4)     # :4-13
    assert q.status() == S.WAITING and q.priority() == 0 and q.last_token_id() == 40
    s.add_sequence(2, [1, 2, 3], 10)                                                        # :16-21
    assert s.get_sequence(2).priority() == 10
    s.append_token(1, 100)
    s.append_token(1, 101)                                                                  # :24-34
    assert (q.output_len(), q.total_len(), q.output_token_ids(), q.last_token_id()) == (2, 6, [100, 101], 101)
    s.add_sequence(3, list(range(1, 9)))                                                    # :37-53
    p = s.get_sequence(3)
    assert (p.num_prefilled_tokens(), p.num_tokens_to_prefill(), p.is_prefill_complete()) == (0, 8, False)
    s.mark_prefilled(3, 4)
    assert (p.num_prefilled_tokens(), p.num_tokens_to_prefill(), p.is_prefill_complete()) == (4, 4, False)
    s.mark_prefilled(3, 4)
    assert (p.num_prefilled_tokens(), p.num_tokens_to_prefill(), p.is_prefill_complete()) == (8, 0, True)
    s.mark_prefilled(3, 4)                       # min(.., prompt_len), sequence.rs:244
    assert p.num_prefilled_tokens() == 8
    with pytest.raises(nv.ConfigError):          # batch.rs:446 "Sequence {id} not found"
        s.append_token(99, 1)
    with pytest.raises(nv.ConfigError):
        s.mark_prefilled(99, 1)


def test_state_transitions():
    s = nv.Scheduler(nv.SchedulerConfig(), 16, 10)
    s.add_sequence(1, [1, 2, 3])
    q = s.get_sequence(1)
    o = oracle.SequenceOracle(1, [1, 2, 3])
    for st in (S.RUNNING, S.SWAPPED, S.RUNNING):                                            # :56-72
        q.set_status(st)
        {S.RUNNING: o.set_running, S.SWAPPED: o.set_swapped}[st]()
        assert q.status() == st == o.status
    s.finish_sequence(1, S.END_OF_SEQUENCE)
    assert q.status() == S.FINISHED and q.finish_reason() == S.END_OF_SEQUENCE
    s.add_sequence(2, [1, 2, 3])                                                            # :75-81
    w = s.get_sequence(2)
    o2 = oracle.SequenceOracle(2, [1, 2, 3])
    for st, fn in ((S.SWAPPED, o2.set_swapped), (S.WAITING, o2.set_waiting)):
        with pytest.raises(nv.ConfigError):
            w.set_status(st)
        with pytest.raises(oracle.InvalidStateTransition):
            fn()
    assert w.status() == S.WAITING


# ---- product == oracle on seeded random traces ----------------------------------------------
@pytest.mark.parametrize("seed", range(6))
def test_random_trace_matches_oracle(seed):
    rng = np.random.default_rng(seed)
    cfg = test_config(max_num_seqs=int(rng.integers(2, 9)), max_prefill_tokens=int(rng.integers(20, 200)),
                      enable_chunked_prefill=bool(seed & 1), chunk_size=int(rng.integers(8, 64)),
                      enable_preemption=bool(seed & 2))
    nb = int(rng.integers(6, 40))
    p, o = Product(cfg, 16, nb), Oracle(cfg, 16, nb)
    next_id, live = 1, []
    for step in range(120):
        r = rng.random()
        if r < 0.35:
            prompt = rng.integers(0, 1000, size=int(rng.integers(1, 90))).tolist()
            prio = int(rng.integers(-3, 4))
            p.add(next_id, prompt, prio)
            o.add(next_id, prompt, prio)
            live.append(next_id)
            next_id += 1
        elif r < 0.75:
            a, b = p.schedule(), o.schedule()
            assert a.prefill_sequences == b.prefill_sequences, step
            assert a.decode_sequences == b.decode_sequences
            assert a.preempted_sequences == b.preempted_sequences
            assert a.prefill_chunk_sizes == b.prefill_chunk_sizes
            assert (a.num_prefill_tokens, a.num_decode_tokens) == (b.num_prefill_tokens, b.num_decode_tokens)
            for sid in a.prefill_sequences:       # the engine's step: prefill the chunk ...
                p.mark_prefilled(sid, a.prefill_chunk_sizes[sid])
                o.mark_prefilled(sid, a.prefill_chunk_sizes[sid])
            for sid in a.decode_sequences:        # ... and append one sampled token per decode
                p.append_token(sid, 7)
                o.append_token(sid, 7)
        elif r < 0.87 and live:
            sid = live[int(rng.integers(len(live)))]
            p.finish_sequence(sid, S.MAX_TOKENS)
            o.finish_sequence(sid, S.MAX_TOKENS)
        elif live:
            sid = live.pop(int(rng.integers(len(live))))
            assert p.remove_sequence(sid) == o.remove_sequence(sid)
        assert (p.num_waiting(), p.num_running(), p.num_swapped()) == (o.num_waiting(), o.num_running(), o.num_swapped())
        assert p.used_blocks() == o.used_blocks()
        assert p.has_unfinished_sequences() == o.has_unfinished_sequences()
        assert p.has_pending_requests() == o.has_pending_requests()
        assert p.s.running_sequence_ids() == o.s.running_ids
        for sid in live:
            assert p.seq(sid) == o.seq(sid), (step, sid)


# ---- scheduler outputs -> batched-kernel metadata --------------------------------------------
def test_fill_batches_from_outputs():
    s = nv.Scheduler(nv.SchedulerConfig(max_num_seqs=8, max_prefill_tokens=1000, chunk_size=20), 16, 64)
    s.add_sequence(1, list(range(50)))
    s.add_sequence(2, list(range(20)))
    s.add_sequence(3, list(range(7)))
    out = s.schedule()
    assert out.prefill_chunk_sizes == {1: 20, 2: 20, 3: 7}
    b20, ids20 = s.prefill_batch(20, 8)
    b7, ids7 = s.prefill_batch(7, 8)
    assert ids20 == [1, 2] and ids7 == [3]
    assert b20.context_lens().tolist() == [20, 20] and b20.num_tokens() == 40
    assert b20.positions().tolist() == list(range(20)) * 2
    t1 = s.get_sequence(1).block_ids()
    assert b20.block_tables()[0].tolist() == t1 + [0] * (8 - len(t1)) and len(t1) == 4   # whole prompt
    assert b20.slot_mapping()[:20].tolist() == [t1[p // 16] * 16 + p % 16 for p in range(20)]
    assert b7.context_lens().tolist() == [7]
    for sid in (2, 3):
        s.mark_prefilled(sid, out.prefill_chunk_sizes[sid])
    out = s.schedule()
    assert out.decode_sequences == [2, 3]
    s.append_token(2, 5)        # the token sampled from the prefill's last position
    s.append_token(3, 6)
    d = s.decode_batch(8)
    assert d.context_lens().tolist() == [21, 8] and d.positions().tolist() == [20, 7]
    t2 = s.get_sequence(2).block_ids()
    assert d.slot_mapping().tolist()[0] == t2[1] * 16 + 4      # block grown by allocate_running_blocks?
    with pytest.raises(nv.ConfigError):
        s.prefill_batch(0, 8)


# ---- extensions (no reference counterpart; off by default) -----------------------------------
def test_default_config_is_the_reference_behaviour_partial_prefill_starves():
    # batch.rs:256-258 / :302: the second chunk is never scheduled -- the quirk the switch repairs
    for impl in (Product, Oracle):
        s = impl(test_config(enable_chunked_prefill=True, chunk_size=4), 16, 100)
        s.add(1, list(range(10)))
        out = s.schedule()
        s.mark_prefilled(1, out.prefill_chunk_sizes[1])
        for _ in range(3):
            out = s.schedule()
            assert out.prefill_sequences == [] and out.decode_sequences == []


def test_reschedule_partial_prefill():
    s = nv.Scheduler(nv.SchedulerConfig(**test_config(enable_chunked_prefill=True, chunk_size=4,
                                                      max_prefill_tokens=6, reschedule_partial_prefill=True)), 16, 100)
    s.add_sequence(1, list(range(10)))
    s.add_sequence(2, list(range(3)))
    chunks = []
    for _ in range(6):
        out = s.schedule()
        chunks.append((out.prefill_sequences, dict(out.prefill_chunk_sizes), out.decode_sequences))
        for sid in out.prefill_sequences:
            s.mark_prefilled(sid, out.prefill_chunk_sizes[sid])
        for sid in out.decode_sequences:
            s.append_token(sid, 1)
    # step 0: seq 1 chunk 4 (seq 2 would exceed the 6-token budget); step 1: seq 1's next chunk
    # first, then seq 2 is over budget again; step 2: last 2 tokens of seq 1 + seq 2 (3) = 5 <= 6
    assert chunks[0] == ([1], {1: 4}, [])
    assert chunks[1] == ([1], {1: 4}, [])
    assert chunks[2] == ([1, 2], {1: 2, 2: 3}, [])
    assert chunks[3] == ([], {}, [1, 2])
    assert s.get_sequence(1).is_prefill_complete() and s.get_sequence(1).num_blocks() == 1


def test_prefix_caching_reuses_full_blocks():
    cfg = nv.SchedulerConfig(**test_config(max_prefill_tokens=1000, enable_prefix_caching=True))
    s = nv.Scheduler(cfg, 16, 32)
    common = list(range(100, 140))                   # 2 full blocks + 8 tokens
    s.add_sequence(1, common + [1, 2, 3])
    out = s.schedule()
    assert out.prefill_chunk_sizes[1] == 43 and s.num_cached_prefixes() == 0   # nothing written yet
    s.mark_prefilled(1, 43)
    assert s.num_cached_prefixes() == 2              # blocks 0 and 1 of seq 1 are now reusable
    b1 = s.get_sequence(1).block_ids()
    s.add_sequence(2, common + [9, 9])               # same first 2 blocks, different tail
    s.add_sequence(3, list(range(100, 132)))         # exactly the 2 shared blocks
    s.add_sequence(4, [7] * 40)                      # nothing shared
    out = s.schedule()
    q2, q3, q4 = (s.get_sequence(i) for i in (2, 3, 4))
    assert q2.block_ids()[:2] == b1[:2] and q2.block_ids()[2] not in b1
    assert q2.num_prefilled_tokens() == 32 and out.prefill_chunk_sizes[2] == 10
    assert q3.block_ids() == b1[:2]
    assert q3.num_prefilled_tokens() == 31 and out.prefill_chunk_sizes[3] == 1   # last token is computed
    assert q4.num_prefilled_tokens() == 0 and out.prefill_chunk_sizes[4] == 40
    assert out.num_prefill_tokens == 10 + 1 + 40
    assert s.block_ref_count(b1[0]) == 3 and s.block_ref_count(b1[1]) == 3
    # hashes are the chained hash_token_block values (block.rs:43-57)
    h0 = nv.hash_token_block(common[:16])
    h1 = nv.hash_token_block(common[16:32], h0)
    assert h0 != h1
    # freeing every owner releases the shared blocks and their cache entries (block_manager.rs:187-204)
    used = s.num_used_blocks()
    for sid in (1, 2, 3, 4):
        s.remove_sequence(sid)
    assert s.num_used_blocks() == 0 and s.num_cached_prefixes() == 0 and used == 3 + 1 + 0 + 3
    # a prompt identical to an earlier one is NOT shared unless that one was prefilled
    s.add_sequence(5, common)
    s.add_sequence(6, common)
    out = s.schedule()
    assert set(s.get_sequence(5).block_ids()).isdisjoint(s.get_sequence(6).block_ids())
