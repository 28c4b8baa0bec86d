"""Harness-side mirror of the scheduler C ABI (nvllm_scheduler_*, csrc/nvllm_sched.cpp).

Names and meanings are the reference's (src/scheduler/batch.rs, src/core/sequence.rs,
src/config.rs:44-70); every decision is taken in the C++ library, nothing is computed here.
"""
import ctypes

import numpy as np

from . import capi
from .core import BatchMetadata, ConfigError, check

WAITING, RUNNING, SWAPPED, FINISHED = 0, 1, 2, 3                    # SequenceStatus, sequence.rs:16-25
END_OF_SEQUENCE, MAX_TOKENS, STOP_SEQUENCE, ABORTED = 0, 1, 2, 3     # FinishReason, sequence.rs:51-60


class _CConfig(ctypes.Structure):
    _fields_ = [("max_num_seqs", ctypes.c_int64), ("max_prefill_tokens", ctypes.c_int64),
                ("enable_chunked_prefill", ctypes.c_int), ("chunk_size", ctypes.c_int64),
                ("enable_priority", ctypes.c_int), ("enable_preemption", ctypes.c_int),
                ("reschedule_partial_prefill", ctypes.c_int), ("enable_prefix_caching", ctypes.c_int)]


class SchedulerConfig:
    """config.rs:44-70 (+ the two opt-in extensions documented in include/nvllm_attn.h)."""

    def __init__(self, **kw):
        c = _CConfig()
        capi.lib().nvllm_scheduler_config_default(ctypes.byref(c))
        for name, _ in _CConfig._fields_:
            v = getattr(c, name)
            setattr(self, name, bool(v) if name.startswith(("enable_", "reschedule_")) else int(v))
        for k, v in kw.items():
            if not hasattr(self, k):
                raise TypeError("SchedulerConfig has no field %r" % k)
            setattr(self, k, v)

    def _c(self):
        c = _CConfig()
        for name, _ in _CConfig._fields_:
            setattr(c, name, int(getattr(self, name)))
        return c


class SchedulerOutputs:
    """batch.rs:60-98, a snapshot of one schedule() call."""

    def __init__(self, prefill=(), decode=(), chunk_sizes=None, preempted=(), num_prefill_tokens=0,
                 num_decode_tokens=0):
        self.prefill_sequences = list(prefill)
        self.decode_sequences = list(decode)
        self.prefill_chunk_sizes = dict(chunk_sizes or {})
        self.preempted_sequences = list(preempted)
        self.num_prefill_tokens = num_prefill_tokens
        self.num_decode_tokens = num_decode_tokens

    def is_empty(self):
        return not self.prefill_sequences and not self.decode_sequences

    def num_sequences(self):
        return len(self.prefill_sequences) + len(self.decode_sequences)

    def all_sequence_ids(self):
        return self.prefill_sequences + self.decode_sequences


class SequenceView:
    """Read-only view of a Sequence held by the scheduler (get_sequence, batch.rs:215)."""

    def __init__(self, sched, seq_id):
        self._s, self._id = sched, seq_id

    def _info(self, what):
        return int(capi.lib().nvllm_scheduler_seq_info(self._s._h, self._id, what))

    def seq_id(self):
        return self._id

    def status(self):
        return int(capi.lib().nvllm_scheduler_seq_status(self._s._h, self._id))

    def finish_reason(self):
        r = int(capi.lib().nvllm_scheduler_seq_finish_reason(self._s._h, self._id))
        return None if r < 0 else r

    def set_status(self, status):
        """set_waiting / set_running / set_swapped (sequence.rs:254-301); ConfigError when the
        reference returns Error::InvalidStateTransition."""
        check(capi.lib().nvllm_scheduler_seq_set_status(self._s._h, self._id, status))

    def prompt_len(self):
        return self._info(0)

    def output_len(self):
        return self._info(1)

    def total_len(self):
        return self._info(2)

    def num_prefilled_tokens(self):
        return self._info(3)

    def num_tokens_to_prefill(self):
        return self._info(4)

    def is_prefill_complete(self):
        return bool(self._info(5))

    def num_blocks(self):
        return self._info(6)

    def priority(self):
        return self._info(7)

    def last_token_id(self):
        t = self._info(8)
        return None if t < 0 else t

    def block_ids(self):
        n = self.num_blocks()
        out = np.zeros(max(n, 1), dtype=np.int64)
        capi.lib().nvllm_scheduler_seq_block_ids(self._s._h, self._id, out.ctypes.data, n)
        return out[:n].tolist()

    def output_token_ids(self):
        n = self.output_len()
        out = np.zeros(max(n, 1), dtype=np.uint32)
        capi.lib().nvllm_scheduler_seq_output_tokens(self._s._h, self._id, out.ctypes.data, n)
        return out[:n].tolist()


class Scheduler:
    """batch.rs:141-512."""

    def __init__(self, config, block_size, num_blocks):
        c = config._c()
        self._h = capi.lib().nvllm_scheduler_new(ctypes.byref(c), block_size, num_blocks)
        if not self._h:
            raise ConfigError(capi.NVLLM_ERR_CONFIG)

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            capi.lib().nvllm_scheduler_free(h)

    def add_sequence(self, seq_id, prompt_token_ids, priority=0):
        """Sequence::new / with_priority (sequence.rs:116-141) + add_sequence (batch.rs:181)."""
        t = np.ascontiguousarray(prompt_token_ids, dtype=np.uint32)
        check(capi.lib().nvllm_scheduler_add_sequence(self._h, seq_id, t.ctypes.data, t.size, priority))

    def remove_sequence(self, seq_id):
        return bool(capi.lib().nvllm_scheduler_remove_sequence(self._h, seq_id))

    def get_sequence(self, seq_id):
        if capi.lib().nvllm_scheduler_seq_status(self._h, seq_id) < 0:
            return None
        return SequenceView(self, seq_id)

    def _ids(self, fn, n, ctype=ctypes.c_uint64):
        if n == 0:
            return []
        p = ctypes.cast(fn(self._h), ctypes.POINTER(ctype))
        return [int(p[i]) for i in range(n)]

    def schedule(self):
        L = capi.lib()
        check(L.nvllm_scheduler_schedule(self._h))
        pre = self._ids(L.nvllm_scheduler_out_prefill_ids, int(L.nvllm_scheduler_out_num_prefill(self._h)))
        sizes = self._ids(L.nvllm_scheduler_out_prefill_chunk_sizes, len(pre), ctypes.c_int64)
        return SchedulerOutputs(
            pre, self._ids(L.nvllm_scheduler_out_decode_ids, int(L.nvllm_scheduler_out_num_decode(self._h))),
            dict(zip(pre, sizes)),
            self._ids(L.nvllm_scheduler_out_preempted_ids, int(L.nvllm_scheduler_out_num_preempted(self._h))),
            int(L.nvllm_scheduler_out_num_prefill_tokens(self._h)),
            int(L.nvllm_scheduler_out_num_decode_tokens(self._h)))

    def finish_sequence(self, seq_id, reason):
        capi.lib().nvllm_scheduler_finish_sequence(self._h, seq_id, reason)

    def append_token(self, seq_id, token_id):
        check(capi.lib().nvllm_scheduler_append_token(self._h, seq_id, token_id))

    def mark_prefilled(self, seq_id, num_tokens):
        check(capi.lib().nvllm_scheduler_mark_prefilled(self._h, seq_id, num_tokens))

    def num_waiting(self):
        return int(capi.lib().nvllm_scheduler_num_waiting(self._h))

    def num_running(self):
        return int(capi.lib().nvllm_scheduler_num_running(self._h))

    def num_swapped(self):
        return int(capi.lib().nvllm_scheduler_num_swapped(self._h))

    def has_unfinished_sequences(self):
        return bool(capi.lib().nvllm_scheduler_has_unfinished_sequences(self._h))

    def has_pending_requests(self):
        return bool(capi.lib().nvllm_scheduler_has_pending_requests(self._h))

    def running_sequence_ids(self):
        n = self.num_running()
        out = np.zeros(max(n, 1), dtype=np.uint64)
        capi.lib().nvllm_scheduler_running_sequence_ids(self._h, out.ctypes.data, n)
        return [int(x) for x in out[:n]]

    def reset(self):
        capi.lib().nvllm_scheduler_reset(self._h)

    # block_manager() view (batch.rs:494)
    def num_free_blocks(self):
        return int(capi.lib().nvllm_scheduler_num_free_blocks(self._h))

    def num_used_blocks(self):
        return int(capi.lib().nvllm_scheduler_num_used_blocks(self._h))

    def num_cached_prefixes(self):
        return int(capi.lib().nvllm_scheduler_num_cached_prefixes(self._h))

    def block_ref_count(self, block_id):
        r = int(capi.lib().nvllm_scheduler_block_ref_count(self._h, block_id))
        return None if r < 0 else r

    # last schedule() result -> batched-kernel metadata
    def decode_batch(self, max_blocks_per_seq):
        b = BatchMetadata(max_blocks_per_seq)
        check(capi.lib().nvllm_scheduler_fill_decode_batch(self._h, b._h))
        return b

    def prefill_batch(self, num_new_tokens, max_blocks_per_seq):
        """(BatchMetadata, seq ids) of the prefill sequences whose chunk is num_new_tokens long."""
        b = BatchMetadata(max_blocks_per_seq)
        cap = int(capi.lib().nvllm_scheduler_out_num_prefill(self._h))
        ids = np.zeros(max(cap, 1), dtype=np.uint64)
        n = int(capi.lib().nvllm_scheduler_fill_prefill_batch(self._h, num_new_tokens, b._h,
                                                               ids.ctypes.data, cap))
        if n < 0:
            check(n)
        return b, [int(x) for x in ids[:n]]
