"""Pure-Python restatement of the reference's Sequence and Scheduler (small cases only).

TEST INFRASTRUCTURE ONLY (see oracle/nvllm_oracle.c header).

Follows /root/reference/src/core/sequence.rs:10-327, src/config.rs:44-70 and
src/scheduler/batch.rs:60-512 statement by statement, on top of BlockManagerOracle.  Quirks kept:
  * num_waiting() is the heap length: entries of removed / no longer waiting sequences count
    until a schedule() pops and drops them (:313-316, :462-464);
  * swapped sequences are never resumed (swapped_ids is only ever appended to and counted);
  * a chunk-prefilled sequence is never scheduled again: schedule_prefill pops the waiting heap
    only (:302) and schedule_decode needs is_prefill_complete() (:256-258);
  * the first admitted sequence may exceed max_prefill_tokens (:326-328);
  * find_lowest_priority_running never picks a sequence of priority i32::MAX (:413-419).
The product's two opt-in extensions (reschedule_partial_prefill, enable_prefix_caching) have no
counterpart in the reference and are NOT restated here; tests/test_scheduler.py checks them
against properties instead.
Pinned by the reference's own tests/scheduler_test.rs and tests/sequence_test.rs (ported in
tests/test_scheduler.py).
"""
import heapq

from .block_manager_oracle import BlockManagerOracle

WAITING, RUNNING, SWAPPED, FINISHED = 0, 1, 2, 3
I32_MAX = 2 ** 31 - 1


class InvalidStateTransition(Exception):
    """Error::InvalidStateTransition (sequence.rs:254-301)."""


class SequenceOracle:
    def __init__(self, seq_id, prompt_token_ids, priority=0):       # :116-141
        self.seq_id = seq_id
        self.prompt_token_ids = list(prompt_token_ids)
        self.output_token_ids = []
        self.block_ids = []                                         # BlockTable, default size 16
        self.num_prefilled_tokens = 0
        self.status = WAITING
        self.priority = priority
        self.finish_reason = None

    def prompt_len(self):
        return len(self.prompt_token_ids)

    def output_len(self):
        return len(self.output_token_ids)

    def total_len(self):
        return self.prompt_len() + self.output_len()

    def num_tokens_to_prefill(self):                                # :218 saturating_sub
        return max(self.prompt_len() - self.num_prefilled_tokens, 0)

    def is_prefill_complete(self):                                  # :223
        return self.num_prefilled_tokens >= self.prompt_len()

    def append_token(self, t):
        self.output_token_ids.append(t)

    def last_token_id(self):                                        # :235-240
        if self.output_token_ids:
            return self.output_token_ids[-1]
        return self.prompt_token_ids[-1] if self.prompt_token_ids else None

    def mark_prefilled(self, n):                                    # :243-245
        self.num_prefilled_tokens = min(self.num_prefilled_tokens + n, self.prompt_len())

    def set_running(self):                                          # :254-266
        if self.status not in (WAITING, SWAPPED):
            raise InvalidStateTransition()
        self.status = RUNNING

    def set_waiting(self):                                          # :272-284
        if self.status != RUNNING:
            raise InvalidStateTransition()
        self.status = WAITING

    def set_swapped(self):                                          # :290-301
        if self.status != RUNNING:
            raise InvalidStateTransition()
        self.status = SWAPPED

    def set_finished(self, reason):                                 # :304-307
        self.status = FINISHED
        self.finish_reason = reason


class SchedulerOutputsOracle:
    def __init__(self):                                             # :60-80
        self.prefill_sequences = []
        self.decode_sequences = []
        self.prefill_chunk_sizes = {}
        self.preempted_sequences = []
        self.num_prefill_tokens = 0
        self.num_decode_tokens = 0

    def num_sequences(self):
        return len(self.prefill_sequences) + len(self.decode_sequences)


def _ceil_div(a, b):
    return -(-a // b)


class SchedulerOracle:
    def __init__(self, max_num_seqs=256, max_prefill_tokens=4096, enable_chunked_prefill=True,
                 chunk_size=512, enable_priority=True, enable_preemption=False, block_size=16,
                 num_blocks=0):
        self.max_num_seqs = max_num_seqs
        self.max_prefill_tokens = max_prefill_tokens
        self.enable_chunked_prefill = enable_chunked_prefill
        self.chunk_size = chunk_size
        self.enable_preemption = enable_preemption
        self.bm = BlockManagerOracle(num_blocks, block_size)
        self.sequences = {}
        self.waiting = []            # BinaryHeap<PriorityEntry>: max by (priority, -arrival)
        self.running_ids = []
        self.swapped_ids = []
        self.arrival_counter = 0

    def add_sequence(self, seq):                                    # :181-193
        heapq.heappush(self.waiting, (-seq.priority, self.arrival_counter, seq.seq_id))
        self.arrival_counter += 1
        self.sequences[seq.seq_id] = seq

    def remove_sequence(self, seq_id):                              # :198-212
        seq = self.sequences.pop(seq_id, None)
        if seq is None:
            return None
        self.bm.free_many(list(seq.block_ids))
        self.running_ids = [i for i in self.running_ids if i != seq_id]
        self.swapped_ids = [i for i in self.swapped_ids if i != seq_id]
        return seq

    def schedule(self):                                             # :227-246
        out = SchedulerOutputsOracle()
        # :249-264
        for seq_id in self.running_ids:
            if out.num_sequences() >= self.max_num_seqs:
                break
            seq = self.sequences.get(seq_id)
            if seq is not None and seq.is_prefill_complete():
                out.decode_sequences.append(seq_id)
                out.num_decode_tokens += 1
        # :267-288
        for seq_id in list(self.running_ids):
            seq = self.sequences.get(seq_id)
            if seq is None:
                continue
            needed = _ceil_div(seq.total_len(), self.bm.block_size)
            have = len(seq.block_ids)
            if needed > have and self.bm.can_allocate(needed - have):
                seq.block_ids.extend(self.bm.allocate_many(needed - have))
        # :291-375
        to_admit = []
        pending = 0
        while (out.num_sequences() + len(to_admit) < self.max_num_seqs
               and out.num_prefill_tokens < self.max_prefill_tokens):
            if not self.waiting:
                break
            entry = heapq.heappop(self.waiting)
            seq = self.sequences.get(entry[2])
            if seq is None or seq.status != WAITING:
                continue
            tokens = seq.num_tokens_to_prefill()
            if self.enable_chunked_prefill:
                tokens = min(tokens, self.chunk_size)
            if out.num_prefill_tokens + tokens > self.max_prefill_tokens and to_admit:
                heapq.heappush(self.waiting, entry)
                break
            blocks = _ceil_div(seq.prompt_len(), self.bm.block_size)
            if not self.bm.can_allocate(pending + blocks):
                heapq.heappush(self.waiting, entry)
                break
            to_admit.append((entry[2], tokens, blocks))
            out.num_prefill_tokens += tokens
            pending += blocks
        for seq_id, tokens, blocks in to_admit:
            seq = self.sequences[seq_id]
            seq.block_ids.extend(self.bm.allocate_many(blocks))
            seq.set_running()
            out.prefill_sequences.append(seq_id)
            out.prefill_chunk_sizes[seq_id] = tokens
            self.running_ids.append(seq_id)
        # :378-409
        if self.enable_preemption:
            while self.waiting and self.bm.num_free_blocks() == 0:
                lowest, at = I32_MAX, None
                for i, seq_id in enumerate(self.running_ids):       # :412-425
                    seq = self.sequences.get(seq_id)
                    if seq is not None and seq.priority < lowest:
                        lowest, at = seq.priority, i
                if at is None:
                    break
                seq_id = self.running_ids.pop(at)
                seq = self.sequences.get(seq_id)
                if seq is not None:
                    self.bm.free_many(list(seq.block_ids))
                    seq.block_ids = []
                    seq.set_swapped()
                    self.swapped_ids.append(seq_id)
                    out.preempted_sequences.append(seq_id)
                    out.decode_sequences = [i for i in out.decode_sequences if i != seq_id]
        return out

    def finish_sequence(self, seq_id, reason):                      # :428-439
        seq = self.sequences.get(seq_id)
        if seq is not None:
            seq.set_finished(reason)
            self.running_ids = [i for i in self.running_ids if i != seq_id]

    def append_token(self, seq_id, token):                          # :442-449 (KeyError = Config)
        self.sequences[seq_id].append_token(token)

    def mark_prefilled(self, seq_id, n):                            # :452-459
        self.sequences[seq_id].mark_prefilled(n)

    def num_waiting(self):
        return len(self.waiting)

    def num_running(self):
        return len(self.running_ids)

    def num_swapped(self):
        return len(self.swapped_ids)

    def has_unfinished_sequences(self):                             # :482-486
        return bool(self.waiting or self.running_ids or self.swapped_ids)

    def has_pending_requests(self):                                 # :489-491
        return bool(self.waiting or self.running_ids)

    def reset(self):                                                # :504-511
        self.sequences.clear()
        self.waiting = []
        self.running_ids = []
        self.swapped_ids = []
        self.bm.reset()
        self.arrival_counter = 0
