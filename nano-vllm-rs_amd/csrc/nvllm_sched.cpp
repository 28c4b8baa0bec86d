// nvllm_sched.cpp -- see nvllm_sched.hpp.  Line references are to the reference under
// src/scheduler/batch.rs and src/core/sequence.rs.
#include "nvllm_sched.hpp"

#include <algorithm>
#include <limits>
#include <string>

namespace nvllm {
namespace core {

std::optional<std::uint32_t> Sequence::last_token_id() const {
    if (!output_.empty()) return output_.back();
    if (!prompt_.empty()) return prompt_.back();
    return std::nullopt;
}

bool Sequence::set_running() {  // sequence.rs:254-266
    if (status_ == SequenceStatus::Waiting || status_ == SequenceStatus::Swapped) {
        status_ = SequenceStatus::Running;
        return true;
    }
    return false;
}
bool Sequence::set_waiting() {  // :272-284
    if (status_ == SequenceStatus::Running) {
        status_ = SequenceStatus::Waiting;
        return true;
    }
    return false;
}
bool Sequence::set_swapped() {  // :290-301
    if (status_ == SequenceStatus::Running) {
        status_ = SequenceStatus::Swapped;
        return true;
    }
    return false;
}

std::vector<SequenceId> SchedulerOutputs::all_sequence_ids() const {  // batch.rs:93-97
    std::vector<SequenceId> ids = prefill_sequences;
    ids.insert(ids.end(), decode_sequences.begin(), decode_sequences.end());
    return ids;
}

void Scheduler::add_sequence(Sequence seq) {  // :181-193
    const SequenceId id = seq.seq_id();
    waiting_.push(PriorityEntry{id, seq.priority(), arrival_counter_});
    ++arrival_counter_;
    sequences_.erase(id);  // HashMap::insert replaces
    sequences_.emplace(id, std::move(seq));
}

std::optional<Sequence> Scheduler::remove_sequence(SequenceId id) {  // :198-212
    auto it = sequences_.find(id);
    if (it == sequences_.end()) return std::nullopt;
    Sequence seq = std::move(it->second);
    sequences_.erase(it);
    const auto &ids = seq.block_table().get_physical_block_ids();
    block_manager_.free_many(ids.data(), ids.size());
    running_ids_.erase(std::remove(running_ids_.begin(), running_ids_.end(), id), running_ids_.end());
    swapped_ids_.erase(std::remove(swapped_ids_.begin(), swapped_ids_.end(), id), swapped_ids_.end());
    return seq;
}

const Sequence *Scheduler::get_sequence(SequenceId id) const {
    auto it = sequences_.find(id);
    return it == sequences_.end() ? nullptr : &it->second;
}
Sequence *Scheduler::get_sequence_mut(SequenceId id) {
    auto it = sequences_.find(id);
    return it == sequences_.end() ? nullptr : &it->second;
}

SchedulerOutputs Scheduler::schedule() {  // :227-246
    SchedulerOutputs out;
    schedule_decode(out);
    allocate_running_blocks();
    if (config_.reschedule_partial_prefill) schedule_partial_prefill(out);
    schedule_prefill(out);
    if (config_.enable_preemption) handle_preemption(out);
    return out;
}

void Scheduler::schedule_decode(SchedulerOutputs &out) {  // :249-264
    for (SequenceId id : running_ids_) {
        if (out.num_sequences() >= config_.max_num_seqs) break;
        const Sequence *seq = get_sequence(id);
        if (seq && seq->is_prefill_complete()) {
            out.decode_sequences.push_back(id);
            out.num_decode_tokens += 1;
        }
    }
}

void Scheduler::allocate_running_blocks() {  // :267-288
    const std::vector<SequenceId> ids = running_ids_;
    for (SequenceId id : ids) {
        Sequence *seq = get_sequence_mut(id);
        if (!seq) continue;
        const std::size_t needed = compute_num_blocks(seq->total_len(), block_manager_.block_size());
        const std::size_t have = seq->block_table().num_blocks();
        if (needed > have) {
            std::vector<std::size_t> fresh;
            if (block_manager_.can_allocate(needed - have) &&
                block_manager_.allocate_many(needed - have, &fresh))
                for (std::size_t b : fresh) seq->block_table_mut().append_block(b);
        }
    }
}

// Extension (off by default): the reference pops sequences from the waiting heap only, so a
// sequence admitted with a first chunk never gets its second one (schedule_decode skips it while
// is_prefill_complete() is false, :256-258).  Here such sequences get their next chunk, in
// running order, inside the same sequence and token budgets as admission.
void Scheduler::schedule_partial_prefill(SchedulerOutputs &out) {
    for (SequenceId id : running_ids_) {
        if (out.num_sequences() >= config_.max_num_seqs) break;
        if (out.num_prefill_tokens >= config_.max_prefill_tokens) break;
        const Sequence *seq = get_sequence(id);
        if (!seq || seq->is_prefill_complete()) continue;
        std::size_t n = seq->num_tokens_to_prefill();
        if (config_.enable_chunked_prefill) n = std::min(n, config_.chunk_size);
        if (out.num_prefill_tokens + n > config_.max_prefill_tokens && !out.prefill_sequences.empty())
            break;
        out.prefill_sequences.push_back(id);
        out.prefill_chunk_sizes[id] = n;
        out.num_prefill_tokens += n;
    }
}

void Scheduler::schedule_prefill(SchedulerOutputs &out) {  // :291-375
    struct Admit {
        SequenceId id;
        std::size_t tokens, blocks;
    };
    std::vector<Admit> to_admit;
    std::size_t pending_blocks = 0;
    while (out.num_sequences() + to_admit.size() < config_.max_num_seqs &&
           out.num_prefill_tokens < config_.max_prefill_tokens) {
        if (waiting_.empty()) break;
        const PriorityEntry entry = waiting_.top();
        waiting_.pop();
        const Sequence *seq = get_sequence(entry.seq_id);
        if (!seq || seq->status() != SequenceStatus::Waiting) continue;  // :313-316 (entry dropped)
        std::size_t tokens = seq->num_tokens_to_prefill();
        if (config_.enable_chunked_prefill) tokens = std::min(tokens, config_.chunk_size);
        // (out.prefill_sequences is empty here unless reschedule_partial_prefill put chunks in)
        if (out.num_prefill_tokens + tokens > config_.max_prefill_tokens &&
            (!to_admit.empty() || !out.prefill_sequences.empty())) {
            waiting_.push(entry);  // :326-332
            break;
        }
        const std::size_t blocks = compute_num_blocks(seq->prompt_len(), block_manager_.block_size());
        if (!block_manager_.can_allocate(pending_blocks + blocks)) {  // :339-347
            waiting_.push(entry);
            break;
        }
        to_admit.push_back(Admit{entry.seq_id, tokens, blocks});
        out.num_prefill_tokens += tokens;
        pending_blocks += blocks;
    }
    for (const Admit &a : to_admit) {  // :356-374
        Sequence *seq = get_sequence_mut(a.id);
        if (!seq) continue;
        std::size_t tokens = a.tokens;
        std::size_t hits = 0;
        if (config_.enable_prefix_caching) {
            // hash chain over the full prompt blocks; reuse the cached leading run
            const std::size_t bs = block_manager_.block_size();
            const auto &prompt = seq->prompt_token_ids();
            seq->block_hashes.clear();
            std::optional<std::uint64_t> parent;
            for (std::size_t b = 0; (b + 1) * bs <= prompt.size(); ++b) {
                const std::uint64_t h = hash_token_block(prompt.data() + b * bs, bs, parent);
                seq->block_hashes.push_back(h);
                parent = h;
            }
            while (hits < seq->block_hashes.size()) {
                const auto cached = block_manager_.get_cached_block(seq->block_hashes[hits]);  // ref++
                if (!cached) break;
                seq->block_table_mut().append_block(*cached);
                ++hits;
            }
            if (hits > 0) {
                // the last prompt token is always computed (its logits start the decode)
                const std::size_t skip = std::min(hits * bs, seq->prompt_len() - 1);
                seq->mark_prefilled(skip);
                seq->num_registered_blocks = hits;
                std::size_t n = seq->num_tokens_to_prefill();
                if (config_.enable_chunked_prefill) n = std::min(n, config_.chunk_size);
                out.num_prefill_tokens -= tokens - n;
                tokens = n;
            }
        }
        std::vector<std::size_t> fresh;
        if (!block_manager_.allocate_many(a.blocks - hits, &fresh)) continue;
        for (std::size_t b : fresh) seq->block_table_mut().append_block(b);
        seq->set_running();
        out.prefill_sequences.push_back(a.id);
        out.prefill_chunk_sizes[a.id] = tokens;
        running_ids_.push_back(a.id);
    }
}

void Scheduler::handle_preemption(SchedulerOutputs &out) {  // :378-409
    while (!waiting_.empty() && block_manager_.num_free_blocks() == 0) {
        const auto idx = find_lowest_priority_running();
        if (!idx) break;
        const SequenceId id = running_ids_[*idx];
        running_ids_.erase(running_ids_.begin() + (std::ptrdiff_t)*idx);
        Sequence *seq = get_sequence_mut(id);
        if (seq) {
            const std::vector<std::size_t> ids = seq->block_table().get_physical_block_ids();
            block_manager_.free_many(ids.data(), ids.size());
            seq->block_table_mut().clear();
            seq->set_swapped();
            swapped_ids_.push_back(id);
            out.preempted_sequences.push_back(id);
            auto &d = out.decode_sequences;
            d.erase(std::remove(d.begin(), d.end(), id), d.end());
        }
    }
}

std::optional<std::size_t> Scheduler::find_lowest_priority_running() const {  // :412-425
    Priority lowest = std::numeric_limits<Priority>::max();
    std::optional<std::size_t> at;
    for (std::size_t i = 0; i < running_ids_.size(); ++i) {
        const Sequence *seq = get_sequence(running_ids_[i]);
        if (seq && seq->priority() < lowest) {
            lowest = seq->priority();
            at = i;
        }
    }
    return at;
}

void Scheduler::finish_sequence(SequenceId id, FinishReason reason) {  // :428-439
    Sequence *seq = get_sequence_mut(id);
    if (!seq) return;
    seq->set_finished(reason);
    running_ids_.erase(std::remove(running_ids_.begin(), running_ids_.end(), id), running_ids_.end());
}

bool Scheduler::append_token(SequenceId id, std::uint32_t token) {  // :442-449
    Sequence *seq = get_sequence_mut(id);
    if (!seq) return false;
    seq->append_token(token);
    return true;
}

bool Scheduler::mark_prefilled(SequenceId id, std::size_t num_tokens) {  // :452-459
    Sequence *seq = get_sequence_mut(id);
    if (!seq) return false;
    seq->mark_prefilled(num_tokens);
    if (config_.enable_prefix_caching) register_prefilled_blocks(*seq);
    return true;
}

// Extension: a full prompt block becomes reusable once its K/V have been written, i.e. once
// mark_prefilled has passed its last token.
void Scheduler::register_prefilled_blocks(Sequence &seq) {
    const std::size_t bs = block_manager_.block_size();
    const auto &ids = seq.block_table().get_physical_block_ids();
    while (seq.num_registered_blocks < seq.block_hashes.size() &&
           (seq.num_registered_blocks + 1) * bs <= seq.num_prefilled_tokens() &&
           seq.num_registered_blocks < ids.size()) {
        const std::uint64_t h = seq.block_hashes[seq.num_registered_blocks];
        if (!block_manager_.is_prefix_cached(h))
            block_manager_.cache_block(ids[seq.num_registered_blocks], h);
        ++seq.num_registered_blocks;
    }
}

void Scheduler::reset() {  // :504-511
    sequences_.clear();
    waiting_ = std::priority_queue<PriorityEntry>();
    running_ids_.clear();
    swapped_ids_.clear();
    block_manager_.reset();
    arrival_counter_ = 0;
}

// Both fill_* helpers are all-or-nothing: on an error the batch is cut back to what it held on
// entry, so a later sequence's failure cannot leave a half-filled step behind.
Error Scheduler::fill_decode_batch(const SchedulerOutputs &out, BatchMetadata *batch) const {
    const std::size_t entry = batch->num_seqs();
    for (SequenceId id : out.decode_sequences) {
        const Sequence *seq = get_sequence(id);
        Error e = seq ? batch->add_sequence(seq->block_table(), seq->total_len(), 1)
                      : Error{ErrorKind::Config, "Sequence " + std::to_string(id) + " not found"};
        if (!e.ok()) {
            batch->truncate(entry);
            return e;
        }
    }
    return Error{};
}

Error Scheduler::fill_prefill_batch(const SchedulerOutputs &out, std::size_t num_new_tokens,
                                    BatchMetadata *batch, std::vector<SequenceId> *added) const {
    const std::size_t entry = batch->num_seqs();
    const std::size_t added_entry = added ? added->size() : 0;
    for (SequenceId id : out.prefill_sequences) {
        const auto it = out.prefill_chunk_sizes.find(id);
        if (it == out.prefill_chunk_sizes.end() || it->second != num_new_tokens) continue;
        const Sequence *seq = get_sequence(id);
        Error e = seq ? batch->add_sequence(seq->block_table(), seq->num_prefilled_tokens() + num_new_tokens,
                                            num_new_tokens)
                      : Error{ErrorKind::Config, "Sequence " + std::to_string(id) + " not found"};
        if (!e.ok()) {
            batch->truncate(entry);
            if (added) added->resize(added_entry);
            return e;
        }
        if (added) added->push_back(id);
    }
    return Error{};
}

}  // namespace core
}  // namespace nvllm
