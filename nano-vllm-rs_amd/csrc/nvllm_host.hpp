// nvllm_host.hpp -- host half of the attention hot path: block table, slot
// mapping, prefix hashing and the block allocator, in C++ (the reference's
// host code is Rust; no Rust toolchain exists in the build image, so the host
// side above the C ABI is C++ and keeps the reference's names, argument
// meaning and error behaviour; see INTEGRATION.md for the Rust binding).
//
// Mirrors (paths under the reference checkout):
//   src/core/block.rs          hash_token_block, Block, BlockTable, compute_num_blocks
//   src/core/block_manager.rs  BlockManager
//   src/error.rs               Error::{OutOfBlocks, BlockIndexOutOfBounds, Config}
#pragma once

#include <cstddef>
#include <cstdint>
#include <deque>
#include <optional>
#include <string>
#include <unordered_map>
#include <vector>

namespace nvllm {

// crate::error::Error for the variants this path can raise (src/error.rs:10-60).
enum class ErrorKind : int {
    Ok = 0,
    Config = -1,
    Unsupported = -2,
    Null = -3,
    Hip = -4,
    OutOfBlocks = -5,
    BlockIndexOutOfBounds = -6,
    Tensor = -7,
};

struct Error {
    ErrorKind kind = ErrorKind::Ok;
    std::string message;
    bool ok() const { return kind == ErrorKind::Ok; }
};

// Thread-local "last error" used by the C ABI (nvllm_get_last_error).
void set_last_error(ErrorKind kind, const std::string &message);
const char *last_error_message();

namespace core {

constexpr std::size_t DEFAULT_BLOCK_SIZE = 16;  // block.rs:14

// block.rs:43-57.  SipHash-1-3 with zero keys == Rust's DefaultHasher::new().
std::uint64_t hash_token_block(const std::uint32_t *token_ids, std::size_t n,
                               std::optional<std::uint64_t> parent_hash);

// block.rs:311-313
inline std::size_t compute_num_blocks(std::size_t seq_len, std::size_t block_size) {
    return (seq_len + block_size - 1) / block_size;
}

// block.rs:72-156
class Block {
  public:
    Block(std::size_t block_id, std::size_t block_size)
        : block_id_(block_id), block_size_(block_size) {}
    std::size_t block_id() const { return block_id_; }
    std::size_t block_size() const { return block_size_; }
    std::size_t ref_count() const { return ref_count_; }
    std::optional<std::uint64_t> prefix_hash() const { return prefix_hash_; }
    bool is_full() const { return is_full_; }
    void set_prefix_hash(std::uint64_t h) { prefix_hash_ = h; }
    void set_full(bool f) { is_full_ = f; }
    void increment_ref() { ++ref_count_; }
    std::size_t decrement_ref() {  // saturating_sub(1), block.rs:152-155
        if (ref_count_ > 0) --ref_count_;
        return ref_count_;
    }

  private:
    std::size_t block_id_;
    std::size_t block_size_;
    std::size_t ref_count_ = 1;  // block.rs:96
    std::optional<std::uint64_t> prefix_hash_;
    bool is_full_ = false;
};

// block.rs:183-289
class BlockTable {
  public:
    explicit BlockTable(std::size_t block_size = DEFAULT_BLOCK_SIZE) : block_size_(block_size) {}
    std::size_t block_size() const { return block_size_; }
    // block.rs:223-231: false => Error::BlockIndexOutOfBounds
    bool get_block_id(std::size_t logical_block_idx, std::size_t *out) const;
    void append_block(std::size_t block_id) { block_ids_.push_back(block_id); }
    std::size_t num_blocks() const { return block_ids_.size(); }
    bool is_empty() const { return block_ids_.empty(); }
    const std::vector<std::size_t> &get_physical_block_ids() const { return block_ids_; }
    // block.rs:269-283: positions past the last allocated block are dropped.
    std::vector<std::size_t> get_slot_mapping(std::size_t seq_len) const;
    void clear() { block_ids_.clear(); }

  private:
    std::vector<std::size_t> block_ids_;
    std::size_t block_size_;
};

// Batched step metadata.  The reference's scheduler hands out sequences one at a
// time (src/scheduler/batch.rs:227-375) and paged_attention takes one BlockTable per call
// (src/attention/paged.rs:117-192).  The batched kernels take flat arrays instead; this builder
// turns the scheduled sequences of one step into them, with the reference's own slot arithmetic
// (block.rs:269-283) for the tokens whose K/V are written this step:
//   block_tables  int32 [num_seqs, max_blocks_per_seq]   (row padded with 0)
//   context_lens  int32 [num_seqs]                        (context INCLUDING the new tokens)
//   slot_mapping  int64 [num_tokens]                      (new tokens of all sequences, in order)
//   query_start   int32 [num_seqs + 1]                    (first new-token row of each sequence)
// rope.rs:42-80 RotaryEmbedding::new: cos / sin tables [max_seq_len, dim] in f32 (the caller
// converts to the model dtype like to_dtype there).  false when dim is odd or not positive.
bool rope_cache(std::size_t dim, std::size_t max_seq_len, double theta, float *cos_out,
                float *sin_out);

class BatchMetadata {
  public:
    explicit BatchMetadata(std::size_t max_blocks_per_seq) : max_blocks_(max_blocks_per_seq) {
        query_start_.push_back(0);
    }
    void clear();
    // drop every sequence after the first num_seqs (roll-back of a partly filled step)
    void truncate(std::size_t num_seqs);
    // context_len tokens are in the cache after this step, the last num_new_tokens of them are
    // new.  Errors (Config): more new tokens than context, context beyond the table's blocks
    // (the reference's narrow() fails there, paged.rs:150), table longer than max_blocks_per_seq.
    Error add_sequence(const BlockTable &table, std::size_t context_len, std::size_t num_new_tokens);
    std::size_t max_blocks_per_seq() const { return max_blocks_; }
    std::size_t num_seqs() const { return context_lens_.size(); }
    std::size_t num_tokens() const { return slot_mapping_.size(); }
    const std::vector<std::int32_t> &block_tables() const { return block_tables_; }
    const std::vector<std::int32_t> &context_lens() const { return context_lens_; }
    const std::vector<std::int64_t> &slot_mapping() const { return slot_mapping_; }
    const std::vector<std::int32_t> &query_start() const { return query_start_; }
    // position of every new token inside its sequence (RoPE row; rope.rs:107-113 start_pos + i)
    const std::vector<std::int32_t> &positions() const { return positions_; }

  private:
    std::size_t max_blocks_;
    std::vector<std::int32_t> block_tables_, context_lens_, query_start_, positions_;
    std::vector<std::int64_t> slot_mapping_;
};

// block_manager.rs:40-383
class BlockManager {
  public:
    BlockManager(std::size_t num_blocks, std::size_t block_size);
    std::size_t block_size() const { return block_size_; }
    std::size_t num_blocks() const { return num_blocks_; }
    std::size_t num_free_blocks() const { return free_list_.size(); }
    std::size_t num_used_blocks() const { return blocks_.size(); }
    bool has_free_blocks() const { return !free_list_.empty(); }
    bool can_allocate(std::size_t n) const { return free_list_.size() >= n; }

    bool allocate(std::size_t *block_id);                                // :140 (false => OutOfBlocks)
    bool allocate_many(std::size_t n, std::vector<std::size_t> *ids);    // :162
    bool free(std::size_t block_id);                                     // :187
    std::size_t free_many(const std::size_t *ids, std::size_t n);        // :215
    const Block *get_block(std::size_t block_id) const;                  // :220
    Block *get_block_mut(std::size_t block_id);                          // :225
    std::optional<std::size_t> increment_ref(std::size_t block_id);      // :236
    std::optional<std::size_t> decrement_ref(std::size_t block_id);      // :251
    std::optional<std::size_t> get_cached_block(std::uint64_t hash);     // :292
    void cache_block(std::size_t block_id, std::uint64_t hash);          // :312
    bool allocate_with_prefix(std::uint64_t hash, std::size_t *block_id, bool *was_cached);  // :337
    bool is_prefix_cached(std::uint64_t hash) const { return prefix_cache_.count(hash) != 0; }
    std::size_t num_cached_prefixes() const { return prefix_cache_.size(); }
    void clear_prefix_cache();                                           // :362
    void reset();                                                        // :377

  private:
    std::unordered_map<std::size_t, Block> blocks_;
    std::deque<std::size_t> free_list_;  // FIFO: pop_front / push_back (:141, :199)
    std::unordered_map<std::uint64_t, std::size_t> prefix_cache_;
    std::size_t block_size_;
    std::size_t num_blocks_;
};

// Sequential "last writer wins" of paged.rs:231-262 expressed as a mask: out[i]
// = slots[i] unless a later j > i has slots[j] == slots[i], then -1.
std::int64_t slot_mapping_last_writer(const std::int64_t *slots, std::int64_t n,
                                      std::int64_t *out);

}  // namespace core
}  // namespace nvllm
