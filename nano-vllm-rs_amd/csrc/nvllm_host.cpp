// nvllm_host.cpp -- see nvllm_host.hpp.
#include "nvllm_host.hpp"

#include <cmath>

#include <unordered_map>

namespace nvllm {

namespace {
thread_local std::string g_last_error = "no error";
}

void set_last_error(ErrorKind kind, const std::string &message) {
    (void)kind;
    g_last_error = message;
}

const char *last_error_message() { return g_last_error.c_str(); }

namespace core {

namespace {

inline std::uint64_t rotl(std::uint64_t x, int b) { return (x << b) | (x >> (64 - b)); }

// Streaming SipHash-1-3 state: std::hash::DefaultHasher (SipHasher13, k0=k1=0).
class SipHasher13 {
  public:
    SipHasher13() = default;
    void write(const std::uint8_t *bytes, std::size_t n) {
        for (std::size_t i = 0; i < n; ++i) {
            tail_ |= static_cast<std::uint64_t>(bytes[i]) << (8 * ntail_);
            ++ntail_;
            ++length_;
            if (ntail_ == 8) {
                compress(tail_);
                tail_ = 0;
                ntail_ = 0;
            }
        }
    }
    void write_u32(std::uint32_t v) {
        std::uint8_t b[4];
        for (int j = 0; j < 4; ++j) b[j] = static_cast<std::uint8_t>(v >> (8 * j));
        write(b, 4);
    }
    void write_u64(std::uint64_t v) {
        std::uint8_t b[8];
        for (int j = 0; j < 8; ++j) b[j] = static_cast<std::uint8_t>(v >> (8 * j));
        write(b, 8);
    }
    std::uint64_t finish() const {
        std::uint64_t v0 = v0_, v1 = v1_, v2 = v2_, v3 = v3_;
        const std::uint64_t b = (static_cast<std::uint64_t>(length_ & 0xff) << 56) | tail_;
        v3 ^= b;
        round(v0, v1, v2, v3);  // 1 compression round
        v0 ^= b;
        v2 ^= 0xff;
        round(v0, v1, v2, v3);  // 3 finalisation rounds
        round(v0, v1, v2, v3);
        round(v0, v1, v2, v3);
        return v0 ^ v1 ^ v2 ^ v3;
    }

  private:
    static void round(std::uint64_t &v0, std::uint64_t &v1, std::uint64_t &v2, std::uint64_t &v3) {
        v0 += v1; v1 = rotl(v1, 13); v1 ^= v0; v0 = rotl(v0, 32);
        v2 += v3; v3 = rotl(v3, 16); v3 ^= v2;
        v0 += v3; v3 = rotl(v3, 21); v3 ^= v0;
        v2 += v1; v1 = rotl(v1, 17); v1 ^= v2; v2 = rotl(v2, 32);
    }
    void compress(std::uint64_t m) {
        v3_ ^= m;
        round(v0_, v1_, v2_, v3_);
        v0_ ^= m;
    }
    std::uint64_t v0_ = 0x736f6d6570736575ULL, v1_ = 0x646f72616e646f6dULL,
                  v2_ = 0x6c7967656e657261ULL, v3_ = 0x7465646279746573ULL;
    std::uint64_t tail_ = 0;
    unsigned ntail_ = 0;
    std::uint64_t length_ = 0;
};

}  // namespace

std::uint64_t hash_token_block(const std::uint32_t *token_ids, std::size_t n,
                               std::optional<std::uint64_t> parent_hash) {
    SipHasher13 hasher;
    if (parent_hash) hasher.write_u64(*parent_hash);                  // block.rs:47-49
    for (std::size_t i = 0; i < n; ++i) hasher.write_u32(token_ids[i]);  // block.rs:52-54
    return hasher.finish();
}

bool BlockTable::get_block_id(std::size_t logical_block_idx, std::size_t *out) const {
    if (logical_block_idx >= block_ids_.size()) return false;
    *out = block_ids_[logical_block_idx];
    return true;
}

std::vector<std::size_t> BlockTable::get_slot_mapping(std::size_t seq_len) const {
    std::vector<std::size_t> slots;
    slots.reserve(seq_len);
    for (std::size_t pos = 0; pos < seq_len; ++pos) {
        const std::size_t logical_block = pos / block_size_;
        const std::size_t slot_in_block = pos % block_size_;
        if (logical_block < block_ids_.size())
            slots.push_back(block_ids_[logical_block] * block_size_ + slot_in_block);
    }
    return slots;
}

BlockManager::BlockManager(std::size_t num_blocks, std::size_t block_size)
    : block_size_(block_size), num_blocks_(num_blocks) {
    blocks_.reserve(num_blocks);
    for (std::size_t i = 0; i < num_blocks; ++i) free_list_.push_back(i);  // :71
}

bool BlockManager::allocate(std::size_t *block_id) {
    if (free_list_.empty()) return false;  // Error::OutOfBlocks
    const std::size_t id = free_list_.front();
    free_list_.pop_front();
    blocks_.insert_or_assign(id, Block(id, block_size_));
    *block_id = id;
    return true;
}

bool BlockManager::allocate_many(std::size_t n, std::vector<std::size_t> *ids) {
    if (!can_allocate(n)) return false;
    ids->clear();
    ids->reserve(n);
    for (std::size_t i = 0; i < n; ++i) {
        std::size_t id = 0;
        if (!allocate(&id)) return false;
        ids->push_back(id);
    }
    return true;
}

bool BlockManager::free(std::size_t block_id) {
    auto it = blocks_.find(block_id);
    if (it != blocks_.end()) {
        const std::size_t new_ref = it->second.decrement_ref();
        if (new_ref == 0) {
            if (auto h = it->second.prefix_hash()) prefix_cache_.erase(*h);  // :193-195
            blocks_.erase(it);
            free_list_.push_back(block_id);  // :199
            return true;
        }
    }
    return false;
}

std::size_t BlockManager::free_many(const std::size_t *ids, std::size_t n) {
    std::size_t freed = 0;
    for (std::size_t i = 0; i < n; ++i)
        if (free(ids[i])) ++freed;
    return freed;
}

const Block *BlockManager::get_block(std::size_t block_id) const {
    auto it = blocks_.find(block_id);
    return it == blocks_.end() ? nullptr : &it->second;
}

Block *BlockManager::get_block_mut(std::size_t block_id) {
    auto it = blocks_.find(block_id);
    return it == blocks_.end() ? nullptr : &it->second;
}

std::optional<std::size_t> BlockManager::increment_ref(std::size_t block_id) {
    Block *b = get_block_mut(block_id);
    if (!b) return std::nullopt;
    b->increment_ref();
    return b->ref_count();
}

std::optional<std::size_t> BlockManager::decrement_ref(std::size_t block_id) {
    Block *b = get_block_mut(block_id);
    if (!b) return std::nullopt;
    return b->decrement_ref();
}

std::optional<std::size_t> BlockManager::get_cached_block(std::uint64_t hash) {
    auto it = prefix_cache_.find(hash);
    if (it != prefix_cache_.end()) {
        const std::size_t block_id = it->second;
        if (blocks_.count(block_id)) {
            increment_ref(block_id);
            return block_id;
        }
        prefix_cache_.erase(it);  // :299-300 stale entry
    }
    return std::nullopt;
}

void BlockManager::cache_block(std::size_t block_id, std::uint64_t hash) {
    if (Block *b = get_block_mut(block_id)) {
        b->set_prefix_hash(hash);
        prefix_cache_[hash] = block_id;
    }
}

bool BlockManager::allocate_with_prefix(std::uint64_t hash, std::size_t *block_id,
                                        bool *was_cached) {
    if (auto cached = get_cached_block(hash)) {
        *block_id = *cached;
        *was_cached = true;
        return true;
    }
    if (!allocate(block_id)) return false;
    cache_block(*block_id, hash);
    *was_cached = false;
    return true;
}

void BlockManager::clear_prefix_cache() {
    for (auto &kv : prefix_cache_)
        if (Block *b = get_block_mut(kv.second)) b->set_prefix_hash(0);  // :368 (quirk kept)
    prefix_cache_.clear();
}

void BlockManager::reset() {
    blocks_.clear();
    prefix_cache_.clear();
    free_list_.clear();
    for (std::size_t i = 0; i < num_blocks_; ++i) free_list_.push_back(i);
}

std::int64_t slot_mapping_last_writer(const std::int64_t *slots, std::int64_t n,
                                      std::int64_t *out) {
    std::unordered_map<std::int64_t, std::int64_t> last;
    last.reserve(static_cast<std::size_t>(n) * 2);
    for (std::int64_t i = 0; i < n; ++i)
        if (slots[i] >= 0) last[slots[i]] = i;
    std::int64_t superseded = 0;
    for (std::int64_t i = 0; i < n; ++i) {
        if (slots[i] >= 0 && last[slots[i]] != i) {
            out[i] = -1;
            ++superseded;
        } else {
            out[i] = slots[i];
        }
    }
    return superseded;
}

bool rope_cache(std::size_t dim, std::size_t max_seq_len, double theta, float *cos_out,
                float *sin_out) {
    if (dim == 0 || (dim & 1)) return false;
    const std::size_t half = dim / 2;
    std::vector<float> inv_freq(half);
    // rope.rs:55: `1.0 / theta.powf(..) as f32` -- `as` binds tighter than `/`: the power is taken in
    // f64 and narrowed, the division is an f32 one
    for (std::size_t i = 0; i < half; ++i)
        inv_freq[i] = 1.0f / static_cast<float>(std::pow(theta, 2.0 * (double)i / (double)dim));
    for (std::size_t p = 0; p < max_seq_len; ++p)
        for (std::size_t i = 0; i < half; ++i) {
            const float f = static_cast<float>(p) * inv_freq[i];  // rope.rs:65 broadcast_mul in f32
            const float c = std::cos(f), sn = std::sin(f);
            cos_out[p * dim + i] = cos_out[p * dim + half + i] = c;  // rope.rs:69 cat(freqs, freqs)
            sin_out[p * dim + i] = sin_out[p * dim + half + i] = sn;
        }
    return true;
}

void BatchMetadata::truncate(std::size_t num_seqs) {
    if (num_seqs >= context_lens_.size()) return;
    const std::size_t tokens = static_cast<std::size_t>(query_start_[num_seqs]);
    block_tables_.resize(num_seqs * max_blocks_);
    context_lens_.resize(num_seqs);
    slot_mapping_.resize(tokens);
    positions_.resize(tokens);
    query_start_.resize(num_seqs + 1);
}

void BatchMetadata::clear() {
    block_tables_.clear();
    context_lens_.clear();
    slot_mapping_.clear();
    positions_.clear();
    query_start_.assign(1, 0);
}

Error BatchMetadata::add_sequence(const BlockTable &table, std::size_t context_len,
                                  std::size_t num_new_tokens) {
    const std::size_t bs = table.block_size();
    const auto &ids = table.get_physical_block_ids();
    if (num_new_tokens > context_len)
        return Error{ErrorKind::Config, "batch: " + std::to_string(num_new_tokens) +
                                            " new tokens exceed context_len " +
                                            std::to_string(context_len)};
    if (context_len > ids.size() * bs)
        return Error{ErrorKind::Config, "batch: context_len " + std::to_string(context_len) +
                                            " exceeds the " + std::to_string(ids.size()) +
                                            " allocated blocks"};
    if (ids.size() > max_blocks_)
        return Error{ErrorKind::Config, "batch: block table longer than max_blocks_per_seq"};
    if (context_len > 0x7fffffffu) return Error{ErrorKind::Config, "batch: context_len too large"};
    // every check comes before the first push_back: a rejected sequence leaves the batch untouched
    for (std::size_t id : ids)
        if (id > 0x7fffffffu) return Error{ErrorKind::Config, "batch: block id exceeds int32"};
    for (std::size_t i = 0; i < max_blocks_; ++i)
        block_tables_.push_back(static_cast<std::int32_t>(i < ids.size() ? ids[i] : 0));
    context_lens_.push_back(static_cast<std::int32_t>(context_len));
    for (std::size_t p = context_len - num_new_tokens; p < context_len; ++p) {  // block.rs:272-279
        slot_mapping_.push_back(static_cast<std::int64_t>(ids[p / bs] * bs + p % bs));
        positions_.push_back(static_cast<std::int32_t>(p));
    }
    query_start_.push_back(static_cast<std::int32_t>(slot_mapping_.size()));
    return Error{};
}

}  // namespace core
}  // namespace nvllm
