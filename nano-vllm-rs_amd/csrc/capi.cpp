// capi.cpp -- extern "C" surface declared in include/nvllm_attn.h.
// Argument validation mirrors the reference's error behaviour (which variant
// of crate::error::Error a bad call produces); the kernels themselves live in
// the .hip files.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../include/nvllm_attn.h"
#include "nvllm_host.hpp"
#include "nvllm_kernels.hpp"
#include "nvllm_sched.hpp"

using nvllm::ErrorKind;
using nvllm::set_last_error;
namespace core = nvllm::core;
namespace kernels = nvllm::kernels;

struct nvllm_block_table {
    core::BlockTable table;
    explicit nvllm_block_table(std::size_t bs) : table(bs) {}
};
struct nvllm_batch {
    core::BatchMetadata meta;
    explicit nvllm_batch(std::size_t mb) : meta(mb) {}
};
struct nvllm_scheduler {
    core::Scheduler sched;
    core::SchedulerOutputs out;                  // result of the last schedule()
    std::vector<int64_t> out_chunk_sizes;        // aligned with out.prefill_sequences
    nvllm_scheduler(const core::SchedulerConfig &c, std::size_t bs, std::size_t nb) : sched(c, bs, nb) {}
};
struct nvllm_block_manager {
    core::BlockManager mgr;
    nvllm_block_manager(std::size_t n, std::size_t bs) : mgr(n, bs) {}
};

namespace {

int fail(ErrorKind kind, const std::string &msg) {
    set_last_error(kind, msg);
    return static_cast<int>(kind);
}

int hip_status(hipError_t err, const char *what) {
    if (err == hipSuccess) return NVLLM_OK;
    return fail(ErrorKind::Hip, std::string(what) + ": " + hipGetErrorString(err));
}

hipStream_t as_stream(nvllm_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

bool aligned16(const void *p) { return (reinterpret_cast<std::uintptr_t>(p) & 15u) == 0; }

int flash_common(kernels::DType dt, const void *q, const void *k, const void *v, void *o,
                 float *lse, int batch, int seq_q, int seq_kv, int num_heads, int num_kv_heads,
                 int head_dim, float softmax_scale, int causal, nvllm_stream_t stream) {
    if (!q || !k || !v || !o) return fail(ErrorKind::Null, "flash_attn: null tensor pointer");
    if (batch < 0 || seq_q < 0 || seq_kv < 0 || num_heads <= 0 || num_kv_heads <= 0 ||
        head_dim <= 0)
        return fail(ErrorKind::Config, "flash_attn: negative or zero dimension");
    if (num_heads % num_kv_heads != 0)
        return fail(ErrorKind::Config, "flash_attn: num_heads must be a multiple of num_kv_heads");
    if (head_dim > 256) return fail(ErrorKind::Unsupported, "flash_attn: head_dim > 256");
    if (batch == 0 || seq_q == 0) return NVLLM_OK;
    if (seq_kv == 0) return fail(ErrorKind::Tensor, "flash_attn: seq_kv == 0");
    kernels::FlashAttnArgs a{q, k, v, o, lse, batch, seq_q, seq_kv, num_heads, num_kv_heads,
                             head_dim, softmax_scale, causal};
    return hip_status(kernels::launch_flash_attn(dt, a, as_stream(stream)), "flash_attn launch");
}

int paged_common(kernels::DType dt, const void *q, const void *k_cache, const void *v_cache,
                 const int32_t *block_tables, const int32_t *context_lens, void *out,
                 void *workspace, size_t workspace_bytes, int num_seqs, int num_new_tokens,
                 int num_heads, int num_kv_heads, int head_dim, int block_size,
                 int max_blocks_per_seq, int64_t num_cache_blocks, float scale,
                 nvllm_stream_t stream, bool noalloc = false) {
    if (!q || !k_cache || !v_cache || !block_tables || !context_lens || !out)
        return fail(ErrorKind::Null, "paged_attn: null pointer");
    if (num_seqs < 0 || num_new_tokens <= 0 || num_heads <= 0 || num_kv_heads <= 0 ||
        head_dim <= 0 || block_size <= 0 || num_cache_blocks <= 0)
        return fail(ErrorKind::Config, "paged_attn: negative or zero dimension");
    if (max_blocks_per_seq <= 0)  // paged.rs:133-137 "BlockTable is empty"
        return fail(ErrorKind::Config, "BlockTable is empty - no blocks allocated");
    if (num_heads % num_kv_heads != 0)
        return fail(ErrorKind::Config, "paged_attn: num_heads must be a multiple of num_kv_heads");
    if (head_dim > 256) return fail(ErrorKind::Unsupported, "paged_attn: head_dim > 256");
    if (num_seqs > 65535) return fail(ErrorKind::Unsupported, "paged_attn: num_seqs > 65535 per launch");
    if (num_seqs == 0) return NVLLM_OK;
    kernels::PagedAttnArgs a{q, k_cache, v_cache, block_tables, context_lens, out, workspace,
                             workspace ? workspace_bytes : 0, num_seqs, num_new_tokens, num_heads,
                             num_kv_heads, head_dim, block_size, max_blocks_per_seq,
                             num_cache_blocks, scale};
    if (kernels::paged_attn_fast_path(dt, num_seqs, num_new_tokens, num_heads, num_kv_heads, head_dim,
                                      block_size, max_blocks_per_seq) &&
        !(aligned16(q) && aligned16(k_cache) && aligned16(v_cache) && aligned16(out)))
        return fail(ErrorKind::Config, "paged_attn: tensors must be 16-byte aligned");
    // the split decision is a function of the launch shape (and the launch policy) alone, and
    // nvllm_paged_attn_workspace_bytes is that same function: a smaller workspace is a caller
    // error, not a reason to launch differently
    // (what this launch needs: the shape's figure, or 0 when the generic kernel runs)
    const size_t need = kernels::paged_attn_launch_workspace_bytes(dt, a);
    if (noalloc) {
        if (need > kernels::kNoAllocScratchBytes)
            return fail(ErrorKind::Unsupported,
                        "paged_attn_noalloc: this launch shape needs " + std::to_string(need) +
                            " bytes of split-K scratch, the built-in scratch holds " +
                            std::to_string(kernels::kNoAllocScratchBytes) +
                            "; use nvllm_paged_attn_bf16 with a caller workspace");
        return hip_status(kernels::launch_paged_attn_noalloc(dt, a, as_stream(stream)),
                          "paged_attn_noalloc launch");
    }
    if (need > 0 && (!workspace || workspace_bytes < need))
        return fail(ErrorKind::Config, "paged_attn: workspace of " + std::to_string(workspace ? workspace_bytes : 0) +
                                           " bytes, this launch shape needs " + std::to_string(need) +
                                           " (nvllm_paged_attn_workspace_bytes)");
    if (need > 0 && !aligned16(workspace))
        return fail(ErrorKind::Config, "paged_attn: workspace must be 16-byte aligned");
    return hip_status(kernels::launch_paged_attn(dt, a, as_stream(stream)), "paged_attn launch");
}

int reshape_common(int elt, const void *key, const void *value, void *k_cache, void *v_cache,
                   const int64_t *slot_mapping, int64_t num_tokens, int num_kv_heads, int head_dim,
                   int block_size, int64_t num_cache_blocks, nvllm_stream_t stream) {
    if (num_tokens < 0 || num_kv_heads <= 0 || head_dim <= 0 || block_size <= 0 ||
        num_cache_blocks <= 0)
        return fail(ErrorKind::Config, "reshape_and_cache: negative or zero dimension");
    if (num_tokens == 0) return NVLLM_OK;
    if (!key || !value || !k_cache || !v_cache || !slot_mapping)
        return fail(ErrorKind::Null, "reshape_and_cache: null pointer");
    int64_t row_bytes = (int64_t)num_kv_heads * head_dim * elt;
    if (!(aligned16(key) && aligned16(value) && aligned16(k_cache) && aligned16(v_cache)) &&
        row_bytes % 16 == 0)
        return fail(ErrorKind::Config, "reshape_and_cache: tensors must be 16-byte aligned");
    return hip_status(kernels::launch_reshape_and_cache(key, value, k_cache, v_cache, slot_mapping,
                                                        num_tokens, row_bytes,
                                                        num_cache_blocks * block_size,
                                                        as_stream(stream)),
                      "reshape_and_cache launch");
}

int qk_norm_rope_common(kernels::DType dt, const void *q, const void *k, const void *v, void *q_out,
                        void *k_out, const void *q_weight, const void *k_weight, float eps,
                        const void *cos_cache, const void *sin_cache, int64_t max_positions,
                        const int32_t *positions, void *k_cache, void *v_cache,
                        const int64_t *slot_mapping, int64_t num_tokens, int num_heads,
                        int num_kv_heads, int head_dim, int block_size, int64_t num_cache_blocks,
                        nvllm_stream_t stream) {
    if (num_tokens < 0 || num_heads <= 0 || num_kv_heads <= 0 || head_dim <= 0 || block_size <= 0 ||
        num_cache_blocks <= 0 || max_positions <= 0 || max_positions > 0x7fffffff)
        return fail(ErrorKind::Config, "qk_norm_rope_cache: negative or zero dimension");
    if (head_dim % 2 != 0)  // rope.rs:49 assert!(dim.is_multiple_of(2))
        return fail(ErrorKind::Config, "qk_norm_rope_cache: RoPE dimension must be even");
    if (num_tokens == 0) return NVLLM_OK;
    if (!q || !k || !v || !q_out || !q_weight || !k_weight || !cos_cache || !sin_cache ||
        !positions || !k_cache || !v_cache || !slot_mapping)
        return fail(ErrorKind::Null, "qk_norm_rope_cache: null pointer");
    kernels::QkNormRopeCacheArgs a{q, k, v, q_out, k_out, q_weight, k_weight, eps, cos_cache,
                                   sin_cache, (int)max_positions, positions, k_cache, v_cache,
                                   slot_mapping, num_tokens, num_heads, num_kv_heads, head_dim,
                                   block_size, num_cache_blocks};
    return hip_status(kernels::launch_qk_norm_rope_cache(dt, a, as_stream(stream)),
                      "qk_norm_rope_cache launch");
}

}  // namespace

extern "C" {

int nvllm_flash_attn_fwd_bf16(const void *q, const void *k, const void *v, void *o, float *lse,
                              int batch, int seq_q, int seq_kv, int num_heads, int num_kv_heads,
                              int head_dim, float softmax_scale, int causal,
                              nvllm_stream_t stream) {
    return flash_common(kernels::DType::BF16, q, k, v, o, lse, batch, seq_q, seq_kv, num_heads,
                        num_kv_heads, head_dim, softmax_scale, causal, stream);
}

int nvllm_flash_attn_fwd_f32(const void *q, const void *k, const void *v, void *o, float *lse,
                             int batch, int seq_q, int seq_kv, int num_heads, int num_kv_heads,
                             int head_dim, float softmax_scale, int causal,
                             nvllm_stream_t stream) {
    return flash_common(kernels::DType::F32, q, k, v, o, lse, batch, seq_q, seq_kv, num_heads,
                        num_kv_heads, head_dim, softmax_scale, causal, stream);
}

int nvllm_paged_attn_bf16(const void *q, const void *k_cache, const void *v_cache,
                          const int32_t *block_tables, const int32_t *context_lens, void *out,
                          void *workspace, size_t workspace_bytes, int num_seqs,
                          int num_new_tokens, int num_heads, int num_kv_heads, int head_dim,
                          int block_size, int max_blocks_per_seq, int64_t num_cache_blocks,
                          float scale, nvllm_stream_t stream) {
    return paged_common(kernels::DType::BF16, q, k_cache, v_cache, block_tables, context_lens, out,
                        workspace, workspace_bytes, num_seqs, num_new_tokens, num_heads,
                        num_kv_heads, head_dim, block_size, max_blocks_per_seq, num_cache_blocks,
                        scale, stream);
}

int nvllm_paged_attn_bf16_noalloc(const void *q, const void *k_cache, const void *v_cache,
                                  const int32_t *block_tables, const int32_t *context_lens, void *out,
                                  int num_seqs, int num_new_tokens, int num_heads, int num_kv_heads,
                                  int head_dim, int block_size, int max_blocks_per_seq,
                                  int64_t num_cache_blocks, float scale, nvllm_stream_t stream) {
    return paged_common(kernels::DType::BF16, q, k_cache, v_cache, block_tables, context_lens, out,
                        nullptr, 0, num_seqs, num_new_tokens, num_heads, num_kv_heads, head_dim,
                        block_size, max_blocks_per_seq, num_cache_blocks, scale, stream, true);
}

size_t nvllm_paged_attn_noalloc_max_scratch_bytes(void) { return kernels::kNoAllocScratchBytes; }

int nvllm_paged_attn_f32(const void *q, const void *k_cache, const void *v_cache,
                         const int32_t *block_tables, const int32_t *context_lens, void *out,
                         void *workspace, size_t workspace_bytes, int num_seqs, int num_new_tokens,
                         int num_heads, int num_kv_heads, int head_dim, int block_size,
                         int max_blocks_per_seq, int64_t num_cache_blocks, float scale,
                         nvllm_stream_t stream) {
    return paged_common(kernels::DType::F32, q, k_cache, v_cache, block_tables, context_lens, out,
                        workspace, workspace_bytes, num_seqs, num_new_tokens, num_heads,
                        num_kv_heads, head_dim, block_size, max_blocks_per_seq, num_cache_blocks,
                        scale, stream);
}

size_t nvllm_paged_attn_workspace_bytes(int num_seqs, int num_new_tokens, int num_heads,
                                        int num_kv_heads, int head_dim, int block_size,
                                        int max_blocks_per_seq) {
    if (num_seqs <= 0 || num_new_tokens <= 0 || num_heads <= 0 || num_kv_heads <= 0) return 0;
    return kernels::paged_attn_workspace_bytes(num_seqs, num_new_tokens, num_heads, num_kv_heads,
                                               head_dim, block_size, max_blocks_per_seq);
}

int nvllm_reshape_and_cache_bf16(const void *key, const void *value, void *k_cache, void *v_cache,
                                 const int64_t *slot_mapping, int64_t num_tokens,
                                 int num_kv_heads, int head_dim, int block_size,
                                 int64_t num_cache_blocks, nvllm_stream_t stream) {
    return reshape_common(2, key, value, k_cache, v_cache, slot_mapping, num_tokens, num_kv_heads,
                          head_dim, block_size, num_cache_blocks, stream);
}

int nvllm_reshape_and_cache_f32(const void *key, const void *value, void *k_cache, void *v_cache,
                                const int64_t *slot_mapping, int64_t num_tokens, int num_kv_heads,
                                int head_dim, int block_size, int64_t num_cache_blocks,
                                nvllm_stream_t stream) {
    return reshape_common(4, key, value, k_cache, v_cache, slot_mapping, num_tokens, num_kv_heads,
                          head_dim, block_size, num_cache_blocks, stream);
}

int nvllm_gather_cache_blocks(const void *cache, const int32_t *block_ids, int64_t num_ids, void *out,
                              int64_t num_cache_blocks, int block_size, int num_kv_heads,
                              int head_dim, int dtype, nvllm_stream_t stream) {
    if (num_ids <= 0)  // kv_cache.rs:124-126
        return fail(ErrorKind::Config, "block_ids cannot be empty");
    if (!cache || !block_ids || !out) return fail(ErrorKind::Null, "gather_cache_blocks: null pointer");
    if (num_cache_blocks <= 0 || block_size <= 0 || num_kv_heads <= 0 || head_dim <= 0)
        return fail(ErrorKind::Config, "gather_cache_blocks: negative or zero dimension");
    if (dtype != 0 && dtype != 1) return fail(ErrorKind::Unsupported, "gather_cache_blocks: dtype");
    const int64_t block_bytes = (int64_t)block_size * num_kv_heads * head_dim * (dtype == 0 ? 2 : 4);
    if (block_bytes % 16 != 0 || !aligned16(cache) || !aligned16(out))
        return fail(ErrorKind::Unsupported, "gather_cache_blocks: blocks must be 16-byte multiples, 16-byte aligned");
    return hip_status(kernels::launch_gather_blocks(cache, block_ids, out, num_ids, block_bytes,
                                                    num_cache_blocks, as_stream(stream)),
                      "gather_cache_blocks launch");
}

const char *nvllm_get_last_error(void) { return nvllm::last_error_message(); }

int nvllm_synchronize(void) {
    hipError_t err = hipDeviceSynchronize();
    if (err != hipSuccess) {
        set_last_error(ErrorKind::Hip, std::string("synchronize: ") + hipGetErrorString(err));
        return -1;
    }
    return 0;
}

int nvllm_debug_set_policy(const char *name, int value) {
    if (!kernels::set_launch_policy(name, value))
        return fail(ErrorKind::Config, std::string("unknown launch-policy name: ") + (name ? name : "(null)"));
    return NVLLM_OK;
}

void nvllm_debug_reset_policy(void) { kernels::reset_launch_policy(); }

int nvllm_debug_validate_tables(const int32_t *block_tables, const int32_t *context_lens, int num_seqs,
                                int max_blocks_per_seq, int block_size, int64_t num_cache_blocks,
                                int32_t *counts, nvllm_stream_t stream) {
    if (!block_tables || !context_lens || !counts) return fail(ErrorKind::Null, "debug_validate_tables: null pointer");
    if (num_seqs <= 0 || max_blocks_per_seq <= 0 || block_size <= 0 || num_cache_blocks <= 0)
        return fail(ErrorKind::Config, "debug_validate_tables: negative or zero dimension");
    return hip_status(kernels::launch_validate_tables(block_tables, context_lens, num_seqs, max_blocks_per_seq,
                                                      block_size, num_cache_blocks, counts, as_stream(stream)),
                      "debug_validate_tables launch");
}

int64_t nvllm_kv_cache_layer_bytes(int64_t num_blocks, int64_t block_size, int64_t num_kv_heads,
                                   int64_t head_dim, int dtype) {
    if (num_blocks < 0 || block_size < 0 || num_kv_heads < 0 || head_dim < 0 || (dtype != 0 && dtype != 1))
        return -1;
    return num_blocks * block_size * num_kv_heads * head_dim * (dtype == 0 ? 2 : 4);  // kv_cache.rs:62-65
}

int64_t nvllm_kv_cache_total_bytes(int64_t num_blocks, int64_t block_size, int64_t num_kv_heads,
                                   int64_t head_dim, int64_t num_layers, int dtype) {
    const int64_t layer = nvllm_kv_cache_layer_bytes(num_blocks, block_size, num_kv_heads, head_dim, dtype);
    if (layer < 0 || num_layers < 0) return -1;
    return layer * num_layers * 2;  // kv_cache.rs:68-70: K and V
}

int nvllm_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int nvllm_set_device(int device) {
    const int rc = hip_status(hipSetDevice(device), "set_device");
    if (rc != NVLLM_OK) return rc;
    // the no-allocation entry's scratch address on this device, looked up here so that its launch
    // path never does (a symbol lookup is not a stream operation: not for a capturing stream)
    return hip_status(kernels::noalloc_prepare_device(), "set_device: scratch address");
}

int nvllm_paged_attn_path(int num_seqs, int num_new_tokens, int num_heads, int num_kv_heads,
                          int head_dim, int block_size, int max_blocks_per_seq,
                          int64_t num_cache_blocks, int dtype) {
    if (num_seqs <= 0 || num_new_tokens <= 0 || num_heads <= 0 || num_kv_heads <= 0 || head_dim <= 0 ||
        block_size <= 0 || max_blocks_per_seq <= 0 || num_cache_blocks <= 0 || (dtype != 0 && dtype != 1))
        return fail(ErrorKind::Config, "paged_attn_path: negative or zero dimension, or unknown dtype");
    if (num_heads % num_kv_heads != 0)
        return fail(ErrorKind::Config, "paged_attn_path: num_heads must be a multiple of num_kv_heads");
    kernels::PagedAttnArgs a{};
    a.num_seqs = num_seqs;
    a.num_new_tokens = num_new_tokens;
    a.num_heads = num_heads;
    a.num_kv_heads = num_kv_heads;
    a.head_dim = head_dim;
    a.block_size = block_size;
    a.max_blocks_per_seq = max_blocks_per_seq;
    a.num_cache_blocks = num_cache_blocks;
    return (int)kernels::paged_attn_path(dtype == 0 ? kernels::DType::BF16 : kernels::DType::F32, a);
}

/* ---- host logic --------------------------------------------------------- */

int64_t nvllm_compute_num_blocks(int64_t seq_len, int64_t block_size) {
    if (seq_len <= 0 || block_size <= 0) return 0;
    return (int64_t)core::compute_num_blocks((std::size_t)seq_len, (std::size_t)block_size);
}

int nvllm_qk_norm_rope_cache_bf16(const void *q, const void *k, const void *v, void *q_out,
                                  void *k_out, const void *q_weight, const void *k_weight,
                                  float eps, const void *cos_cache, const void *sin_cache,
                                  int64_t max_positions, const int32_t *positions, void *k_cache,
                                  void *v_cache, const int64_t *slot_mapping, int64_t num_tokens,
                                  int num_heads, int num_kv_heads, int head_dim, int block_size,
                                  int64_t num_cache_blocks, nvllm_stream_t stream) {
    return qk_norm_rope_common(kernels::DType::BF16, q, k, v, q_out, k_out, q_weight, k_weight, eps,
                               cos_cache, sin_cache, max_positions, positions, k_cache, v_cache,
                               slot_mapping, num_tokens, num_heads, num_kv_heads, head_dim,
                               block_size, num_cache_blocks, stream);
}

int nvllm_qk_norm_rope_cache_f32(const void *q, const void *k, const void *v, void *q_out,
                                 void *k_out, const void *q_weight, const void *k_weight, float eps,
                                 const void *cos_cache, const void *sin_cache,
                                 int64_t max_positions, const int32_t *positions, void *k_cache,
                                 void *v_cache, const int64_t *slot_mapping, int64_t num_tokens,
                                 int num_heads, int num_kv_heads, int head_dim, int block_size,
                                 int64_t num_cache_blocks, nvllm_stream_t stream) {
    return qk_norm_rope_common(kernels::DType::F32, q, k, v, q_out, k_out, q_weight, k_weight, eps,
                               cos_cache, sin_cache, max_positions, positions, k_cache, v_cache,
                               slot_mapping, num_tokens, num_heads, num_kv_heads, head_dim,
                               block_size, num_cache_blocks, stream);
}

int nvllm_rope_cache_f32(int64_t dim, int64_t max_seq_len, double theta, float *cos_out,
                         float *sin_out) {
    if (dim <= 0 || max_seq_len < 0) return fail(ErrorKind::Config, "rope_cache: bad dimension");
    if (dim % 2 != 0) return fail(ErrorKind::Config, "RoPE dimension must be even");
    if (!cos_out || !sin_out) return fail(ErrorKind::Null, "rope_cache: null pointer");
    core::rope_cache((std::size_t)dim, (std::size_t)max_seq_len, theta, cos_out, sin_out);
    return NVLLM_OK;
}

uint64_t nvllm_hash_token_block(const uint32_t *token_ids, int64_t num_tokens, int has_parent,
                                uint64_t parent_hash) {
    return core::hash_token_block(token_ids, num_tokens > 0 ? (std::size_t)num_tokens : 0,
                                  has_parent ? std::optional<std::uint64_t>(parent_hash)
                                             : std::nullopt);
}

int64_t nvllm_slot_mapping_last_writer(const int64_t *slots, int64_t n, int64_t *out) {
    if (n <= 0) return 0;
    return core::slot_mapping_last_writer(slots, n, out);
}

nvllm_block_table *nvllm_block_table_new(int64_t block_size) {
    if (block_size <= 0) {
        set_last_error(ErrorKind::Config, "BlockTable: block_size must be positive");
        return nullptr;
    }
    return new (std::nothrow) nvllm_block_table((std::size_t)block_size);
}
void nvllm_block_table_free(nvllm_block_table *t) { delete t; }
int64_t nvllm_block_table_block_size(const nvllm_block_table *t) {
    return (int64_t)t->table.block_size();
}
void nvllm_block_table_append_block(nvllm_block_table *t, int64_t block_id) {
    t->table.append_block((std::size_t)block_id);
}
int64_t nvllm_block_table_num_blocks(const nvllm_block_table *t) {
    return (int64_t)t->table.num_blocks();
}
int nvllm_block_table_is_empty(const nvllm_block_table *t) { return t->table.is_empty() ? 1 : 0; }
void nvllm_block_table_clear(nvllm_block_table *t) { t->table.clear(); }
int nvllm_block_table_get_block_id(const nvllm_block_table *t, int64_t logical_idx,
                                   int64_t *block_id_out) {
    std::size_t id = 0;
    if (logical_idx < 0 || !t->table.get_block_id((std::size_t)logical_idx, &id))
        return fail(ErrorKind::BlockIndexOutOfBounds,
                    "logical block " + std::to_string(logical_idx) + " not allocated (table has " +
                        std::to_string(t->table.num_blocks()) + " blocks)");
    *block_id_out = (int64_t)id;
    return NVLLM_OK;
}
int64_t nvllm_block_table_get_physical_block_ids(const nvllm_block_table *t, int64_t *ids_out,
                                                 int64_t cap) {
    const auto &ids = t->table.get_physical_block_ids();
    for (int64_t i = 0; i < cap && i < (int64_t)ids.size(); ++i) ids_out[i] = (int64_t)ids[i];
    return (int64_t)ids.size();
}
int64_t nvllm_block_table_get_slot_mapping(const nvllm_block_table *t, int64_t seq_len,
                                           int64_t *slots_out, int64_t cap) {
    if (seq_len <= 0) return 0;
    const auto slots = t->table.get_slot_mapping((std::size_t)seq_len);
    for (int64_t i = 0; i < cap && i < (int64_t)slots.size(); ++i) slots_out[i] = (int64_t)slots[i];
    return (int64_t)slots.size();
}

nvllm_batch *nvllm_batch_new(int64_t max_blocks_per_seq) {
    if (max_blocks_per_seq <= 0) {
        set_last_error(ErrorKind::Config, "batch: max_blocks_per_seq must be positive");
        return nullptr;
    }
    return new (std::nothrow) nvllm_batch((std::size_t)max_blocks_per_seq);
}
void nvllm_batch_free(nvllm_batch *b) { delete b; }
void nvllm_batch_clear(nvllm_batch *b) { b->meta.clear(); }
int nvllm_batch_add_sequence(nvllm_batch *b, const nvllm_block_table *t, int64_t context_len,
                             int64_t num_new_tokens) {
    if (!b || !t) return fail(ErrorKind::Null, "batch: null handle");
    if (context_len < 0 || num_new_tokens < 0) return fail(ErrorKind::Config, "batch: negative length");
    const nvllm::Error e = b->meta.add_sequence(t->table, (std::size_t)context_len, (std::size_t)num_new_tokens);
    if (!e.ok()) return fail(e.kind, e.message);
    return NVLLM_OK;
}
int64_t nvllm_batch_num_seqs(const nvllm_batch *b) { return (int64_t)b->meta.num_seqs(); }
int64_t nvllm_batch_num_tokens(const nvllm_batch *b) { return (int64_t)b->meta.num_tokens(); }
int64_t nvllm_batch_max_blocks_per_seq(const nvllm_batch *b) {
    return (int64_t)b->meta.max_blocks_per_seq();
}
const int32_t *nvllm_batch_block_tables(const nvllm_batch *b) { return b->meta.block_tables().data(); }
const int32_t *nvllm_batch_context_lens(const nvllm_batch *b) { return b->meta.context_lens().data(); }
const int64_t *nvllm_batch_slot_mapping(const nvllm_batch *b) { return b->meta.slot_mapping().data(); }
const int32_t *nvllm_batch_query_start(const nvllm_batch *b) { return b->meta.query_start().data(); }
const int32_t *nvllm_batch_positions(const nvllm_batch *b) { return b->meta.positions().data(); }
int nvllm_batch_upload(const nvllm_batch *b, void *d_block_tables, void *d_context_lens,
                       void *d_slot_mapping, void *d_query_start, void *d_positions,
                       nvllm_stream_t stream) {
    if (!b) return fail(ErrorKind::Null, "batch: null handle");
    const auto &m = b->meta;
    hipStream_t st = as_stream(stream);
    hipError_t err = hipSuccess;
    auto copy = [&](void *dst, const void *src, size_t bytes) {
        if (err == hipSuccess && dst && bytes) err = hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, st);
    };
    copy(d_block_tables, m.block_tables().data(), m.block_tables().size() * sizeof(int32_t));
    copy(d_context_lens, m.context_lens().data(), m.context_lens().size() * sizeof(int32_t));
    copy(d_slot_mapping, m.slot_mapping().data(), m.slot_mapping().size() * sizeof(int64_t));
    copy(d_query_start, m.query_start().data(), m.query_start().size() * sizeof(int32_t));
    copy(d_positions, m.positions().data(), m.positions().size() * sizeof(int32_t));
    if (err == hipSuccess) err = hipStreamSynchronize(st);  // the host arrays are pageable
    if (err != hipSuccess) return hip_status(err, "batch upload");
    return NVLLM_OK;
}

void nvllm_scheduler_config_default(nvllm_scheduler_config *c) {
    if (!c) return;
    const core::SchedulerConfig d;  // config.rs:59-69
    c->max_num_seqs = (int64_t)d.max_num_seqs;
    c->max_prefill_tokens = (int64_t)d.max_prefill_tokens;
    c->enable_chunked_prefill = d.enable_chunked_prefill;
    c->chunk_size = (int64_t)d.chunk_size;
    c->enable_priority = d.enable_priority;
    c->enable_preemption = d.enable_preemption;
    c->reschedule_partial_prefill = 0;
    c->enable_prefix_caching = 0;
}
nvllm_scheduler *nvllm_scheduler_new(const nvllm_scheduler_config *c, int64_t block_size,
                                     int64_t num_blocks) {
    if (!c || block_size <= 0 || num_blocks < 0 || c->max_num_seqs < 0 || c->max_prefill_tokens < 0 ||
        c->chunk_size < 0) {
        set_last_error(ErrorKind::Config, "Scheduler: bad configuration");
        return nullptr;
    }
    core::SchedulerConfig k;
    k.max_num_seqs = (std::size_t)c->max_num_seqs;
    k.max_prefill_tokens = (std::size_t)c->max_prefill_tokens;
    k.enable_chunked_prefill = c->enable_chunked_prefill != 0;
    k.chunk_size = (std::size_t)c->chunk_size;
    k.enable_priority = c->enable_priority != 0;
    k.enable_preemption = c->enable_preemption != 0;
    k.reschedule_partial_prefill = c->reschedule_partial_prefill != 0;
    k.enable_prefix_caching = c->enable_prefix_caching != 0;
    return new (std::nothrow) nvllm_scheduler(k, (std::size_t)block_size, (std::size_t)num_blocks);
}
void nvllm_scheduler_free(nvllm_scheduler *s) { delete s; }
int nvllm_scheduler_add_sequence(nvllm_scheduler *s, uint64_t seq_id, const uint32_t *prompt,
                                 int64_t num_tokens, int32_t priority) {
    if (!s || (num_tokens > 0 && !prompt)) return fail(ErrorKind::Null, "scheduler: null pointer");
    if (num_tokens < 0) return fail(ErrorKind::Config, "scheduler: negative prompt length");
    s->sched.add_sequence(core::Sequence(seq_id, std::vector<uint32_t>(prompt, prompt + num_tokens), priority));
    return NVLLM_OK;
}
int nvllm_scheduler_remove_sequence(nvllm_scheduler *s, uint64_t seq_id) {
    return s->sched.remove_sequence(seq_id).has_value() ? 1 : 0;
}
int nvllm_scheduler_schedule(nvllm_scheduler *s) {
    if (!s) return fail(ErrorKind::Null, "scheduler: null handle");
    s->out = s->sched.schedule();
    s->out_chunk_sizes.clear();
    for (uint64_t id : s->out.prefill_sequences) {
        const auto it = s->out.prefill_chunk_sizes.find(id);
        s->out_chunk_sizes.push_back(it == s->out.prefill_chunk_sizes.end() ? 0 : (int64_t)it->second);
    }
    return NVLLM_OK;
}
int64_t nvllm_scheduler_out_num_prefill(const nvllm_scheduler *s) { return (int64_t)s->out.prefill_sequences.size(); }
int64_t nvllm_scheduler_out_num_decode(const nvllm_scheduler *s) { return (int64_t)s->out.decode_sequences.size(); }
int64_t nvllm_scheduler_out_num_preempted(const nvllm_scheduler *s) { return (int64_t)s->out.preempted_sequences.size(); }
const uint64_t *nvllm_scheduler_out_prefill_ids(const nvllm_scheduler *s) { return s->out.prefill_sequences.data(); }
const uint64_t *nvllm_scheduler_out_decode_ids(const nvllm_scheduler *s) { return s->out.decode_sequences.data(); }
const uint64_t *nvllm_scheduler_out_preempted_ids(const nvllm_scheduler *s) { return s->out.preempted_sequences.data(); }
const int64_t *nvllm_scheduler_out_prefill_chunk_sizes(const nvllm_scheduler *s) { return s->out_chunk_sizes.data(); }
int64_t nvllm_scheduler_out_num_prefill_tokens(const nvllm_scheduler *s) { return (int64_t)s->out.num_prefill_tokens; }
int64_t nvllm_scheduler_out_num_decode_tokens(const nvllm_scheduler *s) { return (int64_t)s->out.num_decode_tokens; }
void nvllm_scheduler_finish_sequence(nvllm_scheduler *s, uint64_t seq_id, int reason) {
    s->sched.finish_sequence(seq_id, static_cast<core::FinishReason>(reason));
}
int nvllm_scheduler_append_token(nvllm_scheduler *s, uint64_t seq_id, uint32_t token) {
    if (!s->sched.append_token(seq_id, token))  // batch.rs:446
        return fail(ErrorKind::Config, "Sequence " + std::to_string(seq_id) + " not found");
    return NVLLM_OK;
}
int nvllm_scheduler_mark_prefilled(nvllm_scheduler *s, uint64_t seq_id, int64_t num_tokens) {
    if (num_tokens < 0) return fail(ErrorKind::Config, "scheduler: negative token count");
    if (!s->sched.mark_prefilled(seq_id, (std::size_t)num_tokens))  // batch.rs:456
        return fail(ErrorKind::Config, "Sequence " + std::to_string(seq_id) + " not found");
    return NVLLM_OK;
}
int64_t nvllm_scheduler_num_waiting(const nvllm_scheduler *s) { return (int64_t)s->sched.num_waiting(); }
int64_t nvllm_scheduler_num_running(const nvllm_scheduler *s) { return (int64_t)s->sched.num_running(); }
int64_t nvllm_scheduler_num_swapped(const nvllm_scheduler *s) { return (int64_t)s->sched.num_swapped(); }
int nvllm_scheduler_has_unfinished_sequences(const nvllm_scheduler *s) { return s->sched.has_unfinished_sequences(); }
int nvllm_scheduler_has_pending_requests(const nvllm_scheduler *s) { return s->sched.has_pending_requests(); }
int64_t nvllm_scheduler_running_sequence_ids(const nvllm_scheduler *s, uint64_t *out, int64_t cap) {
    const auto &ids = s->sched.running_sequence_ids();
    for (int64_t i = 0; i < (int64_t)ids.size() && i < cap; ++i) out[i] = ids[(std::size_t)i];
    return (int64_t)ids.size();
}
void nvllm_scheduler_reset(nvllm_scheduler *s) {
    s->sched.reset();
    s->out = core::SchedulerOutputs();
    s->out_chunk_sizes.clear();
}
int nvllm_scheduler_seq_status(const nvllm_scheduler *s, uint64_t seq_id) {
    const core::Sequence *q = s->sched.get_sequence(seq_id);
    return q ? static_cast<int>(q->status()) : -1;
}
int nvllm_scheduler_seq_set_status(nvllm_scheduler *s, uint64_t seq_id, int status) {
    core::Sequence *q = s->sched.get_sequence_mut(seq_id);
    if (!q) return fail(ErrorKind::Config, "Sequence " + std::to_string(seq_id) + " not found");
    static const char *names[] = {"Waiting", "Running", "Swapped", "Finished"};
    const int from = static_cast<int>(q->status());
    bool ok = false;
    switch (status) {
        case 0: ok = q->set_waiting(); break;
        case 1: ok = q->set_running(); break;
        case 2: ok = q->set_swapped(); break;
        default: return fail(ErrorKind::Config, "seq_set_status: use finish_sequence to finish");
    }
    if (!ok)  // Error::InvalidStateTransition (src/error.rs, sequence.rs:254-301)
        return fail(ErrorKind::Config, std::string("Invalid state transition from ") + names[from] +
                                           " to " + names[status]);
    return NVLLM_OK;
}
int nvllm_scheduler_seq_finish_reason(const nvllm_scheduler *s, uint64_t seq_id) {
    const core::Sequence *q = s->sched.get_sequence(seq_id);
    return q ? static_cast<int>(q->finish_reason()) : -1;
}
int64_t nvllm_scheduler_seq_info(const nvllm_scheduler *s, uint64_t seq_id, int what) {
    const core::Sequence *q = s->sched.get_sequence(seq_id);
    if (!q) return -1;
    switch (what) {
        case 0: return (int64_t)q->prompt_len();
        case 1: return (int64_t)q->output_len();
        case 2: return (int64_t)q->total_len();
        case 3: return (int64_t)q->num_prefilled_tokens();
        case 4: return (int64_t)q->num_tokens_to_prefill();
        case 5: return q->is_prefill_complete() ? 1 : 0;
        case 6: return (int64_t)q->block_table().num_blocks();
        case 7: return (int64_t)q->priority();
        case 8: return q->last_token_id() ? (int64_t)*q->last_token_id() : -1;
        default: return -1;
    }
}
int64_t nvllm_scheduler_seq_block_ids(const nvllm_scheduler *s, uint64_t seq_id, int64_t *out, int64_t cap) {
    const core::Sequence *q = s->sched.get_sequence(seq_id);
    if (!q) return -1;
    const auto &ids = q->block_table().get_physical_block_ids();
    for (int64_t i = 0; i < (int64_t)ids.size() && i < cap; ++i) out[i] = (int64_t)ids[(std::size_t)i];
    return (int64_t)ids.size();
}
int64_t nvllm_scheduler_seq_output_tokens(const nvllm_scheduler *s, uint64_t seq_id, uint32_t *out, int64_t cap) {
    const core::Sequence *q = s->sched.get_sequence(seq_id);
    if (!q) return -1;
    const auto &t = q->output_token_ids();
    for (int64_t i = 0; i < (int64_t)t.size() && i < cap; ++i) out[i] = t[(std::size_t)i];
    return (int64_t)t.size();
}
int64_t nvllm_scheduler_num_free_blocks(const nvllm_scheduler *s) { return (int64_t)s->sched.block_manager().num_free_blocks(); }
int64_t nvllm_scheduler_num_used_blocks(const nvllm_scheduler *s) { return (int64_t)s->sched.block_manager().num_used_blocks(); }
int64_t nvllm_scheduler_num_cached_prefixes(const nvllm_scheduler *s) { return (int64_t)s->sched.block_manager().num_cached_prefixes(); }
int64_t nvllm_scheduler_block_ref_count(const nvllm_scheduler *s, int64_t block_id) {
    const core::Block *b = block_id < 0 ? nullptr : s->sched.block_manager().get_block((std::size_t)block_id);
    return b ? (int64_t)b->ref_count() : -1;
}
int nvllm_scheduler_fill_decode_batch(const nvllm_scheduler *s, nvllm_batch *batch) {
    if (!s || !batch) return fail(ErrorKind::Null, "scheduler: null handle");
    const nvllm::Error e = s->sched.fill_decode_batch(s->out, &batch->meta);
    if (!e.ok()) return fail(e.kind, e.message);
    return NVLLM_OK;
}
int64_t nvllm_scheduler_fill_prefill_batch(const nvllm_scheduler *s, int64_t num_new_tokens,
                                           nvllm_batch *batch, uint64_t *ids_out, int64_t cap) {
    if (!s || !batch) return fail(ErrorKind::Null, "scheduler: null handle");
    if (num_new_tokens <= 0) return fail(ErrorKind::Config, "scheduler: chunk length must be positive");
    std::vector<core::SequenceId> added;
    const nvllm::Error e = s->sched.fill_prefill_batch(s->out, (std::size_t)num_new_tokens, &batch->meta, &added);
    if (!e.ok()) return fail(e.kind, e.message);
    for (int64_t i = 0; i < (int64_t)added.size() && ids_out && i < cap; ++i) ids_out[i] = added[(std::size_t)i];
    return (int64_t)added.size();
}

nvllm_block_manager *nvllm_block_manager_new(int64_t num_blocks, int64_t block_size) {
    if (num_blocks < 0 || block_size <= 0) {
        set_last_error(ErrorKind::Config, "BlockManager: bad num_blocks / block_size");
        return nullptr;
    }
    return new (std::nothrow) nvllm_block_manager((std::size_t)num_blocks, (std::size_t)block_size);
}
void nvllm_block_manager_free_handle(nvllm_block_manager *m) { delete m; }
int64_t nvllm_block_manager_block_size(const nvllm_block_manager *m) {
    return (int64_t)m->mgr.block_size();
}
int64_t nvllm_block_manager_num_blocks(const nvllm_block_manager *m) {
    return (int64_t)m->mgr.num_blocks();
}
int64_t nvllm_block_manager_num_free_blocks(const nvllm_block_manager *m) {
    return (int64_t)m->mgr.num_free_blocks();
}
int64_t nvllm_block_manager_num_used_blocks(const nvllm_block_manager *m) {
    return (int64_t)m->mgr.num_used_blocks();
}
int nvllm_block_manager_has_free_blocks(const nvllm_block_manager *m) {
    return m->mgr.has_free_blocks() ? 1 : 0;
}
int nvllm_block_manager_can_allocate(const nvllm_block_manager *m, int64_t n) {
    return (n >= 0 && m->mgr.can_allocate((std::size_t)n)) ? 1 : 0;
}
int nvllm_block_manager_allocate(nvllm_block_manager *m, int64_t *block_id_out) {
    std::size_t id = 0;
    if (!m->mgr.allocate(&id)) return fail(ErrorKind::OutOfBlocks, "out of KV cache blocks");
    *block_id_out = (int64_t)id;
    return NVLLM_OK;
}
int nvllm_block_manager_allocate_many(nvllm_block_manager *m, int64_t n, int64_t *ids_out) {
    std::vector<std::size_t> ids;
    if (n < 0 || !m->mgr.allocate_many((std::size_t)n, &ids))
        return fail(ErrorKind::OutOfBlocks, "out of KV cache blocks");
    for (std::size_t i = 0; i < ids.size(); ++i) ids_out[i] = (int64_t)ids[i];
    return NVLLM_OK;
}
int nvllm_block_manager_free(nvllm_block_manager *m, int64_t block_id) {
    return (block_id >= 0 && m->mgr.free((std::size_t)block_id)) ? 1 : 0;
}
int64_t nvllm_block_manager_free_many(nvllm_block_manager *m, const int64_t *ids, int64_t n) {
    int64_t freed = 0;
    for (int64_t i = 0; i < n; ++i)
        if (ids[i] >= 0 && m->mgr.free((std::size_t)ids[i])) ++freed;
    return freed;
}
int64_t nvllm_block_manager_ref_count(const nvllm_block_manager *m, int64_t block_id) {
    const core::Block *b = block_id >= 0 ? m->mgr.get_block((std::size_t)block_id) : nullptr;
    return b ? (int64_t)b->ref_count() : -1;
}
int nvllm_block_manager_prefix_hash(const nvllm_block_manager *m, int64_t block_id,
                                    uint64_t *hash_out) {
    const core::Block *b = block_id >= 0 ? m->mgr.get_block((std::size_t)block_id) : nullptr;
    if (!b || !b->prefix_hash()) return 0;
    if (hash_out) *hash_out = *b->prefix_hash();
    return 1;
}
int64_t nvllm_block_manager_increment_ref(nvllm_block_manager *m, int64_t block_id) {
    auto r = block_id >= 0 ? m->mgr.increment_ref((std::size_t)block_id) : std::nullopt;
    return r ? (int64_t)*r : -1;
}
int64_t nvllm_block_manager_decrement_ref(nvllm_block_manager *m, int64_t block_id) {
    auto r = block_id >= 0 ? m->mgr.decrement_ref((std::size_t)block_id) : std::nullopt;
    return r ? (int64_t)*r : -1;
}
int64_t nvllm_block_manager_get_cached_block(nvllm_block_manager *m, uint64_t prefix_hash) {
    auto r = m->mgr.get_cached_block(prefix_hash);
    return r ? (int64_t)*r : -1;
}
void nvllm_block_manager_cache_block(nvllm_block_manager *m, int64_t block_id,
                                     uint64_t prefix_hash) {
    if (block_id >= 0) m->mgr.cache_block((std::size_t)block_id, prefix_hash);
}
int nvllm_block_manager_allocate_with_prefix(nvllm_block_manager *m, uint64_t prefix_hash,
                                             int64_t *block_id_out, int *was_cached_out) {
    std::size_t id = 0;
    bool cached = false;
    if (!m->mgr.allocate_with_prefix(prefix_hash, &id, &cached))
        return fail(ErrorKind::OutOfBlocks, "out of KV cache blocks");
    *block_id_out = (int64_t)id;
    if (was_cached_out) *was_cached_out = cached ? 1 : 0;
    return NVLLM_OK;
}
int nvllm_block_manager_is_prefix_cached(const nvllm_block_manager *m, uint64_t prefix_hash) {
    return m->mgr.is_prefix_cached(prefix_hash) ? 1 : 0;
}
int64_t nvllm_block_manager_num_cached_prefixes(const nvllm_block_manager *m) {
    return (int64_t)m->mgr.num_cached_prefixes();
}
void nvllm_block_manager_clear_prefix_cache(nvllm_block_manager *m) { m->mgr.clear_prefix_cache(); }
void nvllm_block_manager_reset(nvllm_block_manager *m) { m->mgr.reset(); }

/* ---- reference-shaped operators ----------------------------------------- */

namespace {
constexpr size_t kAlign = 256;
size_t align_up(size_t x) { return (x + kAlign - 1) / kAlign * kAlign; }
}  // namespace

size_t nvllm_paged_attention_seq_scratch_bytes(int64_t num_table_blocks, int num_new_tokens,
                                               int num_heads, int num_kv_heads, int head_dim,
                                               int block_size) {
    if (num_table_blocks <= 0) return kAlign;
    // [table int32 x n][context_len int32] then the split-K workspace
    size_t meta = align_up(((size_t)num_table_blocks + 1) * sizeof(int32_t));
    return meta + nvllm_paged_attn_workspace_bytes(1, num_new_tokens, num_heads, num_kv_heads,
                                                   head_dim, block_size, (int)num_table_blocks);
}

int nvllm_paged_attention_seq(const void *query, int batch, int num_new_tokens,
                              const void *k_cache, const void *v_cache, int64_t num_cache_blocks,
                              const nvllm_block_table *block_table, int64_t context_len,
                              int num_heads, int num_kv_heads, int head_dim, double scale,
                              int dtype, void *out, void *scratch, size_t scratch_bytes,
                              nvllm_stream_t stream) {
    if (!block_table) return fail(ErrorKind::Null, "paged_attention: null block table");
    const auto &ids = block_table->table.get_physical_block_ids();
    if (ids.empty())  // paged.rs:133-137
        return fail(ErrorKind::Config, "BlockTable is empty - no blocks allocated");
    if (batch != 1)  // K/V get batch dim 1 (paged.rs:155-156); candle matmul rejects the mismatch
        return fail(ErrorKind::Tensor, "paged_attention: query batch must be 1");
    const int64_t bs = (int64_t)block_table->table.block_size();
    for (std::size_t id : ids)
        if ((int64_t)id >= num_cache_blocks)  // index_select out of range (kv_cache.rs:130)
            return fail(ErrorKind::Tensor, "paged_attention: block id out of range");
    if (context_len <= 0 || context_len > (int64_t)ids.size() * bs)  // narrow (paged.rs:151)
        return fail(ErrorKind::Tensor, "paged_attention: context_len exceeds allocated blocks");
    if (num_new_tokens > context_len)  // cached_len underflow (paged.rs:363)
        return fail(ErrorKind::Tensor, "paged_attention: num_new_tokens > context_len");
    if (dtype != 0 && dtype != 1) return fail(ErrorKind::Unsupported, "paged_attention: dtype");
    const size_t need = nvllm_paged_attention_seq_scratch_bytes(
        (int64_t)ids.size(), num_new_tokens, num_heads, num_kv_heads, head_dim, (int)bs);
    if (!scratch || scratch_bytes < need)
        return fail(ErrorKind::Config, "paged_attention: scratch buffer too small");
    std::vector<int32_t> meta(ids.size() + 1);
    for (std::size_t i = 0; i < ids.size(); ++i) meta[i] = (int32_t)ids[i];
    meta[ids.size()] = (int32_t)context_len;
    hipError_t err = hipMemcpyAsync(scratch, meta.data(), meta.size() * sizeof(int32_t),
                                    hipMemcpyHostToDevice, as_stream(stream));
    if (err == hipSuccess) err = hipStreamSynchronize(as_stream(stream));  // `meta` is pageable
    if (err != hipSuccess) return hip_status(err, "paged_attention: table upload");
    const int32_t *d_table = static_cast<const int32_t *>(scratch);
    const int32_t *d_ctx = d_table + ids.size();
    const size_t meta_bytes = align_up(meta.size() * sizeof(int32_t));
    void *ws = static_cast<unsigned char *>(scratch) + meta_bytes;
    const size_t ws_bytes = scratch_bytes - meta_bytes;
    if (dtype == 0)
        return nvllm_paged_attn_bf16(query, k_cache, v_cache, d_table, d_ctx, out, ws, ws_bytes, 1,
                                     num_new_tokens, num_heads, num_kv_heads, head_dim, (int)bs,
                                     (int)ids.size(), num_cache_blocks, (float)scale, stream);
    return nvllm_paged_attn_f32(query, k_cache, v_cache, d_table, d_ctx, out, ws, ws_bytes, 1,
                                num_new_tokens, num_heads, num_kv_heads, head_dim, (int)bs,
                                (int)ids.size(), num_cache_blocks, (float)scale, stream);
}

int nvllm_write_kv_to_cache(const void *key, const void *value, int64_t seq_len, void *k_cache,
                            void *v_cache, int64_t num_cache_blocks, const int64_t *slot_mapping,
                            int64_t slot_mapping_len, int block_size, int num_kv_heads,
                            int head_dim, int dtype, void *scratch, size_t scratch_bytes,
                            nvllm_stream_t stream) {
    if (slot_mapping_len != seq_len)  // paged.rs:218-224
        return fail(ErrorKind::Config, "slot_mapping length " + std::to_string(slot_mapping_len) +
                                           " doesn't match seq_len " + std::to_string(seq_len));
    if (dtype != 0 && dtype != 1) return fail(ErrorKind::Unsupported, "write_kv_to_cache: dtype");
    if (seq_len == 0) return NVLLM_OK;
    if (!slot_mapping) return fail(ErrorKind::Null, "write_kv_to_cache: null slot mapping");
    if (block_size <= 0) return fail(ErrorKind::Config, "write_kv_to_cache: block_size");
    for (int64_t i = 0; i < seq_len; ++i)
        if (slot_mapping[i] < 0 || slot_mapping[i] / block_size >= num_cache_blocks)  // narrow (:287)
            return fail(ErrorKind::Tensor, "write_kv_to_cache: slot outside the cache");
    if (!scratch || scratch_bytes < (size_t)seq_len * sizeof(int64_t))
        return fail(ErrorKind::Config, "write_kv_to_cache: scratch buffer too small");
    std::vector<int64_t> resolved((std::size_t)seq_len);
    core::slot_mapping_last_writer(slot_mapping, seq_len, resolved.data());
    hipError_t err = hipMemcpyAsync(scratch, resolved.data(), (size_t)seq_len * sizeof(int64_t),
                                    hipMemcpyHostToDevice, as_stream(stream));
    if (err == hipSuccess) err = hipStreamSynchronize(as_stream(stream));  // `resolved` is pageable
    if (err != hipSuccess) return hip_status(err, "write_kv_to_cache: slot upload");
    const int64_t *d_slots = static_cast<const int64_t *>(scratch);
    if (dtype == 0)
        return nvllm_reshape_and_cache_bf16(key, value, k_cache, v_cache, d_slots, seq_len,
                                            num_kv_heads, head_dim, block_size, num_cache_blocks,
                                            stream);
    return nvllm_reshape_and_cache_f32(key, value, k_cache, v_cache, d_slots, seq_len,
                                       num_kv_heads, head_dim, block_size, num_cache_blocks, stream);
}

int nvllm_prefill_attention(const void *q, const void *k, const void *v, void *out, int batch,
                            int seq_len, int num_heads, int num_kv_heads, int head_dim,
                            double scale, int causal, int dtype, nvllm_stream_t stream) {
    if (dtype != 0 && dtype != 1) return fail(ErrorKind::Unsupported, "prefill_attention: dtype");
    const int eff_causal = (causal && seq_len > 1) ? 1 : 0;  // paged.rs:79
    if (dtype == 0)
        return nvllm_flash_attn_fwd_bf16(q, k, v, out, nullptr, batch, seq_len, seq_len, num_heads,
                                         num_kv_heads, head_dim, (float)scale, eff_causal, stream);
    return nvllm_flash_attn_fwd_f32(q, k, v, out, nullptr, batch, seq_len, seq_len, num_heads,
                                    num_kv_heads, head_dim, (float)scale, eff_causal, stream);
}

}  // extern "C"
