// nvllm_kernels.hpp -- internal launcher declarations (HIP side of the library).
#pragma once

#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

namespace nvllm {
namespace kernels {

enum class DType : int { BF16 = 0, F32 = 1 };

// Launch-policy overrides (launch_policy.cpp): test and tuning hooks that pin a decision the
// launchers otherwise take from the launch shape.  The NVLLM_* environment is read ONCE, on first
// use; afterwards only nvllm_debug_set_policy() changes a value, so that
// nvllm_paged_attn_workspace_bytes() and the launch always see the same policy and no launch reads the
// environment.  -1 / 0 = "decide from the shape" as noted per field.
struct LaunchPolicy {
    int paged_max_rows;     // NVLLM_PAGED_MAX_ROWS: > 0 pins the decode kernel's query-row limit
    int paged_splits;       // NVLLM_PAGED_SPLITS: > 0 forces the split-K factor
    int paged_wave_target;  // NVLLM_PAGED_WAVE_TARGET: (seq, kv head, split) waves aimed at (2048)
    int paged_nt;           // NVLLM_PAGED_NT: 0 / 1 non-temporal K/V loads, -1 by reach
    int paged_hpb;          // NVLLM_PAGED_HPB: 1, 2, 4, 8 kv heads per workgroup, 0 by grid size
    int paged_flash;        // NVLLM_PAGED_FLASH: 0 keeps long launches off the prefill kernel
    int k1_pairing;         // NVLLM_K1_PAIRING: 0 balanced ('b'), 1 adjacent ('a'), -1 by grid size
    int k1_pack;            // NVLLM_K1_PACK: 0 no head packing, -1 / 1 pack when the group allows
    int k1_ksplit;          // NVLLM_K1_KSPLIT: > 0 forces the key-split part count
    int paged_mid;          // NVLLM_PAGED_MID: 0 never / 1 whenever supported / -1 by launch size (paged_attn_mid.hip);
                            // 2 = as 1 with rows by wave index instead of by SIMD (the kernel's fallback, for tests)
    int k1_chain;           // NVLLM_K1_CHAIN: items per workgroup (0 / 1 = no chaining), -1 by grid size
};
// a snapshot by value (each field read atomically).  A workspace query and the launch it sizes see
// the same policy unless nvllm_debug_set_policy runs between them: that hook is for tests and
// tuning scripts and is not meant to race with launches in flight.
LaunchPolicy launch_policy();
// name = the environment variable's name; false if unknown.  NVLLM_PAGED_WAVE_TARGET is clamped to >= 1.
bool set_launch_policy(const char *name, int value);
void reset_launch_policy();  // back to the environment's values
// compute units of the CURRENT device (cached per device id)
int device_cus();

// K4: reshape_and_cache.hip
hipError_t launch_reshape_and_cache(const void *key, const void *value, void *k_cache,
                                    void *v_cache, const int64_t *slot_mapping,
                                    int64_t num_tokens, int64_t row_bytes, int64_t num_slots,
                                    hipStream_t stream);

// LayerKVCache::gather_keys / gather_values: out[i] = cache[ids[i]] (reshape_and_cache.hip)
hipError_t launch_gather_blocks(const void *cache, const int32_t *ids, void *out, int64_t num_ids,
                                int64_t block_bytes, int64_t num_cache_blocks, hipStream_t stream);

// nvllm_debug_validate_tables (paged_attn.hip): counts[0] = ids outside the pool among the used
// blocks, counts[1] = context lengths outside their table row; counts is device memory (2 ints)
hipError_t launch_validate_tables(const int32_t *block_tables, const int32_t *context_lens, int num_seqs,
                                  int max_blocks_per_seq, int block_size, int64_t num_cache_blocks,
                                  int32_t *counts, hipStream_t stream);

// K5: qk_norm_rope_cache.hip
struct QkNormRopeCacheArgs {
    const void *q;
    const void *k;
    const void *v;
    void *q_out;
    void *k_out;  // optional
    const void *q_weight;
    const void *k_weight;
    float eps;
    const void *cos_cache;
    const void *sin_cache;
    int max_positions;
    const int32_t *positions;
    void *k_cache;
    void *v_cache;
    const int64_t *slot_mapping;
    int64_t num_tokens;
    int num_heads;
    int num_kv_heads;
    int head_dim;
    int block_size;
    int64_t num_cache_blocks;
};
hipError_t launch_qk_norm_rope_cache(DType dt, const QkNormRopeCacheArgs &a, hipStream_t stream);

// K2/K2b/K3: paged_attn.hip
struct PagedAttnArgs {
    const void *q;
    const void *k_cache;
    const void *v_cache;
    const int32_t *block_tables;
    const int32_t *context_lens;
    void *out;
    void *workspace;
    size_t workspace_bytes;
    int num_seqs;
    int num_new_tokens;
    int num_heads;
    int num_kv_heads;
    int head_dim;
    int block_size;
    int max_blocks_per_seq;
    int64_t num_cache_blocks;
    float scale;
};
// true when the bandwidth (MFMA) kernel covers this shape
bool paged_attn_fast_path(DType dt, int num_seqs, int num_new_tokens, int num_heads,
                          int num_kv_heads, int head_dim, int block_size, int max_blocks_per_seq);
// split-K factor chosen for a launch shape (also sizes the workspace)
int paged_attn_num_splits(int num_seqs, int num_kv_heads, int max_blocks_per_seq);
size_t paged_attn_workspace_bytes(int num_seqs, int num_new_tokens, int num_heads,
                                  int num_kv_heads, int head_dim, int block_size,
                                  int max_blocks_per_seq);
// which kernel launch_paged_attn runs for these arguments (pointers may be null in a pure query:
// a null pointer counts as aligned)
enum class PagedPath : int { Decode = 0, K1Paged = 1, Generic = 2, Mid = 3 };
PagedPath paged_attn_path(DType dt, const PagedAttnArgs &a);
// K3b: 49..128 query rows per kv head, two workgroups per CU (paged_attn_mid.hip)
bool paged_attn_mid_supported(DType dt, const PagedAttnArgs &a);
hipError_t launch_paged_attn_mid(const PagedAttnArgs &a, hipStream_t stream);
// what THIS launch needs (<= paged_attn_workspace_bytes of its shape: 0 when the generic kernel runs)
size_t paged_attn_launch_workspace_bytes(DType dt, const PagedAttnArgs &a);
hipError_t launch_paged_attn(DType dt, const PagedAttnArgs &a, hipStream_t stream);
// the same launch with the split-K scratch in a file-scope device array of paged_attn.hip
// (a.workspace is ignored); launches that need more than kNoAllocScratchBytes are refused
constexpr size_t kNoAllocScratchBytes = (size_t)32 << 20;
hipError_t launch_paged_attn_noalloc(DType dt, const PagedAttnArgs &a, hipStream_t stream);
hipError_t noalloc_prepare_device();  // looks the scratch address of the current device up once
// K1 over the paged cache (flash_attn_mfma.hip): chunked prefill / verify with more than 32 query
// rows per kv head; launch_paged_attn dispatches to it
bool flash_attn_paged_supported(DType dt, const PagedAttnArgs &a);
hipError_t launch_flash_attn_paged(const PagedAttnArgs &a, hipStream_t stream);
size_t flash_attn_paged_workspace_bytes(const PagedAttnArgs &a);  // key-split partials (0 = none)

// K1: flash_attn.hip
struct FlashAttnArgs {
    const void *q;
    const void *k;
    const void *v;
    void *o;
    float *lse;
    int batch;
    int seq_q;
    int seq_kv;
    int num_heads;
    int num_kv_heads;
    int head_dim;
    float softmax_scale;
    int causal;
};
bool flash_attn_fast_path(DType dt, const FlashAttnArgs &a);
hipError_t launch_flash_attn(DType dt, const FlashAttnArgs &a, hipStream_t stream);

}  // namespace kernels
}  // namespace nvllm
