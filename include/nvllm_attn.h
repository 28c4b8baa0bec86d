/*
 * nvllm_attn.h -- C ABI of the MI355X (gfx950) attention hot path that drops
 * in behind nano-vllm-rs's `attention` module.
 *
 * This is the boundary a Rust `extern "C"` block (compiled by build.rs with
 * hipcc) binds.  Plain pointers and sizes only; no C++ or
 * torch types.  Every entry cites the reference interface it replaces
 * (paths under the reference checkout).
 *
 * Conventions
 *   - All tensor pointers are DEVICE pointers owned by the caller, laid out
 *     exactly as the reference holds them (row-major, contiguous):
 *        q            [batch, seq_q,  num_heads,    head_dim]
 *        k, v         [batch, seq_kv, num_kv_heads, head_dim]
 *        k/v cache    [num_blocks, block_size, num_kv_heads, head_dim]
 *                     (src/core/kv_cache.rs:75-95)
 *     GQA is native: q head h reads kv head h / (num_heads / num_kv_heads)
 *     (src/attention/flash.rs:147-154, src/attention/paged.rs:326-338).
 *   - Launches are asynchronous on `stream` (NULL = the default stream) and no
 *     entry point allocates device memory.  No entry point synchronises, WITH
 *     THREE EXCEPTIONS, the ones that take a HOST array the way the reference's
 *     API does and must copy it to the device out of pageable memory:
 *     nvllm_batch_upload, nvllm_paged_attention_seq and nvllm_write_kv_to_cache
 *     each call hipStreamSynchronize(stream) once after their upload.  These
 *     three therefore cannot run on a capturing stream; everything else is
 *     HIP-graph capturable (the batched kernels take device arrays only).
 *     nvllm_synchronize is the explicit device-wide wait.
 *   - Return value: NVLLM_OK or a negative nvllm_status.  The message of the
 *     last failure on the calling thread is nvllm_get_last_error().
 *   - Host-logic entries (block tables, allocator, hashing) are pure CPU code
 *     and bit-exact with the reference's integer arithmetic.
 */
#ifndef NVLLM_ATTN_H
#define NVLLM_ATTN_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Opaque HIP stream (identical to hipStream_t; declared here so that binding
 * generators do not need the HIP headers). */
typedef struct ihipStream_t *nvllm_stream_t;

/* Mirrors crate::error::Error (src/error.rs:10-60) for the variants reachable
 * on this path, plus launch-level failures. */
typedef enum nvllm_status {
    NVLLM_OK = 0,
    NVLLM_ERR_CONFIG = -1,            /* Error::Config: shape/len mismatch, empty table        */
    NVLLM_ERR_UNSUPPORTED = -2,       /* head_dim / block_size / T outside the built kernels   */
    NVLLM_ERR_NULL = -3,              /* null pointer argument                                 */
    NVLLM_ERR_HIP = -4,               /* HIP runtime / launch failure                          */
    NVLLM_ERR_OUT_OF_BLOCKS = -5,     /* Error::OutOfBlocks                                    */
    NVLLM_ERR_BLOCK_INDEX_OOB = -6,   /* Error::BlockIndexOutOfBounds                          */
    NVLLM_ERR_TENSOR = -7             /* Error::Tensor: candle narrow/index_select failure     */
} nvllm_status;

/* ======================================================================== */
/* Device kernels                                                           */
/* ======================================================================== */

/* Flash-attention prefill, bf16 in / bf16 out, f32 accumulate on MFMA.
 * Replaces: flash_attention_forward_f16_launch (kernels/flash_attn_fwd.cu:515-534)
 * and the CPU dispatcher flash_attention / flash_attention_cpu
 * (src/attention/flash.rs:85-106, :137-275).
 * Differences from the CUDA launcher, on purpose: layout is the Rust side's
 * [B,S,H,D] (no transposes), GQA is native (num_kv_heads argument), and a
 * status is returned.  Causal mask is TOP-LEFT aligned like flash.rs:296-306:
 * key j is visible to query i iff j <= i.  `lse` may be NULL; otherwise
 * [batch, num_heads, seq_q] f32 receives m + log(l) (flash_attn_fwd.cu:305-309).
 * head_dim: 128 runs the MFMA kernel; any other head_dim <= 256 that is a
 * multiple of 8 runs the generic kernel. */
int nvllm_flash_attn_fwd_bf16(const void *q, const void *k, const void *v, void *o, float *lse,
                              int batch, int seq_q, int seq_kv, int num_heads, int num_kv_heads,
                              int head_dim, float softmax_scale, int causal,
                              nvllm_stream_t stream);

/* Same contract, f32 in / f32 out (generic kernel, f32 FMA).
 * Replaces: flash_attention_forward_f32_launch (kernels/flash_attn_fwd.cu:489-509). */
int nvllm_flash_attn_fwd_f32(const void *q, const void *k, const void *v, void *o, float *lse,
                             int batch, int seq_q, int seq_kv, int num_heads, int num_kv_heads,
                             int head_dim, float softmax_scale, int causal,
                             nvllm_stream_t stream);

/* Paged attention over the block-table KV cache for a BATCH of sequences.
 * Replaces: paged_attention (src/attention/paged.rs:117-192), which handles one
 * sequence per call; here sequence s uses row s of `block_tables`.
 *   q             [num_seqs, num_new_tokens, num_heads, head_dim]
 *   block_tables  int32 [num_seqs, max_blocks_per_seq]  physical block ids
 *                 (BlockTable::get_physical_block_ids, src/core/block.rs:251-253)
 *   context_lens  int32 [num_seqs]  cached + new tokens (paged.rs:107); the new
 *                 tokens' K/V must already be in the cache
 *   out           [num_seqs, num_new_tokens, num_heads * head_dim] (paged.rs:191)
 * When num_new_tokens > 1 the BOTTOM-RIGHT causal mask of paged.rs:357-381
 * applies: query i sees key positions <= context_len - num_new_tokens + i.
 * A sequence with context_len <= 0 produces zeros.  Slots past context_len in
 * the last block are never read into the result (they may hold anything).
 * `workspace` is split-K scratch of at least nvllm_paged_attn_workspace_bytes()
 * bytes (may be NULL when that returns 0).
 * bf16, head_dim 128, block_size 16: num_new_tokens * (num_heads/num_kv_heads)
 * <= 16 always runs the bandwidth kernel (decode, short verify), up to 48 when
 * the launch is large or the contexts long; more query rows
 * (chunked prefill, long verify) run the prefill kernel reading K/V through the
 * block table (any cache size: a block is addressed through its own 64-bit
 * base); other shapes run the generic kernel.  nvllm_paged_attn_path() says
 * which of the three a launch shape takes. */
int nvllm_paged_attn_bf16(const void *q, const void *k_cache, const void *v_cache,
                          const int32_t *block_tables, const int32_t *context_lens, void *out,
                          void *workspace, size_t workspace_bytes, int num_seqs,
                          int num_new_tokens, int num_heads, int num_kv_heads, int head_dim,
                          int block_size, int max_blocks_per_seq, int64_t num_cache_blocks,
                          float scale, nvllm_stream_t stream);

int nvllm_paged_attn_f32(const void *q, const void *k_cache, const void *v_cache,
                         const int32_t *block_tables, const int32_t *context_lens, void *out,
                         void *workspace, size_t workspace_bytes, int num_seqs, int num_new_tokens,
                         int num_heads, int num_kv_heads, int head_dim, int block_size,
                         int max_blocks_per_seq, int64_t num_cache_blocks, float scale,
                         nvllm_stream_t stream);

/* nvllm_paged_attn_bf16 for callers that CANNOT allocate device memory (code under HIP graph
 * capture, tight serving loops, harnesses that forbid hipMalloc).  No counterpart in the
 * reference.  Same arguments, same kernels and bit-identical results, minus the workspace: the
 * split-K partials go to a scratch array built into the library's code object
 * (nvllm_paged_attn_noalloc_max_scratch_bytes() = 32 MiB, held by every GPU the library runs on
 * until the process ends).  A launch shape whose nvllm_paged_attn_workspace_bytes() exceeds that
 * is refused with NVLLM_ERR_UNSUPPORTED (nothing is launched); shapes that need no workspace
 * never touch the scratch.  The call allocates and frees nothing and does not synchronise.  All
 * calls on one GPU share that GPU's array: two calls that overlap in time on one GPU (different
 * streams, or threads) race on it -- serialise them or use the workspace form.  The array's
 * address on a device is looked up once per device: by nvllm_set_device(), else by the first call
 * on that device -- which is then not a pure stream operation, so under stream capture call
 * nvllm_set_device() (or make one call) on the device beforehand. */
int nvllm_paged_attn_bf16_noalloc(const void *q, const void *k_cache, const void *v_cache,
                                  const int32_t *block_tables, const int32_t *context_lens, void *out,
                                  int num_seqs, int num_new_tokens, int num_heads, int num_kv_heads,
                                  int head_dim, int block_size, int max_blocks_per_seq,
                                  int64_t num_cache_blocks, float scale, nvllm_stream_t stream);
size_t nvllm_paged_attn_noalloc_max_scratch_bytes(void);

/* Bytes of split-K scratch nvllm_paged_attn_* needs for this launch shape
 * (an upper bound over all context lengths <= max_blocks_per_seq*block_size and
 * over what the shape does not say: a launch that ends in the generic kernel --
 * misaligned tensors, NVLLM_PAGED_FLASH=0 -- needs none and accepts none).
 * The launch takes its split decision from the same arguments by the same
 * arithmetic; a workspace smaller than what the launch needs (or NULL when that
 * is > 0, or not 16-byte aligned) is rejected with NVLLM_ERR_CONFIG -- never a
 * silently different launch.  The split factors follow the compute-unit count of
 * the CURRENT device: make the query with the launch device current. */
size_t nvllm_paged_attn_workspace_bytes(int num_seqs, int num_new_tokens, int num_heads,
                                        int num_kv_heads, int head_dim, int block_size,
                                        int max_blocks_per_seq);

/* Which kernel nvllm_paged_attn_* runs for a launch shape (a pure query: nothing is launched, no
 * pointer is looked at -- 16-byte aligned tensors are assumed).  No counterpart in the reference
 * (paged.rs:117-192 has one path); it exists so that callers and tests can see that a verify or
 * prefill chunk (src/speculative/engine.rs:146-169) is NOT on the scalar generic kernel.
 *   0 = decode kernel (bandwidth, MFMA 16x16x32, split-K)      csrc/paged_attn.hip
 *   1 = prefill kernel over the cache (MFMA 32x32x16, K1 paged) csrc/flash_attn_mfma.hip
 *   2 = generic kernel (any head_dim / block_size / f32)
 *   3 = mid kernel: 49..128 query rows per kv head (two workgroups per CU): up to 64 rows from
 *       1.5 (sequence, kv head) units per CU, up to 96 rows from 3, up to 128 rows from 4 units per
 *       CU when no context exceeds 2048 tokens                      csrc/paged_attn_mid.hip
 * dtype 0 bf16, 1 f32.  Negative nvllm_status on invalid dimensions. */
int nvllm_paged_attn_path(int num_seqs, int num_new_tokens, int num_heads, int num_kv_heads,
                          int head_dim, int block_size, int max_blocks_per_seq,
                          int64_t num_cache_blocks, int dtype);

/* KV write ("reshape_and_cache"): for each token p,
 *   cache[slot/bs, slot%bs, :, :] = key[p, :, :]   (and value),  slot = slot_mapping[p].
 * Replaces: write_kv_to_cache + scatter_to_cache (src/attention/paged.rs:208-320),
 * IN PLACE (the reference returns new tensors; documented deviation).
 *   key, value    [num_tokens, num_kv_heads, head_dim]
 *   slot_mapping  int64 [num_tokens], from BlockTable::get_slot_mapping
 *                 (src/core/block.rs:269-283)
 * A negative slot skips that token; a slot >= num_cache_blocks*block_size is
 * skipped too (never written out of bounds).  Duplicate non-negative slots in
 * one call race; nvllm_slot_mapping_last_writer() reproduces the reference's
 * sequential last-writer-wins by turning superseded entries into -1. */
int nvllm_reshape_and_cache_bf16(const void *key, const void *value, void *k_cache, void *v_cache,
                                 const int64_t *slot_mapping, int64_t num_tokens,
                                 int num_kv_heads, int head_dim, int block_size,
                                 int64_t num_cache_blocks, nvllm_stream_t stream);

int nvllm_reshape_and_cache_f32(const void *key, const void *value, void *k_cache, void *v_cache,
                                const int64_t *slot_mapping, int64_t num_tokens, int num_kv_heads,
                                int head_dim, int block_size, int64_t num_cache_blocks,
                                nvllm_stream_t stream);

/* LayerKVCache::gather_keys / gather_values (src/core/kv_cache.rs:123-144, an
 * index_select over dim 0): out[i, :, :, :] = cache[block_ids[i], :, :, :].
 *   cache      [num_cache_blocks, block_size, num_kv_heads, head_dim]
 *   block_ids  int32 [num_ids] ON DEVICE (the reference casts usize -> u32)
 *   out        [num_ids, block_size, num_kv_heads, head_dim]
 * num_ids == 0 -> NVLLM_ERR_CONFIG "block_ids cannot be empty" (kv_cache.rs:124).
 * An id outside the pool cannot be reported without a sync (the reference's
 * index_select fails): that block of `out` is zero-filled.  The attention
 * kernels never call this -- they read through the block table; it is the
 * reference's contiguous view for callers that want one.  dtype 0 bf16, 1 f32. */
int nvllm_gather_cache_blocks(const void *cache, const int32_t *block_ids, int64_t num_ids, void *out,
                              int64_t num_cache_blocks, int block_size, int num_kv_heads,
                              int head_dim, int dtype, nvllm_stream_t stream);

/* Producer side of the paged cache in one pass: per-head RMSNorm of Q and K, RoPE, then K (rotated)
 * and V go into their cache slots -- no intermediate tensors.
 * Replaces: steps 3-5 of Qwen3Attention::forward (src/model/attention.rs:229-254):
 *   apply_head_norm = RmsNorm::forward over head_dim (src/model/norm.rs:83-100):
 *       x / sqrt(mean(x^2) + eps) * weight
 *   RotaryEmbedding::apply (src/model/rope.rs:107-152):
 *       x * cos[pos] + cat(-x[D/2:], x[:D/2]) * sin[pos]
 *   the cache append (there Tensor::cat; here the slot write of nvllm_reshape_and_cache_*).
 *   q [num_tokens, num_heads, head_dim] -> q_out (same shape; may be q itself)
 *   k, v [num_tokens, num_kv_heads, head_dim]; k_out optional (NULL, or [num_tokens,
 *       num_kv_heads, head_dim]: the rotated K, contiguous, for a dense prefill)
 *   q_weight, k_weight [head_dim]  (q_norm / k_norm weights, attention.rs:165-166)
 *   cos_cache, sin_cache [max_positions, head_dim] as RotaryEmbedding holds them
 *       (rope.rs:20-23; nvllm_rope_cache_f32 builds them)
 *   positions int32 [num_tokens]: each token's position in its sequence (start_pos + i);
 *       the reference fails on a position past the tables (narrow, rope.rs:111) -- a device
 *       array cannot be validated without a sync, so the kernel clamps into [0, max_positions)
 *   slot_mapping as nvllm_reshape_and_cache_* (negative / out-of-pool slot: the cache write of
 *       that token is skipped, q_out and k_out are still produced)
 * All tensors in the entry's element type; arithmetic in f32, one rounding at the end. */
int nvllm_qk_norm_rope_cache_bf16(const void *q, const void *k, const void *v, void *q_out,
                                  void *k_out, const void *q_weight, const void *k_weight,
                                  float eps, const void *cos_cache, const void *sin_cache,
                                  int64_t max_positions, const int32_t *positions, void *k_cache,
                                  void *v_cache, const int64_t *slot_mapping, int64_t num_tokens,
                                  int num_heads, int num_kv_heads, int head_dim, int block_size,
                                  int64_t num_cache_blocks, nvllm_stream_t stream);

int nvllm_qk_norm_rope_cache_f32(const void *q, const void *k, const void *v, void *q_out,
                                 void *k_out, const void *q_weight, const void *k_weight, float eps,
                                 const void *cos_cache, const void *sin_cache,
                                 int64_t max_positions, const int32_t *positions, void *k_cache,
                                 void *v_cache, const int64_t *slot_mapping, int64_t num_tokens,
                                 int num_heads, int num_kv_heads, int head_dim, int block_size,
                                 int64_t num_cache_blocks, nvllm_stream_t stream);

/* Replaces: flash_attention_get_last_error (kernels/flash_attn_fwd.cu:540-543).
 * Thread-local; never NULL. */
const char *nvllm_get_last_error(void);

/* Replaces: flash_attention_synchronize (kernels/flash_attn_fwd.cu:548-551): 0 / -1. */
int nvllm_synchronize(void);

/* KVCacheConfig::layer_cache_size_bytes / total_cache_size_bytes
 * (src/core/kv_cache.rs:62-70): bytes of ONE layer's K (or V) cache
 * [num_blocks, block_size, num_kv_heads, head_dim], and of all layers' K and V.
 * dtype: 0 = bf16, 1 = f32 (the reference's default, kv_cache.rs:51).  -1 on a
 * negative dimension or unknown dtype. */
int64_t nvllm_kv_cache_layer_bytes(int64_t num_blocks, int64_t block_size, int64_t num_kv_heads,
                                   int64_t head_dim, int dtype);
int64_t nvllm_kv_cache_total_bytes(int64_t num_blocks, int64_t block_size, int64_t num_kv_heads,
                                   int64_t head_dim, int64_t num_layers, int dtype);

/* Test / tuning hook, no counterpart in the reference: pins one launch decision
 * that is otherwise taken from the launch shape.  `name` is one of
 * NVLLM_PAGED_MAX_ROWS, NVLLM_PAGED_SPLITS, NVLLM_PAGED_WAVE_TARGET,
 * NVLLM_PAGED_NT, NVLLM_PAGED_HPB, NVLLM_PAGED_FLASH, NVLLM_K1_PAIRING (0
 * balanced / 1 adjacent), NVLLM_K1_PACK, NVLLM_K1_KSPLIT, NVLLM_K1_CHAIN, NVLLM_PAGED_MID (0 never / 1 whenever the
 * shape allows / -1 by launch size / 2 as 1 with its rows by wave index, a test hook).  Environment variables
 * of the same names seed the values once, at first use; no launch reads the
 * environment.  Process-wide: nvllm_paged_attn_workspace_bytes and the launch
 * both see the value in force when they are called.  _reset restores the
 * environment's values. */
int nvllm_debug_set_policy(const char *name, int value);
void nvllm_debug_reset_policy(void);

/* Debug check of a launch's tables, no counterpart in the reference (whose
 * index_select fails on a bad id, src/attention/paged.rs:134-146): the kernels
 * here never fault on a wrong table -- an id outside the pool reads as zeros or
 * is clamped, a context length is clamped into its table row -- so a wrong
 * table gives plausible wrong output.  This pass counts what a launch over the
 * same tables would have clamped, into `counts` (DEVICE memory, 2 ints, zeroed
 * by the call): counts[0] = ids < 0 or >= num_cache_blocks among the
 * ceil(context_len / block_size) blocks each sequence uses, counts[1] =
 * context lengths < 0 or > max_blocks_per_seq * block_size.  Asynchronous on
 * `stream`; not meant for the serving loop. */
int nvllm_debug_validate_tables(const int32_t *block_tables, const int32_t *context_lens, int num_seqs,
                                int max_blocks_per_seq, int block_size, int64_t num_cache_blocks,
                                int32_t *counts, nvllm_stream_t stream);

/* Device inventory used by the sharded launcher and the bench.  nvllm_set_device also resolves
 * the no-allocation entry's scratch address on that device. */
int nvllm_device_count(void);
int nvllm_set_device(int device);

/* ======================================================================== */
/* Host logic: block table, slot mapping, hashing, allocator (bit-exact)     */
/* ======================================================================== */

/* compute_num_blocks (src/core/block.rs:311-313): ceil(seq_len / block_size). */
int64_t nvllm_compute_num_blocks(int64_t seq_len, int64_t block_size);

/* hash_token_block (src/core/block.rs:43-57): SipHash-1-3 (keys 0,0) over
 * [parent_hash as u64 LE, if has_parent] ++ tokens as u32 LE. */
uint64_t nvllm_hash_token_block(const uint32_t *token_ids, int64_t num_tokens, int has_parent,
                                uint64_t parent_hash);

/* RotaryEmbedding::new (src/model/rope.rs:42-80), host side: cos / sin tables
 * [max_seq_len, dim] in f32 (inv_freq = 1.0f32 / (theta^(2i/dim) in f64 -> f32), as rope.rs:55
 * parses; angle = pos * inv_freq in f32, the row is
 * the half-row twice).  Odd dim -> NVLLM_ERR_CONFIG (the reference panics). */
int nvllm_rope_cache_f32(int64_t dim, int64_t max_seq_len, double theta, float *cos_out,
                         float *sin_out);

/* Host-side helper for nvllm_reshape_and_cache_*: copies `slots` to `out`,
 * replacing every entry that a LATER entry overwrites by -1, so a parallel
 * scatter reproduces the sequential loop of paged.rs:231-262.  Returns the
 * number of superseded entries. */
int64_t nvllm_slot_mapping_last_writer(const int64_t *slots, int64_t n, int64_t *out);

/* --- BlockTable (src/core/block.rs:183-289) ------------------------------ */
typedef struct nvllm_block_table nvllm_block_table;
nvllm_block_table *nvllm_block_table_new(int64_t block_size);                /* :192 */
void nvllm_block_table_free(nvllm_block_table *t);
int64_t nvllm_block_table_block_size(const nvllm_block_table *t);            /* :205 */
void nvllm_block_table_append_block(nvllm_block_table *t, int64_t block_id); /* :236 */
int64_t nvllm_block_table_num_blocks(const nvllm_block_table *t);            /* :241 */
int nvllm_block_table_is_empty(const nvllm_block_table *t);                  /* :246 */
void nvllm_block_table_clear(nvllm_block_table *t);                          /* :286 */
/* :223-231; NVLLM_ERR_BLOCK_INDEX_OOB when logical_idx is not allocated. */
int nvllm_block_table_get_block_id(const nvllm_block_table *t, int64_t logical_idx,
                                   int64_t *block_id_out);
/* :251; copies up to `cap` ids, returns the table length. */
int64_t nvllm_block_table_get_physical_block_ids(const nvllm_block_table *t, int64_t *ids_out,
                                                 int64_t cap);
/* :269-283; writes at most `cap` slots, returns how many the table yields
 * (positions past the last allocated block are dropped, like the reference). */
int64_t nvllm_block_table_get_slot_mapping(const nvllm_block_table *t, int64_t seq_len,
                                           int64_t *slots_out, int64_t cap);

/* --- BlockManager (src/core/block_manager.rs:40-383) --------------------- */
typedef struct nvllm_block_manager nvllm_block_manager;
nvllm_block_manager *nvllm_block_manager_new(int64_t num_blocks, int64_t block_size); /* :69  */
void nvllm_block_manager_free_handle(nvllm_block_manager *m);
int64_t nvllm_block_manager_block_size(const nvllm_block_manager *m);                 /* :88  */
int64_t nvllm_block_manager_num_blocks(const nvllm_block_manager *m);                 /* :93  */
int64_t nvllm_block_manager_num_free_blocks(const nvllm_block_manager *m);            /* :98  */
int64_t nvllm_block_manager_num_used_blocks(const nvllm_block_manager *m);            /* :103 */
int nvllm_block_manager_has_free_blocks(const nvllm_block_manager *m);                /* :108 */
int nvllm_block_manager_can_allocate(const nvllm_block_manager *m, int64_t n);        /* :113 */
/* :140; NVLLM_ERR_OUT_OF_BLOCKS when the free list is empty. */
int nvllm_block_manager_allocate(nvllm_block_manager *m, int64_t *block_id_out);
/* :162; all-or-nothing. */
int nvllm_block_manager_allocate_many(nvllm_block_manager *m, int64_t n, int64_t *ids_out);
/* :187; 1 if the block went back to the free list, 0 otherwise. */
int nvllm_block_manager_free(nvllm_block_manager *m, int64_t block_id);
/* :215; number of blocks actually freed. */
int64_t nvllm_block_manager_free_many(nvllm_block_manager *m, const int64_t *ids, int64_t n);
/* get_block().ref_count() (:220, block.rs:118); -1 if the block is not in use. */
int64_t nvllm_block_manager_ref_count(const nvllm_block_manager *m, int64_t block_id);
/* get_block().prefix_hash() (block.rs:123); returns 1 and writes *hash_out if Some. */
int nvllm_block_manager_prefix_hash(const nvllm_block_manager *m, int64_t block_id,
                                    uint64_t *hash_out);
/* :236 / :251; new ref count, or -1 for None. */
int64_t nvllm_block_manager_increment_ref(nvllm_block_manager *m, int64_t block_id);
int64_t nvllm_block_manager_decrement_ref(nvllm_block_manager *m, int64_t block_id);
/* :292; block id (ref count incremented) or -1 for None. */
int64_t nvllm_block_manager_get_cached_block(nvllm_block_manager *m, uint64_t prefix_hash);
void nvllm_block_manager_cache_block(nvllm_block_manager *m, int64_t block_id,
                                     uint64_t prefix_hash);                           /* :312 */
/* :337; *was_cached_out = 1 on a prefix hit. */
int nvllm_block_manager_allocate_with_prefix(nvllm_block_manager *m, uint64_t prefix_hash,
                                             int64_t *block_id_out, int *was_cached_out);
int nvllm_block_manager_is_prefix_cached(const nvllm_block_manager *m, uint64_t prefix_hash); /* :350 */
int64_t nvllm_block_manager_num_cached_prefixes(const nvllm_block_manager *m);               /* :355 */
void nvllm_block_manager_clear_prefix_cache(nvllm_block_manager *m);                         /* :362 */
void nvllm_block_manager_reset(nvllm_block_manager *m);                                      /* :377 */

/* ======================================================================== */
/* Reference-shaped operators (one sequence per call, host BlockTable)       */
/* ======================================================================== */

/* --- batched step metadata -------------------------------------------------
 * Replaces, for the batched kernels, the per-sequence hand-off of the reference's scheduler
 * (src/scheduler/batch.rs:227-375 -> one BlockTable per paged_attention call,
 * src/attention/paged.rs:117-192): the scheduled sequences of one step are flattened into
 * block_tables int32 [num_seqs, max_blocks_per_seq], context_lens int32 [num_seqs] (context
 * including the new tokens), slot_mapping int64 [num_tokens] (block.rs:269-283 arithmetic for
 * the new tokens, in order) and query_start int32 [num_seqs + 1].  Host only; _upload copies the
 * four arrays to caller-allocated device buffers on `stream`. */
typedef struct nvllm_batch nvllm_batch;
nvllm_batch *nvllm_batch_new(int64_t max_blocks_per_seq);
void nvllm_batch_free(nvllm_batch *b);
void nvllm_batch_clear(nvllm_batch *b);
/* NVLLM_ERR_CONFIG: num_new_tokens > context_len, context_len beyond the table's blocks
 * (paged.rs:150 narrow), table longer than max_blocks_per_seq */
int nvllm_batch_add_sequence(nvllm_batch *b, const nvllm_block_table *t, int64_t context_len,
                             int64_t num_new_tokens);
int64_t nvllm_batch_num_seqs(const nvllm_batch *b);
int64_t nvllm_batch_num_tokens(const nvllm_batch *b);
int64_t nvllm_batch_max_blocks_per_seq(const nvllm_batch *b);
const int32_t *nvllm_batch_block_tables(const nvllm_batch *b);
const int32_t *nvllm_batch_context_lens(const nvllm_batch *b);
const int64_t *nvllm_batch_slot_mapping(const nvllm_batch *b);
const int32_t *nvllm_batch_query_start(const nvllm_batch *b);
/* int32 [num_tokens]: each new token's position in its sequence (the RoPE row) */
const int32_t *nvllm_batch_positions(const nvllm_batch *b);
/* any destination may be NULL (skipped); the copies are issued on `stream` and complete before
 * the call returns (the host arrays are pageable) */
int nvllm_batch_upload(const nvllm_batch *b, void *d_block_tables, void *d_context_lens,
                       void *d_slot_mapping, void *d_query_start, void *d_positions,
                       nvllm_stream_t stream);

/* --- Scheduler + Sequence (src/scheduler/batch.rs:60-512, src/core/sequence.rs:10-327) ------
 * The caller of the attention path: decides, step by step, which sequences prefill (and how
 * many tokens) and which decode, and owns their block tables.  Decisions are the reference's,
 * step for step, with its defaults (src/config.rs:59-69).  Two extra switches, both off by
 * default, repair what SURVEY 8f-1 lists: reschedule_partial_prefill (a chunk-prefilled sequence
 * gets its next chunk; the reference never schedules it again) and enable_prefix_caching
 * (admission reuses cached full prompt blocks through hash_token_block + the BlockManager prefix
 * cache; src/config.rs:22 declares the flag, nothing reads it there). */
typedef struct nvllm_scheduler nvllm_scheduler;
typedef struct nvllm_scheduler_config {      /* SchedulerConfig, src/config.rs:44-57 */
    int64_t max_num_seqs;
    int64_t max_prefill_tokens;
    int enable_chunked_prefill;
    int64_t chunk_size;
    int enable_priority;
    int enable_preemption;
    int reschedule_partial_prefill;          /* extension, default 0 */
    int enable_prefix_caching;               /* extension, default 0 */
} nvllm_scheduler_config;
void nvllm_scheduler_config_default(nvllm_scheduler_config *c);               /* config.rs:59-69 */
nvllm_scheduler *nvllm_scheduler_new(const nvllm_scheduler_config *c, int64_t block_size,
                                     int64_t num_blocks);                      /* batch.rs:166 */
void nvllm_scheduler_free(nvllm_scheduler *s);
/* Sequence::with_priority (sequence.rs:131) + Scheduler::add_sequence (batch.rs:181) */
int nvllm_scheduler_add_sequence(nvllm_scheduler *s, uint64_t seq_id, const uint32_t *prompt,
                                 int64_t num_tokens, int32_t priority);
int nvllm_scheduler_remove_sequence(nvllm_scheduler *s, uint64_t seq_id);      /* :198; 1 / 0 */
/* Scheduler::schedule (batch.rs:227); the SchedulerOutputs stay readable until the next call */
int nvllm_scheduler_schedule(nvllm_scheduler *s);
int64_t nvllm_scheduler_out_num_prefill(const nvllm_scheduler *s);
int64_t nvllm_scheduler_out_num_decode(const nvllm_scheduler *s);
int64_t nvllm_scheduler_out_num_preempted(const nvllm_scheduler *s);
const uint64_t *nvllm_scheduler_out_prefill_ids(const nvllm_scheduler *s);
const uint64_t *nvllm_scheduler_out_decode_ids(const nvllm_scheduler *s);
const uint64_t *nvllm_scheduler_out_preempted_ids(const nvllm_scheduler *s);
/* prefill_chunk_sizes (batch.rs:66-68) in the order of the prefill ids */
const int64_t *nvllm_scheduler_out_prefill_chunk_sizes(const nvllm_scheduler *s);
int64_t nvllm_scheduler_out_num_prefill_tokens(const nvllm_scheduler *s);
int64_t nvllm_scheduler_out_num_decode_tokens(const nvllm_scheduler *s);
/* reason: 0 EndOfSequence, 1 MaxTokens, 2 StopSequence, 3 Aborted (sequence.rs:51-60) */
void nvllm_scheduler_finish_sequence(nvllm_scheduler *s, uint64_t seq_id, int reason); /* :428 */
int nvllm_scheduler_append_token(nvllm_scheduler *s, uint64_t seq_id, uint32_t token); /* :442 */
int nvllm_scheduler_mark_prefilled(nvllm_scheduler *s, uint64_t seq_id, int64_t num_tokens); /* :452 */
int64_t nvllm_scheduler_num_waiting(const nvllm_scheduler *s);                 /* :462 */
int64_t nvllm_scheduler_num_running(const nvllm_scheduler *s);
int64_t nvllm_scheduler_num_swapped(const nvllm_scheduler *s);
int nvllm_scheduler_has_unfinished_sequences(const nvllm_scheduler *s);        /* :482 */
int nvllm_scheduler_has_pending_requests(const nvllm_scheduler *s);            /* :489 */
/* copies up to cap ids, returns how many are running (:477) */
int64_t nvllm_scheduler_running_sequence_ids(const nvllm_scheduler *s, uint64_t *out, int64_t cap);
void nvllm_scheduler_reset(nvllm_scheduler *s);                                /* :504 */
/* Sequence queries (get_sequence, :215): status 0 Waiting, 1 Running, 2 Swapped, 3 Finished
 * (sequence.rs:16-25); -1 = no such sequence */
int nvllm_scheduler_seq_status(const nvllm_scheduler *s, uint64_t seq_id);
int nvllm_scheduler_seq_finish_reason(const nvllm_scheduler *s, uint64_t seq_id);
/* get_sequence_mut + set_waiting / set_running / set_swapped (sequence.rs:254-301); a transition
 * the reference rejects (Error::InvalidStateTransition) -> NVLLM_ERR_CONFIG */
int nvllm_scheduler_seq_set_status(nvllm_scheduler *s, uint64_t seq_id, int status);
/* what: 0 prompt_len, 1 output_len, 2 total_len, 3 num_prefilled_tokens, 4 num_tokens_to_prefill,
 * 5 is_prefill_complete, 6 block_table().num_blocks(), 7 priority, 8 last_token_id (-1 = none) */
int64_t nvllm_scheduler_seq_info(const nvllm_scheduler *s, uint64_t seq_id, int what);
int64_t nvllm_scheduler_seq_block_ids(const nvllm_scheduler *s, uint64_t seq_id, int64_t *out,
                                      int64_t cap);
int64_t nvllm_scheduler_seq_output_tokens(const nvllm_scheduler *s, uint64_t seq_id, uint32_t *out,
                                          int64_t cap);
/* block_manager() view (:494) */
int64_t nvllm_scheduler_num_free_blocks(const nvllm_scheduler *s);
int64_t nvllm_scheduler_num_used_blocks(const nvllm_scheduler *s);
int64_t nvllm_scheduler_num_cached_prefixes(const nvllm_scheduler *s);
int64_t nvllm_scheduler_block_ref_count(const nvllm_scheduler *s, int64_t block_id);
/* Last schedule() result -> batched-kernel metadata (appends to `batch`):
 * decode: every decode sequence with one new token at position total_len - 1;
 * prefill: the prefill sequences whose chunk is exactly num_new_tokens long (the kernels take one
 * T per launch), their ids to ids_out; returns how many were added, or a negative status. */
int nvllm_scheduler_fill_decode_batch(const nvllm_scheduler *s, nvllm_batch *batch);
int64_t nvllm_scheduler_fill_prefill_batch(const nvllm_scheduler *s, int64_t num_new_tokens,
                                           nvllm_batch *batch, uint64_t *ids_out, int64_t cap);

/* paged_attention exactly as src/attention/paged.rs:117-126 takes it: one
 * sequence, a host BlockTable, the layer's cache.  Validates like the
 * reference (empty table -> NVLLM_ERR_CONFIG, paged.rs:133-137; batch != 1 or
 * context_len > blocks*block_size or an id >= num_cache_blocks ->
 * NVLLM_ERR_TENSOR), stages the table into `scratch` (device, at least
 * nvllm_paged_attention_seq_scratch_bytes() bytes) and launches
 * nvllm_paged_attn_*.  dtype: 0 = bf16, 1 = f32. */
size_t nvllm_paged_attention_seq_scratch_bytes(int64_t num_table_blocks, int num_new_tokens,
                                               int num_heads, int num_kv_heads, int head_dim,
                                               int block_size);
int nvllm_paged_attention_seq(const void *query, int batch, int num_new_tokens,
                              const void *k_cache, const void *v_cache, int64_t num_cache_blocks,
                              const nvllm_block_table *block_table, int64_t context_len,
                              int num_heads, int num_kv_heads, int head_dim, double scale,
                              int dtype, void *out, void *scratch, size_t scratch_bytes,
                              nvllm_stream_t stream);

/* write_kv_to_cache exactly as src/attention/paged.rs:208-215 takes it: host
 * slot mapping (`&[usize]`), batch-1 key/value [1, seq_len, Hkv, D].
 * slot_mapping length != seq_len -> NVLLM_ERR_CONFIG (paged.rs:218-224); a
 * slot outside the cache -> NVLLM_ERR_TENSOR.  Duplicates resolve last-writer
 * -wins.  `scratch` (device) needs seq_len * 8 bytes.  Updates the caches in
 * place. */
int nvllm_write_kv_to_cache(const void *key, const void *value, int64_t seq_len, void *k_cache,
                            void *v_cache, int64_t num_cache_blocks, const int64_t *slot_mapping,
                            int64_t slot_mapping_len, int block_size, int num_kv_heads,
                            int head_dim, int dtype, void *scratch, size_t scratch_bytes,
                            nvllm_stream_t stream);

/* prefill_attention (src/attention/paged.rs:55-63): SDPA over contiguous
 * K/V, causal mask only when causal && seq_len > 1 (:79), output flattened to
 * [B, S, Hq*D] (:94).  Runs the flash kernel (same math, no S x S scores). */
int nvllm_prefill_attention(const void *q, const void *k, const void *v, void *out, int batch,
                            int seq_len, int num_heads, int num_kv_heads, int head_dim,
                            double scale, int causal, int dtype, nvllm_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* NVLLM_ATTN_H */
