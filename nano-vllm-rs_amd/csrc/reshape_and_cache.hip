// reshape_and_cache.hip -- K4: the KV write of the paged cache (gfx950).
//
// Replaces write_kv_to_cache + scatter_to_cache (reference
// src/attention/paged.rs:208-320): for each token p,
//   cache[slot / bs, slot % bs, :, :] = key[p, :, :]     (same for value)
// The reference rebuilds the whole cache tensor per token; here it is one
// in-place scatter.  Pure data movement, so the kernel works on 16-byte
// vectors and is element-type agnostic (bf16 and f32 entries share it).
//
// Roofline: HBM-bound.  Algorithmic bytes per token = 2 (K,V) * row_bytes
// read + the same written (+ 8 B of slot index); row_bytes = Hkv*D*elt.
//
// Mapping: flat -- thread g moves 16-byte vector g of the key rows and of the value rows (token
// g / row_vec, found with one division; the slot is a cached vector load shared by the row's
// lanes).  Against one wave per token this halves a 4096-token launch (19 -> 9 us) because every
// CU sees many short independent threads instead of a few waves that each wait on their slot.
// Launches whose K/V rows exceed what the Infinity Cache can still hold from their producer (>= 48 MiB)
// stream: non-temporal loads and stores (4.4 -> 5.1 TB/s at 32k tokens); smaller ones use ordinary
// accesses (a 4096-token launch: 8.0 -> 6.7 us), and a decode step's few rows stay in cache for the
// attention kernel.
//
// Where the remaining gap to the copy ceiling lies (scripts/k4_lab, profiles/r03_k4_lab.txt): with
// slots in order this kernel runs at the speed of a straight copy of the same bytes (6.0 TB/s, 0.75
// of 8 TB/s) and FETCH_SIZE / WRITE_SIZE equal the algorithmic bytes; whole blocks in ascending order
// give 5.8, whole blocks or single rows in random order 5.2-5.3 -- the same for every mapping tried
// (flat, one wave per 1 / 2 / 4 rows with scalar slots and all loads first, resident waves with the
// next row in flight), so the loss is the DRAM's for scattered 2-32 KiB writes, not the kernel's.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "nvllm_kernels.hpp"

namespace nvllm {
namespace kernels {

namespace {

constexpr int kWavesPerBlock = 4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

template <bool NT>
__global__ __launch_bounds__(256) void reshape_and_cache_kernel(
    const u32x4 *__restrict__ key, const u32x4 *__restrict__ value, u32x4 *__restrict__ k_cache,
    u32x4 *__restrict__ v_cache, const int64_t *__restrict__ slot_mapping, int64_t total_vec,
    int row_vec, int64_t num_slots) {
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (gid >= total_vec) return;
    // total_vec < 2^32 is every realistic launch; the 64-bit division is far slower
    const int64_t tok = total_vec <= 0xffffffffLL ? (int64_t)((unsigned)gid / (unsigned)row_vec)
                                                  : gid / row_vec;
    const int idx = (int)(gid - tok * row_vec);
    const u32x4 kx = __builtin_nontemporal_load(key + gid);  // read once
    const u32x4 vx = __builtin_nontemporal_load(value + gid);
    const int64_t slot = slot_mapping[tok];
    if (slot < 0 || slot >= num_slots) return;  // skip marker / never write out of bounds
    const int64_t dst = slot * row_vec + idx;
    if constexpr (NT) {
        __builtin_nontemporal_store(kx, k_cache + dst);
        __builtin_nontemporal_store(vx, v_cache + dst);
    } else {
        k_cache[dst] = kx;
        v_cache[dst] = vx;
    }
}

// Rows that are not a multiple of 16 bytes (tiny test shapes): 2-byte granularity.
__global__ __launch_bounds__(64 * kWavesPerBlock) void reshape_and_cache_small_kernel(
    const unsigned short *__restrict__ key, const unsigned short *__restrict__ value,
    unsigned short *__restrict__ k_cache, unsigned short *__restrict__ v_cache,
    const int64_t *__restrict__ slot_mapping, int64_t num_tokens, int row_halfs,
    int64_t num_slots) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int64_t tok = (int64_t)blockIdx.x * kWavesPerBlock + wave; tok < num_tokens;
         tok += (int64_t)gridDim.x * kWavesPerBlock) {
        const int64_t slot = slot_mapping[tok];
        if (slot < 0 || slot >= num_slots) continue;
        for (int idx = lane; idx < row_halfs; idx += 64) {
            k_cache[slot * row_halfs + idx] = key[tok * row_halfs + idx];
            v_cache[slot * row_halfs + idx] = value[tok * row_halfs + idx];
        }
    }
}

// LayerKVCache::gather_keys / gather_values (reference src/core/kv_cache.rs:123-144, an
// index_select over dim 0): out[i] = cache[ids[i]], whole blocks.  The attention kernels never need
// this copy (they read through the block table); it exists for callers that want the reference's
// contiguous view, e.g. to hand a sequence's K/V to another engine.  Flat 16-byte mapping like the
// scatter above; an id outside the pool yields a zero block.
__global__ __launch_bounds__(256) void gather_blocks_kernel(const u32x4 *__restrict__ cache,
                                                            const int32_t *__restrict__ ids,
                                                            u32x4 *__restrict__ out, int64_t total_vec,
                                                            int block_vec, int64_t num_cache_blocks) {
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (gid >= total_vec) return;
    const int64_t i = total_vec <= 0xffffffffLL ? (int64_t)((unsigned)gid / (unsigned)block_vec)
                                                : gid / block_vec;
    const int idx = (int)(gid - i * block_vec);
    const int64_t id = ids[i];
    u32x4 x = {0u, 0u, 0u, 0u};
    if (id >= 0 && id < num_cache_blocks) x = cache[id * block_vec + idx];
    __builtin_nontemporal_store(x, out + gid);
}

}  // namespace

hipError_t launch_gather_blocks(const void *cache, const int32_t *ids, void *out, int64_t num_ids,
                                int64_t block_bytes, int64_t num_cache_blocks, hipStream_t stream) {
    if (num_ids <= 0) return hipSuccess;
    if (block_bytes % 16 != 0 || block_bytes / 16 > 0x7fffffffLL) return hipErrorInvalidValue;
    const int block_vec = static_cast<int>(block_bytes / 16);
    const int64_t total = num_ids * block_vec;
    const int64_t blocks = (total + 255) / 256;
    if (blocks > 0x7fffffffLL) return hipErrorInvalidValue;
    gather_blocks_kernel<<<dim3(static_cast<unsigned>(blocks)), dim3(256), 0, stream>>>(
        static_cast<const u32x4 *>(cache), ids, static_cast<u32x4 *>(out), total, block_vec,
        num_cache_blocks);
    return hipGetLastError();
}

hipError_t launch_reshape_and_cache(const void *key, const void *value, void *k_cache,
                                    void *v_cache, const int64_t *slot_mapping,
                                    int64_t num_tokens, int64_t row_bytes, int64_t num_slots,
                                    hipStream_t stream) {
    if (num_tokens <= 0) return hipSuccess;
    if (row_bytes % 16 != 0) {
        int64_t blocks = (num_tokens + kWavesPerBlock - 1) / kWavesPerBlock;
        if (blocks > 256 * 32) blocks = 256 * 32;  // grid-stride the rest
        const dim3 grid(static_cast<unsigned>(blocks)), block(64 * kWavesPerBlock);
        reshape_and_cache_small_kernel<<<grid, block, 0, stream>>>(
            static_cast<const unsigned short *>(key), static_cast<const unsigned short *>(value),
            static_cast<unsigned short *>(k_cache), static_cast<unsigned short *>(v_cache),
            slot_mapping, num_tokens, static_cast<int>(row_bytes / 2), num_slots);
        return hipGetLastError();
    }
    const int row_vec = static_cast<int>(row_bytes / 16);
    const int64_t total = num_tokens * row_vec;
    const int64_t blocks = (total + 255) / 256;
    if (blocks > 0x7fffffffLL) return hipErrorInvalidValue;
    const dim3 grid(static_cast<unsigned>(blocks)), block(256);
    const u32x4 *k = static_cast<const u32x4 *>(key), *v = static_cast<const u32x4 *>(value);
    u32x4 *kc = static_cast<u32x4 *>(k_cache), *vc = static_cast<u32x4 *>(v_cache);
    constexpr int64_t kStreamBytes = 48ll << 20;  // K + V rows of the launch
    if (2 * num_tokens * row_bytes >= kStreamBytes)
        reshape_and_cache_kernel<true><<<grid, block, 0, stream>>>(k, v, kc, vc, slot_mapping, total,
                                                                   row_vec, num_slots);
    else
        reshape_and_cache_kernel<false><<<grid, block, 0, stream>>>(k, v, kc, vc, slot_mapping, total,
                                                                    row_vec, num_slots);
    return hipGetLastError();
}

}  // namespace kernels
}  // namespace nvllm
