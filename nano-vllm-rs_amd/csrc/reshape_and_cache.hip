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
// Mapping: one 64-lane wave per token, 16 B per lane per access, so one
// wave-instruction moves 1 KiB of a row (a 2 KiB bf16 Qwen3 row = 2 loads for
// K + 2 for V, all issued before the first store).  The slot is read once per
// wave (wave-uniform scalar load).  Tokens are grid-strided over 4-wave blocks.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "nvllm_kernels.hpp"

namespace nvllm {
namespace kernels {

namespace {

constexpr int kWavesPerBlock = 4;

template <int ITERS>  // ITERS = ceil(row_vec / 64) when <= 4, 0 = generic loop
__global__ __launch_bounds__(64 * kWavesPerBlock) void reshape_and_cache_kernel(
    const uint4 *__restrict__ key, const uint4 *__restrict__ value, uint4 *__restrict__ k_cache,
    uint4 *__restrict__ v_cache, const int64_t *__restrict__ slot_mapping, int64_t num_tokens,
    int row_vec, int64_t num_slots) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int64_t tok = (int64_t)blockIdx.x * kWavesPerBlock + wave; tok < num_tokens;
         tok += (int64_t)gridDim.x * kWavesPerBlock) {
        const int64_t slot = slot_mapping[tok];
        if (slot < 0 || slot >= num_slots) continue;  // skip marker / never write out of bounds
        const uint4 *ks = key + tok * row_vec;
        const uint4 *vs = value + tok * row_vec;
        uint4 *kd = k_cache + slot * row_vec;
        uint4 *vd = v_cache + slot * row_vec;
        if constexpr (ITERS > 0) {
            uint4 kr[ITERS], vr[ITERS];
#pragma unroll
            for (int i = 0; i < ITERS; ++i) {
                const int idx = lane + 64 * i;
                if (idx < row_vec) {
                    kr[i] = ks[idx];
                    vr[i] = vs[idx];
                }
            }
#pragma unroll
            for (int i = 0; i < ITERS; ++i) {
                const int idx = lane + 64 * i;
                if (idx < row_vec) {
                    kd[idx] = kr[i];
                    vd[idx] = vr[i];
                }
            }
        } else {
            for (int idx = lane; idx < row_vec; idx += 64) {
                kd[idx] = ks[idx];
                vd[idx] = vs[idx];
            }
        }
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

}  // namespace

hipError_t launch_reshape_and_cache(const void *key, const void *value, void *k_cache,
                                    void *v_cache, const int64_t *slot_mapping,
                                    int64_t num_tokens, int64_t row_bytes, int64_t num_slots,
                                    hipStream_t stream) {
    if (num_tokens <= 0) return hipSuccess;
    const int row_vec = static_cast<int>(row_bytes / 16);
    int64_t blocks = (num_tokens + kWavesPerBlock - 1) / kWavesPerBlock;
    if (blocks > 256 * 32) blocks = 256 * 32;  // grid-stride the rest
    const dim3 grid(static_cast<unsigned>(blocks)), block(64 * kWavesPerBlock);
    if (row_bytes % 16 != 0) {
        reshape_and_cache_small_kernel<<<grid, block, 0, stream>>>(
            static_cast<const unsigned short *>(key), static_cast<const unsigned short *>(value),
            static_cast<unsigned short *>(k_cache), static_cast<unsigned short *>(v_cache),
            slot_mapping, num_tokens, static_cast<int>(row_bytes / 2), num_slots);
        return hipGetLastError();
    }
    const uint4 *k = static_cast<const uint4 *>(key), *v = static_cast<const uint4 *>(value);
    uint4 *kc = static_cast<uint4 *>(k_cache), *vc = static_cast<uint4 *>(v_cache);
    const int iters = (row_vec + 63) / 64;
    switch (iters) {
        case 1:
            reshape_and_cache_kernel<1><<<grid, block, 0, stream>>>(k, v, kc, vc, slot_mapping,
                                                                    num_tokens, row_vec, num_slots);
            break;
        case 2:
            reshape_and_cache_kernel<2><<<grid, block, 0, stream>>>(k, v, kc, vc, slot_mapping,
                                                                    num_tokens, row_vec, num_slots);
            break;
        case 3:
        case 4:
            reshape_and_cache_kernel<4><<<grid, block, 0, stream>>>(k, v, kc, vc, slot_mapping,
                                                                    num_tokens, row_vec, num_slots);
            break;
        default:
            reshape_and_cache_kernel<0><<<grid, block, 0, stream>>>(k, v, kc, vc, slot_mapping,
                                                                    num_tokens, row_vec, num_slots);
            break;
    }
    return hipGetLastError();
}

}  // namespace kernels
}  // namespace nvllm
