// flash_attn.hip -- K1: flash-attention prefill for gfx950.
//
// Replaces flash_attention_cpu (reference src/attention/flash.rs:137-275) and
// the never-launched CUDA kernels flash_attention_forward_{f32,f16}
// (kernels/flash_attn_fwd.cu:128-466): exact softmax(Q K^T * scale + causal) V
// with online softmax, never materialising the S x S scores.  Layout is the
// Rust side's [B, S, H, D]; GQA is native (q head h reads kv head
// h / (Hq/Hkv), flash.rs:147-154); the causal mask is TOP-LEFT aligned (key j
// visible to query i iff j <= i, flash.rs:296-306).
//
// Generic path (this file, any head_dim <= 256, bf16 or f32): one wave per
// (batch, head, query row), f32 FMA.  Covers the reference's small test
// shapes (head_dim 16/32/64).  The MFMA path for head_dim 128 lives in
// flash_attn_mfma.hip.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>

#include "nvllm_kernels.hpp"

namespace nvllm {
namespace kernels {

hipError_t launch_flash_attn_mfma(const FlashAttnArgs &a, hipStream_t stream);
bool flash_attn_mfma_supported(const FlashAttnArgs &a);

namespace {

template <typename T>
__device__ __forceinline__ float to_f32(T x);
template <>
__device__ __forceinline__ float to_f32<float>(float x) { return x; }
template <>
__device__ __forceinline__ float to_f32<__bf16>(__bf16 x) { return static_cast<float>(x); }
template <typename T>
__device__ __forceinline__ T from_f32(float x);
template <>
__device__ __forceinline__ float from_f32<float>(float x) { return x; }
template <>
__device__ __forceinline__ __bf16 from_f32<__bf16>(float x) { return static_cast<__bf16>(x); }

template <typename ET>
__global__ __launch_bounds__(64) void flash_attn_generic_kernel(
    const ET *__restrict__ q, const ET *__restrict__ k, const ET *__restrict__ v,
    ET *__restrict__ o, float *__restrict__ lse, int Sq, int Skv, int Hq, int Hkv, int D,
    float scale, int causal) {
    __shared__ float q_s[256];
    __shared__ float p_s[64];
    const int lane = threadIdx.x;
    const int i = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
    const int kvh = h / (Hq / Hkv);
    const ET *qrow = q + (((int64_t)b * Sq + i) * Hq + h) * D;
    ET *orow = o + (((int64_t)b * Sq + i) * Hq + h) * D;
    for (int d = lane; d < D; d += 64) q_s[d] = to_f32<ET>(qrow[d]);
    __syncthreads();
    int limit = Skv - 1;
    if (causal && i < limit) limit = i;  // flash.rs:299-306
    float m = -INFINITY, l = 0.0f;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int base = 0; base <= limit; base += 64) {
        const int j = base + lane;
        float s = -INFINITY;
        if (j <= limit) {
            const ET *krow = k + (((int64_t)b * Skv + j) * Hkv + kvh) * D;
            float a = 0.0f;
            for (int d = 0; d < D; ++d) a += q_s[d] * to_f32<ET>(krow[d]);
            s = a * scale;
        }
        float mx = s;
        for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
        const float m_new = fmaxf(m, mx);
        const float alpha = expf(m - m_new);
        const float pj = (j <= limit) ? expf(s - m_new) : 0.0f;
        float ps = pj;
        for (int off = 32; off > 0; off >>= 1) ps += __shfl_xor(ps, off);
        l = l * alpha + ps;
        m = m_new;
        __syncthreads();
        p_s[lane] = pj;
        __syncthreads();
#pragma unroll
        for (int x = 0; x < 4; ++x) acc[x] *= alpha;
        const int n = (limit - base + 1) < 64 ? (limit - base + 1) : 64;
        for (int jj = 0; jj < n; ++jj) {
            const ET *vrow = v + (((int64_t)b * Skv + base + jj) * Hkv + kvh) * D;
            const float pw = p_s[jj];
#pragma unroll
            for (int x = 0; x < 4; ++x) {
                const int d = lane + 64 * x;
                if (d < D) acc[x] += pw * to_f32<ET>(vrow[d]);
            }
        }
    }
    const float inv = l > 0.0f ? 1.0f / l : 0.0f;
#pragma unroll
    for (int x = 0; x < 4; ++x) {
        const int d = lane + 64 * x;
        if (d < D) orow[d] = from_f32<ET>(acc[x] * inv);
    }
    if (lse && lane == 0) lse[((int64_t)b * Hq + h) * Sq + i] = m + logf(l);
}

}  // namespace

bool flash_attn_fast_path(DType dt, const FlashAttnArgs &a) {
    return dt == DType::BF16 && flash_attn_mfma_supported(a);
}

hipError_t launch_flash_attn(DType dt, const FlashAttnArgs &a, hipStream_t stream) {
    if (a.batch <= 0 || a.seq_q <= 0 || a.num_heads <= 0) return hipSuccess;
    if (flash_attn_fast_path(dt, a)) return launch_flash_attn_mfma(a, stream);
    const dim3 grid(a.seq_q, a.num_heads, a.batch), block(64);
    if (dt == DType::BF16)
        flash_attn_generic_kernel<__bf16><<<grid, block, 0, stream>>>(
            static_cast<const __bf16 *>(a.q), static_cast<const __bf16 *>(a.k),
            static_cast<const __bf16 *>(a.v), static_cast<__bf16 *>(a.o), a.lse, a.seq_q, a.seq_kv,
            a.num_heads, a.num_kv_heads, a.head_dim, a.softmax_scale, a.causal);
    else
        flash_attn_generic_kernel<float><<<grid, block, 0, stream>>>(
            static_cast<const float *>(a.q), static_cast<const float *>(a.k),
            static_cast<const float *>(a.v), static_cast<float *>(a.o), a.lse, a.seq_q, a.seq_kv,
            a.num_heads, a.num_kv_heads, a.head_dim, a.softmax_scale, a.causal);
    return hipGetLastError();
}

}  // namespace kernels
}  // namespace nvllm
