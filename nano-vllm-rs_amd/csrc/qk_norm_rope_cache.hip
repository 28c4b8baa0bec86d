// qk_norm_rope_cache.hip -- K5: the producer side of the paged cache (gfx950).
//
// Replaces steps 3-5 of Qwen3Attention::forward (reference src/model/attention.rs:229-254):
//   q = q_norm(q); k = k_norm(k)            per-head RMSNorm over head_dim (src/model/norm.rs:83-100)
//   (q, k) = rotary_emb.apply(q, k, pos)    x*cos + rotate_half(x)*sin     (src/model/rope.rs:107-152)
//   cache append of k and v                 there Tensor::cat on a contiguous cache; here the
//                                           slot-mapped write of reshape_and_cache.hip (K4)
// in ONE pass over the projections' output: Q is normalised and rotated (to q_out, which may be
// q itself), K is normalised, rotated and written straight into its cache slot (and optionally
// also to a contiguous k_out for a dense prefill), V is copied into its cache slot.
//
// Roofline: HBM-bound.  Algorithmic bytes per token =
//   (2*Hq + 4*Hkv) * D * elt   (Q read + written, K and V read, K and V cache rows written)
//   + 2*D*elt of cos/sin (shared by the token's heads) + 12 B (slot, position);
// Qwen3-4B shapes, bf16: 24.5 KiB per token.
//
// Mapping (fast path, bf16, D = 128): a "row" is one head of one token, 256 B.  8 lanes per
// row, 8 rows per wave; lane i holds elements [8i, 8i+8) and [64+8i, 64+8i+8) (two 16-byte
// accesses), i.e. both halves of its rotation pairs, so the rotation is lane-local and the only
// cross-lane step is the 8-lane sum of squares.  Q/K/V are read once: non-temporal loads.  All
// arithmetic in f32, one rounding to bf16 at the end.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "nvllm_kernels.hpp"

namespace nvllm {
namespace kernels {

namespace {

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

__device__ __forceinline__ float bf16_lo(unsigned w) { return __builtin_bit_cast(float, w << 16); }
__device__ __forceinline__ float bf16_hi(unsigned w) {
    return __builtin_bit_cast(float, w & 0xffff0000u);
}
__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
    bf16x2 r;
    r[0] = static_cast<__bf16>(lo);
    r[1] = static_cast<__bf16>(hi);
    return __builtin_bit_cast(unsigned, r);
}
constexpr int kRowsPerBlock = 32;  // 256 threads, 8 lanes per row

__device__ __forceinline__ void unpack8(u32x4 w, float *x) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        x[2 * i] = bf16_lo(w[i]);
        x[2 * i + 1] = bf16_hi(w[i]);
    }
}

// NT: outputs leave as non-temporal stores.  Used for prefill-sized launches, whose outputs are far
// larger than L2 (+8 % bandwidth at 32k tokens); a decode step's few rows keep ordinary stores so
// that the attention kernel right behind finds them in cache.
template <bool NT> __device__ __forceinline__ void st16(__bf16 *p, u32x4 x) {
    if constexpr (NT) __builtin_nontemporal_store(x, reinterpret_cast<u32x4 *>(p));
    else *reinterpret_cast<u32x4 *>(p) = x;
}

template <bool NT>
__global__ __launch_bounds__(256) void qk_norm_rope_cache_d128_kernel(
    // q / q_out may be the same buffer (in-place is the default use), and k / k_out too: no
    // __restrict__ on them; every row is read completely before anything of it is written
    const __bf16 *q, const __bf16 *k, const __bf16 *__restrict__ v,
    __bf16 *q_out, __bf16 *k_out, const __bf16 *__restrict__ q_weight,
    const __bf16 *__restrict__ k_weight, float eps, const __bf16 *__restrict__ cos_cache,
    const __bf16 *__restrict__ sin_cache, int max_pos, const int32_t *__restrict__ positions,
    __bf16 *__restrict__ k_cache, __bf16 *__restrict__ v_cache,
    const int64_t *__restrict__ slot_mapping, int64_t num_rows, int Hq, int Hkv,
    int64_t num_slots) {
    constexpr int D = 128;
    const int li = threadIdx.x & 7;
    const int64_t row = (int64_t)blockIdx.x * kRowsPerBlock + (threadIdx.x >> 3);
    if (row >= num_rows) return;  // whole 8-lane groups leave together
    const int R = Hq + 2 * Hkv;
    const int64_t tok = row / R;
    const int j = (int)(row - tok * R);
    const int off_lo = 8 * li, off_hi = 64 + 8 * li;

    if (j >= Hq + Hkv) {  // V head: copy into the cache slot
        const int hv = j - Hq - Hkv;
        const int64_t slot = slot_mapping[tok];
        if (slot < 0 || slot >= num_slots) return;
        const __bf16 *src = v + (tok * Hkv + hv) * D;
        __bf16 *dst = v_cache + (slot * Hkv + hv) * D;
        const u32x4 lo = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(src + off_lo));
        const u32x4 hi = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(src + off_hi));
        st16<NT>(dst + off_lo, lo);
        st16<NT>(dst + off_hi, hi);
        return;
    }
    const bool is_q = j < Hq;
    const int hk = j - Hq;
    const __bf16 *src = is_q ? q + (tok * Hq + j) * D : k + (tok * Hkv + hk) * D;
    const __bf16 *w = is_q ? q_weight : k_weight;
    int pos = positions[tok];
    pos = pos < 0 ? 0 : (pos >= max_pos ? max_pos - 1 : pos);  // never read outside the tables
    const u32x4 xlo = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(src + off_lo));
    const u32x4 xhi = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(src + off_hi));
    const u32x4 wlo = *reinterpret_cast<const u32x4 *>(w + off_lo);
    const u32x4 whi = *reinterpret_cast<const u32x4 *>(w + off_hi);
    const u32x4 clo = *reinterpret_cast<const u32x4 *>(cos_cache + (int64_t)pos * D + off_lo);
    const u32x4 chi = *reinterpret_cast<const u32x4 *>(cos_cache + (int64_t)pos * D + off_hi);
    const u32x4 slo = *reinterpret_cast<const u32x4 *>(sin_cache + (int64_t)pos * D + off_lo);
    const u32x4 shi = *reinterpret_cast<const u32x4 *>(sin_cache + (int64_t)pos * D + off_hi);
    int64_t slot = -1;
    if (!is_q) slot = slot_mapping[tok];

    float x1[8], x2[8], w1[8], w2[8], c1[8], c2[8], s1[8], s2[8];
    unpack8(xlo, x1);
    unpack8(xhi, x2);
    unpack8(wlo, w1);
    unpack8(whi, w2);
    unpack8(clo, c1);
    unpack8(chi, c2);
    unpack8(slo, s1);
    unpack8(shi, s2);
    float ss = 0.0f;
#pragma unroll
    for (int i = 0; i < 8; ++i) ss += x1[i] * x1[i] + x2[i] * x2[i];
#pragma unroll
    for (int m = 1; m < 8; m <<= 1) ss += __shfl_xor(ss, m);
    const float inv = 1.0f / sqrtf(ss * (1.0f / D) + eps);
    u32x4 olo, ohi;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float y1[2], y2[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int t = 2 * i + e;
            const float n1 = x1[t] * inv * w1[t], n2 = x2[t] * inv * w2[t];
            y1[e] = n1 * c1[t] - n2 * s1[t];  // x*cos + (-x2)*sin   (rope.rs:131-151)
            y2[e] = n2 * c2[t] + n1 * s2[t];  // x*cos + ( x1)*sin
        }
        olo[i] = pack_bf16x2(y1[0], y1[1]);
        ohi[i] = pack_bf16x2(y2[0], y2[1]);
    }
    if (is_q) {
        __bf16 *dst = q_out + (tok * Hq + j) * D;
        st16<NT>(dst + off_lo, olo);
        st16<NT>(dst + off_hi, ohi);
    } else {
        if (k_out) {
            __bf16 *dst = k_out + (tok * Hkv + hk) * D;
            st16<NT>(dst + off_lo, olo);
            st16<NT>(dst + off_hi, ohi);
        }
        if (slot >= 0 && slot < num_slots) {
            __bf16 *dst = k_cache + (slot * Hkv + hk) * D;
            st16<NT>(dst + off_lo, olo);
            st16<NT>(dst + off_hi, ohi);
        }
    }
}

// Any even head_dim, bf16 or f32: one wave per row, lane l takes the rotation pairs l, l+64, ...
// The f32 instantiation follows the reference's operation order (x / rms, then * weight) so that
// its own 1e-5 bound (tests/model_norm_test.rs:47-72) holds with room to spare.
template <typename ET> __device__ __forceinline__ float ld(const ET *p) { return static_cast<float>(*p); }
template <typename ET> __device__ __forceinline__ void st(ET *p, float x) { *p = static_cast<ET>(x); }

template <typename ET>
__global__ __launch_bounds__(256) void qk_norm_rope_cache_generic_kernel(
    const ET *q, const ET *k, const ET *__restrict__ v,  // q / q_out and k / k_out may alias
    ET *q_out, ET *k_out, const ET *__restrict__ q_weight,
    const ET *__restrict__ k_weight, float eps, const ET *__restrict__ cos_cache,
    const ET *__restrict__ sin_cache, int max_pos, const int32_t *__restrict__ positions,
    ET *__restrict__ k_cache, ET *__restrict__ v_cache, const int64_t *__restrict__ slot_mapping,
    int64_t num_rows, int Hq, int Hkv, int D, int64_t num_slots) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= num_rows) return;
    const int R = Hq + 2 * Hkv;
    const int64_t tok = row / R;
    const int j = (int)(row - tok * R);
    const int half = D / 2;
    if (j >= Hq + Hkv) {
        const int hv = j - Hq - Hkv;
        const int64_t slot = slot_mapping[tok];
        if (slot < 0 || slot >= num_slots) return;
        for (int i = lane; i < D; i += 64)
            v_cache[(slot * Hkv + hv) * D + i] = v[(tok * Hkv + hv) * D + i];
        return;
    }
    const bool is_q = j < Hq;
    const int hk = j - Hq;
    const ET *src = is_q ? q + (tok * Hq + j) * (int64_t)D : k + (tok * Hkv + hk) * (int64_t)D;
    const ET *w = is_q ? q_weight : k_weight;
    int pos = positions[tok];
    pos = pos < 0 ? 0 : (pos >= max_pos ? max_pos - 1 : pos);
    const ET *cr = cos_cache + (int64_t)pos * D, *sr = sin_cache + (int64_t)pos * D;
    float ss = 0.0f;
    for (int i = lane; i < D; i += 64) {
        const float x = ld(src + i);
        ss += x * x;
    }
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) ss += __shfl_xor(ss, m);
    const float rms = sqrtf(ss / (float)D + eps);
    const int64_t slot = is_q ? -1 : slot_mapping[tok];
    ET *d0 = is_q ? q_out + (tok * Hq + j) * (int64_t)D : nullptr;
    ET *d1 = (!is_q && k_out) ? k_out + (tok * Hkv + hk) * (int64_t)D : nullptr;
    ET *d2 = (!is_q && slot >= 0 && slot < num_slots) ? k_cache + (slot * Hkv + hk) * (int64_t)D
                                                      : nullptr;
    for (int i = lane; i < half; i += 64) {
        // both halves are read before either is written: q_out may be q
        const float n1 = (ld(src + i) / rms) * ld(w + i);
        const float n2 = (ld(src + i + half) / rms) * ld(w + i + half);
        const float y1 = n1 * ld(cr + i) + (-n2) * ld(sr + i);
        const float y2 = n2 * ld(cr + i + half) + n1 * ld(sr + i + half);
        if (d0) { st(d0 + i, y1); st(d0 + i + half, y2); }
        if (d1) { st(d1 + i, y1); st(d1 + i + half, y2); }
        if (d2) { st(d2 + i, y1); st(d2 + i + half, y2); }
    }
}

}  // namespace

hipError_t launch_qk_norm_rope_cache(DType dt, const QkNormRopeCacheArgs &a, hipStream_t stream) {
    if (a.num_tokens <= 0) return hipSuccess;
    const int64_t rows = a.num_tokens * (a.num_heads + 2 * a.num_kv_heads);
    const int64_t num_slots = a.num_cache_blocks * a.block_size;
    const auto al8 = [](const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; };
    if (dt == DType::BF16 && a.head_dim == 128 && al8(a.q) && al8(a.k) && al8(a.v) && al8(a.q_out) &&
        al8(a.k_out) && al8(a.q_weight) && al8(a.k_weight) && al8(a.cos_cache) &&
        al8(a.sin_cache) && al8(a.k_cache) && al8(a.v_cache)) {
        const dim3 grid((unsigned)((rows + kRowsPerBlock - 1) / kRowsPerBlock)), block(256);
        constexpr int64_t kStreamTokens = 2048;  // >= 48 MiB of outputs at Qwen3-4B shapes
        if (a.num_tokens >= kStreamTokens)
            qk_norm_rope_cache_d128_kernel<true><<<grid, block, 0, stream>>>(
                static_cast<const __bf16 *>(a.q), static_cast<const __bf16 *>(a.k),
                static_cast<const __bf16 *>(a.v), static_cast<__bf16 *>(a.q_out),
                static_cast<__bf16 *>(a.k_out), static_cast<const __bf16 *>(a.q_weight),
                static_cast<const __bf16 *>(a.k_weight), a.eps, static_cast<const __bf16 *>(a.cos_cache),
                static_cast<const __bf16 *>(a.sin_cache), a.max_positions, a.positions,
                static_cast<__bf16 *>(a.k_cache), static_cast<__bf16 *>(a.v_cache), a.slot_mapping, rows,
                a.num_heads, a.num_kv_heads, num_slots);
        else
            qk_norm_rope_cache_d128_kernel<false><<<grid, block, 0, stream>>>(
                static_cast<const __bf16 *>(a.q), static_cast<const __bf16 *>(a.k),
                static_cast<const __bf16 *>(a.v), static_cast<__bf16 *>(a.q_out),
                static_cast<__bf16 *>(a.k_out), static_cast<const __bf16 *>(a.q_weight),
                static_cast<const __bf16 *>(a.k_weight), a.eps, static_cast<const __bf16 *>(a.cos_cache),
                static_cast<const __bf16 *>(a.sin_cache), a.max_positions, a.positions,
                static_cast<__bf16 *>(a.k_cache), static_cast<__bf16 *>(a.v_cache), a.slot_mapping, rows,
                a.num_heads, a.num_kv_heads, num_slots);
        return hipGetLastError();
    }
    const dim3 grid((unsigned)((rows + 3) / 4)), block(256);
    if (dt == DType::BF16)
        qk_norm_rope_cache_generic_kernel<__bf16><<<grid, block, 0, stream>>>(
            static_cast<const __bf16 *>(a.q), static_cast<const __bf16 *>(a.k),
            static_cast<const __bf16 *>(a.v), static_cast<__bf16 *>(a.q_out),
            static_cast<__bf16 *>(a.k_out), static_cast<const __bf16 *>(a.q_weight),
            static_cast<const __bf16 *>(a.k_weight), a.eps, static_cast<const __bf16 *>(a.cos_cache),
            static_cast<const __bf16 *>(a.sin_cache), a.max_positions, a.positions,
            static_cast<__bf16 *>(a.k_cache), static_cast<__bf16 *>(a.v_cache), a.slot_mapping, rows,
            a.num_heads, a.num_kv_heads, a.head_dim, num_slots);
    else
        qk_norm_rope_cache_generic_kernel<float><<<grid, block, 0, stream>>>(
            static_cast<const float *>(a.q), static_cast<const float *>(a.k),
            static_cast<const float *>(a.v), static_cast<float *>(a.q_out),
            static_cast<float *>(a.k_out), static_cast<const float *>(a.q_weight),
            static_cast<const float *>(a.k_weight), a.eps, static_cast<const float *>(a.cos_cache),
            static_cast<const float *>(a.sin_cache), a.max_positions, a.positions,
            static_cast<float *>(a.k_cache), static_cast<float *>(a.v_cache), a.slot_mapping, rows,
            a.num_heads, a.num_kv_heads, a.head_dim, num_slots);
    return hipGetLastError();
}

}  // namespace kernels
}  // namespace nvllm
