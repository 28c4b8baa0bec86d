// v2: K1 with buffer-descriptor staging loads (no address VALU, free clamp), permlane32_swap for
// the row max, loop unrolled by two so every LDS address is base VGPR + immediate, explicit
// fragment prefetch.  NW = 4 or 8 waves per workgroup (128 / 256 query rows).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>

#include "nvllm_kernels.hpp"

#ifndef K1_NW
#define K1_NW 4
#endif
#ifndef K1_PDK
#define K1_PDK 4   // K fragment prefetch distance (MFMAs ahead)
#endif
#ifndef K1_PIN
#define K1_PIN 1
#endif
#ifndef K1_PDV
#define K1_PDV 2   // V fragment prefetch distance (MFMAs ahead)
#endif

namespace nvllm {
namespace kernels {

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;

constexpr int kD = 128;
constexpr int kRowBytes = 256;
constexpr int kKV = 64;                       // keys per tile
constexpr int kTileBytes = kKV * kRowBytes;   // 16 KiB
constexpr float kDeferLog2 = 6.0f;

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
    bf16x2 r;
    r[0] = static_cast<__bf16>(lo);
    r[1] = static_cast<__bf16>(hi);
    return __builtin_bit_cast(unsigned, r);
}

// max over the two lane halves (lane l <-> lane l^32) without LDS
__device__ __forceinline__ float max_across_halves(float x) {
    const unsigned u = __builtin_bit_cast(unsigned, x);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return fmaxf(__builtin_bit_cast(float, r[0]), __builtin_bit_cast(float, r[1]));
}

template <int NW, bool CAUSAL>
__global__ __launch_bounds__(64 * NW, 2) void flash_attn_mfma_kernel(
    const __bf16 *__restrict__ q, const __bf16 *__restrict__ k, const __bf16 *__restrict__ v,
    __bf16 *__restrict__ o, float *__restrict__ lse, int Sq, int Skv, int Hq, int Hkv,
    int num_q_tiles, float scale_log2e) {
    constexpr int kQRows = 32 * NW;
    constexpr int kCH = 1024 / (64 * NW);   // staging chunks per thread per tile
    constexpr int kRS = 64 / kCH;           // row step between a thread's chunks
    __shared__ __attribute__((aligned(16))) unsigned char lds[4 * kTileBytes];
    // K buffers at 0 / 16 KiB, V buffers at 32 / 48 KiB
    constexpr int kVOff = 2 * kTileBytes;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 31;
    const int h = lane >> 5;

    const int G = Hq / Hkv;
    int bid = blockIdx.x;
    const int kvh = bid % Hkv;
    bid /= Hkv;
    const int hg = bid % G;
    int qt = bid / G;
    if (CAUSAL) {
        if (NW == 4) {
            const int half = (num_q_tiles + 1) / 2;
            qt = qt < half ? num_q_tiles - 1 - qt : qt - half;
        } else {
            qt = num_q_tiles - 1 - qt;
        }
    }
    const int hq = kvh * G + hg;
    const int b = blockIdx.y;

    const int q0 = qt * kQRows;
    const int qw0 = q0 + 32 * wave;
    const int qrow = qw0 + c;

    int kv_end = Skv;
    if (CAUSAL && q0 + kQRows < kv_end) kv_end = q0 + kQRows;
    const int ntiles = (kv_end + kKV - 1) / kKV;

    // ---- Q^T fragments
    u32x4 qf[8];
    {
        const __bf16 *qp = q + (((int64_t)b * Sq + (qrow < Sq ? qrow : 0)) * Hq + hq) * kD + 8 * h;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            u32x4 x = {0u, 0u, 0u, 0u};
            if (qrow < Sq) x = *reinterpret_cast<const u32x4 *>(qp + 16 * ks);
            qf[ks] = x;
        }
    }

    // ---- staging through buffer descriptors: rows past Skv read as zero
    const int row_bytes = Hkv * kRowBytes;
    const __amdgpu_buffer_rsrc_t k_rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<__bf16 *>(k + (int64_t)b * Skv * Hkv * kD), 0, Skv * row_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t v_rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<__bf16 *>(v + (int64_t)b * Skv * Hkv * kD), 0, Skv * row_bytes, 0x00020000);
    const int st_ch = tid & 15;
    const int st_row0 = tid >> 4;
    const int st_voff = st_row0 * row_bytes + kvh * kRowBytes + st_ch * 16;
    const unsigned k_wr = (unsigned)(st_row0 * kRowBytes + 16 * (st_ch ^ (st_row0 & 15)));
    const unsigned v_wr = (unsigned)(st_row0 * kRowBytes + 16 * (st_ch ^ (4 * (st_row0 & 3))));
    u32x4 st[kCH];
    auto stage_load = [&](const __amdgpu_buffer_rsrc_t rs, int t) {
        const int soff = t * kKV * row_bytes;
#pragma unroll
        for (int i = 0; i < kCH; ++i)
            st[i] = __builtin_bit_cast(
                u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, st_voff + i * kRS * row_bytes, soff, 0));
    };
    auto stage_write = [&](unsigned base) {
#pragma unroll
        for (int i = 0; i < kCH; ++i)
            *reinterpret_cast<u32x4 *>(lds + base + i * kRS * kRowBytes) = st[i];
    };

    // ---- LDS read addresses (lane VGPR + immediates)
    unsigned k_rd[8];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) k_rd[ks] = (unsigned)(c * kRowBytes + 16 * ((2 * ks + h) ^ (c & 15)));
    const int c16 = lane & 15, qq = c16 >> 2, p = c16 & 3, g16 = lane >> 4;
    const int v_row_in = 4 * h + qq;
    const int v_sw = 4 * (v_row_in & 3);
    const int v_ch_lo = 2 * (g16 & 1) + (p >> 1);
    unsigned v_rd[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
        v_rd[dt] = (unsigned)(v_row_in * kRowBytes + 8 * (p & 1) + 16 * ((4 * dt + v_ch_lo) ^ v_sw));

    float m_run = -INFINITY, l_run = 0.0f;
    f32x16 acc[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[dt][r] = 0.0f;

    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

    // one 64-key tile; BUF is the LDS buffer index (compile-time)
    auto tile = [&](int t, auto buf_tag) {
        constexpr int BUF = decltype(buf_tag)::value;
        const int kv0 = t * kKV;
        const bool more = t + 1 < ntiles;
        if (more) stage_load(k_rs, t + 1);
        const bool wave_active = (!CAUSAL || (kv0 <= qw0 + 31)) && qw0 < Sq;
        f32x16 s[2];
        if (wave_active) {
            const unsigned char *kt = lds + BUF * kTileBytes;
            u32x4 kf[16];
            auto rd = [&](int i) {
                kf[i] = *reinterpret_cast<const u32x4 *>(kt + k_rd[i & 7] + (i >> 3) * 32 * kRowBytes);
            };
#pragma unroll
            for (int i = 0; i < K1_PDK; ++i) rd(i);
#pragma unroll
            for (int r = 0; r < 16; ++r) s[0][r] = s[1][r] = 0.0f;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if (i + K1_PDK < 16) rd(i + K1_PDK);
                s[i >> 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, kf[i]),
                                                                   __builtin_bit_cast(bf16x8, qf[i & 7]),
                                                                   s[i >> 3], 0, 0, 0);
            }
#if K1_PIN
#pragma unroll
            for (int i = 0; i < K1_PDK; ++i) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                if (i + K1_PDK < 16) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
#endif
        }
        if (more) {
            stage_write(BUF == 0 ? kTileBytes + k_wr : k_wr);
            stage_load(v_rs, t + 1);
        }
        if (wave_active) {
            const bool need_mask = (kv0 + kKV > Skv) || (CAUSAL && (kv0 + kKV - 1 > qw0));
            if (need_mask) {
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int key = kv0 + 32 * u + (r & 3) + 8 * (r >> 2) + 4 * h;
                        const bool vis = (key < Skv) && (!CAUSAL || key <= qrow);
                        s[u][r] = vis ? s[u][r] : -INFINITY;
                    }
            }
            float mx = s[0][0];
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[u][r]);
            mx = max_across_halves(mx);
            if (__any((mx - m_run) * scale_log2e > kDeferLog2)) {
                const float m_new = fmaxf(m_run, mx);
                const float m_safe = (m_new == -INFINITY) ? 0.0f : m_new;
                const float alpha = fast_exp2((m_run - m_safe) * scale_log2e);
                l_run *= alpha;
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) acc[dt] *= alpha;
                m_run = m_new;
            }
            const float mc = ((m_run == -INFINITY) ? 0.0f : m_run) * scale_log2e;
            float psum = 0.0f;
            u32x4 pb[4];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                float pe[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    pe[r] = fast_exp2(__builtin_fmaf(s[u][r], scale_log2e, -mc));
                    psum += pe[r];
                }
#pragma unroll
                for (int sh = 0; sh < 2; ++sh)
                    pb[2 * u + sh] = u32x4{pack_bf16x2(pe[8 * sh + 0], pe[8 * sh + 1]),
                                           pack_bf16x2(pe[8 * sh + 2], pe[8 * sh + 3]),
                                           pack_bf16x2(pe[8 * sh + 4], pe[8 * sh + 5]),
                                           pack_bf16x2(pe[8 * sh + 6], pe[8 * sh + 7])};
            }
            l_run += psum;
            // ---- O^T += V^T P^T, MFMA i = 4 dt + sp, V fragments K1_PDV MFMAs ahead
            const unsigned char *vt = lds + kVOff + BUF * kTileBytes;
            s16x4 vf[16][2];
            auto rdv = [&](int i) {
                const unsigned char *p0 = vt + v_rd[i >> 2] + (i & 3) * 16 * kRowBytes;
                vf[i][0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)p0);
                vf[i][1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)(p0 + 8 * kRowBytes));
            };
#pragma unroll
            for (int i = 0; i < K1_PDV; ++i) rdv(i);
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if (i + K1_PDV < 16) rdv(i + K1_PDV);
                const s16x4 v0 = vf[i][0], v1 = vf[i][1];
                const s16x8 va = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
                acc[i >> 2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                    __builtin_bit_cast(bf16x8, va), __builtin_bit_cast(bf16x8, pb[i & 3]), acc[i >> 2], 0,
                    0, 0);
            }
#if K1_PIN
#pragma unroll
            for (int i = 0; i < K1_PDV; ++i) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                if (i + K1_PDV < 16) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            }
#endif
        }
        if (more) stage_write(kVOff + (BUF == 0 ? kTileBytes + v_wr : v_wr));
        __syncthreads();
    };

    stage_load(k_rs, 0);
    stage_write(k_wr);
    stage_load(v_rs, 0);
    stage_write(kVOff + v_wr);
    __syncthreads();

    for (int t = 0; t < ntiles; t += 2) {
        tile(t, std::integral_constant<int, 0>{});
        if (t + 1 < ntiles) tile(t + 1, std::integral_constant<int, 1>{});
    }

    // ---- epilogue
    float l = l_run + __shfl_xor(l_run, 32);
    if (qrow < Sq) {
        const float inv = l > 0.0f ? 1.0f / l : 0.0f;
        __bf16 *op = o + (((int64_t)b * Sq + qrow) * Hq + hq) * kD + 4 * h;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                u32x2 w;
                w[0] = pack_bf16x2(acc[dt][4 * i] * inv, acc[dt][4 * i + 1] * inv);
                w[1] = pack_bf16x2(acc[dt][4 * i + 2] * inv, acc[dt][4 * i + 3] * inv);
                *reinterpret_cast<u32x2 *>(op + 32 * dt + 8 * i) = w;
            }
        if (lse && h == 0)
            lse[((int64_t)b * Hq + hq) * Sq + qrow] =
                m_run * scale_log2e * 0.6931471805599453f + logf(l);
    }
}

}  // namespace

bool flash_attn_mfma_supported(const FlashAttnArgs &a) {
    if (a.head_dim != kD) return false;
    if (a.num_kv_heads <= 0 || a.num_heads % a.num_kv_heads != 0) return false;
    if (a.batch > 65535) return false;
    if ((int64_t)a.seq_kv * a.num_kv_heads * kRowBytes >= (int64_t)1 << 31) return false;
    const auto al = [](const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; };
    return al(a.q) && al(a.k) && al(a.v) && al(a.o);
}

hipError_t launch_flash_attn_mfma(const FlashAttnArgs &a, hipStream_t stream) {
    constexpr int NW = K1_NW;
    constexpr int kQRows = 32 * NW;
    const int num_q_tiles = (a.seq_q + kQRows - 1) / kQRows;
    const dim3 grid((unsigned)(num_q_tiles * a.num_heads), (unsigned)a.batch), block(64 * NW);
    const float scale_log2e = a.softmax_scale * 1.4426950408889634f;
    auto q = static_cast<const __bf16 *>(a.q);
    auto k = static_cast<const __bf16 *>(a.k);
    auto v = static_cast<const __bf16 *>(a.v);
    auto o = static_cast<__bf16 *>(a.o);
    if (a.causal)
        flash_attn_mfma_kernel<NW, true><<<grid, block, 0, stream>>>(
            q, k, v, o, a.lse, a.seq_q, a.seq_kv, a.num_heads, a.num_kv_heads, num_q_tiles,
            scale_log2e);
    else
        flash_attn_mfma_kernel<NW, false><<<grid, block, 0, stream>>>(
            q, k, v, o, a.lse, a.seq_q, a.seq_kv, a.num_heads, a.num_kv_heads, num_q_tiles,
            scale_log2e);
    return hipGetLastError();
}

}  // namespace kernels
}  // namespace nvllm
