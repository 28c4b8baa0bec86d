// flash_attn_mfma.hip -- K1 fast path: bf16 flash-attention prefill on the
// gfx950 matrix cores (mfma_f32_32x32x16_bf16), head_dim 128.
//
// Replaces flash_attention_cpu (reference src/attention/flash.rs:137-275) and
// the never-launched CUDA kernel flash_attention_forward_f16
// (kernels/flash_attn_fwd.cu:319-466; scalar FMAs, [B,H,S,D], no GQA).
// Same math: softmax(Q K^T * scale + causal) V with online softmax, top-left
// aligned causal mask (key j visible to query i iff j <= i, flash.rs:296-306),
// native [B,S,H,D] layout, GQA by kv head = q head / (Hq/Hkv).
//
// Roofline: MFMA.  Algorithmic FLOPs = 4 * B * Hq * Sq * Skv * D (x 1/2 causal).
//
// Structure (wave64, 4 waves per workgroup = 128 query rows of one q head)
//   * each wave owns 32 query rows; the query row lives on the LANE for the
//     whole kernel (lane & 31), so softmax state (m, l) and the O rescale are
//     lane-local:
//        S^T[key][row] = K[key][:] . Q[row][:]    A = K (LDS), B = Q^T (VGPRs)
//        O^T[d][row]  += V^T[d][key] . P^T[key][row]
//                                                 A = V^T (LDS, transposed read),
//                                                 B = P^T straight from the S^T
//                                                 accumulators (k order of the
//                                                 accumulator-as-operand trick)
//   * K/V tiles of 64 keys are shared by the 4 waves through LDS, double
//     buffered (2 x (16 KiB K + 16 KiB V) = 64 KiB, two workgroups per CU);
//     tile t+1 is fetched HBM -> VGPR with coalesced 16-byte loads while tile t
//     is computed and written to LDS after the compute (issue-early /
//     write-late), one barrier per tile.
//   * LDS images: K rows are read as MFMA A fragments with ds_read_b128, 16-byte
//     chunk index XOR (row & 15) (conflict-free); V rows are read transposed
//     with ds_read_b64_tr_b16, chunk index XOR 4*(row & 3) (conflict-free).
//   * causal: workgroups are launched heaviest first (descending query tile);
//     a wave skips tiles that lie entirely above its diagonal; only diagonal /
//     ragged tiles pay for masking.
//   * blockIdx.x enumerates (kv head fastest) so that the workgroups sharing a
//     kv head's K/V stream land on one XCD's L2 when Hkv is a multiple of 8.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>

#include "nvllm_kernels.hpp"

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
constexpr int kWaves = 4;
constexpr int kQRows = 32 * kWaves;   // query rows per workgroup
constexpr int kKV = 64;               // keys per tile
constexpr int kTileBytes = kKV * kRowBytes;  // 16 KiB
constexpr float kDeferLog2 = 6.0f;           // rescale only when a row max grows by > 2^6 (T13)

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
    bf16x2 r;
    r[0] = static_cast<__bf16>(lo);
    r[1] = static_cast<__bf16>(hi);
    return __builtin_bit_cast(unsigned, r);
}

template <bool CAUSAL>
__global__ __launch_bounds__(64 * kWaves, 2) void flash_attn_mfma_kernel(
    const __bf16 *__restrict__ q, const __bf16 *__restrict__ k, const __bf16 *__restrict__ v,
    __bf16 *__restrict__ o, float *__restrict__ lse, int Sq, int Skv, int Hq, int Hkv,
    int num_q_tiles, float scale_log2e) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[4 * kTileBytes];
    unsigned char *const k_lds = lds;                  // 2 buffers
    unsigned char *const v_lds = lds + 2 * kTileBytes;  // 2 buffers

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 31;   // query row within the wave / key row / d row
    const int h = lane >> 5;   // lane half

    // blockIdx.x = kvh + Hkv * (hg + G * qt_rev): kv head fastest, heavy q tiles first
    const int G = Hq / Hkv;
    int bid = blockIdx.x;
    const int kvh = bid % Hkv;
    bid /= Hkv;
    const int hg = bid % G;
    // Launch order of the query tiles: the heavy half in descending order, then the light half
    // in ASCENDING order, so that (under the observed round-robin placement, speed only) the
    // two workgroups that share a CU hold complementary amounts of causal work.
    int qt = bid / G;
    if (CAUSAL) {
        const int half = (num_q_tiles + 1) / 2;
        qt = qt < half ? num_q_tiles - 1 - qt : qt - half;
    }
    const int hq = kvh * G + hg;
    const int b = blockIdx.y;

    const int q0 = qt * kQRows;
    const int qw0 = q0 + 32 * wave;
    const int qrow = qw0 + c;

    int kv_end = Skv;
    if (CAUSAL && q0 + kQRows < kv_end) kv_end = q0 + kQRows;
    const int ntiles = (kv_end + kKV - 1) / kKV;

    // ---- Q^T fragments: lane (c,h) holds Q[qrow][16ks + 8h .. +7], ks = 0..7
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

    // ---- staging: thread handles chunks idx = i*256 + tid (i = 0..3) of a 64x16-chunk tile
    const int st_ch = tid & 15;
    const int st_row0 = tid >> 4;  // + 16*i
    const int64_t kv_row_stride = (int64_t)Hkv * kD;  // elements
    const __bf16 *k_base = k + ((int64_t)b * Skv * Hkv + kvh) * kD + st_ch * 8;
    const __bf16 *v_base = v + ((int64_t)b * Skv * Hkv + kvh) * kD + st_ch * 8;
    unsigned k_wr[4], v_wr[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = st_row0 + 16 * i;
        k_wr[i] = (unsigned)(r * kRowBytes + 16 * (st_ch ^ (r & 15)));
        v_wr[i] = (unsigned)(r * kRowBytes + 16 * (st_ch ^ (4 * (r & 3))));
    }
    // One staging register set, used twice per tile: K(t+1) is fetched at the top of tile t and
    // written to LDS right after the QK^T phase; V(t+1) is fetched then and written at the end.
    u32x4 st[4];
    auto stage_load = [&](const __bf16 *base, int t) {
        const int kv0 = t * kKV;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            // rows past the end are clamped to the last row: always a valid, finite load (no
            // per-load branch); those keys are masked out of S, so their P is exactly 0
            int r = kv0 + st_row0 + 16 * i;
            r = r < Skv ? r : Skv - 1;
            st[i] = *reinterpret_cast<const u32x4 *>(base + (int64_t)r * kv_row_stride);
        }
    };
    auto stage_write = [&](unsigned char *tile, const unsigned (&wr)[4]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<u32x4 *>(tile + wr[i]) = st[i];
    };

    // ---- LDS read addresses
    // K A-fragment: row = 32u + c, chunk (2ks + h) ^ (row & 15); (32u + c) & 15 == c & 15
    const unsigned k_rd_row = (unsigned)(c * kRowBytes);
    const int k_rd_sw = c & 15;
    // V^T A-fragment (transposed read): 16-lane group G16 = lane>>4 covers d chunk
    // 4dt + 2(G16&1) + (p>>1), key rows 16s + 4h + qq (+8 for the second read)
    const int c16 = lane & 15, qq = c16 >> 2, p = c16 & 3, g16 = lane >> 4;
    const int v_row_in = 4 * h + qq;                 // + 16s (+8)
    const int v_sw = 4 * (v_row_in & 3);             // (row & 3) is invariant under +16s, +8
    const int v_ch_lo = 2 * (g16 & 1) + (p >> 1);    // + 4dt
    const unsigned v_rd_base = (unsigned)(v_row_in * kRowBytes + 8 * (p & 1));

    float m_run = -INFINITY, l_run = 0.0f;
    f32x16 acc[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[dt][r] = 0.0f;

    stage_load(k_base, 0);
    stage_write(k_lds, k_wr);
    stage_load(v_base, 0);
    stage_write(v_lds, v_wr);
    __syncthreads();

    for (int t = 0; t < ntiles; ++t) {
        const int buf = t & 1;
        const int kv0 = t * kKV;
        const bool more = t + 1 < ntiles;
        if (more) stage_load(k_base, t + 1);

        const bool wave_active = !CAUSAL || (kv0 <= qw0 + 31);
        f32x16 s[2];
        if (wave_active) {
            // ---- S^T = K Q^T : two 32-key sub-tiles
#pragma unroll
            for (int u = 0; u < 2; ++u) {
#pragma unroll
                for (int r = 0; r < 16; ++r) s[u][r] = 0.0f;
                const unsigned char *kt = k_lds + buf * kTileBytes + u * 32 * kRowBytes + k_rd_row;
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) {
                    const u32x4 a =
                        *reinterpret_cast<const u32x4 *>(kt + 16 * ((2 * ks + h) ^ k_rd_sw));
                    s[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                        __builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, qf[ks]), s[u], 0,
                        0, 0);
                }
            }
        }
        if (more) {
            stage_write(k_lds + (buf ^ 1) * kTileBytes, k_wr);
            stage_load(v_base, t + 1);
        }
        if (wave_active) {
            // ---- mask: one wave-uniform branch, diagonal or ragged tiles only
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
            // ---- online softmax on raw scores; p = exp2(s*c - m*c), c = scale*log2(e)
            float mx = s[0][0];
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[u][r]);
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            // deferred rescale: keep the old reference max while the row max grew by less than
            // kDeferLog2 (in the exp2 domain), so P stays <= 2^kDeferLog2; otherwise rescale
            // O and l once for the whole wave (wave-uniform branch)
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
            u32x4 pb[4];  // P^T B fragments, k-step sp = 2u + sh
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
            // ---- O^T += V^T P^T : V fragments of d-tile dt+1 are read while dt is multiplied
            const unsigned char *vt = v_lds + buf * kTileBytes + v_rd_base;
            typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
            s16x4 vf[2][8];
            auto load_vf = [&](s16x4 (&f)[8], int dt) {
                const unsigned choff = 16u * (unsigned)((4 * dt + v_ch_lo) ^ v_sw);
#pragma unroll
                for (int sp = 0; sp < 4; ++sp) {
                    const unsigned char *p0 = vt + sp * 16 * kRowBytes + choff;
                    f[2 * sp] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)p0);
                    f[2 * sp + 1] =
                        __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)(p0 + 8 * kRowBytes));
                }
            };
            load_vf(vf[0], 0);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                if (dt + 1 < 4) load_vf(vf[(dt + 1) & 1], dt + 1);
#pragma unroll
                for (int sp = 0; sp < 4; ++sp) {
                    const s16x4 v0 = vf[dt & 1][2 * sp], v1 = vf[dt & 1][2 * sp + 1];
                    const s16x8 va = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
                    acc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                        __builtin_bit_cast(bf16x8, va), __builtin_bit_cast(bf16x8, pb[sp]), acc[dt],
                        0, 0, 0);
                }
            }
        }
        if (more) stage_write(v_lds + (buf ^ 1) * kTileBytes, v_wr);
        __syncthreads();
    }

    // ---- epilogue: O^T lane (c,h) holds O[qrow][32dt + 8i + 4h + (0..3)] in regs 4i..4i+3
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
    const auto al = [](const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; };
    return al(a.q) && al(a.k) && al(a.v) && al(a.o);
}

hipError_t launch_flash_attn_mfma(const FlashAttnArgs &a, hipStream_t stream) {
    const int num_q_tiles = (a.seq_q + kQRows - 1) / kQRows;
    const dim3 grid((unsigned)(num_q_tiles * a.num_heads), (unsigned)a.batch), block(64 * kWaves);
    const float scale_log2e = a.softmax_scale * 1.4426950408889634f;
    auto q = static_cast<const __bf16 *>(a.q);
    auto k = static_cast<const __bf16 *>(a.k);
    auto v = static_cast<const __bf16 *>(a.v);
    auto o = static_cast<__bf16 *>(a.o);
    if (a.causal)
        flash_attn_mfma_kernel<true><<<grid, block, 0, stream>>>(
            q, k, v, o, a.lse, a.seq_q, a.seq_kv, a.num_heads, a.num_kv_heads, num_q_tiles,
            scale_log2e);
    else
        flash_attn_mfma_kernel<false><<<grid, block, 0, stream>>>(
            q, k, v, o, a.lse, a.seq_q, a.seq_kv, a.num_heads, a.num_kv_heads, num_q_tiles,
            scale_log2e);
    return hipGetLastError();
}

}  // namespace kernels
}  // namespace nvllm
