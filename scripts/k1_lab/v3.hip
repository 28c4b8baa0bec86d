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
// Structure (wave64; ONE wave per SIMD with the whole 512-register file)
//   * workgroup = 4 waves = TWO 128-row query tiles of one q head: tile J and
//     tile NT-1-J.  Under the causal mask their key ranges are complementary,
//     so every workgroup carries the same number of MFMAs whatever J is: at
//     B=1, S=4096, 16 heads that is 256 equal workgroups, one per CU.
//   * a wave owns two 32-row query blocks, A in the long tile and B in the
//     short one.  The query row lives on the LANE (lane & 31) for the whole
//     kernel, so the online-softmax state and the O rescale are lane-local:
//        S^T[key][row] = K[key][:] . Q[row][:]    A = K (LDS), B = Q^T (VGPRs)
//        O^T[d][row]  += V^T[d][key] . P^T[key][row]
//                                                 A = V^T (LDS, transposed read),
//                                                 B = P^T straight from the S^T
//                                                 accumulators
//   * software pipeline over "items" (block, key tile): while the matrix
//     cores run QK^T of item k+1 and P.V of item k-1, the vector ALU runs the
//     softmax of item k.  While both blocks have keys left the items alternate
//     A,B on the same key tile; after that block A alternates over two score
//     register sets.  Only a block's LAST key tile needs a mask (diagonal or
//     ragged tail); it runs un-pipelined.
//   * K/V tiles of 64 keys are shared by the 4 waves through LDS rings (K x2
//     with rows padded to 272 B, V x3 XOR-swizzled for ds_read_b64_tr_b16),
//     fetched HBM -> VGPR a whole step ahead and written to LDS at the start
//     of the next step; one barrier per key tile.
//   * blockIdx.x enumerates kv head fastest so that the workgroups sharing a
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
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

constexpr int kD = 128;
constexpr int kWaves = 4;
constexpr int kQTile = 32 * kWaves;  // query rows per tile (two tiles per workgroup)
constexpr int kKV = 64;              // keys per tile
constexpr int kVRow = 256, kVTile = kKV * kVRow;  // 16 KiB, swizzled
constexpr int kKRow = 272, kKTile = kKV * kKRow;  // 17 KiB, padded rows (conflict-free b128 reads)
constexpr int kVRing = 3, kKRing = 2;
constexpr int kKBase = kVRing * kVTile;
constexpr int kLdsBytes = kKBase + kKRing * kKTile;  // 83,968 B
constexpr float kDeferLog2 = 6.0f;  // rescale only when a row max grows by > 2^6 (T13)

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
    bf16x2 r;
    r[0] = static_cast<__bf16>(lo);
    r[1] = static_cast<__bf16>(hi);
    return __builtin_bit_cast(unsigned, r);
}

// value of the other lane half (lane ^ 32) combined with this lane's
__device__ __forceinline__ float half_max(float x) {
    const auto r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, x),
                                                    __builtin_bit_cast(unsigned, x), false, false);
    return fmaxf(__builtin_bit_cast(float, r[0]), __builtin_bit_cast(float, r[1]));
}
__device__ __forceinline__ float half_sum(float x) {
    const auto r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, x),
                                                    __builtin_bit_cast(unsigned, x), false, false);
    return __builtin_bit_cast(float, r[0]) + __builtin_bit_cast(float, r[1]);
}

// One score register set: raw S^T of a 64-key tile for one 32-row block, the P^T fragments made
// from it and the softmax constants of that item.
struct Item {
    f32x16 sa, sb;  // keys 0..31 / 32..63 of the tile
    u32x4 p[4];     // P^T B fragments, k-step 2u + sh
    float mc;       // reference max * scale * log2(e)
    float alpha;    // pending O rescale factor
    float psum;     // partial row sum
    bool need;      // wave-uniform: alpha != 1
};

// Per-block state: O^T accumulators, running max / sum, Q^T fragments.
struct Block {
    f32x16 acc[4];
    u32x4 q[8];
    float m, l;
    int qrow;  // this lane's query row
};

// instruction-mix hints for the scheduler (LLVM SchedGroupMask)
#define K1_SGB_MFMA 0x008
#define K1_SGB_VALU 0x002
#define K1_SGB_TRANS 0x400
#define K1_SGB_DSREAD 0x100
#ifndef K1_NO_HINTS
#define K1_MIX_QK()                                                   \
    _Pragma("unroll") for (int i_ = 0; i_ < 16; ++i_) {               \
        __builtin_amdgcn_sched_group_barrier(K1_SGB_MFMA, 1, 0);      \
        __builtin_amdgcn_sched_group_barrier(K1_SGB_DSREAD, 1, 0);    \
        __builtin_amdgcn_sched_group_barrier(K1_SGB_VALU, 3, 0);      \
        __builtin_amdgcn_sched_group_barrier(K1_SGB_TRANS, 1, 0);     \
    }
#define K1_MIX_PV()                                                   \
    _Pragma("unroll") for (int i_ = 0; i_ < 16; ++i_) {               \
        __builtin_amdgcn_sched_group_barrier(K1_SGB_MFMA, 1, 0);      \
        __builtin_amdgcn_sched_group_barrier(K1_SGB_DSREAD, 2, 0);    \
        __builtin_amdgcn_sched_group_barrier(K1_SGB_VALU, 4, 0);      \
        __builtin_amdgcn_sched_group_barrier(K1_SGB_TRANS, 1, 0);     \
    }
#else
#define K1_MIX_QK()
#define K1_MIX_PV()
#endif

template <bool CAUSAL>
__global__ __launch_bounds__(64 * kWaves, 1) void flash_attn_mfma_kernel(
    const __bf16 *__restrict__ q, const __bf16 *__restrict__ k, const __bf16 *__restrict__ v,
    __bf16 *__restrict__ o, float *__restrict__ lse, int Sq, int Skv, int Hq, int Hkv,
    int num_q_tiles, float scale_log2e) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[kLdsBytes];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 31;  // query row within a block / key row / d row
    const int h = lane >> 5;  // lane half

    // blockIdx.x = kvh + Hkv * (hg + G * J): kv head fastest
    const int G = Hq / Hkv;
    int bid = blockIdx.x;
    const int kvh = bid % Hkv;
    bid /= Hkv;
    const int hg = bid % G;
    const int J = bid / G;
    const int hq = kvh * G + hg;
    const int b = blockIdx.y;
    const int tile_long = num_q_tiles - 1 - J, tile_short = J;

    const int nkv = (Skv + kKV - 1) / kKV;
    auto tiles_for = [&](int r0) {
        if (r0 >= Sq) return 0;
        int n = nkv;
        if (CAUSAL) {
            const int d = (r0 + 31) / kKV + 1;
            n = d < n ? d : n;
        }
        return n;
    };
    int rowA0 = tile_long * kQTile + 32 * wave;
    int rowB0 = tile_short * kQTile + 32 * wave;
    int nA = tiles_for(rowA0);
    int nB = (tile_short != tile_long) ? tiles_for(rowB0) : 0;
    if (nB > nA) {  // block A is the one with the longer key range
        int x = rowA0; rowA0 = rowB0; rowB0 = x;
        x = nA; nA = nB; nB = x;
    }
    // key tiles the workgroup walks (same in every wave)
    int nsteps = nkv;
    if (CAUSAL) {
        int last = tile_long * kQTile + kQTile - 1;
        last = last < Sq - 1 ? last : Sq - 1;
        const int d = last / kKV + 1;
        nsteps = d < nsteps ? d : nsteps;
    }

    Block A, B;
    A.qrow = rowA0 + c;
    B.qrow = rowB0 + c;
    auto load_q = [&](Block &blk, bool live) {
        const bool ok = live && blk.qrow < Sq;
        const __bf16 *qp = q + (((int64_t)b * Sq + (ok ? blk.qrow : 0)) * Hq + hq) * kD + 8 * h;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            u32x4 x = {0u, 0u, 0u, 0u};
            if (ok) x = *reinterpret_cast<const u32x4 *>(qp + 16 * ks);
            blk.q[ks] = x;
        }
        blk.m = -INFINITY;
        blk.l = 0.0f;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) blk.acc[dt][r] = 0.0f;
    };
    load_q(A, nA > 0);
    load_q(B, nB > 0);

    // ---- staging: thread handles chunks idx = i*256 + tid (i = 0..3) of a 64 x 16-chunk tile
    const int st_ch = tid & 15;
    const int st_row0 = tid >> 4;  // + 16*i
    const int64_t kv_row_stride = (int64_t)Hkv * kD;  // elements
    const __bf16 *k_base = k + ((int64_t)b * Skv * Hkv + kvh) * kD + st_ch * 8;
    const __bf16 *v_base = v + ((int64_t)b * Skv * Hkv + kvh) * kD + st_ch * 8;
    u32x4 stk[4], stv[4];
    auto stage_load = [&](int t) {
        const int kv0 = t * kKV;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            // rows past the end are clamped to the last row: always a valid, finite load; those
            // keys are masked out of S in the block's last tile, so their P is exactly 0
            int r = kv0 + st_row0 + 16 * i;
            r = r < Skv ? r : Skv - 1;
            stk[i] = *reinterpret_cast<const u32x4 *>(k_base + (int64_t)r * kv_row_stride);
            stv[i] = *reinterpret_cast<const u32x4 *>(v_base + (int64_t)r * kv_row_stride);
        }
    };
    auto stage_write = [&](int kslot, int vslot) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = st_row0 + 16 * i;
            *reinterpret_cast<u32x4 *>(lds + kKBase + kslot * kKTile + r * kKRow + 16 * st_ch) = stk[i];
            *reinterpret_cast<u32x4 *>(lds + vslot * kVTile + r * kVRow +
                                       16 * (st_ch ^ (4 * (r & 3)))) = stv[i];
        }
    };

    // ---- LDS read addresses
    // K A-fragment (u, ks): row 32u + c, 16-byte chunk 2ks + h of a padded row
    const unsigned k_rd = (unsigned)(kKBase + c * kKRow + 16 * h);
    // V^T A-fragment (transposed read): 16-lane group g16 = lane>>4 covers d chunk
    // 4dt + 2(g16&1) + (p>>1), key rows 16sp + 4h + qq (+8 for the second read)
    const int c16 = lane & 15, qq = c16 >> 2, pp = c16 & 3, g16 = lane >> 4;
    const int v_row_in = 4 * h + qq;
    const int v_sw = 4 * (v_row_in & 3);
    const int v_ch_lo = 2 * (g16 & 1) + (pp >> 1);
    unsigned v_rd[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
        v_rd[dt] = (unsigned)(v_row_in * kVRow + 8 * (pp & 1) + 16 * ((4 * dt + v_ch_lo) ^ v_sw));

    // ---- the pipeline's building blocks
    // S^T = K Q^T for one block over the 64 keys of K ring slot `kslot`
    auto qk = [&](Item &it, const Block &blk, int kslot) {
        const unsigned char *kt = lds + k_rd + kslot * kKTile;
#pragma unroll
        for (int r = 0; r < 16; ++r) { it.sa[r] = 0.0f; it.sb[r] = 0.0f; }
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            const u32x4 a = *reinterpret_cast<const u32x4 *>(kt + 32 * ks);
            it.sa = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a),
                                                            __builtin_bit_cast(bf16x8, blk.q[ks]),
                                                            it.sa, 0, 0, 0);
        }
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            const u32x4 a = *reinterpret_cast<const u32x4 *>(kt + 32 * kKRow + 32 * ks);
            it.sb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a),
                                                            __builtin_bit_cast(bf16x8, blk.q[ks]),
                                                            it.sb, 0, 0, 0);
        }
    };
    // O^T += V^T P^T for one block over the 64 keys of V ring slot `vslot`
    auto pv = [&](const Item &it, Block &blk, int vslot) {
        const unsigned char *vt = lds + vslot * kVTile;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
#pragma unroll
            for (int sp = 0; sp < 4; ++sp) {
                const unsigned char *p0 = vt + v_rd[dt] + sp * 16 * kVRow;
                const s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)p0);
                const s16x4 v1 =
                    __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)(p0 + 8 * kVRow));
                const s16x8 va = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
                blk.acc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                    __builtin_bit_cast(bf16x8, va), __builtin_bit_cast(bf16x8, it.p[sp]),
                    blk.acc[dt], 0, 0, 0);
            }
        }
    };
    // mask of a block's last key tile: key < Skv and (causal) key <= query row
    auto mask = [&](Item &it, const Block &blk, int kv0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = kv0 + (r & 3) + 8 * (r >> 2) + 4 * h;
            const bool va = (key < Skv) && (!CAUSAL || key <= blk.qrow);
            const bool vb = (key + 32 < Skv) && (!CAUSAL || key + 32 <= blk.qrow);
            it.sa[r] = va ? it.sa[r] : -INFINITY;
            it.sb[r] = vb ? it.sb[r] : -INFINITY;
        }
    };
    // row max, deferred-rescale decision (wave-uniform), new reference max; l is rescaled here,
    // O later (just before this item's P.V)
    auto stats = [&](Item &it, Block &blk) {
        float mx = fmaxf(it.sa[0], it.sb[0]);
#pragma unroll
        for (int r = 1; r < 16; ++r) mx = fmaxf(fmaxf(mx, it.sa[r]), it.sb[r]);
        mx = half_max(mx);
        // keep the old reference max while every row max of the wave grew by less than
        // kDeferLog2 (exp2 domain): P stays <= 2^kDeferLog2
        const bool need = __any((mx - blk.m) * scale_log2e > kDeferLog2);
        const float m_new = need ? fmaxf(blk.m, mx) : blk.m;
        const float m_safe = (m_new == -INFINITY) ? 0.0f : m_new;
        const float alpha = need ? fast_exp2((blk.m - m_safe) * scale_log2e) : 1.0f;
        blk.m = m_new;
        blk.l *= alpha;
        it.mc = m_safe * scale_log2e;
        it.alpha = alpha;
        it.need = need;
        it.psum = 0.0f;
    };
    auto exps = [&](Item &it, const f32x16 &s, int u) {
        float pe[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            pe[r] = fast_exp2(__builtin_fmaf(s[r], scale_log2e, -it.mc));
            it.psum += pe[r];
        }
#pragma unroll
        for (int sh = 0; sh < 2; ++sh)
            it.p[2 * u + sh] = u32x4{pack_bf16x2(pe[8 * sh + 0], pe[8 * sh + 1]),
                                     pack_bf16x2(pe[8 * sh + 2], pe[8 * sh + 3]),
                                     pack_bf16x2(pe[8 * sh + 4], pe[8 * sh + 5]),
                                     pack_bf16x2(pe[8 * sh + 6], pe[8 * sh + 7])};
    };
    auto exp_a = [&](Item &it) { exps(it, it.sa, 0); };
    auto exp_b = [&](Item &it, Block &blk) {
        exps(it, it.sb, 1);
        blk.l += it.psum;
    };
    auto rescale = [&](const Item &it, Block &blk) {
        if (it.need) {
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) blk.acc[dt] *= it.alpha;
        }
    };
    // finish a pending item: second half of the exponentials, then its P.V
    auto drain = [&](Item &it, Block &blk, int vslot) {
        exp_b(it, blk);
        rescale(it, blk);
        pv(it, blk, vslot);
    };
    // a block's last key tile, masked, un-pipelined
    auto last_item = [&](Item &it, Block &blk, int kslot, int vslot, int kv0) {
        qk(it, blk, kslot);
        mask(it, blk, kv0);
        stats(it, blk);
        exp_a(it);
        exp_b(it, blk);
        rescale(it, blk);
        pv(it, blk, vslot);
    };

    Item I0, I1;
    // "nothing pending" = an item whose P is all zero, multiplied into a zeroed V slot
    auto clear_item = [&](Item &it) {
#pragma unroll
        for (int r = 0; r < 16; ++r) { it.sa[r] = -INFINITY; it.sb[r] = -INFINITY; }
#pragma unroll
        for (int i = 0; i < 4; ++i) it.p[i] = u32x4{0u, 0u, 0u, 0u};
        it.mc = 0.0f; it.alpha = 1.0f; it.psum = 0.0f; it.need = false;
    };
    clear_item(I0);
    clear_item(I1);
    bool pend1 = true;  // the pending item lives in I1 (else I0)

    // ---- prologue: tile 0 into slot 0, V slot 2 ("tile -1") zeroed, tile 1 in flight
    stage_load(0);
    {
        const u32x4 z = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int i = 0; i < 4; ++i)
            *reinterpret_cast<u32x4 *>(lds + 2 * kVTile + (i * 256 + tid) * 16) = z;
    }
    stage_write(0, 0);
    if (nsteps > 1) stage_load(1);
    __syncthreads();

    int vs_prev = 2, vs_cur = 0, vs_next = 1;  // V ring slots of tiles t-1, t, t+1
    for (int t = 0; t < nsteps; ++t) {
        const int ks_cur = t & 1;
        if (t + 1 < nsteps) stage_write(ks_cur ^ 1, vs_next);
        if (t + 2 < nsteps) stage_load(t + 2);

        if (t < nB - 1) {
            // ---- both blocks, interior tile: A then B on the same keys
            __builtin_amdgcn_sched_barrier(0);
            qk(I0, A, ks_cur);
            exp_b(I1, B);
            K1_MIX_QK();
            __builtin_amdgcn_sched_barrier(0);
            rescale(I1, B);
            __builtin_amdgcn_sched_barrier(0);
            pv(I1, B, vs_prev);
            stats(I0, A);
            exp_a(I0);
            K1_MIX_PV();
            __builtin_amdgcn_sched_barrier(0);
            qk(I1, B, ks_cur);
            exp_b(I0, A);
            K1_MIX_QK();
            __builtin_amdgcn_sched_barrier(0);
            rescale(I0, A);
            __builtin_amdgcn_sched_barrier(0);
            pv(I0, A, vs_cur);
            stats(I1, B);
            exp_a(I1);
            K1_MIX_PV();
            __builtin_amdgcn_sched_barrier(0);
        } else if (t == nB - 1) {
            // ---- block B's last tile: finish what is pending, then B and A un-pipelined
            drain(I1, B, vs_prev);
            last_item(I1, B, ks_cur, vs_cur, t * kKV);
            if (t == nA - 1) {
                last_item(I1, A, ks_cur, vs_cur, t * kKV);
            } else {
                qk(I1, A, ks_cur);
                stats(I1, A);
                exp_a(I1);
                pend1 = true;
            }
        } else if (t < nA - 1) {
            // ---- block A alone, interior tile, alternating score register sets
            if (pend1) {
                __builtin_amdgcn_sched_barrier(0);
                qk(I0, A, ks_cur);
                exp_b(I1, A);
                K1_MIX_QK();
                __builtin_amdgcn_sched_barrier(0);
                rescale(I1, A);
                __builtin_amdgcn_sched_barrier(0);
                pv(I1, A, vs_prev);
                stats(I0, A);
                exp_a(I0);
                K1_MIX_PV();
                __builtin_amdgcn_sched_barrier(0);
                pend1 = false;
            } else {
                __builtin_amdgcn_sched_barrier(0);
                qk(I1, A, ks_cur);
                exp_b(I0, A);
                K1_MIX_QK();
                __builtin_amdgcn_sched_barrier(0);
                rescale(I0, A);
                __builtin_amdgcn_sched_barrier(0);
                pv(I0, A, vs_prev);
                stats(I1, A);
                exp_a(I1);
                K1_MIX_PV();
                __builtin_amdgcn_sched_barrier(0);
                pend1 = true;
            }
        } else if (t == nA - 1) {
            // ---- block A's last tile
            if (pend1) {
                drain(I1, A, vs_prev);
                last_item(I1, A, ks_cur, vs_cur, t * kKV);
            } else {
                drain(I0, A, vs_prev);
                last_item(I0, A, ks_cur, vs_cur, t * kKV);
            }
        }
        __syncthreads();
        const int x = vs_prev;
        vs_prev = vs_cur;
        vs_cur = vs_next;
        vs_next = x;
    }

    // ---- epilogue: O^T lane (c,h) holds O[qrow][32dt + 8i + 4h + (0..3)] in regs 4i..4i+3
    auto store = [&](Block &blk, bool live) {
        const float l = half_sum(blk.l);
        if (live && blk.qrow < Sq) {
            const float inv = l > 0.0f ? 1.0f / l : 0.0f;
            __bf16 *op = o + (((int64_t)b * Sq + blk.qrow) * Hq + hq) * kD + 4 * h;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    u32x2 w;
                    w[0] = pack_bf16x2(blk.acc[dt][4 * i] * inv, blk.acc[dt][4 * i + 1] * inv);
                    w[1] = pack_bf16x2(blk.acc[dt][4 * i + 2] * inv, blk.acc[dt][4 * i + 3] * inv);
                    *reinterpret_cast<u32x2 *>(op + 32 * dt + 8 * i) = w;
                }
            if (lse && h == 0)
                lse[((int64_t)b * Hq + hq) * Sq + blk.qrow] =
                    blk.m * scale_log2e * 0.6931471805599453f + logf(l);
        }
    };
    store(A, nA > 0);
    store(B, nB > 0);
}

}  // namespace

bool flash_attn_mfma_supported(const FlashAttnArgs &a) {
    if (a.head_dim != kD) return false;
    if (a.num_kv_heads <= 0 || a.num_heads % a.num_kv_heads != 0) return false;
    if (a.batch > 65535 || a.seq_q <= 0 || a.seq_kv <= 0) return false;
    const auto al = [](const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; };
    return al(a.q) && al(a.k) && al(a.v) && al(a.o);
}

hipError_t launch_flash_attn_mfma(const FlashAttnArgs &a, hipStream_t stream) {
    const int num_q_tiles = (a.seq_q + kQTile - 1) / kQTile;
    const int pairs = (num_q_tiles + 1) / 2;
    const dim3 grid((unsigned)(pairs * a.num_heads), (unsigned)a.batch), block(64 * kWaves);
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
