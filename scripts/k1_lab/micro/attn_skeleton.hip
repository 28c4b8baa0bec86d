// attn_skeleton -- lab A/B of attention-loop STRUCTURES on gfx950 (round 3).
//
// Not the product kernel: a compact, compiler-scheduled flash-attention forward (non-causal walk
// over NS key tiles of 64, head_dim 128, bf16 in / bf16 out, f32 accumulate) that is numerically a
// real attention -- softmax(Q K^T / sqrt(d)) V without a running max (fine for N(0,1) data; the
// max chain is still computed so that the vector-ALU work per item is the product's: 16 max3,
// 32 fma, 32 exp2, 16 cvt_pk) -- and is checked against a CPU reference.  What it varies:
//
//   SHAPE 0 : v_mfma_f32_32x32x16_bf16 (the product's tiles)
//   SHAPE 1 : v_mfma_f32_16x16x32_bf16 (same output tile per wave, 4-register accumulators)
//   MODE  0 : 4 waves, ONE per SIMD, 512 registers; a wave owns two 32-row blocks that share every
//             K / V^T fragment read (the product's dual step); one barrier per key tile
//   MODE  1 : 8 waves, TWO per SIMD, 256 registers; a wave owns one 32-row block; per key tile four
//             segments {read K fragments, QK^T, read V^T fragments + softmax, P.V}, waves 4..7 one
//             segment behind waves 0..3 so that one wave of a SIMD computes while its partner loads
//             (MI355X_MICROARCH.md "Two waves per SIMD"); four barriers per key tile
//
// Both modes: S^T = K Q^T orientation (query row on the lane, P^T straight from the accumulators),
// K/V tiles HBM -> LDS by LDS-DMA (buffer_load ... lds) one tile ahead, XOR-swizzled images.
// A workgroup = 256 query rows of one head; every workgroup walks NS tiles (equal work).
//   config-2-sized run:  256 workgroups x 33 tiles  (= the product's 66 items per wave at B=1 S=4096)
//   B=8-sized run:      2048 workgroups x 33 tiles
// Reported TFLOP/s counts 4 * rows * keys * 128 (no row-sum MFMAs), like the product's metric.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

#define CK(x)                                                                              \
    do {                                                                                   \
        hipError_t e_ = (x);                                                               \
        if (e_ != hipSuccess) {                                                            \
            printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__);             \
            exit(2);                                                                       \
        }                                                                                  \
    } while (0)

constexpr int kD = 128, kKV = 64, kRow = 256, kTile = kKV * kRow;  // 16 KiB per K or V tile

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
    bf16x2 r;
    r[0] = static_cast<__bf16>(lo);
    r[1] = static_cast<__bf16>(hi);
    return __builtin_bit_cast(unsigned, r);
}
__device__ __forceinline__ float max3f(float a, float b, float c) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// One 32-row block of one wave.  SHAPE 0: s[u] = keys 32u.. (16 regs), o[dt] = d 32dt.. (16 regs),
// l = 16 regs (ones . P^T); SHAPE 1: s[rt*4+kb] (4 regs each), o[rt*8+dt], l[rt].
template <int SHAPE> struct Blk;
template <> struct Blk<0> {
    f32x16 s[2], o[4], l;
    u32x4 q[8], p[4];
    float mx;
};
template <> struct Blk<1> {
    f32x4 s[8], o[16], l[2];
    u32x4 q[2][4], p[2][2];
    float mx;
};

template <int SHAPE, int MODE>
__global__ __launch_bounds__(MODE == 0 ? 256 : 512, MODE == 0 ? 1 : 2) void attn_skel(
    const __bf16 *__restrict__ q, const __bf16 *__restrict__ k, const __bf16 *__restrict__ v,
    __bf16 *__restrict__ o, float *__restrict__ dummy, int NS, int nkv, float scale_log2e) {
    constexpr int NW = MODE == 0 ? 4 : 8;
    constexpr int NB = MODE == 0 ? 2 : 1;  // 32-row blocks per wave
    __shared__ __attribute__((aligned(16))) unsigned char lds[4 * kTile];  // K x2, V x2
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wg = blockIdx.x;
    const int kvi = wg % nkv;
    const __bf16 *kp = k + (size_t)kvi * NS * kKV * kD;
    const __bf16 *vp = v + (size_t)kvi * NS * kKV * kD;
    const __bf16 *qp = q + (size_t)wg * 256 * kD;
    __bf16 *op = o + (size_t)wg * 256 * kD;

    // ---- lane geometry
    const int c32 = lane & 31, h = lane >> 5;   // SHAPE 0
    const int c16 = lane & 15, g = lane >> 4;   // SHAPE 1
    Blk<SHAPE> blk[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const int row0 = MODE == 0 ? 128 * b + 32 * wave : 32 * wave;
        if constexpr (SHAPE == 0) {
#pragma unroll
            for (int ks = 0; ks < 8; ++ks)
                blk[b].q[ks] = *reinterpret_cast<const u32x4 *>(qp + (size_t)(row0 + c32) * kD + 16 * ks + 8 * h);
#pragma unroll
            for (int i = 0; i < 4; ++i) blk[b].o[i] = f32x16{};
            blk[b].l = f32x16{};
#pragma unroll
            for (int i = 0; i < 16; ++i) { blk[b].s[0][i] = -INFINITY; blk[b].s[1][i] = -INFINITY; }
#pragma unroll
            for (int i = 0; i < 4; ++i) blk[b].p[i] = u32x4{0u, 0u, 0u, 0u};
        } else {
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int s = 0; s < 4; ++s)
                    blk[b].q[rt][s] = *reinterpret_cast<const u32x4 *>(qp + (size_t)(row0 + 16 * rt + c16) * kD + 32 * s + 8 * g);
#pragma unroll
            for (int i = 0; i < 16; ++i) blk[b].o[i] = f32x4{};
#pragma unroll
            for (int i = 0; i < 8; ++i) blk[b].s[i] = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
            blk[b].l[0] = f32x4{};
            blk[b].l[1] = f32x4{};
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int s = 0; s < 2; ++s) blk[b].p[rt][s] = u32x4{0u, 0u, 0u, 0u};
        }
        blk[b].mx = -INFINITY;
    }
    u32x4 ones;
    asm volatile("v_mov_b32 %0, 0x3f803f80" : "=v"(ones[0]));
    ones[1] = ones[0];
    ones[2] = ones[0];
    ones[3] = ones[0];
    asm volatile("" : "+v"(ones));

    // ---- staging by LDS-DMA: a piece = 4 rows x 256 B = 1 KiB per wave instruction; lane L writes
    // position L & 15 of row L >> 4 of the piece and fetches the chunk the image's swizzle puts there
    const unsigned lds_base = (unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned char *)lds;
    const unsigned rin = (unsigned)lane >> 4, pos = (unsigned)lane & 15u;
    constexpr int PIECES = 16 / NW;  // pieces of K (and of V) per wave and tile
    const unsigned long long ka = (unsigned long long)kp, va = (unsigned long long)vp;
    const unsigned tile_bytes = (unsigned)(kKV * kRow);
    const u32x4 krs = {(unsigned)ka, (unsigned)(ka >> 32) & 0xffffu, (unsigned)NS * tile_bytes, 0x00020000u};
    const u32x4 vrs = {(unsigned)va, (unsigned)(va >> 32) & 0xffffu, (unsigned)NS * tile_bytes, 0x00020000u};
    auto dma_tile = [&](int t, int kslot, int vslot, bool do_k, bool do_v) {
#pragma unroll
        for (int j = 0; j < PIECES; ++j) {
            const unsigned r0 = (unsigned)(wave * PIECES + j) * 4u;  // first row of the piece
            const unsigned r = r0 + rin;
            const unsigned soff = (unsigned)t * tile_bytes + r0 * kRow;
            if (do_k) {
                const unsigned voff = rin * kRow + 16u * (pos ^ (r & 15u));
                const unsigned dst = lds_base + kslot * kTile + r0 * kRow;
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                             : : "s"(dst), "v"(voff), "s"(krs), "s"(soff) : "memory");
            }
            if (do_v) {
                const unsigned sw = SHAPE == 0 ? 4u * (r & 3u) : 2u * (r & 7u);
                const unsigned voff = rin * kRow + 16u * (pos ^ sw);
                const unsigned dst = lds_base + (2 + vslot) * kTile + r0 * kRow;
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                             : : "s"(dst), "v"(voff), "s"(vrs), "s"(soff) : "memory");
            }
        }
    };

    // ---- fragment reads
    // SHAPE 0: K A-fragment (u, ks): row 32u + c32, chunk 2ks + h; V^T (dt, sp): two transposed reads
    // SHAPE 1: K A-fragment (kb, s): row 16kb + c16, chunk 4s + g; V^T (dt, s): two transposed reads
    auto read_k = [&](int slot, int i) -> u32x4 {
        if constexpr (SHAPE == 0) {
            const int u = i >> 3, ks = i & 7;
            return *reinterpret_cast<const u32x4 *>(lds + slot * kTile + (32 * u + c32) * kRow + 16 * ((2 * ks + h) ^ (c32 & 15)));
        } else {
            const int kb = i >> 2, s = i & 3;
            return *reinterpret_cast<const u32x4 *>(lds + slot * kTile + (16 * kb + c16) * kRow + 16 * ((4 * s + g) ^ c16));
        }
    };
    auto read_v = [&](int slot, int i) -> u32x4 {
        const unsigned char *base = lds + (2 + slot) * kTile;
        const int qq = c16 >> 2, pp = c16 & 3;
        const unsigned char *p0, *p1;
        if constexpr (SHAPE == 0) {
            const int dt = i >> 2, sp = i & 3;  // d tile of 32, k-step of 16 keys
            const int g16 = lane >> 4;
            const int row_in = 4 * h + qq;      // + 16 sp (+ 8 for the second read)
            const int ch = 4 * dt + 2 * (g16 & 1) + (pp >> 1);
            p0 = base + (16 * sp + row_in) * kRow + 8 * (pp & 1) + 16 * (ch ^ (4 * (row_in & 3)));
            p1 = p0 + 8 * kRow;
        } else {
            const int dt = i >> 1, s = i & 1;   // d tile of 16, k-step of 32 keys
            const int row = 4 * g + qq;         // + 32 s (+ 16 for the second read)
            const int ch = 2 * dt + (pp >> 1);
            p0 = base + (32 * s + row) * kRow + 8 * (pp & 1) + 16 * (ch ^ (2 * (row & 7)));
            p1 = p0 + 16 * kRow;
        }
        const s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)p0);
        const s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)p1);
        const s16x8 vv = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
        return __builtin_bit_cast(u32x4, vv);
    };

    // ---- MFMA helpers
    auto mfma_qk = [&](Blk<SHAPE> &bk, int i, const u32x4 &kf) {  // K fragment i
        if constexpr (SHAPE == 0) {
            const int u = i >> 3, ks = i & 7;
            if (ks == 0) bk.s[u] = f32x16{};
            bk.s[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, kf),
                                                               __builtin_bit_cast(bf16x8, bk.q[ks]), bk.s[u], 0, 0, 0);
        } else {
            const int kb = i >> 2, s = i & 3;
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
                if (s == 0) bk.s[rt * 4 + kb] = f32x4{};
                bk.s[rt * 4 + kb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                    __builtin_bit_cast(bf16x8, kf), __builtin_bit_cast(bf16x8, bk.q[rt][s]), bk.s[rt * 4 + kb], 0, 0, 0);
            }
        }
    };
    auto mfma_pv = [&](Blk<SHAPE> &bk, int i, const u32x4 &vf) {  // V^T fragment i
        if constexpr (SHAPE == 0) {
            const int dt = i >> 2, sp = i & 3;
            bk.o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, vf),
                                                                __builtin_bit_cast(bf16x8, bk.p[sp]), bk.o[dt], 0, 0, 0);
        } else {
            const int dt = i >> 1, s = i & 1;
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
                bk.o[rt * 8 + dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                    __builtin_bit_cast(bf16x8, vf), __builtin_bit_cast(bf16x8, bk.p[rt][s]), bk.o[rt * 8 + dt], 0, 0, 0);
        }
    };
    auto mfma_rowsum = [&](Blk<SHAPE> &bk) {
        if constexpr (SHAPE == 0) {
#pragma unroll
            for (int sp = 0; sp < 4; ++sp)
                bk.l = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ones),
                                                                __builtin_bit_cast(bf16x8, bk.p[sp]), bk.l, 0, 0, 0);
        } else {
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int s = 0; s < 2; ++s)
                    bk.l[rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ones),
                                                                        __builtin_bit_cast(bf16x8, bk.p[rt][s]), bk.l[rt], 0, 0, 0);
        }
    };
    // ---- softmax of an item, element pairs [e0, e1) of 16 (pair = two scores -> one packed word);
    // with_max: the running-max chain of those pairs (computed like the product's, result unused)
    // Pairs 0..7 = the half of the scores the first 8 QK^T fragments produce (SHAPE 0: keys 0..31;
    // SHAPE 1: key blocks 0, 1 of both row tiles), pairs 8..15 = the half the last 8 produce.
    auto softmax_pairs = [&](Blk<SHAPE> &bk, int e0, int e1, bool with_max) {
#pragma unroll
        for (int e = e0; e < e1; ++e) {
            float s0, s1;
            const int half = e >> 3, w = e & 7;
            const int rt1 = w >> 2, kb1 = 2 * half + ((w >> 1) & 1), pr1 = w & 1;  // SHAPE 1
            if constexpr (SHAPE == 0) {
                s0 = bk.s[e >> 3][(2 * e) & 15];
                s1 = bk.s[e >> 3][(2 * e + 1) & 15];
            } else {
                s0 = bk.s[rt1 * 4 + kb1][2 * pr1];
                s1 = bk.s[rt1 * 4 + kb1][2 * pr1 + 1];
            }
            if (with_max) bk.mx = max3f(bk.mx, s0, s1);
            const float y0 = fast_exp2(s0 * scale_log2e), y1 = fast_exp2(s1 * scale_log2e);
            const unsigned wd = pack_bf16x2(y0, y1);
            if constexpr (SHAPE == 0) {
                bk.p[e >> 2][e & 3] = wd;  // elements 8j..8j+7 of the 32 -> k-step j
            } else {
                bk.p[rt1][kb1 >> 1][(kb1 & 1) * 2 + pr1] = wd;
            }
        }
    };

    if constexpr (MODE == 0) {
        // ================= 4 waves, one per SIMD, two blocks per wave, one barrier per tile =======
        dma_tile(0, 0, 0, true, false);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // V slot 1 ("tile -1") must read as zero for the empty first P.V
        for (int i = tid; i < kTile / 16; i += 256) reinterpret_cast<u32x4 *>(lds + 3 * kTile)[i] = u32x4{0u, 0u, 0u, 0u};
        __syncthreads();
        u32x4 fr[4];
        for (int t = 0; t <= NS; ++t) {
            const int ks = t & 1, vs_prev = (t + 1) & 1, vs_cur = t & 1;
            // K(t+1) into the other K slot, V(t) into the slot P.V(t-2) read last step
            if (t + 1 < NS) dma_tile(t + 1, ks ^ 1, 0, true, false);
            if (t < NS) dma_tile(t, 0, vs_cur, false, true);
            // ---- QK^T of tile t  ||  second half of the pending item's softmax
            if (t < NS) {
#pragma unroll
                for (int j = 0; j < 4; ++j) fr[j] = read_k(ks, j);
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const u32x4 kf = fr[i & 3];
                    if (i + 4 < 16) fr[i & 3] = read_k(ks, i + 4);
                    // the pending item's scores are consumed before this tile's overwrite them:
                    // pairs 8..15 live in s[1] (SHAPE 0) / tiles 4..7 (SHAPE 1), rewritten from i = 8 on
                    if (i < 8) {
                        softmax_pairs(blk[0], 8 + i, 9 + i, false);
                        if (NB == 2) softmax_pairs(blk[NB - 1], 8 + i, 9 + i, false);
                    }
                    mfma_qk(blk[0], i, kf);
                    if (NB == 2) mfma_qk(blk[NB - 1], i, kf);
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    softmax_pairs(blk[0], 8 + i, 9 + i, false);
                    if (NB == 2) softmax_pairs(blk[NB - 1], 8 + i, 9 + i, false);
                }
            }
            // ---- P.V of the pending item (tile t-1)  ||  first half of the new item's softmax
            {
                u32x4 pold[NB][4];
#pragma unroll
                for (int b = 0; b < NB; ++b) {
                    if constexpr (SHAPE == 0) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) pold[b][j] = blk[b].p[j];
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j) pold[b][j] = blk[b].p[j >> 1][j & 1];
                    }
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) fr[j] = read_v(vs_prev, j);
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const u32x4 vf = fr[i & 3];
                    if (i + 4 < 16) fr[i & 3] = read_v(vs_prev, i + 4);
#pragma unroll
                    for (int b = 0; b < NB; ++b) {
                        if constexpr (SHAPE == 0) {
                            const int dt = i >> 2, sp = i & 3;
                            blk[b].o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                                __builtin_bit_cast(bf16x8, vf), __builtin_bit_cast(bf16x8, pold[b][sp]), blk[b].o[dt], 0, 0, 0);
                        } else {
                            const int dt = i >> 1, s = i & 1;
#pragma unroll
                            for (int rt = 0; rt < 2; ++rt)
                                blk[b].o[rt * 8 + dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                                    __builtin_bit_cast(bf16x8, vf), __builtin_bit_cast(bf16x8, pold[b][rt * 2 + s]),
                                    blk[b].o[rt * 8 + dt], 0, 0, 0);
                        }
                    }
                    if (t < NS && i < 8) {
#pragma unroll
                        for (int b = 0; b < NB; ++b) softmax_pairs(blk[b], i, i + 1, true);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                // row sums of the pending item
#pragma unroll
                for (int b = 0; b < NB; ++b) {
                    if constexpr (SHAPE == 0) {
#pragma unroll
                        for (int sp = 0; sp < 4; ++sp)
                            blk[b].l = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                                __builtin_bit_cast(bf16x8, ones), __builtin_bit_cast(bf16x8, pold[b][sp]), blk[b].l, 0, 0, 0);
                    } else {
#pragma unroll
                        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                            for (int s = 0; s < 2; ++s)
                                blk[b].l[rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                                    __builtin_bit_cast(bf16x8, ones), __builtin_bit_cast(bf16x8, pold[b][rt * 2 + s]),
                                    blk[b].l[rt], 0, 0, 0);
                    }
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
    } else {
        // ================= 8 waves, two per SIMD, one block per wave, four segments per tile ======
        // group X = waves 0..3, group Y = waves 4..7 runs ONE segment behind X: while a wave of a
        // SIMD is in a compute segment (MFMAs on register operands only) its partner is in a load
        // segment (LDS fragment reads, softmax, DMA issue).
        const bool grpY = wave >= 4;
        Blk<SHAPE> &bk = blk[0];
        u32x4 fr[16];
        dma_tile(0, 0, 0, true, true);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (grpY) __builtin_amdgcn_s_barrier();  // Y starts one segment late
        for (int t = 0; t < NS; ++t) {
            const int sl = t & 1;
            // segment 1 (load): K fragments of tile t -> registers; tiles t+1 requested
            if (t + 1 < NS) dma_tile(t + 1, sl ^ 1, sl ^ 1, true, true);
#pragma unroll
            for (int i = 0; i < 16; ++i) fr[i] = read_k(sl, i);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            // segment 2 (compute): QK^T
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < 16; ++i) mfma_qk(bk, i, fr[i]);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            // segment 3 (load): V^T fragments of tile t -> registers, the item's whole softmax
#pragma unroll
            for (int i = 0; i < 16; ++i) fr[i] = read_v(sl, i);
            softmax_pairs(bk, 0, 16, true);
            // the tiles requested in segment 1 must have landed -- and every wave must have seen
            // that -- before group X reads them two barriers from here
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            // segment 4 (compute): P.V and row sums
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < 16; ++i) mfma_pv(bk, i, fr[i]);
            mfma_rowsum(bk);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
        }
        if (!grpY) __builtin_amdgcn_s_barrier();  // X waits for Y's last segment: barrier counts match
    }

    // ---- epilogue: O / l -> bf16 (plain per-lane stores; the same in every variant)
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const int row0 = MODE == 0 ? 128 * b + 32 * wave : 32 * wave;
        if constexpr (SHAPE == 0) {
            const float inv = 1.0f / blk[b].l[0];
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    u32x2 w;
                    w[0] = pack_bf16x2(blk[b].o[dt][4 * i] * inv, blk[b].o[dt][4 * i + 1] * inv);
                    w[1] = pack_bf16x2(blk[b].o[dt][4 * i + 2] * inv, blk[b].o[dt][4 * i + 3] * inv);
                    *reinterpret_cast<u32x2 *>(op + (size_t)(row0 + c32) * kD + 32 * dt + 8 * i + 4 * h) = w;
                }
        } else {
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
                const float inv = 1.0f / blk[b].l[rt][0];
#pragma unroll
                for (int dt = 0; dt < 8; ++dt) {
                    u32x2 w;
                    w[0] = pack_bf16x2(blk[b].o[rt * 8 + dt][0] * inv, blk[b].o[rt * 8 + dt][1] * inv);
                    w[1] = pack_bf16x2(blk[b].o[rt * 8 + dt][2] * inv, blk[b].o[rt * 8 + dt][3] * inv);
                    *reinterpret_cast<u32x2 *>(op + (size_t)(row0 + 16 * rt + c16) * kD + 16 * dt + 4 * g) = w;
                }
            }
        }
        if (blk[b].mx == 12345.678f) dummy[tid] = blk[b].mx;  // keeps the max chain alive
    }
}

static uint16_t f2bf(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    u += 0x7FFF + ((u >> 16) & 1);
    return (uint16_t)(u >> 16);
}
static float bf2f(uint16_t hh) {
    uint32_t u = (uint32_t)hh << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}

struct Variant {
    const char *name;
    void (*launch)(int, const __bf16 *, const __bf16 *, const __bf16 *, __bf16 *, float *, int, int, float);
};
template <int SHAPE, int MODE>
static void launch(int nwg, const __bf16 *q, const __bf16 *k, const __bf16 *v, __bf16 *o, float *dummy, int NS, int nkv,
                   float sc) {
    attn_skel<SHAPE, MODE><<<nwg, MODE == 0 ? 256 : 512>>>(q, k, v, o, dummy, NS, nkv, sc);
}

int main(int argc, char **argv) {
    int nwg = 256, NS = 33, rounds = 7, iters = 20, nkv = 8;
    for (int i = 1; i < argc; ++i) {
        if (!strncmp(argv[i], "wg=", 3)) nwg = atoi(argv[i] + 3);
        if (!strncmp(argv[i], "ns=", 3)) NS = atoi(argv[i] + 3);
        if (!strncmp(argv[i], "rounds=", 7)) rounds = atoi(argv[i] + 7);
        if (!strncmp(argv[i], "iters=", 6)) iters = atoi(argv[i] + 6);
    }
    const size_t nq = (size_t)nwg * 256 * kD, nk = (size_t)nkv * NS * kKV * kD;
    std::vector<uint16_t> hq(nq), hk(nk), hv(nk), ho(nq);
    std::mt19937 rng(7);
    std::normal_distribution<float> nd(0.f, 1.f);
    for (auto &x : hq) x = f2bf(nd(rng));
    for (auto &x : hk) x = f2bf(nd(rng));
    for (auto &x : hv) x = f2bf(nd(rng));
    uint16_t *dq, *dk, *dv, *dout;
    float *dd;
    CK(hipMalloc(&dq, nq * 2));
    CK(hipMalloc(&dk, nk * 2));
    CK(hipMalloc(&dv, nk * 2));
    CK(hipMalloc(&dout, nq * 2));
    CK(hipMalloc(&dd, 4096));
    CK(hipMemcpy(dq, hq.data(), nq * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dk, hk.data(), nk * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dv, hv.data(), nk * 2, hipMemcpyHostToDevice));
    const float scale = 1.0f / sqrtf((float)kD), sc = scale * 1.4426950408889634f;
    // CPU reference for workgroup 1 (kv stream 1), rows 0..255
    const int refwg = nwg > 1 ? 1 : 0;
    const int S = NS * kKV;
    std::vector<float> ref(256 * kD);
    {
        const uint16_t *Q = hq.data() + (size_t)refwg * 256 * kD;
        const uint16_t *K = hk.data() + (size_t)(refwg % nkv) * S * kD, *V = hv.data() + (size_t)(refwg % nkv) * S * kD;
        std::vector<double> p(S);
        for (int r = 0; r < 256; ++r) {
            double mx = -1e30;
            for (int j = 0; j < S; ++j) {
                double s = 0;
                for (int d = 0; d < kD; ++d) s += (double)bf2f(Q[r * kD + d]) * bf2f(K[(size_t)j * kD + d]);
                p[j] = s * scale;
                mx = std::max(mx, p[j]);
            }
            double den = 0;
            for (int j = 0; j < S; ++j) { p[j] = exp(p[j] - mx); den += p[j]; }
            for (int d = 0; d < kD; ++d) {
                double a = 0;
                for (int j = 0; j < S; ++j) a += p[j] * bf2f(V[(size_t)j * kD + d]);
                ref[r * kD + d] = (float)(a / den);
            }
        }
    }
    const Variant vars[] = {
        {"32x32x16 4w 1/SIMD dual", launch<0, 0>},
        {"16x16x32 4w 1/SIMD dual", launch<1, 0>},
        {"32x32x16 8w 2/SIMD pingpong", launch<0, 1>},
        {"16x16x32 8w 2/SIMD pingpong", launch<1, 1>},
    };
    const int nv = sizeof(vars) / sizeof(vars[0]);
    const double flops = 4.0 * nwg * 256.0 * S * kD;
    printf("attention skeleton: %d workgroups x 256 rows x %d keys (%d tiles), %.2f GFLOP\n", nwg, S, NS, flops / 1e9);
    auto Q = reinterpret_cast<const __bf16 *>(dq);
    auto K = reinterpret_cast<const __bf16 *>(dk);
    auto V = reinterpret_cast<const __bf16 *>(dv);
    auto O = reinterpret_cast<__bf16 *>(dout);
    for (int vi = 0; vi < nv; ++vi) {
        CK(hipMemset(dout, 0xFF, nq * 2));
        vars[vi].launch(nwg, Q, K, V, O, dd, NS, nkv, sc);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(ho.data(), dout, nq * 2, hipMemcpyDeviceToHost));
        double mx = 0, sum = 0;
        for (int i = 0; i < 256 * kD; ++i) {
            double e = fabs((double)bf2f(ho[(size_t)refwg * 256 * kD + i]) - ref[i]);
            if (!(e == e)) e = 1e9;
            mx = std::max(mx, e);
            sum += e;
        }
        printf("%-30s check (workgroup %d vs f64 CPU) max_abs=%.3e mean_abs=%.3e %s\n", vars[vi].name, refwg, mx,
               sum / (256 * kD), (mx < 2e-2 && sum / (256 * kD) < 1e-3) ? "OK" : "FAIL");
    }
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    std::vector<std::vector<double>> times(nv);
    for (int r = 0; r < rounds + 1; ++r)
        for (int vi = 0; vi < nv; ++vi) {
            CK(hipEventRecord(e0, 0));
            for (int i = 0; i < iters; ++i) vars[vi].launch(nwg, Q, K, V, O, dd, NS, nkv, sc);
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (r > 0) times[vi].push_back(ms / iters);
        }
    for (int vi = 0; vi < nv; ++vi) {
        auto &t = times[vi];
        std::sort(t.begin(), t.end());
        const double med = t[t.size() / 2], mn = t[0];
        printf("%-30s median %8.1f us  %7.1f TF   best %8.1f us  %7.1f TF\n", vars[vi].name, med * 1e3,
               flops / (med * 1e-3) / 1e12, mn * 1e3, flops / (mn * 1e-3) / 1e12);
    }
    return 0;
}
