// flash_attn_pp.hip -- LAB VARIANT of K1's dense path (round 3), not part
// of the product library: bf16 flash-attention prefill on the gfx950 matrix cores
// (mfma_f32_16x16x32_bf16), head_dim 128, as an 8-wave workgroup whose two wave groups ping-pong
// on every SIMD ("pp").  A complete kernel (causal / non-causal, GQA, batch, ragged sizes, LSE,
// deferred rescale) that passes the lab's checks on every shape; built and timed by
// scripts/k1_lab/build.sh + k1_lab next to the product kernel (csrc/flash_attn_mfma.hip).
// RESULT (profiles/r03_k1_lab_pp.txt): correct everywhere, but 8-11 % SLOWER than the product
// kernel at config 2 (83.8 against 75.8 us) and 16 % slower at B = 8 (600 against 517 us), so the
// product keeps the 4-wave kernel.  Where the time goes (cycle stamps, -DNVLLM_PP_STAMPS): a wave's
// QK^T segment takes 1680 cycles for 512 cycles of MFMAs -- every LDS-DMA request costs the
// issuing wave 100+ cycles and phase 2 doubles them (two K/V streams) --, the vector-ALU segment
// 1010, the P.V segment 1180; without the fragment sharing of the product's two-block waves every
// MFMA needs its own LDS fragment traffic, and that, not the matrix pipe, sets the step.
//
// Replaces flash_attention_cpu (reference src/attention/flash.rs:137-275) and the never-launched
// CUDA kernel flash_attention_forward_f16 (kernels/flash_attn_fwd.cu:319-466).  Same math:
// softmax(Q K^T * scale + causal) V with online softmax, TOP-LEFT aligned causal mask (key j is
// visible to query i iff j <= i, flash.rs:296-306), native [B,S,H,D] layout, GQA by
// kv head = q head / (Hq/Hkv).
//
// Roofline: MFMA.  Algorithmic FLOPs = 4 * B * Hq * Sq * Skv * D (x 1/2 causal).
//
// Why this structure (scripts/k1_lab/micro/attn_skeleton.hip, profiles/r03_attn_skeleton_ab.txt):
// four structures were timed on the same work -- {32x32x16, 16x16x32 tiles} x {4 waves, one per
// SIMD, two row blocks sharing fragment reads; 8 waves, two per SIMD, compute / load ping-pong}.
// The 8-wave ping-pong on 16x16x32 tiles was fastest at both sizes (config-2-sized 65 us against
// 79 us for the 4-wave 32x32x16 form the round-2 kernel has; B=8-sized 504 against 587 us): a wave
// that loads fragments, runs its softmax and issues the staging while its SIMD partner holds the
// matrix pipe needs no hand-placed instruction stream, and the chip holds a higher clock on the
// 16x16x32 shape (MI355X_MICROARCH.md, DVFS give-back item 7).
//
// Structure (wave64, 512 threads, TWO waves per SIMD, 256 registers each)
//   * a wave owns ONE block of 32 query rows (two 16-row MFMA column tiles).  The query row lives
//     on the LANE (lane & 15) for the whole kernel, so the online-softmax state, the O rescale and
//     the final 1/l are lane-local:
//        S^T[key][row] = K[key][:] . Q[row][:]     A = K (LDS, row reads), B = Q^T (registers)
//        O^T[d][row]  += V^T[d][key] . P^T[key][row]
//                                                  A = V^T (LDS, transposed reads),
//                                                  B = P^T straight from the S^T accumulators
//        l[row]       += 1 . P^T[key][row]         (an MFMA too: the row sums cost no vector ALU)
//   * per key tile of 64 keys a wave runs four SEGMENTS, a barrier behind each:
//        1 load     K fragments of the tile -> registers; request the next tile (LDS-DMA)
//        2 compute  QK^T (32 MFMAs on register operands only)
//        3 load     V^T fragments -> registers; mask (last tiles only), softmax -> P^T
//        4 compute  P.V and row sums (36 MFMAs)
//     Waves 0..3 (group S) and 4..7 (group L) share the SIMDs pairwise; L runs ONE segment
//     behind S, so on every SIMD one wave computes while the other loads.
//   * workgroup = TWO 128-row query tiles of one q head, tile J (group S, the short one) and tile
//     NT-1-J (group L, the long one).  Under the causal mask group S runs out of keys after
//     nS tiles while group L needs nL >= nS.  Phase 1: both groups walk tiles [0, nS) off ONE
//     K/V stream in LDS.  Phase 2: group S has stored its own rows and takes over HALF of group
//     L's remaining keys -- fresh accumulators for L's rows over tiles [nS + hL, nL), its own K/V
//     stream -- while L walks [nS, nS + hL); the two partial results (O, m, l) of a row meet in
//     LDS at the end.  Every workgroup thus runs nS + ceil((nL - nS) / 2) = (NT + 1) tile steps
//     (give or take one) with all 8 waves busy, whatever J is: B = 1, S = 4096, 16 heads is 256
//     equal workgroups, one per CU, and no wave ever walks alone on its SIMD.
//   * K/V tiles of 64 keys go HBM -> LDS directly (buffer_load ... lds), two slots per stream;
//     256-byte rows, 16-byte chunks XOR-swizzled by the row through each lane's SOURCE address
//     (K: chunk ^ (row & 15), conflict-free row-per-lane ds_read_b128; V: chunk ^ 2 (row & 7),
//     conflict-free ds_read_b64_tr_b16).  The tile's offset rides in the buffer descriptor's base
//     and its byte count in num_records, so rows past Skv read as zero.
//   * deferred rescale (T13): a block keeps its reference max while every row max stays within
//     2^6 of it; the check is one compare per lane and one wave-uniform branch per tile, taken
//     before any P of the tile exists (nothing pending: P.V of the tile before is complete).
//   * blockIdx.x enumerates kv head fastest so that the workgroups sharing a kv head's K/V stream
//     land on one XCD's L2 when Hkv is a multiple of 8.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>

#include "nvllm_kernels.hpp"

namespace nvllm {
namespace kernels {

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

// Diagnostic build only (scripts/k1_lab, -DNVLLM_K1_LAB -DNVLLM_PP_STAMPS): per-wave cycle sums of
// the segments, written through the lse pointer.  In the product build no stamp executes.
#if defined(NVLLM_PP_STAMPS) && !defined(NVLLM_K1_LAB)
#error "NVLLM_PP_STAMPS is a lab switch (scripts/k1_lab): not a product build"
#endif
#ifdef NVLLM_PP_STAMPS
#define PP_STAMP(i)                                                                        \
    do {                                                                                   \
        unsigned long long t_;                                                             \
        __builtin_amdgcn_sched_barrier(0);                                                 \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");          \
        __builtin_amdgcn_sched_barrier(0);                                                 \
        seg_sum[i] += t_ - t_prev;                                                         \
        t_prev = t_;                                                                       \
    } while (0)
#else
#define PP_STAMP(i)
#endif

#ifdef PP_NO_SETPRIO
#define PP_SETPRIO(x)
#else
#define PP_SETPRIO(x) __builtin_amdgcn_s_setprio(x)
#endif

constexpr int kD = 128;
constexpr int kKV = 64;                    // keys per tile
constexpr int kRow = 256;                  // bytes of a K / V row of one head
constexpr int kTile = kKV * kRow;          // 16 KiB
constexpr int kGroupRows = 128;            // query rows of a wave group (4 waves x 32)
constexpr int kLdsBytes = 8 * kTile;       // two streams x (K x2, V x2) = 128 KiB
constexpr float kDeferLog2 = 6.0f;         // rescale only when a row max grows by > 2^6 (T13)
constexpr int kMergeFloats = 64 * 68;      // phase-2 partial of a wave: O (64) + m, l (4) per lane
static_assert(4 * kMergeFloats * 4 <= 5 * kTile, "the merge area ends below the O staging of the final epilogue");

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
    bf16x2 r;
    r[0] = static_cast<__bf16>(lo);
    r[1] = static_cast<__bf16>(hi);
    return __builtin_bit_cast(unsigned, r);
}
// fmaxf on MFMA outputs makes hipcc canonicalise every input first: single instructions instead
__device__ __forceinline__ float max3_asm(float a, float b, float c) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// hipcc hoists every loop-invariant address computation out of the tile loop and keeps the results
// in registers (the epilogue's 16 staging addresses, the mask's 32 key offsets, ...): with 256
// registers per wave that is what spills.  A value passed through here is opaque, so whatever is
// computed from it is computed where it is used.
__device__ __forceinline__ int fresh(int x) {
    asm volatile("" : "+v"(x));
    return x;
}
__device__ __forceinline__ unsigned fresh(unsigned x) {
    asm volatile("" : "+v"(x));
    return x;
}

// Per-wave state of one 32-row block: scores and outputs as 16x16 accumulator tiles
//   s[rt * 4 + kb] : S^T, keys 16 kb + 4 g + j, row 16 rt + c      (g = lane >> 4, c = lane & 15)
//   o[rt * 8 + dt] : O^T, d 16 dt + 4 g + j, row 16 rt + c
//   l[rt]          : row sums (ones . P^T: every register holds the row's sum)
//   p[rt][ks]      : P^T B fragments, k-step ks = keys 32 ks .. 32 ks + 31 of the tile
struct Block {
    f32x4 s[8], o[16], l[2];
    u32x4 q[2][4], p[2][2];
    float m[2];    // reference max of the running softmax (raw score units)
    float mc[2];   // m * scale * log2(e) (0 while m is -inf)
    float thr[2];  // a score above this makes the wave re-reference: m + kDeferLog2 / (scale * log2(e))
};

template <bool CAUSAL>
__global__ __launch_bounds__(512, 2) void flash_attn_pp_kernel(
    const __bf16 *__restrict__ q, const __bf16 *__restrict__ k, const __bf16 *__restrict__ v,
    __bf16 *__restrict__ o, float *__restrict__ lse, int Sq, int Skv, int Hq, int Hkv, int num_q_tiles,
    float scale_log2e) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[kLdsBytes];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef NVLLM_PP_STAMPS
    unsigned long long seg_sum[16] = {0}, t_prev;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_prev)::"memory");
#endif
    const int grp = wave >> 2;  // 0 = group S (short tile), 1 = group L (long tile)
    const int w4 = wave & 3;
    const int g = lane >> 4, c = lane & 15;

    // blockIdx.x = kvh + Hkv * (hg + G * J): kv head fastest; blockIdx.y = batch element
    const int G = Hq / Hkv;
    int bid = blockIdx.x;
    const int kvh = bid % Hkv;
    bid /= Hkv;
    const int hg = bid % G;
    const int J = bid / G;
    const int hq = kvh * G + hg;
    const int b = blockIdx.y;
    const int tile_short = J, tile_long = num_q_tiles - 1 - J;
    const bool has_short = tile_short != tile_long;

    const int nkv = (Skv + kKV - 1) / kKV;
    // key tiles a row block starting at r0 (32 rows) / a group's tile needs
    auto tiles_for = [&](int r0, int rows) {
        if (r0 >= Sq) return 0;
        if (!CAUSAL) return nkv;
        int last = r0 + rows - 1;
        last = last < Sq - 1 ? last : Sq - 1;
        const int d = last / kKV + 1;
        return d < nkv ? d : nkv;
    };
    const int nL = tiles_for(tile_long * kGroupRows, kGroupRows);
    const int nS = has_short ? tiles_for(tile_short * kGroupRows, kGroupRows) : 0;  // nS <= nL
    const int R = nL - nS;
    const int hL = (R + 1) >> 1, hS = R - hL;  // phase 2: L walks [nS, nS + hL), S walks [nS + hL, nL)

    // ---- staging by LDS-DMA.  A piece = 4 rows x 256 B = 1 KiB per wave instruction: lane L writes
    // position L & 15 of row L >> 4 of the piece and fetches the chunk the image's swizzle puts there.
    // A wave always stages pieces 4 w4 .. 4 w4 + 3 (rows 16 w4 .. 16 w4 + 15 of the tile).  Phase 1
    // (one stream for all 8 waves): group S stages the K image, group L the V image.  Phase 2 (a
    // stream per group): the group's 4 waves stage both images of their stream.
    const unsigned lds_base = (unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned char *)lds;
    const unsigned rin = (unsigned)lane >> 4, pos = (unsigned)lane & 15u;
    const unsigned row_bytes = (unsigned)(Hkv * kD * 2);
    const long long head_off = ((long long)b * Skv * Hkv + kvh) * kD * 2;
    const unsigned long long ka = (unsigned long long)k + head_off, va = (unsigned long long)v + head_off;
    const long long kv_bytes = (long long)(Skv - 1) * row_bytes + kRow;
    // Per-lane source offset of piece 4 i' + j (j = 0..3; row r = 4 j + rin, row & 15 = r):
    //   K: rin * row_bytes + 16 (pos ^ r)           = dma_k0 ^ (64 j)
    //   V: rin * row_bytes + 16 (pos ^ 2 (r & 7))   = dma_v0 ^ (128 (j & 1))
    // (row_bytes is a multiple of 256, so the XOR only touches the chunk bits): one register each.
    const unsigned dma_k0 = rin * row_bytes + 16u * (pos ^ rin);
    const unsigned dma_v0 = rin * row_bytes + 16u * (pos ^ (2u * rin));
    // descriptor of the rows [row, end of this head's K or V): base advanced in 64-bit scalar
    // arithmetic, num_records = what is left (0 past Skv: reads as zero)
    auto dma_desc = [&](unsigned long long base, int row) -> u32x4 {
        const long long off = (long long)row * row_bytes;
        long long left = kv_bytes - off;
        left = left < 0 ? 0 : left;
        const unsigned long long a = base + (unsigned long long)off;
        return u32x4{(unsigned)a, (unsigned)(a >> 32) & 0xffffu, (unsigned)left, 0x00020000u};
    };
    // stage this wave's pieces of tile `t` into slot `slot` (0 / 1) of stream `stream`: ONE
    // descriptor per image (the wave's 16 rows onward), the piece's row offset in the lane offset
    auto dma_tile = [&](int t, int stream, int slot, bool do_k, bool do_v) {
        const unsigned dst0 = lds_base + (unsigned)((stream * 4 + slot) * kTile + w4 * 16 * kRow);
        if (do_k) {
            const u32x4 krs = dma_desc(ka, t * kKV + 16 * w4);
            const unsigned k0 = fresh(dma_k0);
#pragma unroll
            for (int j = 0; j < 4; ++j)
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds"
                             : : "s"(dst0 + 4u * j * kRow), "v"((k0 ^ (64u * j)) + 4u * j * row_bytes), "s"(krs) : "memory");
        }
        if (do_v) {
            const u32x4 vrs = dma_desc(va, t * kKV + 16 * w4);
            const unsigned v0 = fresh(dma_v0);
#pragma unroll
            for (int j = 0; j < 4; ++j)
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds"
                             : : "s"(dst0 + 2u * kTile + 4u * j * kRow), "v"((v0 ^ (128u * (j & 1))) + 4u * j * row_bytes), "s"(vrs)
                             : "memory");
        }
    };

    // ---- LDS read addresses (relative to a tile's base)
    // K A-fragment (kb, ks): row 16 kb + c, 16-byte chunk 4 ks + g at position (4 ks + g) ^ c:
    //   byte offset c * 256 + 16 ((4 ks + g) ^ c) = k_rd0 ^ (64 ks)
    const unsigned k_rd0 = (unsigned)(c * kRow + 16 * (g ^ c));
    // V^T A-fragment (dt, ks), transposed reads: lane c = 4 qq + pp of group g addresses row
    // 32 ks + 4 g + qq (+16 for the second read), d 16 dt + 4 pp .. + 3 = chunk 2 dt + (pp >> 1) at
    // position chunk ^ 2 (row & 7):  byte offset = v_rd0 ^ (32 dt)
    const int qq = c >> 2, pp = c & 3;
    const int v_row = 4 * g + qq;
    const unsigned v_rd0 = (unsigned)(v_row * kRow + 8 * (pp & 1) + 16 * ((pp >> 1) ^ (2 * (v_row & 7))));

    u32x4 ones;  // bf16 1.0 x 8, opaque so that it stays in registers
    asm volatile("v_mov_b32 %0, 0x3f803f80" : "=v"(ones[0]));
    ones[1] = ones[0];
    ones[2] = ones[0];
    ones[3] = ones[0];
    asm volatile("" : "+v"(ones));

#ifdef NVLLM_PP_STAMPS
    float *lse_dbg = lse;
    lse = nullptr;
#endif
    const float defer_raw = kDeferLog2 / scale_log2e;
    Block blk;
    int row0 = 0;  // first row of the block this wave is working on
    auto init_block = [&](int r0) {
        row0 = r0;
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
            const int row = r0 + 16 * rt + c;
            const bool ok = row < Sq;
            const __bf16 *qp = q + (((int64_t)b * Sq + (ok ? row : 0)) * Hq + hq) * kD + 8 * g;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                blk.q[rt][ks] = u32x4{0u, 0u, 0u, 0u};
                if (ok) blk.q[rt][ks] = *reinterpret_cast<const u32x4 *>(qp + 32 * ks);
            }
            blk.l[rt] = f32x4{0.f, 0.f, 0.f, 0.f};
            blk.m[rt] = -INFINITY;
            blk.mc[rt] = 0.0f;
            blk.thr[rt] = -INFINITY;
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) blk.o[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        // Q is here before the walk goes on (vmcnt(0) as the BUILTIN, which hipcc's wait insertion
        // understands): left pending, hipcc waits for Q at its first use -- inside the QK^T segment
        // of EVERY step, since this function also runs inside the tile loop -- and a vmcnt wait
        // there waits for the tile requests of that step as well
        __builtin_amdgcn_s_waitcnt(0x0F70);
    };

    u32x4 fr[16];  // the tile's K fragments (segments 1-2), then its V^T fragments (segments 3-4)
    // ---- the four segments of one key tile.  `tile` = key tile index, `lt` = LDS base of the
    // tile's K image (its V image sits 2 tiles further), `active` = this wave has keys in the tile.
    auto seg_load_k = [&](const unsigned char *lt) {
        const unsigned a0 = fresh(k_rd0);
#pragma unroll
        for (int i = 0; i < 16; ++i)  // i = kb * 4 + ks
            fr[i] = *reinterpret_cast<const u32x4 *>(lt + (i >> 2) * 16 * kRow + (a0 ^ (64u * (i & 3))));
    };
    // k-step outermost: eight independent accumulators between two MFMAs of one chain.  Head =
    // k-step 0 (8 MFMAs), tail = k-steps 1..3 (24): the step's LDS-DMA requests go between them.
    auto seg_qk_part = [&](int ks0, int ks1) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                for (int rt = 0; rt < 2; ++rt) {
                    if (ks < ks0 || ks >= ks1) continue;
                    const f32x4 a = ks == 0 ? f32x4{0.f, 0.f, 0.f, 0.f} : blk.s[rt * 4 + kb];
                    blk.s[rt * 4 + kb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                        __builtin_bit_cast(bf16x8, fr[kb * 4 + ks]), __builtin_bit_cast(bf16x8, blk.q[rt][ks]), a, 0, 0, 0);
                }
    };
    auto seg_qk_head = [&]() { seg_qk_part(0, 1); };
    auto seg_qk_tail = [&]() { seg_qk_part(1, 4); };
    auto seg_load_v = [&](const unsigned char *lt) {
        const unsigned char *vb = lt + 2 * kTile;
        const unsigned a0 = fresh(v_rd0);
#pragma unroll
        for (int i = 0; i < 16; ++i) {  // i = dt * 2 + ks
            const unsigned char *p0 = vb + (i & 1) * 32 * kRow + (a0 ^ (32u * (i >> 1)));
            const s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)p0);
            const s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)(p0 + 16 * kRow));
            const s16x8 vv = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
            fr[i] = __builtin_bit_cast(u32x4, vv);
        }
    };
    // mask of a block's last key tiles: key < Skv and (causal) key <= query row
    auto mask = [&](int tile) {
        const int key0 = fresh(tile * kKV + 4 * g);
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
            const int row = fresh(row0 + 16 * rt + c);
#pragma unroll
            for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int key = key0 + 16 * kb + j;
                    const bool vis = key < Skv && (!CAUSAL || key <= row);
                    blk.s[rt * 4 + kb][j] = vis ? blk.s[rt * 4 + kb][j] : -INFINITY;
                }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // Deferred rescale (T13).  Each lane tests the max of ITS 16 scores of a row against the row's
    // threshold; only when some lane of the wave fails does the wave take the out-of-line path:
    // the row max over the four lane groups, a new reference, O and l rescaled on the spot.
    // Nothing else is pending at this point (P.V of the tile before is complete), so every value
    // still at the old max is scaled exactly once.
    auto softmax = [&]() {
        float pm[2];
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
            const f32x4 *s = blk.s + rt * 4;
            float a = max3_asm(s[0][0], s[0][1], s[0][2]);
            float bb = max3_asm(s[0][3], s[1][0], s[1][1]);
            a = max3_asm(a, s[1][2], s[1][3]);
            bb = max3_asm(bb, s[2][0], s[2][1]);
            a = max3_asm(a, s[2][2], s[2][3]);
            bb = max3_asm(bb, s[3][0], s[3][1]);
            a = max3_asm(a, s[3][2], s[3][3]);
            pm[rt] = fmaxf(a, bb);
        }
        if (__builtin_expect(__any((pm[0] > blk.thr[0]) | (pm[1] > blk.thr[1])), 0)) {
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
                float mx = pm[rt];
                mx = fmaxf(mx, __shfl_xor(mx, 16));
                mx = fmaxf(mx, __shfl_xor(mx, 32));
                const float m_new = fmaxf(blk.m[rt], mx);
                const float m_safe = (m_new == -INFINITY) ? 0.0f : m_new;
                const float alpha = fast_exp2((blk.m[rt] - m_safe) * scale_log2e);
                blk.m[rt] = m_new;
                blk.mc[rt] = m_safe * scale_log2e;
                blk.thr[rt] = m_new + defer_raw;
#pragma unroll
                for (int dt = 0; dt < 8; ++dt) blk.o[rt * 8 + dt] *= alpha;
                blk.l[rt] *= alpha;
            }
        }
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) {
                const f32x4 s = blk.s[rt * 4 + kb];
                const float y0 = fast_exp2(__builtin_fmaf(s[0], scale_log2e, -blk.mc[rt]));
                const float y1 = fast_exp2(__builtin_fmaf(s[1], scale_log2e, -blk.mc[rt]));
                const float y2 = fast_exp2(__builtin_fmaf(s[2], scale_log2e, -blk.mc[rt]));
                const float y3 = fast_exp2(__builtin_fmaf(s[3], scale_log2e, -blk.mc[rt]));
                blk.p[rt][kb >> 1][(kb & 1) * 2] = pack_bf16x2(y0, y1);
                blk.p[rt][kb >> 1][(kb & 1) * 2 + 1] = pack_bf16x2(y2, y3);
                // one 16x16 tile at a time: hipcc would otherwise hoist all 32 exponentials in front
                // of the packing, and the 30 temporaries that takes are more than the wave has left
                __builtin_amdgcn_sched_barrier(0);
            }
    };
    auto seg_pv = [&]() {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int dt = 0; dt < 8; ++dt)
#pragma unroll
                for (int rt = 0; rt < 2; ++rt)
                    blk.o[rt * 8 + dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                        __builtin_bit_cast(bf16x8, fr[dt * 2 + ks]), __builtin_bit_cast(bf16x8, blk.p[rt][ks]),
                        blk.o[rt * 8 + dt], 0, 0, 0);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
                blk.l[rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                    __builtin_bit_cast(bf16x8, ones), __builtin_bit_cast(bf16x8, blk.p[rt][ks]), blk.l[rt], 0, 0, 0);
    };
    // One key tile = THREE segments, a barrier behind each; every wave of the workgroup runs the
    // same three barriers per step whether it has keys in the tile or not.
    //   A  QK^T (32 MFMAs; the tile's K fragments were requested at the tail of the step before),
    //      the next tile's LDS-DMA requests issued between the MFMAs, then the tile's V^T
    //      fragment reads issued (they land during segment B)
    //   B  mask (last tiles only), softmax -> P^T: vector ALU only; ends with the wait that says
    //      the next tile has landed
    //   C  P.V and row sums (36 MFMAs), then the NEXT tile's K fragment reads issued
    // Group L runs one segment behind group S: S.A | L.C, S.B | L.A, S.C | L.B -- the vector-ALU
    // segment of one wave always lies beside a matrix segment of its SIMD partner.
    // next_tile >= 0: the tile this wave helps to stage during the step (its K and / or V pieces
    // into slot next_slot of `stream`); extra_tile >= 0: group S's request of the first tile of
    // stream 1 (last step of phase 1).  k_next: LDS base of the tile whose K fragments this wave
    // needs in the NEXT step (nullptr: none / not known to have landed -- the next step reads them
    // itself, have_k says which).
    bool have_k = false;
    auto step = [&](int tile, const unsigned char *lt, bool active, int next_tile, int stream, int next_slot,
                    bool do_k, bool do_v, int extra_tile, const unsigned char *k_next) {
        // ---- segment A
        if (active) {
            if (!have_k) seg_load_k(lt);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            seg_qk_head();
        }
        __builtin_amdgcn_sched_barrier(0);
        if (next_tile >= 0) dma_tile(next_tile, stream, next_slot, do_k, do_v);
        if (extra_tile >= 0) dma_tile(extra_tile, 1, 0, true, true);
        __builtin_amdgcn_sched_barrier(0);
        if (active) {
            seg_qk_tail();
            __builtin_amdgcn_sched_barrier(0);
            seg_load_v(lt);
        }
        __builtin_amdgcn_sched_barrier(0);
        PP_STAMP(0);
        __builtin_amdgcn_s_barrier();
        PP_STAMP(1);
        // ---- segment B
        if (active) {
            const bool need_mask = (tile * kKV + kKV > Skv) || (CAUSAL && tile * kKV + kKV - 1 > row0);
            if (need_mask) mask(tile);
            softmax();
        }
        PP_STAMP(2);
        // the tiles requested in segment A have landed -- and every wave of the staging group has
        // seen that -- before any wave reads them (behind the next barrier at the earliest)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        PP_STAMP(3);
        __builtin_amdgcn_s_barrier();
        PP_STAMP(4);
        // ---- segment C
        if (active) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            seg_pv();
        }
        __builtin_amdgcn_sched_barrier(0);
        have_k = k_next != nullptr;
        if (have_k) seg_load_k(k_next);
        __builtin_amdgcn_sched_barrier(0);
        PP_STAMP(5);
        __builtin_amdgcn_s_barrier();
        PP_STAMP(6);
#ifdef NVLLM_PP_STAMPS
        seg_sum[15] += 1;
#endif
    };
    // O / l of the block -> bf16 rows in global memory (+ the log-sum-exp), through a wave-private
    // 8 KiB LDS image (256-byte rows, chunk ch of row r at position ch ^ (r & 15)) so that whole
    // rows leave, four per store instruction
    auto store_block = [&](unsigned char *stg) {
        const int c = fresh(lane & 15), g = fresh(lane >> 4);  // (shadow the kernel's: see fresh())
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
            const float l = blk.l[rt][0];
            const float inv = l > 0.0f ? 1.0f / l : 0.0f;
#pragma unroll
            for (int dt = 0; dt < 8; ++dt) {
                u32x2 w;
                w[0] = pack_bf16x2(blk.o[rt * 8 + dt][0] * inv, blk.o[rt * 8 + dt][1] * inv);
                w[1] = pack_bf16x2(blk.o[rt * 8 + dt][2] * inv, blk.o[rt * 8 + dt][3] * inv);
                *reinterpret_cast<u32x2 *>(stg + (16 * rt + c) * kRow + 16 * ((2 * dt + (g >> 1)) ^ c) + 8 * (g & 1)) = w;
            }
            const int row = row0 + 16 * rt + c;
            if (lse && g == 0 && row < Sq)
                lse[((int64_t)b * Hq + hq) * Sq + row] = blk.m[rt] * scale_log2e * 0.6931471805599453f + logf(l);
        }
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int ri = 4 * r + g;
            const u32x4 x = *reinterpret_cast<const u32x4 *>(stg + ri * kRow + 16 * (c ^ (ri & 15)));
            const int orow = row0 + ri;
            if (orow < Sq) *reinterpret_cast<u32x4 *>(o + (((int64_t)b * Sq + orow) * Hq + hq) * kD + c * 8) = x;
        }
    };

    // ================================ the walk ================================================
    // group S without a tile of its own (odd tile count, middle pair) starts on L's rows right away
    const int own_tile = (grp || !has_short) ? tile_long : tile_short;
    init_block(own_tile * kGroupRows + 32 * w4);
    int nW = tiles_for(row0, 32);  // key tiles this wave's block needs
    // prologue: tile 0 of stream 0 (with a phase 1 group S stages K and group L stages V; without one
    // stream 0 is group L's alone), and with no phase 1 group S's first tile of stream 1 too
    const int total0 = nS + hL;  // tiles of stream 0 = steps of the workgroup
    if (total0 > 0) dma_tile(0, 0, 0, nS > 0 ? grp == 0 : grp == 1, grp == 1);
    if (nS == 0 && hS > 0 && grp == 0) dma_tile(hL, 1, 0, true, true);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (grp) __builtin_amdgcn_s_barrier();  // group L runs one segment behind group S
    PP_STAMP(9);  // prologue

    // ---- phase 1: steps [0, nS), one stream; group S stages the K image, group L the V image.
    // Group S requests its first stream-1 tile beside the last step's staging.
    for (int it = 0; it < nS; ++it) {
        const bool more = it + 1 < total0;
        const int extra = (grp == 0 && it == nS - 1 && hS > 0) ? nS + hL : -1;
        // K fragments of the next step: tile it + 1 of stream 0 -- unless this wave changes over to
        // stream 1 there (group S after its last phase-1 step reads them itself)
        const bool same_next = it + 1 < nS || (grp == 1 && it + 1 < total0);
        const unsigned char *kn = (same_next && it + 1 < nW) ? lds + ((it + 1) & 1) * kTile : nullptr;
        step(it, lds + (it & 1) * kTile, it < nW, more ? it + 1 : -1, 0, (it + 1) & 1, grp == 0, grp == 1, extra, kn);
    }
    if (grp == 0 && has_short && R > 0) {
        // group S: its own rows are done (O staging: slot 1 of stream 1 is still free -- its K image
        // for waves 0, 1, its V image for waves 2, 3); it takes over the far half of group L's
        // remaining keys with fresh accumulators
        store_block(lds + (5 + 2 * (w4 >> 1)) * kTile + (w4 & 1) * 32 * kRow);
        init_block(tile_long * kGroupRows + 32 * w4);
        nW = tiles_for(row0, 32);
        have_k = false;
        PP_STAMP(10);  // transition
    }
    // ---- phase 2: hL steps; L on stream 0 tiles [nS, nS + hL), S on stream 1 tiles [nS + hL, nL)
    {
        const int base2 = grp ? nS : nS + hL, cnt2 = grp ? hL : hS, st2 = grp ? 0 : 1;
        for (int i = 0; i < hL; ++i) {
            const int tile = base2 + i;
            const int slot = (grp ? tile : i) & 1;
            const unsigned char *lt = lds + (st2 * 4 + slot) * kTile;
            const bool more = i + 1 < cnt2;
            const unsigned char *kn = (more && tile + 1 < nW) ? lds + (st2 * 4 + (slot ^ 1)) * kTile : nullptr;
            step(tile, lt, i < cnt2 && tile < nW, more ? tile + 1 : -1, st2, slot ^ 1, true, true, -1, kn);
        }
    }
    if (!grp) __builtin_amdgcn_s_barrier();  // matches group L's late start
    __syncthreads();                          // every LDS tile is dead from here on
    const bool merged = hS > 0;               // group S holds a partial of L's rows

    // ---- the two partial results of L's rows meet: S parks (O, m, l) in LDS, L combines and stores
    float *mg = reinterpret_cast<float *>(lds) + w4 * kMergeFloats + fresh(lane);
    if (grp == 0 && merged) {
#pragma unroll
        for (int i = 0; i < 16; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) mg[(i * 4 + j) * 64] = blk.o[i][j];
        mg[(64 + 0) * 64] = blk.m[0];
        mg[(64 + 1) * 64] = blk.m[1];
        mg[(64 + 2) * 64] = blk.l[0][0];
        mg[(64 + 3) * 64] = blk.l[1][0];
    }
    __syncthreads();
    if (grp == 0 && has_short && R == 0) store_block(lds + w4 * 32 * kRow);  // no phase 2: S's own rows
    if (grp == 1) {
        if (merged) {
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
                const float m2 = mg[(64 + rt) * 64], l2 = mg[(64 + 2 + rt) * 64];
                const float m_new = fmaxf(blk.m[rt], m2);
                const float m_safe = (m_new == -INFINITY) ? 0.0f : m_new;
                const float a1 = fast_exp2((blk.m[rt] - m_safe) * scale_log2e);
                const float a2 = fast_exp2((m2 - m_safe) * scale_log2e);
#pragma unroll
                for (int dt = 0; dt < 8; ++dt)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        blk.o[rt * 8 + dt][j] = blk.o[rt * 8 + dt][j] * a1 + mg[((rt * 8 + dt) * 4 + j) * 64] * a2;
                blk.l[rt][0] = blk.l[rt][0] * a1 + l2 * a2;
                blk.m[rt] = m_new;
            }
        }
        // the merge area (first 68 KiB) is still being read by other L waves: stage O above it
        store_block(lds + 5 * kTile + w4 * 32 * kRow);
    }
#ifdef NVLLM_PP_STAMPS
    PP_STAMP(11);  // merge + epilogue
    if (lse_dbg && lane == 0) {
        unsigned long long *dbg = reinterpret_cast<unsigned long long *>(lse_dbg) +
                                  ((((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 8 + wave) * 16);
        for (int i = 0; i < 16; ++i) dbg[i] = seg_sum[i];
    }
#endif
}

}  // namespace

bool flash_attn_pp_supported(const FlashAttnArgs &a) {
    if (a.head_dim != kD) return false;
    if (a.num_kv_heads <= 0 || a.num_heads % a.num_kv_heads != 0) return false;
    if (a.batch > 65535 || a.seq_q <= 0 || a.seq_kv <= 0) return false;
    // num_records of a buffer descriptor is 32 bits: one (batch, kv head)'s rows span less than 4 GiB
    if ((int64_t)a.seq_kv * a.num_kv_heads * kD * 2 >= (int64_t)1 << 32) return false;
    const auto al = [](const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; };
    return al(a.q) && al(a.k) && al(a.v) && al(a.o);
}

hipError_t launch_flash_attn_pp(const FlashAttnArgs &a, hipStream_t stream) {
    const int num_q_tiles = (a.seq_q + kGroupRows - 1) / kGroupRows;
    const int pairs = (num_q_tiles + 1) / 2;
    const dim3 grid((unsigned)(pairs * a.num_heads), (unsigned)a.batch), block(512);
    const float scale_log2e = a.softmax_scale * 1.4426950408889634f;
    auto q = static_cast<const __bf16 *>(a.q);
    auto k = static_cast<const __bf16 *>(a.k);
    auto v = static_cast<const __bf16 *>(a.v);
    auto o = static_cast<__bf16 *>(a.o);
    if (a.causal)
        flash_attn_pp_kernel<true><<<grid, block, 0, stream>>>(q, k, v, o, a.lse, a.seq_q, a.seq_kv, a.num_heads,
                                                                a.num_kv_heads, num_q_tiles, scale_log2e);
    else
        flash_attn_pp_kernel<false><<<grid, block, 0, stream>>>(q, k, v, o, a.lse, a.seq_q, a.seq_kv, a.num_heads,
                                                                 a.num_kv_heads, num_q_tiles, scale_log2e);
    return hipGetLastError();
}

#ifdef NVLLM_K1_LAB  // scripts/k1_lab/lab.cpp wants a paged launcher per variant: this kernel has none
hipError_t launch_flash_attn_pp_paged_stub(const PagedAttnArgs &, hipStream_t) { return hipErrorNotSupported; }
#endif

}  // namespace kernels
}  // namespace nvllm
