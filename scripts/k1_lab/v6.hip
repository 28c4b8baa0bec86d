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
#include <type_traits>
#include <utility>

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

// Diagnostic build only (scripts/k1_lab, -DNVLLM_K1_STAMPS): per-wave cycle sums of the loop's
// segments, written through the lse pointer.  In the product build no stamp executes.
#ifdef NVLLM_K1_STAMPS
#define K1_STAMP(i)                                                                        \
    do {                                                                                   \
        unsigned long long t_;                                                             \
        __builtin_amdgcn_sched_barrier(0);                                                 \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");          \
        __builtin_amdgcn_sched_barrier(0);                                                 \
        seg_sum[i] += t_ - t_prev;                                                         \
        t_prev = t_;                                                                       \
    } while (0)
#else
#define K1_STAMP(i)
#endif

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
    bf16x2 r;
    r[0] = static_cast<__bf16>(lo);
    r[1] = static_cast<__bf16>(hi);
    return __builtin_bit_cast(unsigned, r);
}

// max over the two lane halves (lane ^ 32): v_permlane32_swap exchanges the upper half of its
// first operand with the lower half of its second, so with x in both, every lane ends up with
// its own value in one register and the other half's in the other.  Inline asm: hipcc (ROCm 7.2)
// folds the builtin's two results into one.  s_nop 1 = the VALU-write -> permlane wait states.
__device__ __forceinline__ float half_max(float x) {
    float a = x, b = x;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    return fmaxf(a, b);
}
__device__ __forceinline__ float half_sum(float x) { return x + __shfl_xor(x, 32); }

// One score register set: raw S^T of a 64-key tile for one 32-row block, the P^T fragments made
// from it and the softmax constants of that item.
struct Item {
    f32x16 sa, sb;  // keys 0..31 / 32..63 of the tile
    u32x4 p[4];     // P^T B fragments, k-step 2u + sh
    float mc;       // reference max * scale * log2(e)
    float alpha;    // pending O rescale factor
    float mx;       // running max of the raw scores
    float hold;     // even element waiting for its pair (bf16 packing)
    bool need;      // wave-uniform: alpha != 1
};

// Per-block softmax state.  The O^T accumulators and the Q^T fragments of both blocks live in the
// accumulator half of the register file under LITERAL names, out of the compiler's sight:
//   O^T of block b, d-tile dt : a[64b + 16dt : 64b + 16dt + 15]
//   Q^T of block b, k-step ks : a[128 + 32b + 4ks : 128 + 32b + 4ks + 3]
//   row sums l of block b     : a[192 + 16b : 192 + 16b + 15]  (ones . P^T, every register the same)
// so that the vector ALU half (256 registers) is left to the softmax and the staging.  The
// compiler never spills (checked at build time: zero spills, no v_accvgpr outside the asm
// statements), so nothing else ever lands in a[0:223].
struct Block {
    float m;   // reference max of the running softmax
#ifdef K1_DBG_VALU_L
    float l;
#endif
    int qrow;  // this lane's query row
};
#define K1_AGPRS "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47", "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63", "a64", "a65", "a66", "a67", "a68", "a69", "a70", "a71", "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79", "a80", "a81", "a82", "a83", "a84", "a85", "a86", "a87", "a88", "a89", "a90", "a91", "a92", "a93", "a94", "a95", "a96", "a97", "a98", "a99", "a100", "a101", "a102", "a103", "a104", "a105", "a106", "a107", "a108", "a109", "a110", "a111", "a112", "a113", "a114", "a115", "a116", "a117", "a118", "a119", "a120", "a121", "a122", "a123", "a124", "a125", "a126", "a127", "a128", "a129", "a130", "a131", "a132", "a133", "a134", "a135", "a136", "a137", "a138", "a139", "a140", "a141", "a142", "a143", "a144", "a145", "a146", "a147", "a148", "a149", "a150", "a151", "a152", "a153", "a154", "a155", "a156", "a157", "a158", "a159", "a160", "a161", "a162", "a163", "a164", "a165", "a166", "a167", "a168", "a169", "a170", "a171", "a172", "a173", "a174", "a175", "a176", "a177", "a178", "a179", "a180", "a181", "a182", "a183", "a184", "a185", "a186", "a187", "a188", "a189", "a190", "a191", "a192", "a193", "a194", "a195", "a196", "a197", "a198", "a199", "a200", "a201", "a202", "a203", "a204", "a205", "a206", "a207", "a208", "a209", "a210", "a211", "a212", "a213", "a214", "a215", "a216", "a217", "a218", "a219", "a220", "a221", "a222", "a223"

template <int I> using ic = std::integral_constant<int, I>;
template <int... Is, class F>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, Is...>, F &&f) {
    (f(ic<Is>{}), ...);
}
template <int N, class F> __device__ __forceinline__ void static_for(F &&f) {
    static_for_impl(std::make_integer_sequence<int, N>{}, static_cast<F &&>(f));
}

// The MFMAs are inline asm so that the operand register files can be chosen.  hipcc neither
// schedules around them nor pads their hazards (cdna_hip_programming.md 5.7): every VALU read of
// an MFMA result below sits at least 12 wait states after the MFMA, by construction in the
// pipelined bodies and by mfma_fence() elsewhere.
// Every MFMA statement also takes the softmax values the vector ALU has produced so far (Deps) as
// unused inputs: that pins the vector-ALU work of an MFMA gap in front of the next MFMA (hipcc
// would otherwise sink the exponentials to their first use, a whole slot later), at no
// instruction cost.  The first MFMA of a 4-fragment group names the group's other three fragments
// the same way, so that one s_waitcnt covers the group.
struct Deps {
    float a, b, c;
    u32x4 p0, p1, p2, p3;
};
#define K1_DEPS(d) "v"(d.a), "v"(d.b), "v"(d.c), "v"(d.p0), "v"(d.p1), "v"(d.p2), "v"(d.p3)
template <int QR, bool FIRST>
__device__ __forceinline__ void mfma_qk(f32x16 &s, const u32x4 &kf, const Deps &d) {
    if constexpr (FIRST)
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, a[%c2:%c3], 0"
                     : "=&v"(s) : "v"(kf), "i"(QR), "i"(QR + 3), K1_DEPS(d) : K1_AGPRS);
    else
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, a[%c2:%c3], %0"
                     : "+v"(s) : "v"(kf), "i"(QR), "i"(QR + 3), K1_DEPS(d) : K1_AGPRS);
}
template <int QR, bool FIRST>
__device__ __forceinline__ void mfma_qk_g(f32x16 &s, const u32x4 &kf, const u32x4 &f1,
                                          const u32x4 &f2, const u32x4 &f3, const Deps &d) {
    if constexpr (FIRST)
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, a[%c2:%c3], 0"
                     : "=&v"(s) : "v"(kf), "i"(QR), "i"(QR + 3), "v"(f1), "v"(f2), "v"(f3), K1_DEPS(d)
                     : K1_AGPRS);
    else
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, a[%c2:%c3], %0"
                     : "+v"(s) : "v"(kf), "i"(QR), "i"(QR + 3), "v"(f1), "v"(f2), "v"(f3), K1_DEPS(d)
                     : K1_AGPRS);
}
template <int AR>
__device__ __forceinline__ void mfma_pv(const u32x4 &vf, const u32x4 &p, const Deps &d) {
    asm volatile("v_mfma_f32_32x32x16_bf16 a[%c2:%c3], %0, %1, a[%c2:%c3]"
                 : : "v"(vf), "v"(p), "i"(AR), "i"(AR + 15), K1_DEPS(d) : K1_AGPRS);
}
template <int AR>
__device__ __forceinline__ void mfma_pv_g(const u32x4 &vf, const u32x4 &p, const u32x4 &f1,
                                          const u32x4 &f2, const u32x4 &f3, const Deps &d) {
    asm volatile("v_mfma_f32_32x32x16_bf16 a[%c2:%c3], %0, %1, a[%c2:%c3]"
                 : : "v"(vf), "v"(p), "i"(AR), "i"(AR + 15), "v"(f1), "v"(f2), "v"(f3), K1_DEPS(d)
                 : K1_AGPRS);
}
template <int R> __device__ __forceinline__ void agpr_write(unsigned x) {
    asm volatile("v_accvgpr_write_b32 a%c1, %0" : : "v"(x), "i"(R));
}
template <int R> __device__ __forceinline__ float agpr_read() {
    float x;
    asm volatile("v_accvgpr_read_b32 %0, a%c1" : "=v"(x) : "i"(R));
    return x;
}
template <int R> __device__ __forceinline__ void agpr_scale(float f) {
    float t;
    asm volatile("v_accvgpr_read_b32 %0, a%c2\n\ts_nop 1\n\tv_mul_f32 %0, %0, %1\n\ts_nop 1\n\t"
                 "v_accvgpr_write_b32 a%c2, %0"
                 : "=&v"(t) : "v"(f), "i"(R));
}
// Pins a value's computation at this point of the instruction stream: volatile asm statements
// keep their order, so the vector-ALU work stays in the MFMA gap it was dealt to (without it
// hipcc sinks the exponentials to their first use, past a whole slot of MFMAs).
__device__ __forceinline__ void pin(float &x) { asm volatile("" : "+v"(x)); }
__device__ __forceinline__ void pin(float &x, float &y, u32x4 &z) {
    asm volatile("" : "+v"(x), "+v"(y), "+v"(z));
}
__device__ __forceinline__ void mfma_fence() { asm volatile("s_nop 15\n\ts_nop 3" ::: "memory"); }

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
    // reserves a[0:223] in the kernel descriptor; every later statement names them literally
    asm volatile("" ::: K1_AGPRS);
    constexpr ic<0> kA{};
    constexpr ic<1> kB{};
    auto load_q = [&](Block &blk, auto bt, bool live) {
        constexpr int bi = decltype(bt)::value;
        const bool ok = live && blk.qrow < Sq;
        const __bf16 *qp = q + (((int64_t)b * Sq + (ok ? blk.qrow : 0)) * Hq + hq) * kD + 8 * h;
        u32x4 x[8];
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            x[ks] = u32x4{0u, 0u, 0u, 0u};
            if (ok) x[ks] = *reinterpret_cast<const u32x4 *>(qp + 16 * ks);
        }
        static_for<32>([&](auto R) {
            constexpr int r = decltype(R)::value;
            agpr_write<128 + 32 * bi + r>(x[r >> 2][r & 3]);
        });
        static_for<64>([&](auto R) { agpr_write<64 * bi + decltype(R)::value>(0u); });
        static_for<16>([&](auto R) { agpr_write<192 + 16 * bi + decltype(R)::value>(0u); });
        blk.m = -INFINITY;
    };
    load_q(A, kA, nA > 0);
    load_q(B, kB, nB > 0);

    // ---- staging: thread handles rows st_row0 + 16i (i = 0..3), 16-byte chunk st_ch, of the K and
    // of the V tile: 8 pieces per tile and thread (0..3 = K rows, 4..7 = V rows).  Buffer loads
    // over this (batch, kv head)'s rows: the tile offset rides in the scalar offset (no address
    // arithmetic per load) and rows past Skv read as zero (those keys are masked out of S in the
    // block's last tile, so their P is exactly 0).
    const int st_ch = tid & 15;
    const int st_row0 = tid >> 4;
    const unsigned row_bytes = (unsigned)(Hkv * kD * 2);
    const unsigned kv_bytes = (unsigned)(Skv - 1) * row_bytes + (unsigned)(kD * 2);
    const int64_t head_off = ((int64_t)b * Skv * Hkv + kvh) * kD;
    const auto k_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16 *>(k + head_off), 0,
                                                          (int)kv_bytes, 0x00020000);
    const auto v_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16 *>(v + head_off), 0,
                                                          (int)kv_bytes, 0x00020000);
    const unsigned st_voff = (unsigned)st_row0 * row_bytes + 16u * (unsigned)st_ch;
    const unsigned st_kw = (unsigned)(kKBase + st_row0 * kKRow + 16 * st_ch);
    const unsigned st_vw = (unsigned)(st_row0 * kVRow + 16 * (st_ch ^ (4 * (st_row0 & 3))));
    u32x4 st[8];
    // fetch piece j of key tile t
    auto stage_l = [&](int t, int j) {
        const unsigned soff = ((unsigned)t * kKV + 16u * (unsigned)(j & 3)) * row_bytes;
        st[j] = __builtin_bit_cast(
            u32x4, __builtin_amdgcn_raw_buffer_load_b128(j < 4 ? k_rsrc : v_rsrc, st_voff, soff, 0));
    };
    // write piece j to K ring slot kslot / V ring slot vslot
    auto stage_w = [&](int kslot, int vslot, int j) {
        if (j < 4)
            *reinterpret_cast<u32x4 *>(lds + st_kw + kslot * kKTile + j * 16 * kKRow) = st[j];
        else
            *reinterpret_cast<u32x4 *>(lds + st_vw + vslot * kVTile + (j - 4) * 16 * kVRow) = st[j];
    };
    auto stage_load = [&](int t) {
#pragma unroll
        for (int j = 0; j < 8; ++j) stage_l(t, j);
    };
    auto stage_write = [&](int kslot, int vslot) {
#pragma unroll
        for (int j = 0; j < 8; ++j) stage_w(kslot, vslot, j);
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

    // fragment ring: two groups of four fragments; a group's LDS reads are issued while the group
    // before it is multiplied
    u32x4 fr[8];
    // K fragment i (u = i>>3, ks = i&7) of a QK slot
    auto read_k = [&](int slot, int i) {
        const unsigned char *kt = lds + k_rd + slot * kKTile;
        fr[i & 7] = *reinterpret_cast<const u32x4 *>(kt + (i >> 3) * 32 * kKRow + 32 * (i & 7));
    };
    // V^T fragment i (dt = i>>2, sp = i&3) of a PV slot
    auto read_v = [&](int slot, int i) {
        const unsigned char *p0 = lds + slot * kVTile + v_rd[i >> 2] + (i & 3) * 16 * kVRow;
        const s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)p0);
        const s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)(p0 + 8 * kVRow));
        const s16x8 va = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
        fr[i & 7] = __builtin_bit_cast(u32x4, va);
    };
    // bf16 1.0 x 8, made once and opaque: a constant hipcc re-creates with v_mov right in front of
    // the MFMA statement would be read before it has landed (VALU write -> MFMA operand hazard)
    u32x4 ones;
    asm volatile("v_mov_b32 %0, 0x3f803f80" : "=v"(ones[0]));
    ones[1] = ones[0];
    ones[2] = ones[0];
    ones[3] = ones[0];
    asm volatile("" : "+v"(ones));

#ifdef K1_DBG_VALU_L
    float dbg_l = 0.f;  // only valid for single-block (nB == 0) waves without rescale
#endif
    // ---- vector-ALU steps of one item's softmax, cut so that they can be dealt out between MFMAs
    // running max, step j of 16: sa pairs first (their MFMAs finished long ago), then sb pairs
    auto max_step = [&](Item &it, int j) {
        const int r = 2 * (j & 7);
        if (j == 0) it.mx = fmaxf(it.sa[0], it.sa[1]);
        else if (j < 8) it.mx = fmaxf(fmaxf(it.mx, it.sa[r]), it.sa[r + 1]);
        else it.mx = fmaxf(fmaxf(it.mx, it.sb[r]), it.sb[r + 1]);
    };
    // deferred-rescale decision (wave-uniform), new reference max; O and l are rescaled later
    // (just before this item's P.V)
    auto decide = [&](Item &it, Block &blk) {
        const float mx = half_max(it.mx);
        // keep the old reference max while every row max of the wave grew by less than
        // kDeferLog2 (exp2 domain): P stays <= 2^kDeferLog2
        const bool need = __any((mx - blk.m) * scale_log2e > kDeferLog2);
        const float m_new = need ? fmaxf(blk.m, mx) : blk.m;
        const float m_safe = (m_new == -INFINITY) ? 0.0f : m_new;
        const float alpha = need ? fast_exp2((blk.m - m_safe) * scale_log2e) : 1.0f;
        blk.m = m_new;
        it.mc = m_safe * scale_log2e;
        it.alpha = alpha;
        it.need = need;
    };
    // element e of 32: p = exp2(s*c - m*c), bf16 pack of pairs (the row sum is an MFMA: ones . P^T)
    auto exp_elem = [&](Item &it, int e) {
        const float s = e < 16 ? it.sa[e & 15] : it.sb[e & 15];
        const float pe = fast_exp2(__builtin_fmaf(s, scale_log2e, -it.mc));
        if (e & 1) it.p[e >> 3][(e & 7) >> 1] = pack_bf16x2(it.hold, pe);
        else it.hold = pe;
#ifdef K1_DBG_VALU_L
        dbg_l += pe;
#endif
    };
    auto deps_of = [&](const Item &it) {
        return Deps{it.mx, it.mc, it.hold, it.p[0], it.p[1], it.p[2], it.p[3]};
    };
    // O^T *= alpha, l *= alpha: rare (deferred), so done register by register through the vector ALU
    auto rescale = [&](const Item &it, auto bt) {
        if (it.need) {
            constexpr int bi = decltype(bt)::value;
            const float f = it.alpha;
            mfma_fence();
            static_for<64>([&](auto R) { agpr_scale<64 * bi + decltype(R)::value>(f); });
            static_for<16>([&](auto R) { agpr_scale<192 + 16 * bi + decltype(R)::value>(f); });
            mfma_fence();
        }
    };
    // VALU share of slice i (of 20) of a P.V slot: statistics of the item whose QK^T just
    // finished, then its first kFirst elements
    constexpr int kFirst = 15;
    auto valu_pv = [&](Item &it, Block &blk, int i) {
#ifdef K1_ABL_NOVALU
        return;
#endif
        if (i < 4) {
#pragma unroll
            for (int j = 4 * i; j < 4 * i + 4; ++j) max_step(it, j);
        } else if (i == 4) {
            decide(it, blk);
        } else {
            exp_elem(it, i - 5);
        }
    };
    // VALU share of slice i (of 16) of a QK^T slot: the remaining 17 elements of the other item
    auto valu_qk = [&](Item &it, int i) {
#ifdef K1_ABL_NOVALU
        return;
#endif
        exp_elem(it, kFirst + i);
        if (i == 15) exp_elem(it, 31);
    };

    // ---- pipelined slots.  Slice = { MFMA ; LDS reads ; a few VALU }.  Fragment group g + 1 is read
    // during the first two slices of group g; next_rd(j) issues the read of fragment j (j < 4) of
    // the slot that follows.  `work` is the item whose softmax runs in this slot's MFMA gaps.
    auto slot_qk = [&](Item &dst, auto bt, int kslot, const Item &work, auto &&valu, auto &&next_rd) {
        constexpr int bi = decltype(bt)::value;
        static_for<16>([&](auto I) {
            constexpr int i = decltype(I)::value;
            constexpr int qr = 128 + 32 * bi + 4 * (i & 7);
            constexpr int g0 = (i & 4);  // ring half of this group
            const Deps d = deps_of(work);
            if constexpr ((i & 3) == 0)
                mfma_qk_g<qr, (i & 7) == 0>(i < 8 ? dst.sa : dst.sb, fr[g0], fr[g0 + 1], fr[g0 + 2],
                                            fr[g0 + 3], d);
            else
                mfma_qk<qr, false>(i < 8 ? dst.sa : dst.sb, fr[i & 7], d);
            if constexpr ((i & 3) < 2) {
                constexpr int j0 = (i & ~3) + 4 + 2 * (i & 3);  // two fragments of the next group
                if constexpr (j0 < 16) { read_k(kslot, j0); read_k(kslot, j0 + 1); }
                else { next_rd(j0 - 16); next_rd(j0 - 15); }
            }
#ifdef K1_DBG_FENCE
            mfma_fence();
#endif
            valu(i);
            // hipcc may copy the fresh scores right behind the last MFMA (it does not know the
            // statement is one): keep the MFMA -> VALU wait states here, once per slot
            if constexpr (i == 15) mfma_fence();
            __builtin_amdgcn_sched_barrier(0);
        });
    };
    // O^T += V^T P^T (16 MFMAs) and l += ones P^T (4 MFMAs) for the item `src`
    auto slot_pv = [&](const Item &src, auto bt, int vslot, const Item &work, auto &&valu,
                       auto &&next_rd) {
        constexpr int bi = decltype(bt)::value;
        static_for<20>([&](auto I) {
            constexpr int i = decltype(I)::value;
            const Deps d = deps_of(work);
            if constexpr (i < 16) {
                constexpr int g0 = (i & 4);
                constexpr int ar = 64 * bi + 16 * (i >> 2);
                if constexpr ((i & 3) == 0)
                    mfma_pv_g<ar>(fr[g0], src.p[0], fr[g0 + 1], fr[g0 + 2], fr[g0 + 3], d);
                else
                    mfma_pv<ar>(fr[i & 7], src.p[i & 3], d);
                if constexpr ((i & 3) < 2) {
                    constexpr int j0 = (i & ~3) + 4 + 2 * (i & 3);
                    if constexpr (j0 < 16) { read_v(vslot, j0); read_v(vslot, j0 + 1); }
                    else { next_rd(j0 - 16); next_rd(j0 - 15); }
                }
            } else {
                mfma_pv<192 + 16 * bi>(ones, src.p[i & 3], d);
            }
#ifdef K1_DBG_FENCE
            mfma_fence();
#endif
            valu(i);
            __builtin_amdgcn_sched_barrier(0);
        });
    };
    auto no_rd = [&](int) {};
    auto no_valu = [&](int) {};

    // ---- un-pipelined pieces (a block's last, masked key tile; pipeline drain)
    auto qk_plain = [&](Item &it, auto bt, int kslot) {
#pragma unroll
        for (int j = 0; j < 4; ++j) read_k(kslot, j);
        slot_qk(it, bt, kslot, it, no_valu, no_rd);
        mfma_fence();
    };
    auto pv_plain = [&](Item &it, auto bt, int vslot) {
        // P must be complete before the fence below (VALU write -> MFMA operand wait states)
        asm volatile("" : "+v"(it.p[0]), "+v"(it.p[1]), "+v"(it.p[2]), "+v"(it.p[3]));
#pragma unroll
        for (int j = 0; j < 4; ++j) read_v(vslot, j);
        mfma_fence();  // P was just written by the vector ALU
        slot_pv(it, bt, vslot, it, no_valu, no_rd);
        mfma_fence();
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
    auto softmax_plain = [&](Item &it, Block &blk, int e_from) {
        if (e_from == 0) {
#pragma unroll
            for (int j = 0; j < 16; ++j) max_step(it, j);
            decide(it, blk);
        }
#pragma unroll
        for (int e = 0; e < 32; ++e)
            if (e >= e_from) exp_elem(it, e);
    };
    // finish a pending item (its first kFirst elements are done), then its P.V
    auto drain = [&](Item &it, Block &blk, auto bt, int vslot) {
        mfma_fence();
        softmax_plain(it, blk, kFirst);
        rescale(it, bt);
        pv_plain(it, bt, vslot);
    };
    // a block's last key tile, masked, un-pipelined
    auto last_item = [&](Item &it, Block &blk, auto bt, int kslot, int vslot, int kv0) {
        qk_plain(it, bt, kslot);
        mask(it, blk, kv0);
        softmax_plain(it, blk, 0);
        rescale(it, bt);
        pv_plain(it, bt, vslot);
    };
    // an interior tile started un-pipelined: leaves the item pending with kFirst elements done
    auto start_item = [&](Item &it, Block &blk, auto bt, int kslot) {
        qk_plain(it, bt, kslot);
#pragma unroll
        for (int j = 0; j < 16; ++j) max_step(it, j);
        decide(it, blk);
#pragma unroll
        for (int e = 0; e < kFirst; ++e) exp_elem(it, e);
    };

    Item I0, I1;
    // "nothing pending" = an item whose P is all zero, multiplied into a zeroed V slot
    auto clear_item = [&](Item &it) {
#pragma unroll
        for (int r = 0; r < 16; ++r) { it.sa[r] = -INFINITY; it.sb[r] = -INFINITY; }
#pragma unroll
        for (int i = 0; i < 4; ++i) it.p[i] = u32x4{0u, 0u, 0u, 0u};
        it.mc = 0.0f; it.alpha = 1.0f; it.need = false; it.mx = -INFINITY;
        it.hold = 0.0f;
    };
    clear_item(I0);
    clear_item(I1);

    // ---- prologue: tile 0 into slot 0, V slot 2 ("tile -1") zeroed, tile 1 in flight
    stage_load(0);
    {
        const u32x4 z = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int i = 0; i < 4; ++i)
            *reinterpret_cast<u32x4 *>(lds + 2 * kVTile + (i * 256 + tid) * 16) = z;
    }
    stage_write(0, 0);
    stage_load(1);
    __syncthreads();

    int vs_prev = 2, vs_cur = 0, vs_next = 1;  // V ring slots of tiles t-1, t, t+1
    int ks_cur = 0;                            // K ring slot of tile t
#ifdef NVLLM_K1_STAMPS
    unsigned long long seg_sum[16] = {0}, t_prev;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_prev)::"memory");
#endif
    // Every wave runs the same number of steps (nsteps), one barrier each, but walks them in its
    // own phases -- separate loops, so that each hot loop is one straight body whose registers
    // (scores, staging) have one definition per iteration.
    int t = 0;
    // end of a step: publish tile t+1 (written during this step), rotate the rings
    auto step_end = [&]() {
        __syncthreads();
        K1_STAMP(11);  // barrier
#ifdef NVLLM_K1_STAMPS
        seg_sum[15] += 1;
#endif
        const int x = vs_prev;
        vs_prev = vs_cur;
        vs_cur = vs_next;
        vs_next = x;
        ks_cur ^= 1;
        ++t;
    };
    // staging of a step, all at once (un-pipelined steps): tile t+1 (fetched during step t-1) goes
    // to LDS, then tile t+2 is fetched into the same registers
    auto stage_step = [&]() {
        stage_write(ks_cur ^ 1, vs_next);
        stage_load(t + 2);
    };
    // ... or dealt out over the MFMA gaps of the pipelined bodies
    // ... or dealt out over the MFMA gaps of the pipelined bodies, into slices that issue no
    // fragment reads (2, 3 mod 4, and the four l slices of a P.V slot)
#ifdef K1_ABL_NOSTAGE
    auto st_w8 = [&](int) {};
    auto st_l8 = [&](int) {};
#else
    auto st_w8 = [&](int i) { if ((i & 2) && i < 16) stage_w(ks_cur ^ 1, vs_next, (i >> 2) * 2 + (i & 1)); };
    auto st_l8 = [&](int i) { if ((i & 2) && i < 16) stage_l(t + 2, (i >> 2) * 2 + (i & 1)); };
#endif
    auto prime_k = [&]() {
#pragma unroll
        for (int j = 0; j < 4; ++j) read_k(ks_cur, j);
        __builtin_amdgcn_sched_barrier(0);
    };
    // block A alone, interior tile: QK^T into `nxt` while `cur` (pending) is finished and multiplied
    auto single_step = [&](Item &cur, Item &nxt) {
        prime_k();
        slot_qk(nxt, kA, ks_cur, cur, [&](int i) { valu_qk(cur, i); st_w8(i); },
                [&](int j) { read_v(vs_prev, j); });
        K1_STAMP(5);
        rescale(cur, kA);
        slot_pv(cur, kA, vs_prev, nxt, [&](int i) { valu_pv(nxt, A, i); st_l8(i); }, no_rd);
        K1_STAMP(6);
#ifdef NVLLM_K1_STAMPS
        seg_sum[13] += 1;
#endif
        step_end();
    };
    auto final_step = [&](Item &cur) {
        stage_step();
        drain(cur, A, kA, vs_prev);
        last_item(cur, A, kA, ks_cur, vs_cur, t * kKV);
        K1_STAMP(8);
        step_end();
    };

    // ---- phase 1: both blocks, interior tiles: A then B on the same keys
    while (t < nB - 1) {
        prime_k();
        slot_qk(I0, kA, ks_cur, I1, [&](int i) { valu_qk(I1, i); st_w8(i); },
                [&](int j) { read_v(vs_prev, j); });
        K1_STAMP(0);
        rescale(I1, kB);
        slot_pv(I1, kB, vs_prev, I0, [&](int i) { valu_pv(I0, A, i); st_l8(i); },
                [&](int j) { read_k(ks_cur, j); });
        K1_STAMP(1);
        slot_qk(I1, kB, ks_cur, I0, [&](int i) { valu_qk(I0, i); },
                [&](int j) { read_v(vs_cur, j); });
        K1_STAMP(2);
        rescale(I0, kA);
        slot_pv(I0, kA, vs_cur, I1, [&](int i) { valu_pv(I1, B, i); }, no_rd);
        K1_STAMP(3);
#ifdef NVLLM_K1_STAMPS
        seg_sum[12] += 1;
#endif
        step_end();
    }
    // ---- block B's last tile: finish what is pending, then B and A un-pipelined
    if (nB > 0) {
        stage_step();
        drain(I1, B, kB, vs_prev);
        last_item(I1, B, kB, ks_cur, vs_cur, t * kKV);
        if (t == nA - 1) last_item(I1, A, kA, ks_cur, vs_cur, t * kKV);
        else start_item(I1, A, kA, ks_cur);
        K1_STAMP(7);
        step_end();
    }
    // ---- phase 2: block A alone, interior tiles, alternating the two score register sets
    // (entered with the pending item, real or empty, in I1)
    while (t + 1 < nA - 1) {
        single_step(I1, I0);
        single_step(I0, I1);
    }
    if (t < nA - 1) {
        single_step(I1, I0);
        if (t == nA - 1) final_step(I0);
    } else if (t == nA - 1) {
        final_step(I1);
    }
    // ---- steps this wave has no keys for (the workgroup's other waves do)
    while (t < nsteps) {
        stage_step();
        K1_STAMP(9);
        step_end();
    }

#ifdef NVLLM_K1_STAMPS
    if (lse && lane == 0) {
        unsigned long long *dbg = reinterpret_cast<unsigned long long *>(lse) +
                                  ((((size_t)blockIdx.y * gridDim.x + blockIdx.x) * kWaves + wave) * 16);
        for (int i = 0; i < 16; ++i) dbg[i] = seg_sum[i];
    }
    lse = nullptr;
#endif
    // ---- epilogue: O^T lane (c,h) holds O[qrow][32dt + 8i + 4h + (0..3)] in regs 4i..4i+3
    mfma_fence();
    auto store = [&](Block &blk, auto bt, bool live) {
        constexpr int bi = decltype(bt)::value;
        float l = agpr_read<192 + 16 * bi>();
#ifdef K1_DBG_VALU_L
        if (bi == 0 && nB == 0) l = half_sum(dbg_l);
#endif
        float ov[64];
        static_for<64>([&](auto R) {
            constexpr int r = decltype(R)::value;
            ov[r] = agpr_read<64 * bi + r>();
        });
        if (live && blk.qrow < Sq) {
            const float inv = l > 0.0f ? 1.0f / l : 0.0f;
            __bf16 *op = o + (((int64_t)b * Sq + blk.qrow) * Hq + hq) * kD + 4 * h;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float *a4 = ov + 16 * dt + 4 * i;
                    u32x2 w;
                    w[0] = pack_bf16x2(a4[0] * inv, a4[1] * inv);
                    w[1] = pack_bf16x2(a4[2] * inv, a4[3] * inv);
                    *reinterpret_cast<u32x2 *>(op + 32 * dt + 8 * i) = w;
                }
            if (lse && h == 0)
                lse[((int64_t)b * Hq + hq) * Sq + blk.qrow] =
                    blk.m * scale_log2e * 0.6931471805599453f + logf(l);
        }
    };
    store(A, kA, nA > 0);
    store(B, kB, nB > 0);
}

}  // namespace

bool flash_attn_mfma_supported(const FlashAttnArgs &a) {
    if (a.head_dim != kD) return false;
    if (a.num_kv_heads <= 0 || a.num_heads % a.num_kv_heads != 0) return false;
    if (a.batch > 65535 || a.seq_q <= 0 || a.seq_kv <= 0) return false;
    // K/V of one batch element are addressed with 32-bit buffer offsets
    if ((int64_t)(a.seq_kv + 3 * kKV) * a.num_kv_heads * kD * 2 >= (int64_t)1 << 31) return false;
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
