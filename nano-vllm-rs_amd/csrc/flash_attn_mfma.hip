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
//     ragged tail): a pipelined step like the others, masked between its slots.
//   * K/V tiles of 64 keys are shared by the 4 waves through LDS rings (K x3,
//     V x3; 256-byte rows, chunks XOR-swizzled by the row), filled HBM -> LDS
//     directly by buffer_load ... lds one (V) or two (K) steps ahead; one
//     barrier per key tile behind a counted s_waitcnt vmcnt.
//   * blockIdx.x enumerates kv head fastest so that the workgroups sharing a
//     kv head's K/V stream land on one XCD's L2 when Hkv is a multiple of 8.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <type_traits>
#include <utility>

#include "nvllm_kernels.hpp"

// The lab switches below (scripts/k1_lab: cycle stamps, ablations that produce WRONG results by
// design) must never reach the product library: they compile only when the lab says who it is.
#if !defined(NVLLM_K1_LAB) &&                                                                      \
    (defined(NVLLM_K1_STAMPS) || defined(K1_FINE) || defined(K1_RT) || defined(K1_ABL_NOVALU) ||   \
     defined(K1_DEV_ONLY_CHAIN) || defined(K1_STAMP_SETUP) ||                                                                 \
     defined(K1_ABL_NOSTAGE) || defined(K1_ABL_NOBRANCH) || defined(K1_ABL_SKIP_KINDS) ||          \
     defined(K1_DBG_VALU_L) || defined(K1_DBG_FENCE))
#error "K1 lab switch (NVLLM_K1_STAMPS / K1_ABL_* / K1_DBG_*) defined without NVLLM_K1_LAB: not a product build"
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
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

constexpr int kD = 128;
constexpr int kWaves = 4;
constexpr int kQTile = 32 * kWaves;  // query rows per tile (two tiles per workgroup)
constexpr int kKV = 64;              // keys per tile
constexpr int kVRow = 256, kVTile = kKV * kVRow;  // 16 KiB, chunks XOR-swizzled by the row
constexpr int kKRow = 256, kKTile = kKV * kKRow;  // 16 KiB, likewise (conflict-free b128 row reads)
// K/V tiles go HBM -> LDS directly (buffer_load ... lds, no staging registers, no ds_write): an
// instruction writes 1 KiB linearly, so both images have plain 256-byte rows and the swizzle is
// applied through each lane's SOURCE address.  K ring of 3: a tile is requested two steps before
// it is read; V ring of 3: one step before, and read a step later than K.
// Ring depth: 3 + 3 (98,304 B; K requested two tiles ahead, V one).  The kernel also builds with
// rings of 4 + 4 (K three tiles, V two tiles ahead, 131,072 B), tried in round 3 for the paged
// kernel's short-chunk shapes, which stream K/V from HBM at one workgroup per CU: SLOWER everywhere
// (256 x 2048: T = 16 412.7 -> 428.8 us, T = 32 442.8 -> 448.9, T = 64 636.5 -> 647.2; 16 x 4096
// T = 64 101.6 -> 119.5) -- the prologue has two more tiles to wait for and the steps gain nothing,
// so those shapes are not bound by the bytes a CU keeps in flight.
constexpr int ring_depth(bool paged, bool chain) { return (void)paged, (void)chain, 3; }
constexpr int ring_bytes(int depth) { return depth * (kVTile + kKTile); }
constexpr int kORow = 272, kOStage = 32 * kORow;  // epilogue: O rows of a block, padded
constexpr int kArgsCopy = 128;                    // CHAIN: a wave's copy of the kernel arguments (KArgs)
static_assert(ring_bytes(3) + kWaves * (kOStage + kArgsCopy) <= 160 * 1024 && ring_bytes(4) <= 160 * 1024 &&
                  kWaves * kOStage <= ring_bytes(3),
              "rings + O staging must fit the CU's LDS");
constexpr float kDeferLog2 = 6.0f;  // rescale only when a row max grows by > 2^6 (T13)

// Diagnostic build only (scripts/k1_lab, -DNVLLM_K1_STAMPS): per-wave cycle sums of the loop's
// segments, written through the lse pointer.  In the product build no stamp executes.
#ifdef NVLLM_K1_STAMPS
#define K1_STAMP_IMPL(i)                                                                   \
    do {                                                                                   \
        unsigned long long t_;                                                             \
        __builtin_amdgcn_sched_barrier(0);                                                 \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");          \
        __builtin_amdgcn_sched_barrier(0);                                                 \
        seg_sum[i] += t_ - t_prev;                                                         \
        t_prev = t_;                                                                       \
    } while (0)
#ifdef K1_FINE
#define K1_STAMP(i)                                                                        \
    do {                                                                                   \
        if ((i) == 11 || (i) == 2) K1_STAMP_IMPL(i);                                        \
        else K1_STAMP_IMPL(10);                                                            \
    } while (0)
#else
#define K1_STAMP(i) K1_STAMP_IMPL(i)
#endif
#else
#define K1_STAMP(i)
#endif
// finer diagnostic (-DNVLLM_K1_STAMPS -DK1_FINE): cycles per 8-slice group of the dual slots
// (0..3 = QK^T groups, 4..8 = P.V groups) instead of the per-segment sums; 11 = barrier
#if defined(NVLLM_K1_STAMPS) && defined(K1_FINE)
#define K1_FSTAMP(i) K1_STAMP_IMPL(i)
#else
#define K1_FSTAMP(i)
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
    float mx;       // running max of the raw scores
    float xf;       // exponent of the element in flight (s*c - m*c)
    float ya, yb;   // exponentials of an even / odd element waiting to be packed
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
    float m;    // reference max of the running softmax
    float mc;   // m * scale * log2(e) (0 while m is -inf)
    float thr;  // a row max above this makes the wave re-reference: m + kDeferLog2 / (scale * log2(e))
#ifdef K1_DBG_VALU_L
    float l;
#endif
    int qrow;  // this lane's query row (position)
    int r0;    // first (packed) row of the block in the sequence's tiling
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
    float a, b, c, e, f;
    u32x4 p0, p1, p2, p3;
};
#define K1_DEPS(d) \
    "v"(d.a), "v"(d.b), "v"(d.c), "v"(d.e), "v"(d.f), "v"(d.p0), "v"(d.p1), "v"(d.p2), "v"(d.p3)
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
// An opaque copy: what is computed from it cannot be hoisted out of the loop it sits in (hipcc
// otherwise keeps every loop-invariant lane constant of the item loop's prologue and epilogue
// live across the steps, where no register is free)
__device__ __forceinline__ int fresh(int x) {
    asm volatile("" : "+v"(x));
    return x;
}
__device__ __forceinline__ void mfma_fence() { asm volatile("s_nop 15\n\ts_nop 3" ::: "memory"); }

// fmaxf on MFMA outputs makes hipcc canonicalise every input first (v_max_f32 x, x, x): the running
// max is written as the single instructions it should be
__device__ __forceinline__ float max2_asm(float a, float b) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float max3_asm(float a, float b, float c) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// ---- how an item's softmax is dealt out over the MFMA gaps ------------------------------------
// The vector-ALU work of one item (32 rows x 64 keys) is a fixed sequence of single instructions
// ("units"): 16 x running max (two scores per v_max3), then the element pipeline tick by tick, which
// starts SPECULATIVELY against the block's current reference max (the decision whether that
// reference still holds is one compare and one branch at the very end of the P.V slot, see
// decide_after_pv) -- tick n packs the pair that ended at element
// n - 2 (odd n), exponentiates element n - 1 and multiplies element n, so that no instruction waits
// for the one in front of it.  The units up to and including tick kFirstTicks - 1 run in the gaps of
// the P.V slot that follows the item's QK^T, the rest in the gaps of the next QK^T slot.  One wave
// per SIMD hides about 24 issue cycles per 32-cycle MFMA and nothing averages out between gaps (an
// over-full gap idles the matrix pipe, an under-full one stalls the wave at its next MFMA), so the
// units are dealt by COST: every gap of a slot gets the same total of vector-ALU issue cycles plus
// the LDS / global instructions that gap issues anyway.  Costs are issue cycles measured on gfx950
// (MI355X_MICROARCH.md: v_exp 8, v_fma / v_max3 4, v_cvt_pk 4.5, ds_read 4, ds_write_b128 13,
// buffer_load ~8).
enum UnitKind { kUMax = 0, kUPack = 2, kUExp = 3, kUFma = 4 };
constexpr int kTicksTotal = 34, kFirstTicks = 17;  // ticks in all / beside the P.V slot
constexpr int kMaxUnits = 16 + 80, kPvSlices = 20, kQkSlices = 16;
struct UnitTab {
    int kind[kMaxUnits], arg[kMaxUnits];
    int count, split;  // split = first unit of the QK^T slot's share
};
constexpr UnitTab make_units() {
    UnitTab t{};
    int u = 0;
    for (int j = 0; j < 16; ++j) { t.kind[u] = kUMax; t.arg[u++] = j; }
    for (int n = 0; n < kTicksTotal; ++n) {
        if (n == kFirstTicks) t.split = u;
        if (n >= 3 && (n & 1)) { t.kind[u] = kUPack; t.arg[u++] = n - 2; }   // pair ending at element n - 2
        if (n >= 1 && n <= 32) { t.kind[u] = kUExp; t.arg[u++] = n - 1; }    // element n - 1
        if (n < 32) { t.kind[u] = kUFma; t.arg[u++] = n; }                   // element n
    }
    t.count = u;
    return t;
}
constexpr UnitTab kUnitTab = make_units();
static_assert(kUnitTab.count == kMaxUnits, "unit count");
// issue cost beside an MFMA stream, one wave per SIMD, in tenths of a cycle (scripts/k1_lab/micro/
// gap_cost.hip: the MFMA itself holds the issue port for ~8.5 of its 32 cycles, a plain VALU
// instruction costs ~5.3, v_exp_f32 ~10, and -- whatever the sum -- a second v_exp_f32 in a gap
// that already holds more than three fillers costs another ~8: {exp fma exp fma cvt} runs 42.5
// cycles per MFMA, {exp fma cvt} and {exp fma} 34 and 33.5)
constexpr int unit_cost(int u) { return kUnitTab.kind[u] == kUExp ? 100 : 53; }
constexpr int kGapBudget = 235;  // tenths of a cycle of fillers per gap
enum DealMode { kDealSingle = 0, kDealDualA = 1, kDealDualB = 2 };
// memory instructions a slice issues besides its MFMA: fragment reads of the next group (~4.5
// cycles each), staging writes (ds_write_b128 ~13) / loads (~8)
// staging = the LDS-DMA pieces in the first four staging positions of a slot
constexpr int kDmaK = 180, kDmaV = 250;  // s_mov m0 + s_nop + buffer_load (+ index arithmetic, paged)
constexpr int pv_mem_cost(int mode, int i) {
    if (i >= 16) return 0;
    if (mode == kDealSingle) return (i & 3) < 2 ? 180 : (i < 8 ? kDmaV : 0);
    return ((i & 3) < 2 ? 90 : 0) + ((mode == kDealDualB && (i & 1) && i < 8) ? kDmaV : 0);
}
constexpr int qk_mem_cost(int mode, int i) {
    const bool last_group = i >= 12;                                   // prefetch of the first V^T fragments
    if (mode == kDealSingle) return (i & 3) < 2 ? (last_group ? 180 : 90) : (i < 8 ? kDmaK : 0);
    return ((i & 3) < 2 ? (last_group ? 90 : 45) : 0) + ((mode == kDealDualB && (i & 1) && i < 8) ? kDmaK : 0);
}
struct DealPlan {
    int cum[kPvSlices + kQkSlices + 2];  // units finished before P.V slice i / before QK^T slice i - 20
};
constexpr DealPlan make_deal_plan(int mode) {
    DealPlan p{};
    // P.V slot: units [0, split) over its 20 slices; QK^T slot: the rest over its 16 slices.  A gap
    // takes units in order while they fit its budget, at most one v_exp_f32 -- unless the exps still
    // to come outnumber the gaps still to come, and then as late as possible.
    for (int slot = 0; slot < 2; ++slot) {
        const int n = slot == 0 ? kPvSlices : kQkSlices;
        const int u0 = slot == 0 ? 0 : kUnitTab.split, u1 = slot == 0 ? kUnitTab.split : kUnitTab.count;
        const int base = slot == 0 ? 0 : kPvSlices;
        int exps_left = 0, work_left = 0;
        for (int u = u0; u < u1; ++u) {
            exps_left += kUnitTab.kind[u] == kUExp;
            work_left += unit_cost(u);
        }
        int u = u0;
        for (int i = 0; i < n; ++i) {
            p.cum[base + i] = u;
            const int mem = slot == 0 ? pv_mem_cost(mode, i) : qk_mem_cost(mode, i);
            // budget of this gap: the common one, or this gap's fair share of what is left if that is more
            int cap_left = 0;
            for (int j = i; j < n; ++j) {
                const int m = slot == 0 ? pv_mem_cost(mode, j) : qk_mem_cost(mode, j);
                cap_left += kGapBudget > m ? kGapBudget - m : 0;
            }
            int budget = kGapBudget - mem;
            if (work_left > cap_left) budget += (work_left - cap_left) / (n - i);
            int spent = 0, exps = 0;
            while (u < u1) {
                const bool is_exp = kUnitTab.kind[u] == kUExp;
                if (i < n - 1) {
                    if (is_exp && exps >= 1 && exps_left <= n - 1 - i) break;
                    if (spent + unit_cost(u) > budget && !(is_exp && exps == 0 && exps_left > n - 1 - i)) break;
                }
                spent += unit_cost(u);
                work_left -= unit_cost(u);
                exps += is_exp;
                exps_left -= is_exp;
                ++u;
            }
        }
        if (slot == 0) p.cum[kPvSlices] = kUnitTab.split;
    }
    p.cum[kPvSlices + kQkSlices] = kUnitTab.count;
    return p;
}

// PAGED = K/V come from the paged cache ([num_blocks, 16, Hkv, 128], k / v = cache bases) through
// this sequence's block table, Skv is the sequence's context length and the causal mask is
// bottom-right aligned: query row i sits at position Skv - Sq + i (paged.rs:357-381).  The kernel
// then tiles the query POSITIONS, from the largest multiple of 32 below the first one, so that a
// 32-row block's diagonal stays inside one 64-key tile exactly as in the top-left case.
// KSPLIT (paged only) = blockIdx.z cuts every workgroup's key walk into gridDim.z equal parts, for
// launches with fewer workgroups than CUs (one 512-token chunk against a long context is 64 of
// them): part p walks key tiles [p*n/P, (p+1)*n/P) of its unit's n and leaves a normalised partial
// (bf16 O; f32 m, l) in the workspace; flash_attn_ksplit_merge_kernel combines the P parts.
constexpr int kSplitSlotBytes = 2 * kQTile * kD * 2 + 2 * kQTile * 2 * 4;  // 256 rows: O + (m, l)

// The kernel's arguments as they lie in the kernarg segment.  CHAIN reads what an item's set-up and
// epilogue need again, item by item: held in scalar registers across the steps (where the plain
// kernel has long forgotten them) they cost ~70 spilled SGPRs and, with those, vector registers the
// steps do not have.  Not from the kernarg segment itself (host memory: 2500 cycles a read, measured)
// but from a copy every wave puts into LDS when it starts.
struct KArgs {
    const __bf16 *q, *k, *v;
    __bf16 *o;
    float *lse;
    int Sq, Skv, Hq, Hkv, num_q_tiles, pair_adjacent;
    float scale_log2e;
    const int32_t *block_tables, *context_lens;
    int max_blocks_per_seq;
    unsigned num_cache_blocks;
    unsigned char *split_ws;
    int pack_shift, num_seqs, chain_items, chain_gx;
};
static_assert(offsetof(KArgs, scale_log2e) == 64 && offsetof(KArgs, block_tables) == 72 &&
                  offsetof(KArgs, split_ws) == 96 && offsetof(KArgs, chain_gx) == 116 && sizeof(KArgs) <= 128,
              "KArgs mirrors the kernel's parameter list");
typedef const __attribute__((address_space(4))) KArgs *KArgsPtr;

// NT (paged only) = the K/V requests carry the non-temporal hint: the launch reads more K/V than the
// Infinity Cache holds, each byte once.
template <bool CAUSAL, bool PAGED, bool KSPLIT, bool CHAIN, bool NT = false>
__global__ __launch_bounds__(64 * kWaves, 1) void flash_attn_mfma_kernel(
    const __bf16 *__restrict__ q, const __bf16 *__restrict__ k, const __bf16 *__restrict__ v,
    __bf16 *__restrict__ o, float *__restrict__ lse, int Sq, int Skv_arg, int Hq, int Hkv,
    int num_q_tiles_arg, int pair_adjacent, float scale_log2e,
    const int32_t *__restrict__ block_tables, const int32_t *__restrict__ context_lens,
    int max_blocks_per_seq, unsigned num_cache_blocks, unsigned char *__restrict__ split_ws,
    int pack_shift, int num_seqs, int chain_items, int chain_gx) {
    static_assert(CAUSAL || !PAGED, "the paged variant is causal");
    static_assert(!KSPLIT || PAGED, "the key split is wired for the paged variant");
    static_assert(!CHAIN || !KSPLIT, "a key-split launch has fewer workgroups than CUs: nothing to chain");
    static_assert(PAGED || !NT, "the dense kernel's K/V is re-read by the other heads' workgroups: no hint");
    constexpr int kRing = ring_depth(PAGED, CHAIN);          // K and V ring slots
    constexpr int kKAhead = kRing - 1, kVAhead = kRing - 2;  // step t requests K(t + kKAhead), V(t + kVAhead)
    constexpr int kKBase = kRing * kVTile;
    constexpr int kLdsBytes = ring_bytes(kRing);
    // O staging: behind the rings when the next item's tiles may already be landing in them (CHAIN),
    // else in the rings themselves, free after an item's last barrier
    constexpr int kOStageBase = CHAIN ? kLdsBytes : 0;
    __shared__ __attribute__((aligned(16))) unsigned char lds[kLdsBytes + (CHAIN ? kWaves * (kOStage + kArgsCopy) : 0)];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 31;  // query row within a block / key row / d row
    const int h = lane >> 5;  // lane half
#ifdef NVLLM_K1_STAMPS
    unsigned long long seg_sum[16] = {0}, t_prev;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_prev)::"memory");
#ifdef K1_RT  // wall-clock (100 MHz) start and end of the wave in slots 4 and 9
    unsigned long long rt0_;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rt0_)::"memory");
#endif
#endif

    // PAGED launches may pack the gp = 2^gsh query heads of a group into the rows: packed row rho
    // is position rho >> gsh, head rho & (gp - 1) of the group.  A chunk of T tokens then fills
    // T * gp rows of one kv head's workgroups instead of T rows in each of gp of them, which is what
    // short chunks need (T = 64 at 32Q/8KV: one full workgroup per kv head, not four with a quarter
    // of their rows live).  A 32-row block spans 32 / gp positions, so its diagonal still lies in
    // one 64-key tile.  gsh = 0: a row is a position, as in the dense kernel.
    // blockIdx.x = kvh + Hkv * (hg + Gq * J): kv head fastest; Gq head groups of gp heads per kv head;
    // blockIdx.y = batch element.  PAGED: blockIdx.x = kvh + Hkv * (hg + Gq * (b + num_seqs * J)), the
    // pair index slowest over the WHOLE grid: a paged grid is sized for the worst alignment of the
    // chunk (31 dead positions in front), so its highest pair often has no rows at all, and
    // workgroups are dealt out in order, round-robin over the XCDs and then over each XCD's shader
    // engines, waiting for a slot on the engine whose turn it is.  With J inside a sequence's
    // index, live and empty workgroups alternated and all the live ones queued on half of the
    // engines (T = 64 at 32Q/8KV: 128 of 256 CUs busy, measured with wall-clock stamps); with J
    // slowest the empty ones come last and cost nothing.
    // CHAIN: a workgroup keeps its unit and pair index and works through several batch elements
    // (sequences) one after the other; it requests an item's Q rows and first K/V tiles before it
    // writes the finished item's O (see the loop at the bottom).
    // the arguments an item's set-up / epilogue works with.  CHAIN reads them afresh from the wave's
    // copy in LDS (see KArgs; dword i in lane i, then one v_readlane per field used)
    constexpr bool kArgsInLds = CHAIN && !PAGED;
    if constexpr (kArgsInLds) {
        typedef const __attribute__((address_space(4))) unsigned *KWords;
        const unsigned w = ((KWords)__builtin_amdgcn_kernarg_segment_ptr())[lane & 31];
        *reinterpret_cast<unsigned *>(lds + kLdsBytes + kWaves * kOStage + wave * kArgsCopy + (lane & 31) * 4) = w;
    }
    auto args = [&]() -> KArgs {
        if constexpr (CHAIN && !kArgsInLds) {
            // (the paged kernel: with its arguments born in vector registers hipcc builds the block
            // descriptors there too, which the DMA statements' "s" operands do not survive; it reads
            // the kernarg segment, once per item)
            KArgsPtr p = (KArgsPtr)__builtin_amdgcn_kernarg_segment_ptr();
            asm volatile("" : "+s"(p));
            return KArgs{p->q, p->k, p->v, p->o, p->lse, p->Sq, p->Skv, p->Hq, p->Hkv, p->num_q_tiles,
                         p->pair_adjacent, p->scale_log2e, p->block_tables, p->context_lens,
                         p->max_blocks_per_seq, p->num_cache_blocks, p->split_ws, p->pack_shift, p->num_seqs,
                         p->chain_items, p->chain_gx};
        } else if constexpr (CHAIN) {
            const int l = fresh(tid) & 31;
            const unsigned w = *reinterpret_cast<const unsigned *>(lds + kLdsBytes + kWaves * kOStage + wave * kArgsCopy + l * 4);
            auto S = [&](int i) { return (unsigned)__builtin_amdgcn_readlane((int)w, i); };
            auto P = [&](int i) { return (unsigned long long)S(i) | ((unsigned long long)S(i + 1) << 32); };
            return KArgs{(const __bf16 *)P(0), (const __bf16 *)P(2), (const __bf16 *)P(4), (__bf16 *)P(6), (float *)P(8),
                         (int)S(10), (int)S(11), (int)S(12), (int)S(13), (int)S(14), (int)S(15),
                         __builtin_bit_cast(float, S(16)), (const int32_t *)P(18), (const int32_t *)P(20), (int)S(22),
                         S(23), (unsigned char *)P(24), (int)S(26), (int)S(27), (int)S(28), (int)S(29)};
        } else {
            return KArgs{q, k, v, o, lse, Sq, Skv_arg, Hq, Hkv, num_q_tiles_arg, pair_adjacent, scale_log2e,
                         block_tables, context_lens, max_blocks_per_seq, num_cache_blocks, split_ws, pack_shift,
                         num_seqs, chain_items, chain_gx};
        }
    };
    // ---- the item and everything that follows from it (set by setup(), read by everything below)
    int kvh = 0, hg = 0, J = 0, hu = 0, hq = 0, b = 0;
    int Skv = Skv_arg, num_q_tiles = num_q_tiles_arg;
    // position of query row 0 minus row 0 (kv_off), split into a multiple of 32 (pos_base, where
    // the tiling starts) and the dead rows in front of the first block (pos_shift)
    int kv_off = 0, pos_base = 0, pos_shift = 0;
    unsigned early_id[5] = {0u, 0u, 0u, 0u, 0u};
    int tile_long = 0, tile_short = 0, nA = 0, nB = 0, nsteps = 0;
    int tile0 = 0;  // KSPLIT: first key tile of this part
    bool liveA = false, liveB = false;
    Block A, B;  // qrow = the lane's query POSITION (= row unless PAGED)
    // Q goes to the accumulator file in two phases so that its HBM latency (about 3000 cycles on
    // a first touch) overlaps the first K/V tile's fetch (and, CHAIN, the epilogue of the item
    // before): the loads are issued by setup(), the Q^T registers are written in the prologue.
    u32x4 qa[8], qb[8];
    // reserves a[0:223] in the kernel descriptor; every later statement names them literally
    asm volatile("" ::: K1_AGPRS);
    constexpr ic<0> kA{};
    constexpr ic<1> kB{};

    // ---- staging: thread handles rows st_row0 + 16i (i = 0..3), 16-byte chunk st_ch, of the K and
    // of the V tile: 8 pieces per tile and thread (0..3 = K rows, 4..7 = V rows).  Buffer loads
    // over this (batch, kv head)'s rows: the tile offset rides in the scalar offset (no address
    // arithmetic per load) and rows past Skv read as zero (those keys are masked out of S in the
    // block's last tile, so their P is exactly 0).
    const unsigned row_bytes = (unsigned)(Hkv * kD * 2);
    // PAGED: a wave's four pieces of a tile are the 16 rows of ONE cache block (block 4 tile + wave
    // of the sequence), so the block is addressed through the buffer descriptor itself: base = cache
    // + id * block bytes + the kv head's 256 bytes (64-bit scalar arithmetic, any cache size),
    // num_records = what is left of the block from there, and a lane's 32-bit offset (a lane constant
    // per piece) stays inside the block.  The id is a scalar, read through the scalar cache three
    // tiles ahead.  An id outside the pool gets num_records = 0 and reads as zero, like a row past Skv
    // of a dense launch; rows of V past the context get an offset past the block (true zeros), K needs
    // no such care (those scores are masked).
    const unsigned pg_block_bytes = (unsigned)(16 * Hkv * kD * 2);
    // Piece j of a tile = rows 16 wave + 4 (j & 3) .. + 3 (j < 4 of K, else of V), one
    // instruction: lane L writes LDS bytes [16 L, 16 L + 16) of the piece's 1 KiB, i.e. position
    // L & 15 of row L >> 4, and fetches the chunk that the swizzle puts there
    unsigned dma_kv[4] = {0u, 0u, 0u, 0u}, dma_vv = 0u;
    u32x4 dma_krs = {0u, 0u, 0u, 0u}, dma_vrs = {0u, 0u, 0u, 0u};  // dense: this item's K / V rows
    unsigned dma_pk[4] = {0u, 0u, 0u, 0u}, dma_pv[4] = {0u, 0u, 0u, 0u};
    unsigned pg_s0 = 0, pg_s1 = 0, pg_s2 = 0, pg_s3 = 0;  // id of the wave's block in tiles t+1 .. t+4 (the last one only with rings of 4)
    unsigned long long pg_kbase = 0, pg_vbase = 0;  // PAGED: the caches from this item's kv head on
    unsigned pg_nrec = 0;                           // ... and what is left of a block from there
    constexpr unsigned kPgOutOfBlock = 0x40000000u;  // an offset no block reaches (Hkv < 2^15)
    {
        const unsigned rin = (unsigned)lane >> 4, pos = (unsigned)lane & 15u;
        if constexpr (PAGED) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const unsigned r = 4u * j + rin;
                dma_pk[j] = r * row_bytes + 16u * (pos ^ (r & 15u));
                dma_pv[j] = r * row_bytes + 16u * (pos ^ (4u * rin));
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) dma_kv[j] = rin * row_bytes + 16u * (pos ^ ((4u * j + rin) & 15u));
            dma_vv = rin * row_bytes + 16u * (pos ^ (4u * rin));
        }
    }

    // CHAIN: the workgroup's unit (kv head, head group), pair index and first batch element, decoded
    // once -- blockIdx.x = unit + Hu * (c + chain_gx * J) with chain_gx chains per (unit, J), J slowest
    // (see above); the chain's items are batch elements c, c + chain_gx, ... < chain_items.  An item
    // then costs no integer division (a dozen of them were 3 us of scalar latency per item).
    int ch_kvh = 0, ch_hg = 0, ch_J = 0, ch_b = 0;
    // PAGED: what the next item's set-up waits for longest -- its context length and the ids of its
    // first blocks -- requested one store earlier (pf_b = the sequence they belong to, -1 none)
    int pf_b = -1, pf_ctx = 0;
    unsigned pf_id[4] = {0u, 0u, 0u, 0u};
    if constexpr (CHAIN) {
        const int gsh = PAGED ? pack_shift : 0, Gq = (Hq / Hkv) >> gsh, Hu = Hkv * Gq;
        int bid = blockIdx.x;
        const int u = bid % Hu;
        bid /= Hu;
        ch_b = bid % chain_gx;
        ch_J = bid / chain_gx;
        ch_kvh = u % Hkv;
        ch_hg = u / Hkv;
    }
    // Takes up an item: geometry, the Q loads (into qa / qb), softmax state.  CHAIN: batch element
    // `cb`, every other one with the pair index mirrored (`flip`: a heavy pair of tiles is followed by
    // a light one).  false = the item has no rows (PAGED: the grid is sized for pos_shift = 31).
    auto setup = [&](int cb, bool flip, const KArgs &ka) -> bool {
        const int tid_f = CHAIN ? fresh(tid) : tid;
        const int c = tid_f & 31;  // (shadows the kernel's: not hoistable)
        const int Sq = ka.Sq, Hq = ka.Hq, Hkv = ka.Hkv, num_seqs = ka.num_seqs;
        const int max_blocks_per_seq = ka.max_blocks_per_seq, pair_adjacent = ka.pair_adjacent;
        const int gsh = PAGED ? ka.pack_shift : 0, gp = 1 << gsh, G = Hq / Hkv;
        const int32_t *block_tables = ka.block_tables, *context_lens = ka.context_lens;
        const __bf16 *k = ka.k, *v = ka.v;
        const unsigned row_bytes = (unsigned)(Hkv * kD * 2);
        if constexpr (CHAIN) {
            kvh = ch_kvh;
            hg = ch_hg;
            J = ch_J;
            b = cb;
            (void)num_seqs;
        } else {
            const int Gq = G >> gsh;
            int bid = blockIdx.x;
            kvh = bid % Hkv;
            bid /= Hkv;
            hg = bid % Gq;
            bid /= Gq;
            J = PAGED ? bid / num_seqs : bid;
            b = PAGED ? bid % num_seqs : (int)blockIdx.y;
        }
        hu = kvh + Hkv * hg;                                        // unit (kv head, head group)
        hq = kvh * G + hg * gp + (c & (gp - 1));         // this lane's query head
        // PAGED, no key split: the ids of the wave's blocks in key tiles 0..3 depend on nothing but the
        // sequence, so they are requested here, beside the context length, not behind it (one scalar
        // cache miss less in the prologue's chain length -> ids -> tiles)
        const bool prefetched = CHAIN && PAGED && pf_b == b;
        if constexpr (PAGED && !KSPLIT) {
            if (prefetched) {
#pragma unroll
                for (int i = 0; i < 4; ++i) early_id[i] = pf_id[i];
            } else {
#pragma unroll
                for (int i = 0; i < kRing + 1; ++i) {
                    int bi = 4 * i + wave;
                    bi = bi < max_blocks_per_seq ? bi : max_blocks_per_seq - 1;
                    early_id[i] = (unsigned)block_tables[(int64_t)b * max_blocks_per_seq + bi];
                }
            }
        }
        if constexpr (!PAGED) {
            Skv = ka.Skv;
            num_q_tiles = ka.num_q_tiles;
        }
        if constexpr (PAGED) {
            Skv = prefetched ? pf_ctx : context_lens[b];
            const int cap = max_blocks_per_seq * 16;
            Skv = Skv < 0 ? 0 : (Skv > cap ? cap : Skv);
            kv_off = Skv - Sq;
            pos_shift = kv_off & 31;
            pos_base = kv_off - pos_shift;
            num_q_tiles = (((Sq + pos_shift) << gsh) + kQTile - 1) / kQTile;
            if (J >= (num_q_tiles + 1) / 2) return false;
        }
        // Which two query tiles this workgroup owns.  Balanced pairing (J, NT-1-J) gives every
        // workgroup the same causal work -- the choice when the grid is about one workgroup per CU.
        // Adjacent pairing (2J, 2J+1), heaviest pairs first, keeps both blocks on the same key
        // range for almost the whole walk, so nearly every fragment read feeds two MFMAs -- the
        // choice when there are many workgroups per CU and the tail evens out by itself.
        if (CHAIN && flip) J = (num_q_tiles + 1) / 2 - 1 - J;
        tile_long = num_q_tiles - 1 - J, tile_short = J;
        if (pair_adjacent) {
            const int pairs = (num_q_tiles + 1) / 2;
            const int jr = CAUSAL ? pairs - 1 - J : J;
            tile_short = 2 * jr;
            tile_long = 2 * jr + 1 < num_q_tiles ? 2 * jr + 1 : 2 * jr;
        }
        const int nkv = (Skv + kKV - 1) / kKV;
        // r0 = first row of a block, counted from pos_base - kv_off (= row 0 unless PAGED)
        auto tiles_for = [&](int r0) {
            if ((r0 >> gsh) - pos_shift >= Sq) return 0;
            int n = nkv;
            if (CAUSAL) {
                const int last = ((r0 + 31) >> gsh) + pos_base;  // last position of the block
                if (last < 0) return 0;
                const int d = last / kKV + 1;
                n = d < n ? d : n;
            }
            return n;
        };
        int rowA0 = tile_long * kQTile + 32 * wave;
        int rowB0 = tile_short * kQTile + 32 * wave;
        nA = tiles_for(rowA0);
        nB = (tile_short != tile_long) ? tiles_for(rowB0) : 0;
        if (nB > nA) {  // block A is the one with the longer key range
            int x = rowA0; rowA0 = rowB0; rowB0 = x;
            x = nA; nA = nB; nB = x;
        }
        // key tiles the workgroup walks (same in every wave)
        nsteps = nkv;
        if (CAUSAL) {
            int last = ((tile_long * kQTile + kQTile - 1) >> gsh) + pos_base;
            last = last < Sq - 1 + kv_off ? last : Sq - 1 + kv_off;
            const int d = last < 0 ? 0 : last / kKV + 1;
            nsteps = d < nsteps ? d : nsteps;
        }
        if constexpr (KSPLIT) {
            const int parts = (int)gridDim.z, part = (int)blockIdx.z;
            tile0 = (int)((long long)part * nsteps / parts);
            const int k1 = (int)((long long)(part + 1) * nsteps / parts);
            nA = (nA < k1 ? nA : k1) - tile0;
            nB = (nB < k1 ? nB : k1) - tile0;
            nA = nA > 0 ? nA : 0;
            nB = nB > 0 ? nB : 0;
            nsteps = k1 - tile0;
        }
        // a block with rows to store may have no keys at all (PAGED, context shorter than the chunk)
        liveA = (rowA0 >> gsh) - pos_shift < Sq;
        liveB = tile_short != tile_long && (rowB0 >> gsh) - pos_shift < Sq;
        A.qrow = ((rowA0 + c) >> gsh) + pos_base;
        B.qrow = ((rowB0 + c) >> gsh) + pos_base;
        A.r0 = rowA0;
        B.r0 = rowB0;
        A.m = B.m = -INFINITY;
        A.mc = B.mc = 0.0f;
        A.thr = B.thr = -INFINITY;
        // (uni: these feed "s" operands of the DMA statements; CHAIN's arguments come out of LDS and
        // hipcc would otherwise carry what is computed from them in vector registers)
        auto uni = [](unsigned x) { return (unsigned)__builtin_amdgcn_readfirstlane((int)x); };
        auto uni64 = [&](unsigned long long x) {
            return (unsigned long long)uni((unsigned)x) | ((unsigned long long)uni((unsigned)(x >> 32)) << 32);
        };
        if constexpr (PAGED) {
            const unsigned head = (unsigned)kvh * (unsigned)(kD * 2);
            pg_kbase = uni64((unsigned long long)k + head);
            pg_vbase = uni64((unsigned long long)v + head);
            pg_nrec = uni(pg_block_bytes - head);
        } else {
            const unsigned kv_bytes = (unsigned)(Skv - 1) * row_bytes + (unsigned)(kD * 2);
            const int64_t head_off = ((int64_t)b * Skv * Hkv + kvh) * kD;
            const unsigned long long ka = (unsigned long long)(k + head_off), va = (unsigned long long)(v + head_off);
            dma_krs = u32x4{uni((unsigned)ka), uni((unsigned)(ka >> 32) & 0xffffu), uni(kv_bytes), 0x00020000u};
            dma_vrs = u32x4{uni((unsigned)va), uni((unsigned)(va >> 32) & 0xffffu), uni(kv_bytes), 0x00020000u};
        }
        return true;
    };
    // The item's Q rows, into qa / qb.  Issued AFTER the first tiles' requests: hipcc shuffles the
    // staging registers of a chained item (they are loop-carried) and waits for the loads where it
    // does, and the DMA requests must not stand behind that wait.
    auto load_q = [&](const KArgs &ka) {
        const int tid_f = CHAIN ? fresh(tid) : tid;
        const int h = (tid_f >> 5) & 1;
        // buffer loads over this batch element's Q rows: a row the block does not have gets an
        // offset past the buffer and reads as zero -- no branch around the loads (hipcc waits for
        // every outstanding load where such a branch joins)
        const unsigned seq_bytes = (unsigned)ka.Sq * (unsigned)ka.Hq * (unsigned)(kD * 2);
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
            (void *)(ka.q + (int64_t)b * ka.Sq * ka.Hq * kD), 0, (int)seq_bytes, 0x00020000);
        auto issue_q = [&](const Block &blk, bool live, u32x4 (&x)[8]) {
            const int row = blk.qrow - kv_off;
            const bool ok = live && row >= 0 && row < ka.Sq;
            const unsigned off = ok ? ((unsigned)row * (unsigned)ka.Hq + (unsigned)hq) * (unsigned)(kD * 2) + 16u * h
                                    : 0xffffff00u;
#pragma unroll
            for (int ks = 0; ks < 8; ++ks)
                x[ks] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(off + 32u * ks), 0, 0));
        };
        issue_q(A, liveA, qa);
        issue_q(B, liveB, qb);
    };
    auto prefetch_scalars = [&](int cb, const KArgs &ka) {
        if constexpr (CHAIN && PAGED) {
            pf_b = cb;
            pf_ctx = ka.context_lens[cb];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int bi = 4 * i + wave;
                bi = bi < ka.max_blocks_per_seq ? bi : ka.max_blocks_per_seq - 1;
                pf_id[i] = (unsigned)ka.block_tables[(int64_t)cb * ka.max_blocks_per_seq + bi];
            }
        }
    };
    // the accumulators of a block start at zero (O, row sums)
    auto zero_acc = [&](auto bt) {
        constexpr int bi = decltype(bt)::value;
        static_for<64>([&](auto R) { agpr_write<64 * bi + decltype(R)::value>(0u); });
        static_for<16>([&](auto R) { agpr_write<192 + 16 * bi + decltype(R)::value>(0u); });
    };
    auto commit_q = [&](auto bt, const u32x4 (&x)[8]) {
        constexpr int bi = decltype(bt)::value;
        static_for<32>([&](auto R) {
            constexpr int r = decltype(R)::value;
            agpr_write<128 + 32 * bi + r>(x[r >> 2][r & 3]);
        });
    };
    // PAGED: id of the block this wave fetches in key tile `tile` (clamped into the table row)
    auto dma_id = [&](int tile) -> unsigned {
        int bi = 4 * (tile + tile0) + wave;
        bi = bi < max_blocks_per_seq ? bi : max_blocks_per_seq - 1;
        return (unsigned)block_tables[(int64_t)b * max_blocks_per_seq + bi];
    };
    // PAGED: descriptor of this kv head's rows in block `id` of a cache (all scalar: the id came
    // through the scalar cache)
    auto pg_desc = [&](unsigned long long base, unsigned id) -> u32x4 {
        // (under register pressure hipcc parks a prefetched id in a vector register and would build
        // the descriptor there, which no "s" operand survives: the id is uniform, say so)
        id = (unsigned)__builtin_amdgcn_readfirstlane((int)id);
        const unsigned long long a = base + (unsigned long long)id * pg_block_bytes;
        return u32x4{(unsigned)a, (unsigned)(a >> 32) & 0xffffu, id < num_cache_blocks ? pg_nrec : 0u, 0x00020000u};
    };
    const unsigned lds_base = (unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned char *)lds;
    // settle = the scalar operands may have just come back from a spill lane (v_readlane: a VALU
    // write of an SGPR, which a buffer instruction may read only 5 wait states later; hipcc does not
    // look into the asm statement): one s_nop 4 between.  For the pieces issued outside the
    // hand-scheduled bodies, whose operands are long in place (scripts/k1_lab/check_asm.py audits both).
    // The four pieces of a request go behind ONE write of M0 (27 -> 13.5 cycles of issue per piece on
    // an idle queue, scripts/diag/ldsdma_issue.hip; -1.2 % dense, -1.4...-2.7 % over the cache): piece
    // j & 3 carries the immediate offset 1024 (j & 3), which moves the LDS address and the memory
    // address alike; the memory side is set right through the scalar offset (dense) or the lane offset
    // (paged).  The pieces of a request are always issued in order with no other request in between
    // (check_asm.py audits that too).
    auto dma_piece = [&](int t, int kslot, int vslot, int j, unsigned sid = 0u, bool settle = false) {
        const int jj = j & 3;
        const unsigned r0 = 16u * (unsigned)wave + 4u * (unsigned)jj;
        unsigned dst = lds_base + (j < 4 ? (unsigned)(kKBase + kslot * kKTile) : (unsigned)(vslot * kVTile)) + 16u * (unsigned)wave * 256u;
        if constexpr (PAGED) {
            u32x4 rs = pg_desc(j < 4 ? pg_kbase : pg_vbase, sid);
            if (settle) asm volatile("s_nop 4" : "+s"(dst));
            unsigned off;
            if (j < 4) off = dma_pk[jj] - 1024u * jj;
            else {
                const int rows_left = Skv - 16 * (4 * (t + tile0) + wave);
                const int r = 4 * jj + (lane >> 4);
                off = r < rows_left ? dma_pv[jj] - 1024u * jj : kPgOutOfBlock;
            }
#define K1_PG_PIECE(OFFS, NTS)                                                                                     \
    asm volatile("buffer_load_dwordx4 %0, %1, 0 offen" OFFS NTS " lds" : : "v"(off), "s"(rs) : "memory")
            if (jj == 0) {
                if constexpr (NT)
                    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen nt lds" : : "s"(dst), "v"(off), "s"(rs) : "memory");
                else
                    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" : : "s"(dst), "v"(off), "s"(rs) : "memory");
            } else if (jj == 1) {
                if constexpr (NT) K1_PG_PIECE(" offset:1024", " nt"); else K1_PG_PIECE(" offset:1024", "");
            } else if (jj == 2) {
                if constexpr (NT) K1_PG_PIECE(" offset:2048", " nt"); else K1_PG_PIECE(" offset:2048", "");
            } else {
                if constexpr (NT) K1_PG_PIECE(" offset:3072", " nt"); else K1_PG_PIECE(" offset:3072", "");
            }
#undef K1_PG_PIECE
        } else {
            u32x4 rs = j < 4 ? dma_krs : dma_vrs;
            unsigned soff = (unsigned)__builtin_amdgcn_readfirstlane((int)(((unsigned)t * kKV + r0) * row_bytes - 1024u * jj));   // (a row has >= 256 bytes)
            if (settle) asm volatile("s_nop 4" : "+s"(dst), "+s"(rs), "+s"(soff));
            unsigned so;
            const unsigned vo = j < 4 ? dma_kv[jj] : dma_vv;
            if (jj == 0)
                asm volatile("s_mov_b32 m0, %1\n\ts_mov_b32 %0, %4\n\tbuffer_load_dwordx4 %2, %3, %0 offen lds"
                             : "=&s"(so) : "s"(dst), "v"(vo), "s"(rs), "s"(soff) : "memory");
            else if (jj == 1)
                asm volatile("s_mov_b32 %0, %3\n\tbuffer_load_dwordx4 %1, %2, %0 offen offset:1024 lds" : "=&s"(so) : "v"(vo), "s"(rs), "s"(soff) : "memory");
            else if (jj == 2)
                asm volatile("s_mov_b32 %0, %3\n\tbuffer_load_dwordx4 %1, %2, %0 offen offset:2048 lds" : "=&s"(so) : "v"(vo), "s"(rs), "s"(soff) : "memory");
            else
                asm volatile("s_mov_b32 %0, %3\n\tbuffer_load_dwordx4 %1, %2, %0 offen offset:3072 lds" : "=&s"(so) : "v"(vo), "s"(rs), "s"(soff) : "memory");
        }
    };

    // ---- LDS read addresses
    // K A-fragment (u, ks): row 32u + c, 16-byte chunk 2ks + h, which sits at position (2ks + h) ^ (c & 15)
    unsigned k_rd8[8];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) k_rd8[ks] = (unsigned)(kKBase + c * kKRow + 16 * ((2 * ks + h) ^ (c & 15)));
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
        fr[i & 7] = *reinterpret_cast<const u32x4 *>(lds + k_rd8[i & 7] + slot * kKTile + (i >> 3) * 32 * kKRow);
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
    // the rescale threshold in raw score units (only the rare path needs it: computed there, so that
    // it does not hold a register -- CHAIN: or a scratch slot -- across the steps)
    auto defer_raw = [&]() {
        float sc = scale_log2e;
        asm volatile("" : "+s"(sc));
        return kDeferLog2 / sc;
    };
    // ---- vector-ALU steps of one item's softmax, cut so that they can be dealt out between MFMAs
    // running max, step j of 16: sa pairs first (their MFMAs finished long ago), then sb pairs
    // (two alternating chains -- it.mx on even steps, it.xf, idle until the first element, on odd
    // ones -- so that no v_max3 reads the result of the instruction just before it; step 15 joins them)
    auto max_step = [&](Item &it, int j) {
        const int r = 2 * (j & 7);
        float &acc = (j & 1) ? it.xf : it.mx;
        if (j < 2) acc = max2_asm(it.sa[r], it.sa[r + 1]);
        else if (j < 8) acc = max3_asm(acc, it.sa[r], it.sa[r + 1]);
        else acc = max3_asm(acc, it.sb[r], it.sb[r + 1]);
        if (j == 15) it.mx = max2_asm(it.mx, it.xf);
    };
    // The 32 elements of an item, p = exp2(s*c - m*c) packed to bf16 pairs (the row sum is an MFMA:
    // ones . P^T), as a 3-stage pipeline over 34 ticks so that no instruction waits for the one
    // just before it: tick n multiplies element n, exponentiates element n-1 and packs the pair
    // that ended at element n-2.
    auto el_pack = [&](Item &it, int e) {  // e = odd element that closes a pair
        it.p[e >> 3][(e & 7) >> 1] = pack_bf16x2(it.ya, it.yb);
    };
    auto el_exp = [&](Item &it, int e) {
        const float y = fast_exp2(it.xf);
        if (e & 1) it.yb = y;
        else it.ya = y;
    };
    auto el_fma = [&](Item &it, const Block &blk, int e) {
        const float sv = e < 16 ? it.sa[e & 15] : it.sb[e & 15];
        it.xf = __builtin_fmaf(sv, scale_log2e, -blk.mc);
    };
    auto tick = [&](Item &it, const Block &blk, int n) {
        if (n >= 3 && (n & 1)) el_pack(it, n - 2);
        if (n >= 1 && n <= 32) el_exp(it, n - 1);
        if (n < 32) el_fma(it, blk, n);
    };
    constexpr int kTicks = kTicksTotal, kFirst = kFirstTicks;
    auto deps_of = [&](const Item &it) {
        return Deps{it.mx, it.mx, it.xf, it.ya, it.yb, it.p[0], it.p[1], it.p[2], it.p[3]};
    };
    // Deferred rescale (T13).  The old reference max of a block is kept while every row max of the
    // wave stays within kDeferLog2 (exp2 domain) of it: P <= 2^kDeferLog2.  That is the common case,
    // so the element pipeline of a fresh item starts against the old reference before anyone has
    // looked, and the check is ONE compare (each lane tests the max of its half of the row's keys
    // against the row's threshold; the halves only meet when the branch is taken) and ONE branch per
    // step for both blocks -- a branch costs this kernel 75-150 cycles (instruction fetch restarts
    // with nothing to hide it at one wave per SIMD), so there is exactly one in a step.  It sits
    // behind the last MFMA of the P.V slot: everything in O and l is then at the old reference and
    // the only pending values are the item's own speculative elements, so the rare path rescales
    // O and l on the spot (register by register through the vector ALU) and redoes those elements
    // against the new reference -- every value still at the old max is scaled exactly once.
    auto rereference = [&](Item &it, Block &blk, auto bt, int ticks_done) {
        if (!__any(it.mx > blk.thr)) return;  // (inside the rare branch: which of the blocks it was)
        constexpr int bi = decltype(bt)::value;
        const float mx = half_max(it.mx);
        const float m_new = fmaxf(blk.m, mx);
        const float m_safe = (m_new == -INFINITY) ? 0.0f : m_new;
        const float alpha = fast_exp2((blk.m - m_safe) * scale_log2e);
        blk.m = m_new;
        blk.mc = m_safe * scale_log2e;
        blk.thr = m_new + defer_raw();
        mfma_fence();
        static_for<64>([&](auto R) { agpr_scale<64 * bi + decltype(R)::value>(alpha); });
        static_for<16>([&](auto R) { agpr_scale<192 + 16 * bi + decltype(R)::value>(alpha); });
        mfma_fence();
#pragma unroll
        for (int n = 0; n < kTicks; ++n)
            if (n < ticks_done) tick(it, blk, n);
    };
#ifdef K1_ABL_NOBRANCH
    int abl_sticky = 0;
#endif
    auto decide_after_pv = [&](Item &it, Block &blk, auto bt) {
#ifdef K1_ABL_NOBRANCH  // lab only: the compare without the branch (wrong results, prices the branch)
        abl_sticky |= __any(it.mx > blk.thr);
        return;
#endif
        if (__builtin_expect(__any(it.mx > blk.thr), 0)) rereference(it, blk, bt, kFirst);
    };
    auto decide_after_pv2 = [&](Item &ia, Item &ib) {
#ifdef K1_ABL_NOBRANCH
        abl_sticky |= __any((ia.mx > A.thr) | (ib.mx > B.thr));
        return;
#endif
        if (__builtin_expect(__any((ia.mx > A.thr) | (ib.mx > B.thr)), 0)) {
            rereference(ia, A, kA, kFirst);
            rereference(ib, B, kB, kFirst);
        }
    };
    // VALU share of slice i (of 20) of a P.V slot: statistics of the item whose QK^T just
    // finished, then its first kFirst elements
    // unit u of an item's softmax (kUnitTab)
    auto unit = [&](Item &it, Block &blk, auto U) {
        constexpr int u = decltype(U)::value, kind = kUnitTab.kind[u], arg = kUnitTab.arg[u];
#ifdef K1_ABL_SKIP_KINDS  // lab only: drop unit kinds (bit mask) to price them; results are wrong
        if constexpr ((K1_ABL_SKIP_KINDS >> kind) & 1) return;
#endif
        if constexpr (kind == kUMax) max_step(it, arg);
        else if constexpr (kind == kUPack) el_pack(it, arg);
        else if constexpr (kind == kUExp) el_exp(it, arg);
        else el_fma(it, blk, arg);
    };
    // the units make_deal_plan(MODE) gives slice I of a P.V slot (of 20) / of a QK^T slot (of 16)
    auto valu_pv = [&](Item &it, Block &blk, auto M, auto I) {
#ifdef K1_ABL_NOVALU
        return;
#endif
        constexpr DealPlan plan = make_deal_plan(decltype(M)::value);
        constexpr int i = decltype(I)::value, u0 = plan.cum[i], u1 = plan.cum[i + 1];
        static_for<u1 - u0>([&](auto J) { unit(it, blk, ic<u0 + decltype(J)::value>{}); });
    };
    auto valu_qk = [&](Item &it, Block &blk, auto M, auto I) {
#ifdef K1_ABL_NOVALU
        return;
#endif
        constexpr DealPlan plan = make_deal_plan(decltype(M)::value);
        constexpr int i = decltype(I)::value, u0 = plan.cum[kPvSlices + i], u1 = plan.cum[kPvSlices + i + 1];
        static_for<u1 - u0>([&](auto J) { unit(it, blk, ic<u0 + decltype(J)::value>{}); });
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
            valu(ic<i>{});
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
            valu(ic<i>{});
            __builtin_amdgcn_sched_barrier(0);
        });
    };
    auto no_rd = [&](int) {};
    auto no_valu = [&](auto) {};

    // ---- the same two slots for BOTH blocks on one key tile: every K / V^T fragment read from
    // LDS feeds two MFMAs (block A, block B), which is what brings a slice's LDS + VALU issue
    // under the MFMA's 32 cycles.  Slice k works on block k & 1; the MFMA of slice k takes the
    // softmax values of slice k - 1 as its dependency operands.
    auto slot_qk2 = [&](Item &dA, Item &dB, int kslot, Item &wA, Item &wB, auto &&valu,
                        auto &&next_rd) {
        static_for<32>([&](auto K) {
            constexpr int k = decltype(K)::value;
            constexpr int f = k >> 1;           // fragment 0..15 (u = f>>3, ks = f&7)
            constexpr int bi = k & 1;
            constexpr int qr = 128 + 32 * bi + 4 * (f & 7);
            constexpr int g0 = (f & 4);
            Item &dst = bi ? dB : dA;
            const Deps d = deps_of(bi ? wA : wB);
            if constexpr ((k & 7) == 0)
                mfma_qk_g<qr, (f & 7) == 0>(f < 8 ? dst.sa : dst.sb, fr[g0], fr[g0 + 1], fr[g0 + 2],
                                            fr[g0 + 3], d);
            else
                mfma_qk<qr, (f & 7) == 0>(f < 8 ? dst.sa : dst.sb, fr[f & 7], d);
            if constexpr ((k & 7) < 4) {  // one fragment of the next group per slice
                constexpr int j = (f & ~3) + 4 + (k & 7);
                if constexpr (j < 16) read_k(kslot, j);
                else next_rd(j - 16);
            }
            valu(ic<k>{});
            if constexpr (k == 31) mfma_fence();
            __builtin_amdgcn_sched_barrier(0);
            if constexpr ((k & 7) == 7) K1_FSTAMP(k >> 3);
        });
    };
    auto slot_pv2 = [&](const Item &sA, const Item &sB, int vslot, Item &wA, Item &wB, auto &&valu,
                        auto &&next_rd) {
        static_for<40>([&](auto K) {
            constexpr int k = decltype(K)::value;
            const Deps d = deps_of((k & 1) ? wA : wB);
            if constexpr (k < 32) {
                constexpr int f = k >> 1;  // fragment 0..15 (dt = f>>2, sp = f&3)
                constexpr int bi = k & 1;
                constexpr int g0 = (f & 4);
                constexpr int ar = 64 * bi + 16 * (f >> 2);
                const Item &src = bi ? sB : sA;
                if constexpr ((k & 7) == 0)
                    mfma_pv_g<ar>(fr[g0], src.p[0], fr[g0 + 1], fr[g0 + 2], fr[g0 + 3], d);
                else
                    mfma_pv<ar>(fr[f & 7], src.p[f & 3], d);
                if constexpr ((k & 7) < 4) {
                    constexpr int j = (f & ~3) + 4 + (k & 7);
                    if constexpr (j < 16) read_v(vslot, j);
                    else next_rd(j - 16);
                }
            } else {
                constexpr int bi = (k >> 2) & 1;  // 32..35 block A, 36..39 block B
                mfma_pv<192 + 16 * bi>(ones, (bi ? sB : sA).p[k & 3], d);
            }
            valu(ic<k>{});
            __builtin_amdgcn_sched_barrier(0);
            if constexpr ((k & 7) == 7) K1_FSTAMP(4 + (k >> 3));
        });
    };

    // ---- un-pipelined pieces (pipeline drain) and the mask of a block's last key tile
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
        const int hh = CHAIN ? (fresh(tid) >> 5) & 1 : h;  // (CHAIN: keeps the 16 key offsets out of the item loop's live set)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = kv0 + (r & 3) + 8 * (r >> 2) + 4 * hh;
            const bool va = (key < Skv) && (!CAUSAL || key <= blk.qrow);
            const bool vb = (key + 32 < Skv) && (!CAUSAL || key + 32 <= blk.qrow);
            it.sa[r] = va ? it.sa[r] : -INFINITY;
            it.sb[r] = vb ? it.sb[r] : -INFINITY;
        }
    };
    // finish a pending item (its first kFirst ticks are done and its reference is settled: the step
    // that produced it ended with decide_after_pv), then its P.V
    auto drain = [&](Item &it, Block &blk, auto bt, int vslot) {
        mfma_fence();
#pragma unroll
        for (int n = kFirst; n < kTicks; ++n) tick(it, blk, n);
        pv_plain(it, bt, vslot);
    };
    // ---- prologue of an item, first half: K(0) -> K slot 0, K(1) -> K slot 1, V(0) -> V slot 0, all
    // requested together (PAGED: through the ids of the wave's blocks in tiles 0..3).  The rings must
    // be free: nobody reads them any more (CHAIN: behind the barrier that ends the item before).
    auto request_first_tiles = [&]() {
        unsigned id0 = 0;
        if constexpr (PAGED && !KSPLIT) {
            id0 = early_id[0];
            pg_s0 = early_id[1];
            pg_s1 = early_id[2];
            pg_s2 = early_id[3];
            if constexpr (kRing == 4) pg_s3 = early_id[4];
        } else if constexpr (PAGED) {
            id0 = dma_id(0);
            pg_s0 = dma_id(1);
            pg_s1 = dma_id(2);
            pg_s2 = dma_id(3);
            if constexpr (kRing == 4) pg_s3 = dma_id(4);
        }
        // K tiles 0 .. kKAhead - 1 and V tiles 0 .. kVAhead - 1, tile i into slot i, oldest first
#pragma unroll
        for (int j = 0; j < 4; ++j) dma_piece(0, 0, 0, j, id0, true);
#pragma unroll
        for (int j = 0; j < 4; ++j) dma_piece(1, 1, 0, j, pg_s0, true);
#pragma unroll
        for (int j = 4; j < 8; ++j) dma_piece(0, 0, 0, j, id0, true);
        if constexpr (kRing == 4) {
#pragma unroll
            for (int j = 0; j < 4; ++j) dma_piece(2, 2, 0, j, pg_s1, true);
#pragma unroll
            for (int j = 4; j < 8; ++j) dma_piece(1, 0, 1, j, pg_s0, true);
        }
    };
    // ... second half: accumulators zeroed, Q^T written, the last V slot ("tile -1") zeroed, tiles landed
    auto finish_prologue = [&]() {
        zero_acc(kA);
        zero_acc(kB);
        commit_q(kA, qa);
        commit_q(kB, qb);
        if constexpr (CHAIN) {
            // the staging registers are dead from here to the next item's set-up; hipcc cannot see
            // that (the set-up sits behind a loop condition) and would carry them through the steps
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("" : "=v"(qa[i]), "=v"(qb[i]));
        }
        const u32x4 z = {0u, 0u, 0u, 0u};
        const int tid_f = CHAIN ? fresh(tid) : tid;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            *reinterpret_cast<u32x4 *>(lds + (kRing - 1) * kVTile + (i * 256 + tid_f) * 16) = z;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    };
    Item I0, I1;    // block A's two score register sets
    Item J0, J1;    // block B's
    // "nothing pending" = an item whose P is all zero, multiplied into a zeroed V slot
    auto clear_item = [&](Item &it) {
#pragma unroll
        for (int r = 0; r < 16; ++r) { it.sa[r] = -INFINITY; it.sb[r] = -INFINITY; }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            it.p[i] = u32x4{0u, 0u, 0u, 0u};
            // opaque: a constant that hipcc re-creates with v_mov right in front of an MFMA statement
            // would be read before it has landed (VALU write -> MFMA operand hazard)
            asm volatile("" : "+v"(it.p[i]));
        }
        it.mx = -INFINITY;
        it.xf = -INFINITY; it.ya = 0.0f; it.yb = 0.0f;
    };

    // ring slots: tile i of an item lives in slot i mod kRing of its ring
    int vs_prev = kRing - 1;  // V slot of tile t-1 (read by this step's P.V)
    int vs_tgt = kVAhead;     // V slot of tile t + kVAhead (requested in this step)
    int ks_cur = 0;           // K slot of tile t
    int ks_tgt = kKAhead;     // K slot of tile t + kKAhead (requested in this step)
    auto next_slot = [](int x) { return x + 1 == kRing ? 0 : x + 1; };
    // pieces that may stay in flight over a step's barrier: everything requested after K(t+1) and
    // after V(t), which the next step reads (a step issues its 4 K pieces before its 4 V pieces)
    constexpr int kStepWait = (4 + 8 * (kKAhead - 1)) < 8 * kVAhead ? (4 + 8 * (kKAhead - 1)) : 8 * kVAhead;
    static_assert(kStepWait == (kRing == 3 ? 8 : 16), "counted wait");
    K1_STAMP(2);  // kernel start .. first step (Q load, first two tiles)
    // Every wave runs the same number of steps (nsteps), one barrier each, but walks them in its
    // own phases -- separate loops, so that each hot loop is one straight body whose registers
    // (scores, staging) have one definition per iteration.
    int t = 0;
    // end of a step: publish tile t+1 (written during this step), rotate the rings
    auto step_end = [&]() {
        // the wave's pieces of this step (and, rings of 4, of the step before) may stay in flight;
        // K of t+1 and V of t, read in the next step, must have landed; hipcc does not count
        // LDS-DMA issued from asm, so __syncthreads() below is a bare s_barrier
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kStepWait) : "memory");
        __syncthreads();
        K1_STAMP(11);  // barrier
#ifdef NVLLM_K1_STAMPS
        seg_sum[15] += 1;
#endif
        vs_prev = next_slot(vs_prev);
        vs_tgt = next_slot(vs_tgt);
        ks_cur = next_slot(ks_cur);
        ks_tgt = next_slot(ks_tgt);
        ++t;
        if constexpr (PAGED) {
            pg_s0 = pg_s1;
            pg_s1 = pg_s2;
            if constexpr (kRing == 4) {
                pg_s2 = pg_s3;
                pg_s3 = dma_id(t + 4);
            } else {
                pg_s2 = dma_id(t + 3);
            }
        }
    };
    // PAGED: ids of the blocks this step's requests go to (pg_s0 = tile t+1, ...)
    auto k_id = [&]() { return kKAhead == 3 ? pg_s2 : pg_s1; };
    auto v_id = [&]() { return kVAhead == 2 ? pg_s1 : pg_s0; };
    // staging of a step, all at once (steps this wave has no keys for): K of tile t + kKAhead, V of tile t + kVAhead
    auto stage_step = [&]() {
#pragma unroll
        for (int j = 0; j < 4; ++j) dma_piece(t + kKAhead, ks_tgt, 0, j, k_id());
#pragma unroll
        for (int j = 4; j < 8; ++j) dma_piece(t + kVAhead, 0, vs_tgt, j, v_id());
    };
    // ... or dealt out over the MFMA gaps of the pipelined bodies, into slices that issue no
    // fragment reads: the K pieces in the first four staging positions of the QK^T slot, the V
    // pieces in the first four of the P.V slot
#ifdef K1_ABL_NOSTAGE
    auto st_pos = [&](int, int) {};
#else
    auto st_pos = [&](int p, int base) {
        if (p < 4) {
            if (base == 0) dma_piece(t + kKAhead, ks_tgt, 0, p, k_id());
            else dma_piece(t + kVAhead, 0, vs_tgt, 4 + p, v_id());
        }
    };
#endif
    auto st_qk8 = [&](int i) { if ((i & 2) && i < 16) st_pos((i >> 2) * 2 + (i & 1), 0); };
    auto st_pv8 = [&](int i) { if ((i & 2) && i < 16) st_pos((i >> 2) * 2 + (i & 1), 4); };
    auto prime_k = [&]() {
#pragma unroll
        for (int j = 0; j < 4; ++j) read_k(ks_cur, j);
        __builtin_amdgcn_sched_barrier(0);
    };
    // block A alone: QK^T of tile t into `nxt` while `cur` (pending, tile t-1) is finished and
    // multiplied.  MASKED (A's last tile): the fresh scores are masked between the two slots, the
    // rest of the step is the same, and the caller drains `nxt` afterwards.
    auto single_step = [&](Item &cur, Item &nxt, auto masked) {
        prime_k();
        slot_qk(nxt, kA, ks_cur, cur,
                [&](auto I) {
                    valu_qk(cur, A, ic<kDealSingle>{}, I);
                    st_qk8(decltype(I)::value);
                },
                [&](int j) { read_v(vs_prev, j); });
        K1_STAMP(5);
        if constexpr (decltype(masked)::value) mask(nxt, A, (t + tile0) * kKV);
        slot_pv(cur, kA, vs_prev, nxt,
                [&](auto I) {
                    valu_pv(nxt, A, ic<kDealSingle>{}, I);
                    st_pv8(decltype(I)::value);
                },
                no_rd);
        decide_after_pv(nxt, A, kA);
        K1_STAMP(6);
#ifdef NVLLM_K1_STAMPS
        seg_sum[13] += 1;
#endif
        step_end();
    };

    // both blocks: QK^T of tile t into (nA_, nB_) while the pending items (pA, pB: tile t-1) are
    // finished and multiplied.  MASKED = block B's last tile (and block A's too when mask_a).
    auto dual_step = [&](Item &pA, Item &pB, Item &nA_, Item &nB_, auto masked, bool mask_a) {
        prime_k();
        slot_qk2(nA_, nB_, ks_cur, pA, pB,
                 [&](auto K) {
                     constexpr int k = decltype(K)::value;
                     if constexpr (k & 1) valu_qk(pB, B, ic<kDealDualB>{}, ic<(k >> 1)>{});
                     else valu_qk(pA, A, ic<kDealDualA>{}, ic<(k >> 1)>{});
#ifndef K1_ABL_NOSTAGE
                     if constexpr ((k & 3) == 3 && (k >> 2) < 4) dma_piece(t + kKAhead, ks_tgt, 0, k >> 2, k_id());
#endif
                 },
                 [&](int j) { read_v(vs_prev, j); });
        K1_STAMP(0);
        if constexpr (decltype(masked)::value) {
            mask(nB_, B, (t + tile0) * kKV);
            if (mask_a) mask(nA_, A, (t + tile0) * kKV);
        }
        slot_pv2(pA, pB, vs_prev, nA_, nB_,
                 [&](auto K) {
                     constexpr int k = decltype(K)::value;
                     if constexpr (k & 1) valu_pv(nB_, B, ic<kDealDualB>{}, ic<(k >> 1)>{});
                     else valu_pv(nA_, A, ic<kDealDualA>{}, ic<(k >> 1)>{});
#ifndef K1_ABL_NOSTAGE
                     if constexpr ((k & 3) == 3 && (k >> 2) < 4) dma_piece(t + kVAhead, 0, vs_tgt, 4 + (k >> 2), v_id());
#endif
                 },
                 no_rd);
        decide_after_pv2(nA_, nB_);
        K1_STAMP(1);
#ifdef NVLLM_K1_STAMPS
        seg_sum[12] += 1;
#endif
        step_end();
    };
    constexpr std::false_type kInterior{};
    constexpr std::true_type kLastTile{};

    // ---- epilogue: O^T lane (c,h) holds O[qrow][32dt + 8i + 4h + (0..3)] in regs 4i..4i+3.
    // What it needs of its item is saved in an Epi first: CHAIN takes up the next item before it runs.
    struct Epi {
        Block A, B;
        int kv_off, pos_base, b, kvh, hg, hq, J, hu, tile_long, tile_short;
        bool liveA, liveB;
    };
#ifdef NVLLM_K1_STAMPS
    float *lse_dbg = lse;
    lse = nullptr;
#endif
    auto store = [&](const Epi &e, const Block &blk, auto bt, bool live, const KArgs &ka) {
        constexpr int bi = decltype(bt)::value;
        const int tid_f = CHAIN ? fresh(tid) : tid;
        const int lane = tid_f & 63, c = lane & 31, h = lane >> 5;  // (shadow the kernel's: not hoistable)
        const int Sq = ka.Sq, Hq = ka.Hq, Hkv = ka.Hkv, num_seqs = ka.num_seqs;
        const int gsh = PAGED ? ka.pack_shift : 0, gp = 1 << gsh, G = Hq / Hkv, Gq = G >> gsh;
        __bf16 *o = ka.o;
        unsigned char *split_ws = ka.split_ws;
        (void)num_seqs; (void)Gq; (void)split_ws;
        float l = agpr_read<192 + 16 * bi>();
#ifdef K1_DBG_VALU_L
        if (bi == 0 && e.tile_long == e.tile_short) l = half_sum(dbg_l);
#endif
        // four accumulator registers at a time, as they are packed: the whole block at once would
        // need 64 registers beside the next item's Q rows (CHAIN)
        auto acc4 = [&](auto DT, auto I, float (&a4)[4]) {
            constexpr int r0 = 64 * bi + 16 * decltype(DT)::value + 4 * decltype(I)::value;
            a4[0] = agpr_read<r0>();
            a4[1] = agpr_read<r0 + 1>();
            a4[2] = agpr_read<r0 + 2>();
            a4[3] = agpr_read<r0 + 3>();
        };
        const int row = blk.qrow - e.kv_off;
        const float inv = l > 0.0f ? 1.0f / l : 0.0f;
        if constexpr (KSPLIT) {
            // this part's share of the rows: normalised O in bf16, then (m in the exp2 domain, l);
            // a block without keys in the part carries l = 0 and drops out of the merge
            if (live) {
                const int vr = blk.r0 + c;  // (packed) row inside the sequence's tiling
                const int tile = vr / kQTile;
                const int ur = ((tile == e.tile_long && e.tile_long != e.tile_short) ? kQTile : 0) + (vr & (kQTile - 1));
                const int Hu = Hkv * Gq;
                const size_t pairs_max = gridDim.x / (unsigned)(Hu * num_seqs);
                unsigned char *slot = split_ws + ((((size_t)e.b * Hu + e.hu) * pairs_max + e.J) * gridDim.z + blockIdx.z) *
                                                     (size_t)kSplitSlotBytes;
                __bf16 *op = reinterpret_cast<__bf16 *>(slot) + ur * kD + 4 * h;
                static_for<16>([&](auto X) {
                    constexpr int dt = decltype(X)::value >> 2, i = decltype(X)::value & 3;
                    float a4[4];
                    acc4(ic<dt>{}, ic<i>{}, a4);
                    u32x2 w;
                    w[0] = pack_bf16x2(a4[0] * inv, a4[1] * inv);
                    w[1] = pack_bf16x2(a4[2] * inv, a4[3] * inv);
                    *reinterpret_cast<u32x2 *>(op + 32 * dt + 8 * i) = w;
                });
                if (h == 0) {
                    float *ml = reinterpret_cast<float *>(slot + 2 * kQTile * kD * 2) + 2 * ur;
                    ml[0] = blk.m * scale_log2e;
                    ml[1] = l;
                }
            }
        } else {
            // O goes through a wave-private LDS image (behind the K/V rings, which may already be
            // taking the next item's tiles; block B reuses block A's image, a wave's LDS accesses
            // keep their order) and leaves as whole 256-byte rows, four per store instruction:
            // stored straight from the accumulator layout every instruction would touch 32 rows,
            // 16 bytes each
            unsigned char *stg = lds + kOStageBase + wave * kOStage;
            static_for<16>([&](auto X) {
                constexpr int dt = decltype(X)::value >> 2, i = decltype(X)::value & 3;
                float a4[4];
                acc4(ic<dt>{}, ic<i>{}, a4);
                u32x2 w;
                w[0] = pack_bf16x2(a4[0] * inv, a4[1] * inv);
                w[1] = pack_bf16x2(a4[2] * inv, a4[3] * inv);
                *reinterpret_cast<u32x2 *>(stg + c * kORow + (32 * dt + 8 * i + 4 * h) * 2) = w;
            });
            const int head0 = e.kvh * G + e.hg * gp;
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const int ri = 4 * r + (lane >> 4);  // row of the block: position and head of the group
                const u32x4 x = *reinterpret_cast<const u32x4 *>(stg + ri * kORow + (lane & 15) * 16);
                const int orow = ((blk.r0 + ri) >> gsh) + e.pos_base - e.kv_off;
                const int ohead = head0 + (ri & (gp - 1));
                if (live && orow >= 0 && orow < Sq)
                    *reinterpret_cast<u32x4 *>(o + (((int64_t)e.b * Sq + orow) * Hq + ohead) * kD + (lane & 15) * 8) = x;
            }
        }
        if (!KSPLIT && live && row >= 0 && row < Sq) {
            if (lse && h == 0)
                lse[((int64_t)e.b * Hq + e.hq) * Sq + row] =
                    blk.m * scale_log2e * 0.6931471805599453f + logf(l);
        }
    };

    // ---- the items of this workgroup
    int item = ch_b, nth = 0;  // CHAIN: batch element of the current item, its number in the chain
    {
        const KArgs ka = args();
        bool have = false;
        if constexpr (CHAIN) {
            while (item < ka.chain_items && !(have = setup(item, !PAGED && (nth & 1), ka))) {
                item += ka.chain_gx;
                ++nth;
            }
        } else {
            have = setup(0, false, ka);
        }
        if (!have) return;
        request_first_tiles();
        load_q(ka);
    }
    for (;;) {
        finish_prologue();
        clear_item(I0);
        clear_item(I1);
        clear_item(J0);
        clear_item(J1);
        vs_prev = kRing - 1, vs_tgt = kVAhead;
        ks_cur = 0, ks_tgt = kKAhead;
        t = 0;
        K1_STAMP(2);  // kernel start (or the epilogue's end) .. first step (Q load, first two tiles)
        // ---- phase 1: both blocks, alternating the two score register sets.  The (empty) pending items
        // start in whichever set lets block B's last tile go from set 0 to set 1.  That tile is a
        // pipelined step like the others, only masked; its item is then finished on its own (drain)
        // during the following step, while its V tile is still in the ring.
        if (nB > 0) {
            if ((nB - 1) & 1) dual_step(I1, J1, I0, J0, kInterior, false);
            while (t < nB - 1) {
                dual_step(I0, J0, I1, J1, kInterior, false);
                dual_step(I1, J1, I0, J0, kInterior, false);
            }
            const bool a_too = (t == nA - 1);
            dual_step(I0, J0, I1, J1, kLastTile, a_too);
            drain(J1, B, kB, vs_prev);
            if (a_too) drain(I1, A, kA, vs_prev);
            K1_STAMP(7);
        }
        // ---- phase 2: block A alone; its pending item is in set 1, the last tile goes from set 0 to
        // set 1, so an even number of interior tiles in between needs the two sets exchanged (only
        // happens when a key split cuts the walks; with no block B both sets are empty anyway)
        if (t < nA) {
            const int n2 = nA - 1 - t;
            if (!(n2 & 1) && nB > 0) I0 = I1;  // set 0's old content is dead: its P.V is done
            if (n2 & 1) single_step(I1, I0, kInterior);
            while (t < nA - 1) {
                single_step(I0, I1, kInterior);
                single_step(I1, I0, kInterior);
            }
            single_step(I0, I1, kLastTile);
            drain(I1, A, kA, vs_prev);
            K1_STAMP(8);
        }
        // ---- steps this wave has no keys for (the workgroup's other waves do)
        while (t < nsteps) {
            stage_step();
            K1_STAMP(9);
            step_end();
        }
        // a drain after the last step still reads the V ring, and the pieces requested past the last
        // tile must have landed before the rings are filled again
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
#ifdef K1_ABL_NOBRANCH
        asm volatile("" ::"s"(abl_sticky));
#endif
        mfma_fence();
        K1_STAMP(10);  // last wait + barrier
        const Epi e = {A, B, kv_off, pos_base, b, kvh, hg, hq, J, hu, tile_long, tile_short, liveA, liveB};
        // CHAIN: the rings are free from here on (nobody reads them after the barrier above), so the
        // next item is taken up between the two halves of the epilogue: its scalars are requested,
        // block A's O goes out, its Q rows and first tiles are requested, block B's O goes out -- each
        // request's latency behind a store's worth of work
        bool more = false;
        const KArgs ka = args();
        if constexpr (CHAIN) {
            item += ka.chain_gx;
            ++nth;
            if (item < ka.chain_items) prefetch_scalars(item, ka);
        }
        store(e, e.A, kA, e.liveA, ka);
#ifdef K1_STAMP_SETUP  // lab: store A -> 4, set-up -> 14, requests -> 1 (only meaningful where no dual step runs)
        K1_STAMP(4);
#endif
        if constexpr (CHAIN) {
            while (item < ka.chain_items && !(more = setup(item, !PAGED && (nth & 1), ka))) {
                item += ka.chain_gx;
                ++nth;
            }
#ifdef K1_STAMP_SETUP
            K1_STAMP(14);
#endif
            if (more) {
                request_first_tiles();
                load_q(ka);
            }
#ifdef K1_STAMP_SETUP
            K1_STAMP(1);
#endif
        }
#if !defined(K1_RT) && !defined(K1_STAMP_SETUP)
        K1_STAMP(4);  // block A stored; next item taken up: geometry, Q loads and first tiles requested
#endif
        store(e, e.B, kB, e.liveB, ka);
        K1_STAMP(3);  // epilogue
        if (!more) break;
    }
#ifdef NVLLM_K1_STAMPS
#ifdef K1_RT
    seg_sum[4] = rt0_;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(seg_sum[9])::"memory");
#endif
    if (lse_dbg && lane == 0) {
        unsigned long long *dbg = reinterpret_cast<unsigned long long *>(lse_dbg) +
                                  ((((size_t)blockIdx.y * gridDim.x + blockIdx.x) * kWaves + wave) * 16);
        for (int i = 0; i < 16; ++i) dbg[i] = seg_sum[i];
    }
#endif
}

// Combines the P partials of every row (KSPLIT): grid = (64, pairs_max * Hu, num_seqs) with
// Hu = Hq >> pack_shift units (kv head, head group); 4 rows of a unit per workgroup, one wave per
// row.  The wave's four 16-lane groups take every fourth part (at most kMaxParts = 16 parts, so
// four independent 16-byte loads per lane) and meet through two shuffles.  Tile geometry and row
// packing as in the kernel above.
constexpr int kMaxParts = 16;
__global__ __launch_bounds__(256) void flash_attn_ksplit_merge_kernel(
    const unsigned char *__restrict__ ws, __bf16 *__restrict__ o,
    const int32_t *__restrict__ context_lens, int Sq, int Hq, int Hkv, int max_blocks_per_seq,
    int pair_adjacent, int parts, int pack_shift) {
    const int b = blockIdx.z;
    const int gsh = pack_shift, gp = 1 << gsh;
    const int Hu = Hq >> gsh;
    const int hu = blockIdx.y % Hu, J = blockIdx.y / Hu;
    const int pairs_max = gridDim.y / Hu;
    int Skv = context_lens[b];
    const int cap = max_blocks_per_seq * 16;
    Skv = Skv < 0 ? 0 : (Skv > cap ? cap : Skv);
    const int kv_off = Skv - Sq;
    const int pos_shift = kv_off & 31;
    const int num_q_tiles = (((Sq + pos_shift) << gsh) + kQTile - 1) / kQTile;
    const int pairs = (num_q_tiles + 1) / 2;
    if (J >= pairs) return;
    int tile_long = num_q_tiles - 1 - J, tile_short = J;
    if (pair_adjacent) {
        const int jr = pairs - 1 - J;
        tile_short = 2 * jr;
        tile_long = 2 * jr + 1 < num_q_tiles ? 2 * jr + 1 : 2 * jr;
    }
    const int lane = threadIdx.x & 63;
    const int ur = (blockIdx.x << 2) + (threadIdx.x >> 6);  // one wave per row
    const int grp = lane >> 4, ch = lane & 15;
    if (ur >= kQTile && tile_long == tile_short) return;
    const int rho = (ur >= kQTile ? tile_long : tile_short) * kQTile + (ur & (kQTile - 1));  // packed row
    const int row = (rho >> gsh) - pos_shift;
    if (row < 0 || row >= Sq) return;
    const int hq = (hu % Hkv) * (Hq / Hkv) + (hu / Hkv) * gp + (rho & (gp - 1));
    const unsigned char *slot0 = ws + (((size_t)b * Hu + hu) * pairs_max + J) * parts * (size_t)kSplitSlotBytes;
    float m[kMaxParts / 4], l[kMaxParts / 4];
    u32x4 x[kMaxParts / 4];
    float mmax = -INFINITY;
#pragma unroll
    for (int i = 0; i < kMaxParts / 4; ++i) {
        const int p = grp + 4 * i;
        m[i] = -INFINITY;
        l[i] = 0.0f;
        x[i] = u32x4{0u, 0u, 0u, 0u};
        if (p < parts) {
            const unsigned char *slot = slot0 + (size_t)p * kSplitSlotBytes;
            const float *ml = reinterpret_cast<const float *>(slot + 2 * kQTile * kD * 2) + 2 * ur;
            m[i] = ml[0];
            l[i] = ml[1];
            x[i] = *reinterpret_cast<const u32x4 *>(slot + ((size_t)ur * kD + 8 * ch) * 2);
            if (l[i] > 0.0f) mmax = fmaxf(mmax, m[i]);
        }
    }
    mmax = fmaxf(mmax, __shfl_xor(mmax, 16));
    mmax = fmaxf(mmax, __shfl_xor(mmax, 32));
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float den = 0.0f;
#pragma unroll
    for (int i = 0; i < kMaxParts / 4; ++i) {
        if (!(l[i] > 0.0f)) continue;
        const float w = l[i] * fast_exp2(m[i] - mmax);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            acc[2 * j] += w * __builtin_bit_cast(float, x[i][j] << 16);
            acc[2 * j + 1] += w * __builtin_bit_cast(float, x[i][j] & 0xffff0000u);
        }
        den += w;
    }
#pragma unroll
    for (int off = 16; off <= 32; off <<= 1) {
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] += __shfl_xor(acc[j], off);
        den += __shfl_xor(den, off);
    }
    if (grp != 0) return;
    const float inv = den > 0.0f ? 1.0f / den : 0.0f;
    u32x4 y;
#pragma unroll
    for (int i = 0; i < 4; ++i) y[i] = pack_bf16x2(acc[2 * i] * inv, acc[2 * i + 1] * inv);
    *reinterpret_cast<u32x4 *>(o + (((int64_t)b * Sq + row) * Hq + hq) * kD + 8 * ch) = y;
}

}  // namespace

namespace {
// Chains per (unit, pair index) of a CHAIN launch, 0 = do not chain.  A workgroup that works through
// several batch elements pays the hand-over between workgroups once, and every item after its first
// has its Q rows and first K/V tiles on the way while the item before writes its O.  Workgroups are
// still dealt out by the hardware as CUs fall free, so at least two rounds of them stay for the tail
// to even out: chains of up to 4 items once there are four rounds of items.  NVLLM_K1_CHAIN = items
// per chain (0 / 1 = no chaining).
// Measured (profiles/r03_k1_chain.txt): dense B = 8 x 4096 520.9 -> 511.0 us with chains of 4; over
// the cache (256 x 2048, T = 32 / 64) nothing: taking up an item costs 3700-7000 cycles wherever it
// stands (one wave per SIMD, nothing else to issue), so the chain saves the hand-over between
// workgroups and little else, and the paged kernel's spill reloads wait for the next item's Q
// loads.  Default: the dense kernel chains, the paged one does not unless NVLLM_K1_CHAIN asks.
int chain_count(int64_t units, int batch, bool by_default) {
    const int64_t cus = device_cus();
    int64_t len = by_default ? units * batch / (2 * cus) : 0;
    if (len > 4) len = 4;
    if (const int forced = launch_policy().k1_chain; forced >= 0) len = forced;
    if (len > batch) len = batch;
    if (len < 2) return 0;
    const int64_t nch = (batch + len - 1) / len;
    return units * nch > 0x7fffffffLL ? 0 : (int)nch;
}
}  // namespace

bool flash_attn_mfma_supported(const FlashAttnArgs &a) {
    if (a.head_dim != kD) return false;
    if (a.num_kv_heads <= 0 || a.num_heads % a.num_kv_heads != 0) return false;
    if (a.batch > 65535 || a.seq_q <= 0 || a.seq_kv <= 0) return false;
    // K/V (and Q) of one batch element are addressed with 32-bit buffer offsets
    if ((int64_t)(a.seq_kv + 3 * kKV) * a.num_kv_heads * kD * 2 >= (int64_t)1 << 31) return false;
    if ((int64_t)a.seq_q * a.num_heads * kD * 2 >= (int64_t)1 << 31) return false;
    const auto al = [](const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; };
    return al(a.q) && al(a.k) && al(a.v) && al(a.o);
}

hipError_t launch_flash_attn_mfma(const FlashAttnArgs &a, hipStream_t stream) {
    const int num_q_tiles = (a.seq_q + kQTile - 1) / kQTile;
    const int pairs = (num_q_tiles + 1) / 2;
    // adjacent pairing once there are at least two workgroups per CU (MI355X: 256 CUs)
    int pair_adjacent = (int64_t)pairs * a.num_heads * a.batch >= (int64_t)2 * device_cus();
    if (const int forced = launch_policy().k1_pairing; forced >= 0) pair_adjacent = forced;
    const dim3 grid((unsigned)(pairs * a.num_heads), (unsigned)a.batch), block(64 * kWaves);
    const float scale_log2e = a.softmax_scale * 1.4426950408889634f;
    auto q = static_cast<const __bf16 *>(a.q);
    auto k = static_cast<const __bf16 *>(a.k);
    auto v = static_cast<const __bf16 *>(a.v);
    auto o = static_cast<__bf16 *>(a.o);
    if (const int nch = chain_count((int64_t)grid.x, a.batch, true); a.causal && nch > 0) {
        flash_attn_mfma_kernel<true, false, false, true><<<dim3(grid.x * (unsigned)nch), block, 0, stream>>>(
            q, k, v, o, a.lse, a.seq_q, a.seq_kv, a.num_heads, a.num_kv_heads, num_q_tiles,
            pair_adjacent, scale_log2e, nullptr, nullptr, 0, 0u, nullptr, 0, 0, a.batch, nch);
        return hipGetLastError();
    }
#ifdef K1_DEV_ONLY_CHAIN
    return hipErrorUnknown;
#else
    if (a.causal)
        flash_attn_mfma_kernel<true, false, false, false><<<grid, block, 0, stream>>>(
            q, k, v, o, a.lse, a.seq_q, a.seq_kv, a.num_heads, a.num_kv_heads, num_q_tiles,
            pair_adjacent, scale_log2e, nullptr, nullptr, 0, 0u, nullptr, 0, 0, 0, 0);
    else
        flash_attn_mfma_kernel<false, false, false, false><<<grid, block, 0, stream>>>(
            q, k, v, o, a.lse, a.seq_q, a.seq_kv, a.num_heads, a.num_kv_heads, num_q_tiles,
            pair_adjacent, scale_log2e, nullptr, nullptr, 0, 0u, nullptr, 0, 0, 0, 0);
    return hipGetLastError();
#endif
}

// Chunked prefill / speculative verify over the paged cache: T new tokens per sequence against
// context_lens[seq] cached keys (the new tokens' K/V already written), more query rows per kv head
// than the decode kernel's 32 (paged_attn.hip).  Same contract as launch_paged_attn.
bool flash_attn_paged_supported(DType dt, const PagedAttnArgs &a) {
    if (dt != DType::BF16 || a.head_dim != kD || a.block_size != 16) return false;
    if (a.num_kv_heads <= 0 || a.num_heads % a.num_kv_heads != 0) return false;
    if (a.num_seqs <= 0 || a.num_seqs > 65535 || a.num_new_tokens <= 0) return false;
    if (a.max_blocks_per_seq <= 0 || a.max_blocks_per_seq > (1 << 26)) return false;
    if ((int64_t)a.num_new_tokens * a.num_heads * kD * 2 >= (int64_t)1 << 31) return false;  // Q rows of a sequence
    // a block is addressed through its own buffer descriptor (64-bit base, 32-bit offset inside the
    // block), so the cache may have any size; ids are unsigned 32-bit values below num_cache_blocks
    if (a.num_cache_blocks <= 0 || a.num_cache_blocks > (int64_t)0x7fffffff || a.num_kv_heads >= (1 << 15))
        return false;
    const auto al = [](const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; };
    return al(a.q) && al(a.k_cache) && al(a.v_cache) && al(a.out);
}

namespace {
// Head packing of a paged launch (see the kernel): the group's heads go into the rows when the
// group size is 2, 4 or 8.  Short chunks gain most (T = 32 at 32Q/8KV, 256 sequences: 1420 -> 500
// us); long ones gain a little from the shorter diagonal (NVLLM_K1_PACK=0 switches it off).
int paged_pack_shift(const PagedAttnArgs &a) {
    const int G = a.num_heads / a.num_kv_heads;
    const int gsh = G == 2 ? 1 : (G == 4 ? 2 : (G == 8 ? 3 : 0));
    return launch_policy().k1_pack == 0 ? 0 : gsh;
}
struct PagedGeom {
    int gsh, units, num_q_tiles, pairs;
};
PagedGeom paged_geom(const PagedAttnArgs &a) {
    PagedGeom g;
    g.gsh = paged_pack_shift(a);
    g.units = a.num_heads >> g.gsh;
    // query positions are tiled from a multiple of 32: up to 31 dead positions in front of a sequence
    g.num_q_tiles = (((a.num_new_tokens + 31) << g.gsh) + kQTile - 1) / kQTile;
    g.pairs = (g.num_q_tiles + 1) / 2;
    return g;
}
// Key-split factor of a paged launch: enough parts to put about one workgroup on every CU, at
// least 8 key tiles (512 keys) per part by the longest context the tables allow, at most 8 (16).
int paged_ksplit_parts(const PagedAttnArgs &a, const PagedGeom &g) {
    if (const int v = launch_policy().k1_ksplit; v >= 1) return v > kMaxParts ? kMaxParts : v;
    const long long units = (long long)g.pairs * g.units * a.num_seqs;
    const long long cus = device_cus();
    if (units * 2 > cus) return 1;
    long long parts = cus / units;
    const long long max_steps = ((long long)a.max_blocks_per_seq * 16 + kKV - 1) / kKV;
    if (parts > max_steps / 8) parts = max_steps / 8;
    // at most 8 parts; 16 for a handful of units (one sequence's short chunk on a long context:
    // 1 x 32768, 16 new tokens, 78 -> 58 us), where more parts than that stopped paying
    const long long cap = units <= 16 ? 16 : 8;
    if (parts > cap) parts = cap;
    return parts < 1 ? 1 : (int)parts;
}
}  // namespace

size_t flash_attn_paged_workspace_bytes(const PagedAttnArgs &a) {
    const PagedGeom g = paged_geom(a);
    const int parts = paged_ksplit_parts(a, g);
    if (parts <= 1) return 0;
    return (size_t)a.num_seqs * g.units * g.pairs * parts * kSplitSlotBytes;
}

hipError_t launch_flash_attn_paged(const PagedAttnArgs &a, hipStream_t stream) {
    const PagedGeom g = paged_geom(a);
    int pair_adjacent = (int64_t)g.pairs * g.units * a.num_seqs >= (int64_t)2 * device_cus();
    if (const int forced = launch_policy().k1_pairing; forced >= 0) pair_adjacent = forced;
    const dim3 block(64 * kWaves);
    const float scale_log2e = a.scale * 1.4426950408889634f;
    const unsigned pool = (unsigned)a.num_cache_blocks;
    auto q = static_cast<const __bf16 *>(a.q);
    auto kc = static_cast<const __bf16 *>(a.k_cache);
    auto vc = static_cast<const __bf16 *>(a.v_cache);
    auto o = static_cast<__bf16 *>(a.out);
    const int parts = paged_ksplit_parts(a, g);
    const size_t need = (size_t)a.num_seqs * g.units * g.pairs * parts * kSplitSlotBytes;
    // non-temporal K/V requests unless the K/V the tables can reach fits the 256 MiB Infinity Cache
    // anyway (the decode kernel's rule); NVLLM_PAGED_NT forces either
    const int64_t reach = (int64_t)a.num_seqs * a.max_blocks_per_seq * 2 * 16 * a.num_kv_heads * (kD * 2);
    const int nt_forced = launch_policy().paged_nt;
    const bool nt = nt_forced >= 0 ? nt_forced != 0 : reach > ((int64_t)256 << 20);
#ifndef K1_DEV_ONLY_CHAIN
    if (parts > 1) {
        // checked against flash_attn_paged_workspace_bytes by the C ABI (same policy, same arithmetic)
        if (!a.workspace || a.workspace_bytes < need || (reinterpret_cast<uintptr_t>(a.workspace) & 15u) != 0)
            return hipErrorInvalidValue;
        auto ws = static_cast<unsigned char *>(a.workspace);
        const dim3 grid((unsigned)(g.pairs * g.units * a.num_seqs), 1u, (unsigned)parts);
        auto kern = nt ? flash_attn_mfma_kernel<true, true, true, false, true>
                       : flash_attn_mfma_kernel<true, true, true, false, false>;
        kern<<<grid, block, 0, stream>>>(
            q, kc, vc, o, nullptr, a.num_new_tokens, 0, a.num_heads, a.num_kv_heads, g.num_q_tiles,
            pair_adjacent, scale_log2e, a.block_tables, a.context_lens, a.max_blocks_per_seq, pool, ws,
            g.gsh, a.num_seqs, 0, 0);
        hipError_t err = hipGetLastError();
        if (err != hipSuccess) return err;
        const dim3 mgrid(2 * kQTile / 4, (unsigned)(g.pairs * g.units), (unsigned)a.num_seqs);
        flash_attn_ksplit_merge_kernel<<<mgrid, dim3(256), 0, stream>>>(
            ws, o, a.context_lens, a.num_new_tokens, a.num_heads, a.num_kv_heads, a.max_blocks_per_seq,
            pair_adjacent, parts, g.gsh);
        return hipGetLastError();
    }
#endif
    const dim3 grid((unsigned)(g.pairs * g.units * a.num_seqs));
    float *lse_arg = nullptr;
#ifdef NVLLM_K1_STAMPS  // diagnostic build only: the stamp buffer rides in the (unused) workspace
    lse_arg = static_cast<float *>(a.workspace);
#endif
    if (const int nch = chain_count((int64_t)g.pairs * g.units, a.num_seqs, false); nch > 0) {
        flash_attn_mfma_kernel<true, true, false, true><<<dim3((unsigned)(g.pairs * g.units * nch)), block, 0, stream>>>(
            q, kc, vc, o, lse_arg, a.num_new_tokens, 0, a.num_heads, a.num_kv_heads, g.num_q_tiles,
            pair_adjacent, scale_log2e, a.block_tables, a.context_lens, a.max_blocks_per_seq, pool, nullptr,
            g.gsh, a.num_seqs, a.num_seqs, nch);
        return hipGetLastError();
    }
#ifdef K1_DEV_ONLY_CHAIN
    return hipErrorUnknown;
#else
    auto kern = nt ? flash_attn_mfma_kernel<true, true, false, false, true>
                   : flash_attn_mfma_kernel<true, true, false, false, false>;
    kern<<<grid, block, 0, stream>>>(
        q, kc, vc, o, lse_arg, a.num_new_tokens, 0, a.num_heads, a.num_kv_heads, g.num_q_tiles,
        pair_adjacent, scale_log2e, a.block_tables, a.context_lens, a.max_blocks_per_seq, pool, nullptr,
        g.gsh, a.num_seqs, 0, 0);
    return hipGetLastError();
#endif
}

}  // namespace kernels
}  // namespace nvllm
