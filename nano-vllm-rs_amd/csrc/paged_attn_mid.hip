// paged_attn_mid.hip -- K3b: paged attention for 49..128 query rows per kv head when K/V streams
// from HBM (many sequences, short chunks: speculative verify with T > 12, chunked prefill with
// chunks of 13..32 tokens at 32Q/8KV), hand-written for gfx950.
//
// Same contract as the other paged kernels (reference src/attention/paged.rs:117-192, bottom-right
// causal mask :357-381): T new tokens per sequence attend to context_lens[seq] cached keys, query
// t at position ctx - T + t.
//
// Why a third kernel.  The decode kernel (paged_attn.hip) holds at most 48 rows per wave; the
// prefill kernel over the cache (flash_attn_mfma.hip, PAGED) gives one workgroup the whole
// 512-register file and therefore the CU: its prologue, epilogue and the hand-over between
// workgroups (17 % of an item's life on these shapes, profiles/r03_k1_chain.txt) have nothing to
// overlap with, and chaining items inside the workgroup only moves them around.  These shapes are
// HBM-bound, not MFMA-bound -- so this kernel gives up the prefill kernel's software pipeline
// (256 registers per wave, compiler-scheduled MFMA builtins, one score register set) to fit TWO
// workgroups on a CU, and lets the hardware overlap one workgroup's fixed costs with the other's
// streaming.
//
// Roofline: HBM.  Algorithmic bytes per unit (one cached token of one sequence) as in paged_attn.hip.
//
// Work decomposition: grid = num_seqs * Hkv workgroups (sequence fastest), 4 waves each; every wave
// owns 32 of the kv head's R = T * G packed rows (row rho = token rho / G, head rho % G of the
// group) -- which 32 is decided by the SIMD it landed on, see the kernel.  Key tiles of 64 tokens go
// HBM -> LDS by LDS-DMA exactly as in the prefill kernel (block 4 tile + b of the sequence
// through a per-block buffer descriptor, fetched by wave b when all four waves compute and by the
// waves without rows otherwise; the bank-conflict swizzle is applied through each
// lane's source address), rings K x3 + V x2 = 80 KiB per workgroup.
// Per tile and wave: S^T = K Q^T (16 x mfma_f32_32x32x16_bf16, Q^T in registers), online softmax
// in the exp2 domain with the query row on the lane, O^T += V^T P^T (16 MFMAs, V^T by
// ds_read_b64_tr_b16, P^T straight from the S^T accumulators).  P.V runs one tile behind Q.K^T: step t
// requests V(t) and K(t+2), computes S(t), then multiplies P(t-1) V(t-1) with the row maxima and the
// exponentials of S(t) dealt out between its MFMAs (independent work for the wave's two pipes), and
// rescales O afterwards; one counted wait and one barrier per step.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>

#include "nvllm_kernels.hpp"

namespace nvllm {
namespace kernels {

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

constexpr int kD = 128;
constexpr int kKV = 64;                  // keys per tile
constexpr int kRow = 256;                // bytes of a K / V row in LDS
constexpr int kTile = kKV * kRow;        // 16 KiB
constexpr int kWaves = 4;
constexpr int kKRing = 3, kVRing = 2;    // K is requested two tiles ahead; V(t) in step t, for step t+1's P.V
constexpr int kVBase = kKRing * kTile;
constexpr int kLds = (kKRing + kVRing) * kTile;  // 80 KiB: two workgroups fill a CU's 160 KiB
constexpr int kORow = 272, kOStage = 32 * kORow;
static_assert(kWaves * kOStage <= kKRing * kTile, "O staging reuses the K ring");

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
    bf16x2 r;
    r[0] = static_cast<__bf16>(lo);
    r[1] = static_cast<__bf16>(hi);
    return __builtin_bit_cast(unsigned, r);
}

// NT = K/V requests carry the non-temporal hint (read once: do not keep them in L2 / Infinity Cache)
template <bool NT>
__global__ __launch_bounds__(64 * kWaves, 2) void paged_attn_mid_kernel(
    const __bf16 *__restrict__ q, const unsigned char *__restrict__ k_cache,
    const unsigned char *__restrict__ v_cache, const int32_t *__restrict__ block_tables,
    const int32_t *__restrict__ context_lens, __bf16 *__restrict__ out, int T, int Hq, int Hkv,
    int max_blocks_per_seq, unsigned num_cache_blocks, float scale_log2e, int by_simd) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[kLds];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 31;  // query row within the wave's block / key row / d row
    const int h = lane >> 5;  // lane half
    // sequence fastest: the workgroups in flight together walk the same kv head of different sequences
    // (measured 403 against 413 us for 256 x 2048, T = 13, with the kv head fastest)
    const int num_seqs = gridDim.x / Hkv;
    const int seq = blockIdx.x % num_seqs;
    const int kvh = blockIdx.x / num_seqs;
    const int G = Hq / Hkv;
    const int R = T * G;
    int ctx = context_lens[seq];
    const int cap = max_blocks_per_seq * 16;
    ctx = ctx < 0 ? 0 : (ctx > cap ? cap : ctx);
    const int nt = (ctx + kKV - 1) / kKV;
    const int32_t *table = block_tables + (int64_t)seq * max_blocks_per_seq;
    auto block_id = [&](int tile, int blk) -> unsigned {  // id of the sequence's block 4 tile + blk
        int bi = 4 * tile + blk;
        bi = bi < max_blocks_per_seq ? bi : max_blocks_per_seq - 1;
        return (unsigned)table[bi];
    };

    // Which 32 rows this wave owns.  With fewer than 97 rows some waves only fetch; a workgroup's four
    // waves sit on the CU's four SIMDs in one of four rotations, so the computing waves of the two
    // workgroups that share a CU meet on the same SIMDs as often as not.  Row block by SIMD instead:
    // block (SIMD_ID + 2 TG_ID) & 3 -- TG_ID of HW_REG_HW_ID is the workgroup's slot on the CU, 0 or 1 for
    // the two neighbours -- puts one workgroup's blocks 0, 1 on SIMDs 0, 1 and the other's on 2, 3.
    // The waves agree through LDS (the third K slot is idle until step 0) and fall back to block = wave unless their SIMD ids are four different ones: only
    // the balance rests on the hardware's placement (scripts/diag/hwid_probe.hip shows it; K/V from L2,
    // 256 x 2048 T = 16: 232 -> 208 us).
    int rblk = wave;
    {
        unsigned hw_id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_id));
        volatile unsigned *agree = reinterpret_cast<volatile unsigned *>(lds + 2 * kTile);
        if (lane == 0) agree[wave] = (hw_id >> 4) & 3u;
        __syncthreads();
        const unsigned s0 = agree[0], s1 = agree[1], s2 = agree[2], s3 = agree[3];
        __syncthreads();  // (read before step 0 requests K(2) into this slot)
        const bool distinct = ((1u << s0) | (1u << s1) | (1u << s2) | (1u << s3)) == 0xfu;
        const int turned = (int)((((hw_id >> 4) & 3u) + 2u * ((hw_id >> 16) & 1u)) & 3u);
        rblk = __builtin_amdgcn_readfirstlane((distinct && by_simd) ? turned : wave);  // (by_simd = 0: the fallback, for tests)
    }
    // Who fetches.  With 97..128 rows all four waves compute, and wave w fetches block w of every tile.
    // With fewer, the waves without rows fetch EVERYTHING (two of them: blocks {0, 2} and {1, 3}; one:
    // all four) and the computing waves issue no request at all: a wave stands still while its LDS-DMA
    // requests wait for room among the CU's outstanding ones (700-1200 cycles per step in the stamps,
    // profiles/r03_paged_mid_stamps.txt), which costs a fetch-only wave nothing.
    const int n_live = (R + 31) >> 5;                    // waves with rows: 2, 3 or 4 (R >= 33)
    const int n_idle = 4 - n_live;
    const bool i_fetch = n_idle == 0 || rblk >= n_live;
    const int blk0 = n_idle == 0 ? wave : rblk - n_live; // my first block, then every blk_step-th
    const int blk_step = n_idle == 0 ? 4 : n_idle;
    const int n_mine = !i_fetch ? 0 : (n_idle == 0 ? 1 : 4 / n_idle);   // 0, 1, 2 or 4 blocks per tile
    // ---- staging (see flash_attn_mfma.hip): piece j = rows 16 wave + 4 j .. + 3 of the tile, lane L
    // fetches the 16-byte chunk that the swizzle puts at position L & 15 of row L >> 4
    const unsigned row_bytes = (unsigned)(Hkv * kD * 2);
    const unsigned block_bytes = 16u * row_bytes;
    const unsigned head = (unsigned)kvh * (unsigned)(kD * 2);
    const unsigned long long kbase = (unsigned long long)k_cache + head, vbase = (unsigned long long)v_cache + head;
    const unsigned nrec = block_bytes - head;
    unsigned dma_pk[4], dma_pv[4];
    {
        const unsigned rin = (unsigned)lane >> 4, pos = (unsigned)lane & 15u;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const unsigned r = 4u * j + rin;
            dma_pk[j] = r * row_bytes + 16u * (pos ^ (r & 15u)) - 1024u * j;  // (r >= 4 j and a row has >= 256 bytes)
            dma_pv[j] = r * row_bytes + 16u * (pos ^ (4u * rin)) - 1024u * j;
        }
    }
    const unsigned lds_base = (unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned char *)lds;
    // descriptor of this kv head's rows in block `id`; a tile past the context (live = false) or an id
    // outside the pool gets num_records = 0: the request still counts in vmcnt, but fetches nothing
    auto desc = [&](unsigned long long base, unsigned id, bool live) -> u32x4 {
        id = (unsigned)__builtin_amdgcn_readfirstlane((int)id);
        const unsigned long long a = base + (unsigned long long)id * block_bytes;
        return u32x4{(unsigned)a, (unsigned)(a >> 32) & 0xffffu, (live && id < num_cache_blocks) ? nrec : 0u, 0x00020000u};
    };
    // this wave's 16 rows of K tile `tile` -> K ring slot `slot` / of V tile -> V ring slot: four pieces
    // of 1 KiB behind ONE write of M0 -- the instruction's immediate offset moves the LDS address and the
    // memory address alike, so piece j's lane offset is stored less 1024 j (dma_pk / dma_pv below)
#define NVLLM_MID_DMA4(NTBIT)                                                              \
    "s_nop 4\n\ts_mov_b32 m0, %0\n\ts_nop 0\n\t"                                          \
    "buffer_load_dwordx4 %1, %5, 0 offen" NTBIT " lds\n\t"                                  \
    "buffer_load_dwordx4 %2, %5, 0 offen offset:1024" NTBIT " lds\n\t"                      \
    "buffer_load_dwordx4 %3, %5, 0 offen offset:2048" NTBIT " lds\n\t"                      \
    "buffer_load_dwordx4 %4, %5, 0 offen offset:3072" NTBIT " lds"
    auto request4 = [&](unsigned dst, unsigned o0, unsigned o1, unsigned o2, unsigned o3, const u32x4 &rs) {
        if constexpr (NT)
            asm volatile(NVLLM_MID_DMA4(" nt") : : "s"(dst), "v"(o0), "v"(o1), "v"(o2), "v"(o3), "s"(rs) : "memory");
        else
            asm volatile(NVLLM_MID_DMA4("") : : "s"(dst), "v"(o0), "v"(o1), "v"(o2), "v"(o3), "s"(rs) : "memory");
    };
#undef NVLLM_MID_DMA4
    auto request_k = [&](int tile, int slot, unsigned id, int blk) {
        const u32x4 rk = desc(kbase, id, tile < nt);
        request4(lds_base + (unsigned)(slot * kTile) + 16u * blk * 256u, dma_pk[0], dma_pk[1], dma_pk[2], dma_pk[3], rk);
    };
    auto request_v = [&](int tile, int slot, unsigned id, int blk) {
        const u32x4 rv = desc(vbase, id, tile < nt);
        const int rows_left = ctx - 16 * (4 * tile + blk);
        unsigned off[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)  // rows past the context: true zeros
            off[j] = 4 * j + (lane >> 4) < rows_left ? dma_pv[j] : 0x40000000u;
        request4(lds_base + (unsigned)(kVBase + slot * kTile) + 16u * blk * 256u, off[0], off[1], off[2], off[3], rv);
    };
    // Counted waits: what may still fly is this wave's LAST request group, four pieces per block it fetches
    auto wait_all_but_last_group = [&]() {
        if (n_mine == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else if (n_mine == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (n_mine == 4) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    };

    // ids of my blocks in tiles t .. t+3 (slot k = my k-th block)
    unsigned id0[4], id1[4], id2[4], id3[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        id0[k] = id1[k] = id2[k] = id3[k] = 0u;
        if (k < n_mine) {
            const int blk = blk0 + k * blk_step;
            id0[k] = block_id(0, blk);
            id1[k] = block_id(1, blk);
            id2[k] = block_id(2, blk);
            id3[k] = block_id(3, blk);
        }
    }
    {
        // K(0) and K(1); V(t) is requested in step t (P.V of tile t runs in step t+1).  V slot 1 is
        // read by step 0's (empty) P.V: zeros
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (k < n_mine) request_k(0, 0, id0[k], blk0 + k * blk_step);
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (k < n_mine) request_k(1, 1, id1[k], blk0 + k * blk_step);
        const u32x4 z = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int j = 0; j < 4; ++j)
            *reinterpret_cast<u32x4 *>(lds + kVBase + kTile + (16 * wave + 4 * j) * kRow + 16 * lane) = z;
    }

    const int rho = 32 * rblk + c;
    const bool row_ok = rho < R;
    const int t_row = row_ok ? rho / G : 0;
    const int g_row = rho - t_row * G;
    const int qpos = ctx - T + t_row;          // position: keys 0 .. qpos are visible
    const bool wave_live = 32 * rblk < R;      // (uniform)

    // ---- Q^T fragments (B operand of S^T = K Q^T): lane (c, h) holds Q[row c][16 ks + 8 h .. + 7]
    u32x4 qf[8];
    {
        const unsigned seq_bytes = (unsigned)T * (unsigned)Hq * (unsigned)(kD * 2);
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
            (void *)(q + (int64_t)seq * T * Hq * kD), 0, (int)seq_bytes, 0x00020000);
        const unsigned off = row_ok ? ((unsigned)t_row * (unsigned)Hq + (unsigned)(kvh * G + g_row)) * (unsigned)(kD * 2) + 16u * h
                                    : 0xffffff00u;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks)
            qf[ks] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(off + 32u * ks), 0, 0));
    }
    // ---- LDS read addresses (as in flash_attn_mfma.hip)
    unsigned k_rd8[8];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) k_rd8[ks] = (unsigned)(c * kRow + 16 * ((2 * ks + h) ^ (c & 15)));
    const int c16 = lane & 15, qq = c16 >> 2, pp = c16 & 3, g16 = lane >> 4;
    const int v_row_in = 4 * h + qq;
    const int v_sw = 4 * (v_row_in & 3);
    const int v_ch_lo = 2 * (g16 & 1) + (pp >> 1);
    unsigned v_rd[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
        v_rd[dt] = (unsigned)(v_row_in * kRow + 8 * (pp & 1) + 16 * ((4 * dt + v_ch_lo) ^ v_sw));

    // ---- running state: O^T of the wave's 32 rows (d tile dt in acc[dt]), row max, partial row sum
    f32x16 acc[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[dt][i] = 0.0f;
    float m_run = -INFINITY, l_run = 0.0f;

    // (the Q rows are the only loads hipcc knows of: consumed here once, or it would wait for them --
    // and with them for every request in flight -- in front of each step's first MFMAs)
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) asm volatile("" : "+v"(qf[ks]));

    // ---- S^T = K Q^T of one tile: keys 0..31 in sa, 32..63 in sb.  The fragments of two k-steps are
    // read while the two before them are multiplied (the order is pinned: left to itself hipcc puts
    // every read right in front of its MFMA and waits for it there)
    auto qk_tile = [&](const unsigned char *kt, f32x16 &sa, f32x16 &sb) {
#pragma unroll
        for (int i = 0; i < 16; ++i) { sa[i] = 0.0f; sb[i] = 0.0f; }
        u32x4 kf[2][4];  // [set][2 * (ks & 1) + key half]
        auto read_k = [&](int set, int ks0) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                kf[set][2 * i] = *reinterpret_cast<const u32x4 *>(kt + k_rd8[ks0 + i]);
                kf[set][2 * i + 1] = *reinterpret_cast<const u32x4 *>(kt + k_rd8[ks0 + i] + 32 * kRow);
            }
        };
        read_k(0, 0);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if (g < 3) read_k((g + 1) & 1, 2 * g + 2);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                sa = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, kf[g & 1][2 * i]),
                                                             __builtin_bit_cast(bf16x8, qf[2 * g + i]), sa, 0, 0, 0);
                sb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, kf[g & 1][2 * i + 1]),
                                                             __builtin_bit_cast(bf16x8, qf[2 * g + i]), sb, 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // ---- mask: only tiles that reach past the first query position or the context (a real branch:
    // the asm statement keeps hipcc from turning it into 130 predicated instructions for every tile)
    auto mask_tile = [&](int t, f32x16 &sa, f32x16 &sb) {
        const int kv0 = t * kKV;
        if (kv0 + kKV > ctx - T + 1) {
            asm volatile("" ::: "memory");
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = kv0 + (r & 3) + 8 * (r >> 2) + 4 * h;
                sa[r] = (key < ctx && key <= qpos) ? sa[r] : -INFINITY;
                sb[r] = (key + 32 < ctx && key + 32 <= qpos) ? sb[r] : -INFINITY;
            }
        }
    };
    auto row_max = [&](const f32x16 &sa, const f32x16 &sb) -> float {  // 15 v_max3 in two chains + 1
        auto max3 = [](float a, float b, float c2) {
            float r;
            asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c2));
            return r;
        };
        float ma = max3(sa[0], sa[1], sa[2]), mb = max3(sb[0], sb[1], sb[2]);
#pragma unroll
        for (int r = 3; r < 15; r += 2) {
            ma = max3(ma, sa[r], sa[r + 1]);
            mb = max3(mb, sb[r], sb[r + 1]);
        }
        const float mx = max3(ma, mb, fmaxf(sa[15], sb[15]));
        // the other lane half's maximum: v_permlane32_swap (hipcc's __shfl_xor goes through LDS, a round
        // trip the exponentials would wait for); s_nop 1 = the VALU-write -> permlane wait states
        float a = mx, b2 = mx;
        asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b2));
        return fmaxf(a, b2);
    };
    // the four V^T fragments of d tile dt
    auto read_v = [&](const unsigned char *vt, s16x8 (&vf)[4], int dt) {
#pragma unroll
        for (int sp = 0; sp < 4; ++sp) {
            const unsigned char *p0 = vt + v_rd[dt] + sp * 16 * kRow;
            const s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)p0);
            const s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)(p0 + 8 * kRow));
            vf[sp] = s16x8{v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
        }
    };
    // p (bf16 pairs) of score elements e0 .. e0 + n - 1 (e < 16: sa, else sb), their sum added to psum
    auto exp_n = [&](const f32x16 &sa, const f32x16 &sb, int e0, int n, float mc, u32x4 (&pd)[4], f32x2 &psum2) {
        // (two scores at a time: the scale-and-shift and the row sum as packed f32 instructions)
        const f32x2 sc2 = {scale_log2e, scale_log2e}, mc2 = {-mc, -mc};
#pragma unroll
        for (int e = e0; e < e0 + n; e += 2) {
            const f32x2 s2 = {e < 16 ? sa[e & 15] : sb[e & 15], e < 16 ? sa[(e + 1) & 15] : sb[(e + 1) & 15]};
            const f32x2 x2 = __builtin_elementwise_fma(s2, sc2, mc2);
            const f32x2 y2 = {fast_exp2(x2[0]), fast_exp2(x2[1])};
            psum2 += y2;
            pd[e >> 3][(e & 7) >> 1] = pack_bf16x2(y2[0], y2[1]);
        }
    };
    auto rescale = [&](float m_new, float mc) {  // exp2(-inf) = 0 for a fresh row
        const float alpha = fast_exp2(m_run * scale_log2e - mc);
        l_run *= alpha;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[dt][i] *= alpha;
        m_run = m_new;
    };

    {
        // Step t: request V(t) and K(t+2); S = Q K(t)^T; then O += P(t-1) V(t-1), its MFMAs interleaved
        // with the exponentials of S (independent work for the wave's two pipes); O is rescaled for
        // tile t's maxima after that.  One wait + barrier per step: K(t+1) and V(t) have landed, the
        // pieces of K(t+2) may fly (four per block the wave fetches), and every wave is done reading K(t) and V(t-1), whose slots
        // the next step's requests overwrite.  K ring x3, V ring x2.
        wait_all_but_last_group();  // K(0) is in
        __syncthreads();
        u32x4 p[4] = {{0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}};  // P^T of the tile before
        int kslot = 0, kslot_req = 2;
        for (int t = 0; t < nt; ++t) {
            const int vslot = t & 1;
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (k < n_mine) request_v(t, vslot, id0[k], blk0 + k * blk_step);
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (k < n_mine) {
                    request_k(t + 2, kslot_req, id2[k], blk0 + k * blk_step);
                    id0[k] = id1[k];
                    id1[k] = id2[k];
                    id2[k] = id3[k];
                    id3[k] = block_id(t + 4, blk0 + k * blk_step);
                }
            const unsigned char *kt = lds + kslot * kTile;
            const unsigned char *vt = lds + kVBase + (vslot ^ 1) * kTile;  // V(t-1)
            kslot = kslot == kKRing - 1 ? 0 : kslot + 1;
            kslot_req = kslot_req == kKRing - 1 ? 0 : kslot_req + 1;
            if (wave_live) {
                f32x16 sa, sb;
                qk_tile(kt, sa, sb);
                mask_tile(t, sa, sb);
                // d tile 0 of P(t-1) V(t-1) under the row maxima of S, d tiles 1..3 under its exponentials
                // (12 + 12 + 8 of the 32 scores a lane holds)
                u32x4 pn[4];
                f32x2 psum = {0.0f, 0.0f};
                s16x8 vf[2][4];
                read_v(vt, vf[0], 0);
                read_v(vt, vf[1], 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int sp = 0; sp < 4; ++sp)
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, vf[0][sp]),
                                                                     __builtin_bit_cast(bf16x8, p[sp]), acc[0], 0, 0, 0);
                float m_new = fmaxf(m_run, row_max(sa, sb));
                asm volatile("" : "+v"(m_new));
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                const float mc = ((m_new == -INFINITY) ? 0.0f : m_new) * scale_log2e;
#pragma unroll
                for (int dt = 1; dt < 4; ++dt) {
                    if (dt < 3) read_v(vt, vf[(dt + 1) & 1], dt + 1);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int sp = 0; sp < 4; ++sp)
                        acc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, vf[dt & 1][sp]),
                                                                          __builtin_bit_cast(bf16x8, p[sp]), acc[dt], 0, 0, 0);
                    exp_n(sa, sb, 12 * (dt - 1), dt < 3 ? 12 : 8, mc, pn, psum);
                    asm volatile("" : "+v"(psum));  // (here, not sunk below the rescale branch)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {  // one MFMA, then its share of the exponentials
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x002, 10, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int sp = 0; sp < 4; ++sp) asm volatile("" : "+v"(pn[sp]));
                if (__any(m_new > m_run)) rescale(m_new, mc);  // (uniform branch: most tiles leave every row's max alone)
                l_run += psum[0] + psum[1];
#pragma unroll
                for (int sp = 0; sp < 4; ++sp) p[sp] = pn[sp];
            }
            wait_all_but_last_group();
            __syncthreads();
        }
        if (wave_live && nt > 0) {  // P.V of the last tile
            const unsigned char *vt = lds + kVBase + ((nt - 1) & 1) * kTile;
            s16x8 vf[2][4];
            read_v(vt, vf[0], 0);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                if (dt < 3) read_v(vt, vf[(dt + 1) & 1], dt + 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int sp = 0; sp < 4; ++sp)
                    acc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, vf[dt & 1][sp]),
                                                                      __builtin_bit_cast(bf16x8, p[sp]), acc[dt], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }

    // ---- epilogue: O^T lane (c, h) holds O[row c][32 dt + 8 i + 4 h + (0..3)] in acc[dt][4 i .. 4 i + 3];
    // through a wave-private LDS image (the rings are free), out as whole 256-byte rows
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the requests past the last tile (they fetch nothing) are in
    __syncthreads();
    if (!wave_live) return;
    float l = l_run + __shfl_xor(l_run, 32);
    const float inv = l > 0.0f ? 1.0f / l : 0.0f;
    unsigned char *stg = lds + wave * kOStage;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            u32x2 w;
            w[0] = pack_bf16x2(acc[dt][4 * i] * inv, acc[dt][4 * i + 1] * inv);
            w[1] = pack_bf16x2(acc[dt][4 * i + 2] * inv, acc[dt][4 * i + 3] * inv);
            *reinterpret_cast<u32x2 *>(stg + c * kORow + (32 * dt + 8 * i + 4 * h) * 2) = w;
        }
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const int ri = 4 * r + (lane >> 4);
        const u32x4 x = *reinterpret_cast<const u32x4 *>(stg + ri * kORow + (lane & 15) * 16);
        const int orho = 32 * rblk + ri;
        if (orho < R) {
            const int ot = orho / G, og = orho - ot * G;
            *reinterpret_cast<u32x4 *>(out + (((int64_t)seq * T + ot) * Hq + kvh * G + og) * kD + (lane & 15) * 8) = x;
        }
    }
}

}  // namespace

// Shapes the kernel takes: bf16, head_dim 128, block 16, 49..128 query rows per kv head.  Whether
// it is the one to run is the dispatcher's decision (enough units that K/V streams from HBM in
// several rounds of workgroups; fewer go to the prefill kernel and its key split).
bool paged_attn_mid_supported(DType dt, const PagedAttnArgs &a) {
    if (dt != DType::BF16 || a.head_dim != kD || a.block_size != 16) return false;
    if (a.num_kv_heads <= 0 || a.num_heads % a.num_kv_heads != 0 || a.num_kv_heads >= (1 << 15)) return false;
    const int64_t rows = (int64_t)a.num_new_tokens * (a.num_heads / a.num_kv_heads);
    // (33..48 rows are the decode kernel's by default; NVLLM_PAGED_MID=1 sends them here for comparison)
    if (rows <= (launch_policy().paged_mid > 0 ? 32 : 48) || rows > 128) return false;
    if (a.num_seqs <= 0 || (int64_t)a.num_seqs * a.num_kv_heads > 0x7fffffffLL) return false;
    if (a.max_blocks_per_seq <= 0 || a.max_blocks_per_seq > (1 << 26)) return false;
    if (a.num_cache_blocks <= 0 || a.num_cache_blocks > (int64_t)0x7fffffff) return false;
    if ((int64_t)a.num_new_tokens * a.num_heads * kD * 2 >= (int64_t)1 << 31) return false;
    const auto al = [](const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; };
    return al(a.q) && al(a.k_cache) && al(a.v_cache) && al(a.out);
}

hipError_t launch_paged_attn_mid(const PagedAttnArgs &a, hipStream_t stream) {
    const dim3 grid((unsigned)(a.num_seqs * a.num_kv_heads)), block(64 * kWaves);
    // non-temporal requests unless the K/V the tables can reach fits the 256 MiB Infinity Cache anyway
    // (the decode kernel's rule, paged_attn.hip); NVLLM_PAGED_NT forces either
    const int64_t reach = (int64_t)a.num_seqs * a.max_blocks_per_seq * 2 * 16 * a.num_kv_heads * (kD * 2);
    const int nt_forced = launch_policy().paged_nt;
    const bool nt = nt_forced >= 0 ? nt_forced != 0 : reach > ((int64_t)256 << 20);
    auto kern = nt ? paged_attn_mid_kernel<true> : paged_attn_mid_kernel<false>;
    kern<<<grid, block, 0, stream>>>(
        static_cast<const __bf16 *>(a.q), static_cast<const unsigned char *>(a.k_cache),
        static_cast<const unsigned char *>(a.v_cache), a.block_tables, a.context_lens,
        static_cast<__bf16 *>(a.out), a.num_new_tokens, a.num_heads, a.num_kv_heads, a.max_blocks_per_seq,
        (unsigned)a.num_cache_blocks, a.scale * 1.4426950408889634f, launch_policy().paged_mid == 2 ? 0 : 1);
    return hipGetLastError();
}

}  // namespace kernels
}  // namespace nvllm
