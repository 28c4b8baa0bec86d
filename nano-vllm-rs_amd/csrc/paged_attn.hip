// paged_attn.hip -- K2 / K2b / K3: paged attention over the block-table KV
// cache, hand-written for gfx950 (MI355X, wave64, MFMA 16x16x32 bf16).
//
// Replaces paged_attention (reference src/attention/paged.rs:117-192), which
// per sequence gathers the blocks (index_select), trims to context_len,
// expands GQA, materialises QK^T, masks (bottom-right causal iff T > 1,
// :357-381), softmaxes and multiplies by V -- five full copies of K and V.
// Here K and V are read exactly once, straight from their 16-token blocks via
// the block table, for a whole batch of sequences per launch.
//
// ---------------------------------------------------------------------------
// Fast path (bf16, head_dim 128, block_size 16, T * Hq/Hkv <= 48 query rows)
// ---------------------------------------------------------------------------
// Roofline: HBM.  Algorithmic bytes per unit (one cached token of one sequence)
// = 2 (K,V) * Hkv * 128 * 2 B (4 KiB at Hkv = 8); Q/O/table bytes are < 0.3 %.
//
// Work decomposition
//   grid  = (splits, num_seqs, ceil(Hkv / HPB)),  block = HPB waves
//   wave w of a workgroup owns ONE kv head and the G = Hq/Hkv query heads of
//   its group for all T new tokens (R = T*G query rows, padded to 16 per MFMA
//   tile), over this split's range of logical blocks.  The HPB waves of a
//   workgroup walk the same block-table entries, so together they read the
//   whole contiguous 16 x Hkv x 256 B block (DRAM-page friendly) while every
//   wave keeps its own online-softmax state: no barriers, no cross-wave
//   traffic.  Waves never share LDS; each has a private 16 KiB staging area:
//   an 8 KiB K image and an 8 KiB V image of the current 32-token tile.
//
// Per wave, per 32-token tile (two 16-token blocks):
//   S^T[token][row] = K[token][:] . Q[row][:]      8 x mfma_f32_16x16x32_bf16
//       A = K: loaded like V with fully coalesced 16 B/lane loads (4 whole
//       256 B rows per instruction; fragment-shaped loads of 16 rows x 64 B ask
//       L2 for every 128 B line twice), written to the wave's K image with the
//       16 B chunk ch of row r at position ch ^ r, and read back as
//       row-per-lane A fragments (lane (g = lane>>4, c = lane&15): 16 B of token
//       row c at dims 32s+8g.. for k-step s) with conflict-free ds_read_b128.
//       B = Q^T: kept in VGPRs for the whole kernel.
//   online softmax in the exp2 domain; state (m, l) lives in the lane that
//       owns query row c, replicated over the four lane groups g (the row max
//       needs two cross-group shuffles per tile; l stays a per-group partial
//       until the end).
//   O^T[d][row] += V^T[d][token] . P^T[token][row]  8 x mfma_f32_16x16x32_bf16
//       B = P^T comes straight from the S^T accumulators (token 4g+j of block
//       0/1 <-> k slot 8g + j / 8g+4+j): no lane movement.
//       A = V^T needs tokens along k, but memory has d contiguous: V is loaded
//       with fully coalesced 16 B/lane loads (4 whole 256 B rows per
//       instruction), written to the wave's private LDS tile (XOR-swizzled
//       16 B chunks) and read back transposed with ds_read_b64_tr_b16.
//   Because O^T keeps the query row on the lane (col = lane&15), the rescale
//   by alpha = 2^(m_old - m_new) and the final 1/l are lane-local.
// Loads for tile i+1 are issued before tile i is consumed (register double
// buffering, 16 KiB in flight per wave, then the ds_write into the two LDS
// images); block-table entries are fetched two
// tiles ahead with scalar loads.  Only the last tile(s) of a sequence take
// the masked path (context tail, bottom-right causal mask for T > 1, V rows
// past context_len zeroed so stale cache contents can never reach O).
//
// Split-K (K2b): when splits > 1 every (seq, split, row) writes an
// unnormalised partial (O[128] f32, m, l) to the workspace and
// paged_attn_merge_kernel combines them with the usual log-sum-exp weights.
//
// ---------------------------------------------------------------------------
// Generic path (any head_dim <= 256, any block_size, bf16 or f32)
// ---------------------------------------------------------------------------
// One wave per (sequence, new token, query head); f32 FMA.  Used for the
// reference's small test shapes; not a performance path.
#include <hip/hip_runtime.h>

#include <type_traits>

#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdlib>

#include "nvllm_kernels.hpp"

namespace nvllm {
namespace kernels {

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;  // one 16-byte load / MFMA operand

constexpr int kD = 128;          // head_dim of the fast path
constexpr int kBS = 16;          // block_size of the fast path
constexpr int kRowBytes = kD * 2;
constexpr int kTileBlocks = 2;   // 32 tokens per tile
constexpr int kLdsTile = kTileBlocks * kBS * kRowBytes;     // 8 KiB: one tile of K or of V
constexpr int kLdsPerWave = 2 * kLdsTile;                    // the wave's K image and V image
constexpr int64_t kMultiTileMinTokens = 384 * 1024;  // see paged_attn_fast_path
constexpr int64_t kMultiTileMinContext = 16384;
constexpr int kMaxFastRows = 48;  // query rows (T * Hq/Hkv) per kv head the decode kernel takes
                                  // (a fourth query tile no longer fits 512 registers)

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {
    // round-to-nearest-even via the hardware convert (v_cvt_pk_bf16_f32)
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
    bf16x2 r;
    r[0] = static_cast<__bf16>(lo);
    r[1] = static_cast<__bf16>(hi);
    return __builtin_bit_cast(unsigned, r);
}

struct TileRegs {
    u32x4 k[kTileBlocks][4];  // [block][row quad] coalesced rows of K
    u32x4 v[kTileBlocks][4];  // [block][row quad] coalesced rows of V
};

// NQT = number of 16-row query tiles (R <= 16 NQT).  NQT >= 2 needs more than 256 registers, so
// it runs with at most 4 waves per workgroup (one wave per SIMD, 512 registers).
template <int NQT, int HPB, bool NT>
__global__ __launch_bounds__(64 * HPB) void paged_attn_kernel(
    const __bf16 *__restrict__ q, const unsigned char *__restrict__ k_cache,
    const unsigned char *__restrict__ v_cache, const int32_t *__restrict__ block_tables,
    const int32_t *__restrict__ context_lens, __bf16 *__restrict__ out,
    float *__restrict__ ws_o, float *__restrict__ ws_ml, int T, int Hq, int Hkv,
    int max_blocks_per_seq, int64_t num_cache_blocks, float scale_log2e, int num_splits,
    int blocks_per_split) {
    __shared__ __attribute__((aligned(16))) unsigned char lds_all[HPB * kLdsPerWave];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = lane >> 4;   // lane group
    const int c = lane & 15;   // query row within a q tile / token row / d within a d tile
    const int split = blockIdx.x;
    const int seq = blockIdx.y;
    const int kvh = blockIdx.z * HPB + wave;
    if (kvh >= Hkv) return;  // whole wave; no barriers are used anywhere

    const int G = Hq / Hkv;
    const int R = T * G;
    const int ctx = context_lens[seq];
    int nblk_seq = ctx > 0 ? (ctx + kBS - 1) / kBS : 0;
    if (nblk_seq > max_blocks_per_seq) nblk_seq = max_blocks_per_seq;  // never read past the table row
    const int blk_begin = split * blocks_per_split;
    int blk_end = blk_begin + blocks_per_split;
    if (blk_end > nblk_seq) blk_end = nblk_seq;
    const int ntiles = blk_end > blk_begin ? (blk_end - blk_begin + kTileBlocks - 1) / kTileBlocks : 0;

    unsigned char *lds = lds_all + wave * kLdsPerWave;
    const int32_t *table = block_tables + (int64_t)seq * max_blocks_per_seq;

    // ---- Q^T fragments (B operand): lane (g,c) holds Q[row c][32s+8g .. +7]
    u32x4 qf[NQT][4];
#pragma unroll
    for (int qt = 0; qt < NQT; ++qt) {
        const int r = qt * 16 + c;
        const bool valid = r < R;
        const int t = valid ? r / G : 0;
        const int hg = valid ? r - t * G : 0;
        const __bf16 *qrow = q + (((int64_t)seq * T + t) * Hq + kvh * G + hg) * kD;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            u32x4 x = u32x4{0u, 0u, 0u, 0u};
            if (valid) x = *reinterpret_cast<const u32x4 *>(qrow + 32 * s + 8 * g);
            qf[qt][s] = x;
        }
    }
    // new-token index of this lane's query rows (for the causal mask)
    int t_of_row[NQT];
#pragma unroll
    for (int qt = 0; qt < NQT; ++qt) {
        const int r = qt * 16 + c;
        t_of_row[qt] = r < R ? r / G : 0;
    }

    const int64_t block_bytes = (int64_t)kBS * Hkv * kRowBytes;
    const unsigned row_stride = (unsigned)Hkv * kRowBytes;
    const unsigned v_lane_off = (unsigned)g * row_stride + (unsigned)kvh * kRowBytes + 16u * c;
    // LDS addresses.  Row r of block b sits at b*4096 + r*256; its 16-byte chunk
    // ch is stored at chunk position ch ^ (2*(r&7)) (conflict-free tr reads).
    unsigned lds_wr[4];  // write address of V row (4i+g), chunk c
    // K image (second half of the wave's LDS): chunk ch of row r at position ch ^ r, so that the
    // row-per-lane A-fragment reads below (16 lanes, 16 rows, the same chunk) hit 16 different
    // positions, as the whole-row writes do.
    unsigned lds_kwr[4], lds_krd[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = 4 * i + g;
        lds_wr[i] = (unsigned)(r * kRowBytes + 16 * (c ^ (2 * (r & 7))));
        lds_kwr[i] = (unsigned)(kLdsTile + r * kRowBytes + 16 * (c ^ r));
        lds_krd[i] = (unsigned)(kLdsTile + c * kRowBytes + 16 * ((4 * i + g) ^ c));  // k-step i
    }
    // transposed read: lane c = 4*qq + p of group g addresses row 4g+qq,
    // elements 16*dt + 4p .. +3  => chunk 2*dt + (p>>1), byte 8*(p&1) in chunk.
    const int qq = c >> 2, p = c & 3;
    const int tr_row = 4 * g + qq;
    const unsigned tr_base = (unsigned)(tr_row * kRowBytes + 8 * (p & 1));
    const int tr_sw = 2 * (tr_row & 7);
    const int tr_ch0 = p >> 1;

    auto clamp_id = [&](int id) -> int64_t {
        int64_t x = id;
        if (x < 0) x = 0;
        if (x >= num_cache_blocks) x = num_cache_blocks - 1;
        return x;
    };
    // physical ids of tile t (second id duplicated when the split ends on an odd block)
    auto load_ids = [&](int t, int &p0, int &p1) {
        const int b0 = blk_begin + kTileBlocks * t;
        if (t < ntiles) {
            p0 = table[b0];
            p1 = (b0 + 1 < blk_end) ? table[b0 + 1] : p0;
        } else {
            p0 = 0;
            p1 = 0;
        }
    };
    auto load_tile = [&](auto stream_tag, TileRegs &tr, int p0, int p1) {
        constexpr bool STREAM = decltype(stream_tag)::value;
        const int64_t pid[kTileBlocks] = {clamp_id(p0), clamp_id(p1)};
#pragma unroll
        for (int b = 0; b < kTileBlocks; ++b) {
            // K and V alike: an instruction reads four whole 256-byte rows.  (K used to be read
            // straight into the MFMA's row-per-lane layout, 64 bytes of 16 rows per instruction:
            // every 128-byte line was then asked for twice, and where the reads are served by L2
            // -- shared prefixes -- L2's request rate was the bound: 1024 shared tokens x 256
            // sequences 88 -> 56 us.)
            const unsigned char *kb = k_cache + pid[b] * block_bytes + v_lane_off;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const u32x4 *ptr = reinterpret_cast<const u32x4 *>(kb + (size_t)(4 * s) * row_stride);
                tr.k[b][s] = STREAM ? __builtin_nontemporal_load(ptr) : *ptr;
            }
        }
#pragma unroll
        for (int b = 0; b < kTileBlocks; ++b) {
            const unsigned char *vb = v_cache + pid[b] * block_bytes + v_lane_off;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const u32x4 *ptr = reinterpret_cast<const u32x4 *>(vb + (size_t)(4 * i) * row_stride);
                tr.v[b][i] = STREAM ? __builtin_nontemporal_load(ptr) : *ptr;
            }
        }
    };

    // ---- running state (exp2 domain)
    float m_run[NQT], l_run[NQT];
    f32x4 acc[NQT][8];
#pragma unroll
    for (int qt = 0; qt < NQT; ++qt) {
        m_run[qt] = -INFINITY;
        l_run[qt] = 0.0f;
#pragma unroll
        for (int dt = 0; dt < 8; ++dt) acc[qt][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    // first token that some query row cannot see (bottom-right mask); tiles
    // that end at or before it need no masking at all
    const int first_masked_tok = ctx - (T - 1);

    auto compute_tile = [&](TileRegs &tr, int t) {
        const int tok0 = (blk_begin + kTileBlocks * t) * kBS;
        const bool tail = (tok0 + kTileBlocks * kBS > first_masked_tok) ||
                          (blk_begin + kTileBlocks * t + 1 >= blk_end);
        // ---- K through the wave's LDS image into A fragments: lane (g,c) = K[row c][32s+8g .. +7]
#pragma unroll
        for (int b = 0; b < kTileBlocks; ++b)
#pragma unroll
            for (int i = 0; i < 4; ++i)
                *reinterpret_cast<u32x4 *>(lds + b * (kBS * kRowBytes) + lds_kwr[i]) = tr.k[b][i];
#pragma unroll
        for (int b = 0; b < kTileBlocks; ++b)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
                tr.k[b][ks] = *reinterpret_cast<const u32x4 *>(lds + b * (kBS * kRowBytes) + lds_krd[ks]);
        // ---- S^T = K Q^T
        f32x4 s[NQT][kTileBlocks];
#pragma unroll
        for (int qt = 0; qt < NQT; ++qt)
#pragma unroll
            for (int b = 0; b < kTileBlocks; ++b) {
                f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < 4; ++ks)
                    a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                        __builtin_bit_cast(bf16x8, tr.k[b][ks]),
                        __builtin_bit_cast(bf16x8, qf[qt][ks]), a, 0, 0, 0);
                s[qt][b] = a;
            }
        // ---- stage V into LDS (zero rows past the context in the tail)
        if (tail) {
#pragma unroll
            for (int b = 0; b < kTileBlocks; ++b)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int tok = tok0 + b * kBS + 4 * i + g;
                    const bool live = (tok < ctx) && (blk_begin + kTileBlocks * t + b < blk_end);
                    if (!live) tr.v[b][i] = u32x4{0u, 0u, 0u, 0u};
                }
        }
#pragma unroll
        for (int b = 0; b < kTileBlocks; ++b)
#pragma unroll
            for (int i = 0; i < 4; ++i)
                *reinterpret_cast<u32x4 *>(lds + b * (kBS * kRowBytes) + lds_wr[i]) = tr.v[b][i];

        // ---- online softmax (scores scaled into the exp2 domain)
        u32x4 pfrag[NQT];
#pragma unroll
        for (int qt = 0; qt < NQT; ++qt) {
            float sv[8];
#pragma unroll
            for (int b = 0; b < kTileBlocks; ++b)
#pragma unroll
                for (int j = 0; j < 4; ++j) sv[4 * b + j] = s[qt][b][j] * scale_log2e;
            if (tail) {
                const int limit = ctx - T + t_of_row[qt];  // last visible key position
#pragma unroll
                for (int b = 0; b < kTileBlocks; ++b)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int tok = tok0 + b * kBS + 4 * g + j;
                        const bool vis = (tok <= limit) && (blk_begin + kTileBlocks * t + b < blk_end);
                        if (!vis) sv[4 * b + j] = -INFINITY;
                    }
            }
            float mx = sv[0];
#pragma unroll
            for (int j = 1; j < 8; ++j) mx = fmaxf(mx, sv[j]);
            mx = fmaxf(mx, __shfl_xor(mx, 16));
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            const float m_new = fmaxf(m_run[qt], mx);
            const float m_safe = (m_new == -INFINITY) ? 0.0f : m_new;
            const float alpha = fast_exp2(m_run[qt] - m_safe);
            float pe[8], psum = 0.0f;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                pe[j] = fast_exp2(sv[j] - m_safe);
                psum += pe[j];
            }
            l_run[qt] = l_run[qt] * alpha + psum;
            m_run[qt] = m_new;
#pragma unroll
            for (int dt = 0; dt < 8; ++dt) acc[qt][dt] *= alpha;
            pfrag[qt] = u32x4{pack_bf16x2(pe[0], pe[1]), pack_bf16x2(pe[2], pe[3]),
                              pack_bf16x2(pe[4], pe[5]), pack_bf16x2(pe[6], pe[7])};
        }
        // ---- O^T += V^T P^T
#pragma unroll
        for (int dt = 0; dt < 8; ++dt) {
            const unsigned off = tr_base + 16u * (unsigned)((2 * dt + tr_ch0) ^ tr_sw);
            typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
            const s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                (lds_s16x4 *)(lds + off));
            const s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                (lds_s16x4 *)(lds + kBS * kRowBytes + off));
            const s16x8 va = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
#pragma unroll
            for (int qt = 0; qt < NQT; ++qt)
                acc[qt][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                    __builtin_bit_cast(bf16x8, va), __builtin_bit_cast(bf16x8, pfrag[qt]),
                    acc[qt][dt], 0, 0, 0);
        }
    };

    // ---- cache policy (NT = adaptive).  The leading tiles whose blocks the neighbouring
    // sequence's table holds at the same positions are a shared prefix
    // (BlockManager::allocate_with_prefix): they are read with the default policy, so that the
    // sequences sharing them hit in L2 / Infinity Cache.  Everything after is private, read once,
    // and streams non-temporally (+11 % on config 3).  Speed only: either policy returns the same
    // bytes.  Each lane compares one tile's first block id; one ballot finds the prefix length.
    int nshared = NT ? 0 : ntiles;
    if (NT && ntiles > 0) {
        const int nb_seq = (seq ^ 1) < (int)gridDim.y ? (seq ^ 1) : seq;
        if (nb_seq != seq) {
            const int32_t *table_nb = block_tables + (int64_t)nb_seq * max_blocks_per_seq;
            nshared = ntiles;
            for (int base = 0; base < ntiles; base += 64) {
                const int t = base + lane;
                bool differs = false;
                if (t < ntiles) {
                    const int b0 = blk_begin + kTileBlocks * t;
                    differs = table[b0] != table_nb[b0];
                }
                const unsigned long long m = __ballot(differs);
                if (m) {
                    nshared = base + __builtin_ctzll(m);
                    break;
                }
            }
            nshared = __builtin_amdgcn_readfirstlane(nshared);
        }
    }

    // ---- main loop: register double buffering, ids two tiles ahead; tiles [t0, t1)
    auto run_tiles = [&](auto stream_tag, int t0, int t1) {
        if (t0 >= t1) return;
        if constexpr (NQT == 2) {
            // two query tiles run one wave per SIMD (512 registers): four waves per CU instead of
            // eight, so each keeps TWO tiles in flight (three register sets) to have the same
            // 128 KiB per CU on their way
            TileRegs ta, tb, tc;
            int a0, a1, b0, b1, c0, c1;  // ids of the tiles that go into ta / tb / tc next
            load_ids(t0, a0, a1);
            load_ids(t0 + 1, b0, b1);
            load_ids(t0 + 2, c0, c1);
            load_tile(stream_tag, ta, a0, a1);
            if (t0 + 1 < t1) load_tile(stream_tag, tb, b0, b1);
            load_ids(t0 + 3, a0, a1);
            load_ids(t0 + 4, b0, b1);
            for (int t = t0; t < t1; t += 3) {
                if (t + 2 < t1) load_tile(stream_tag, tc, c0, c1);
                load_ids(t + 5, c0, c1);
                compute_tile(ta, t);
                if (t + 1 >= t1) break;
                if (t + 3 < t1) load_tile(stream_tag, ta, a0, a1);
                load_ids(t + 6, a0, a1);
                compute_tile(tb, t + 1);
                if (t + 2 >= t1) break;
                if (t + 4 < t1) load_tile(stream_tag, tb, b0, b1);
                load_ids(t + 7, b0, b1);
                compute_tile(tc, t + 2);
            }
            return;
        }
        TileRegs ta, tb;
        int a0, a1, n0, n1;
        load_ids(t0, a0, a1);
        load_ids(t0 + 1, n0, n1);
        load_tile(stream_tag, ta, a0, a1);
        for (int t = t0; t < t1; t += 2) {
            int nn0, nn1;
            load_ids(t + 2, nn0, nn1);
            if (t + 1 < t1) load_tile(stream_tag, tb, n0, n1);
            compute_tile(ta, t);
            if (t + 1 >= t1) break;
            load_ids(t + 3, n0, n1);
            if (t + 2 < t1) load_tile(stream_tag, ta, nn0, nn1);
            compute_tile(tb, t + 1);
        }
    };
    // Half of the sequences (every other group of eight: workgroups go round-robin over the eight
    // XCDs, so each XCD's L2 sees both kinds) walk their private tiles first and the shared prefix
    // last; online softmax does not care about the order.  With everybody on the prefix first, HBM
    // sat idle while 256 workgroups drew the same 4 MiB from L2, and then all of them queued on
    // HBM: config 4 took the sum of the two phases, now the phases of the two halves overlap.
    const bool private_first = NT && nshared > 0 && nshared < ntiles && ((seq >> 3) & 1);
    const int cut = private_first ? ntiles : nshared;  // private tiles [nshared, cut) go first
    run_tiles(std::true_type{}, nshared, cut);
    run_tiles(std::false_type{}, 0, nshared);
    run_tiles(std::true_type{}, cut, ntiles);

    // ---- epilogue
#pragma unroll
    for (int qt = 0; qt < NQT; ++qt) {
        float l = l_run[qt];
        l += __shfl_xor(l, 16);
        l += __shfl_xor(l, 32);
        const int r = qt * 16 + c;
        if (r >= R) continue;
        const int t = r / G, hg = r - t * G;
        const int64_t orow = ((int64_t)seq * T + t) * Hq + kvh * G + hg;
        if (num_splits == 1) {
            const float inv = l > 0.0f ? 1.0f / l : 0.0f;
            __bf16 *op = out + orow * kD + 4 * g;
#pragma unroll
            for (int dt = 0; dt < 8; ++dt) {
                uint2 w;
                w.x = pack_bf16x2(acc[qt][dt][0] * inv, acc[qt][dt][1] * inv);
                w.y = pack_bf16x2(acc[qt][dt][2] * inv, acc[qt][dt][3] * inv);
                *reinterpret_cast<uint2 *>(op + 16 * dt) = w;
            }
        } else {
            const int64_t prow = orow * num_splits + split;
            float *op = ws_o + prow * kD + 4 * g;
#pragma unroll
            for (int dt = 0; dt < 8; ++dt)
                *reinterpret_cast<f32x4 *>(op + 16 * dt) = acc[qt][dt];
            if (g == 0) {
                ws_ml[2 * prow] = m_run[qt];
                ws_ml[2 * prow + 1] = l;
            }
        }
    }
}

// K2b: combine split-K partials.  One 4-wave workgroup per output row (seq, t, head): small
// launches are pure latency, so the row's splits are spread over the whole workgroup instead of
// walked one after the other.  Every wave reads the (m, l) pairs of all splits (at most 64,
// paged_attn_num_splits), one per lane, and reduces them to the row's max and denominator; then
// half-wave (w, hf) takes the splits s = 8i + 2w + hf, its lane dl owning d = 4 dl .. 4 dl + 3
// (16-byte loads, all independent), and the eight partial sums meet in LDS.
__global__ __launch_bounds__(256) void paged_attn_merge_kernel(const float *__restrict__ ws_o,
                                                               const float *__restrict__ ws_ml,
                                                               __bf16 *__restrict__ out,
                                                               int64_t num_rows, int num_splits) {
    __shared__ __attribute__((aligned(16))) float part[8][kD];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int hf = lane >> 5, dl = lane & 31;
    const int64_t row = blockIdx.x;
    float ms = -INFINITY, ls = 0.0f;
    if (lane < num_splits) {
        const float2 x = *reinterpret_cast<const float2 *>(ws_ml + 2 * (row * num_splits + lane));
        ms = x.x;
        ls = x.y;
    }
    float m = ms;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
    const float w_lane = (ms == -INFINITY) ? 0.0f : fast_exp2(ms - m);
    float L = ls * w_lane;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) L += __shfl_xor(L, off);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const int s0 = 2 * wave + hf;
#pragma unroll
    for (int i = 0; i < 8; ++i) {  // 64 splits at most
        const int sp = 8 * i + s0;
        const float w = __shfl(w_lane, sp & 63);
        if (sp < num_splits) {
            const f32x4 o = *reinterpret_cast<const f32x4 *>(ws_o + (row * num_splits + sp) * kD + 4 * dl);
            acc += o * w;
        }
    }
    *reinterpret_cast<f32x4 *>(&part[2 * wave + hf][4 * dl]) = acc;
    __syncthreads();
    if (threadIdx.x < 32) {
        f32x4 t = *reinterpret_cast<const f32x4 *>(&part[0][4 * dl]);
#pragma unroll
        for (int p = 1; p < 8; ++p) t += *reinterpret_cast<const f32x4 *>(&part[p][4 * dl]);
        const float inv = L > 0.0f ? 1.0f / L : 0.0f;
        uint2 y;
        y.x = pack_bf16x2(t[0] * inv, t[1] * inv);
        y.y = pack_bf16x2(t[2] * inv, t[3] * inv);
        *reinterpret_cast<uint2 *>(out + row * kD + 4 * dl) = y;
    }
}

// ---------------------------------------------------------------------------
// Generic path: one wave per (seq, t, head); any D <= 256, any block size.
// ---------------------------------------------------------------------------
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
__global__ __launch_bounds__(64) void paged_attn_generic_kernel(
    const ET *__restrict__ q, const ET *__restrict__ k_cache, const ET *__restrict__ v_cache,
    const int32_t *__restrict__ block_tables, const int32_t *__restrict__ context_lens,
    ET *__restrict__ out, int T, int Hq, int Hkv, int D, int block_size, int max_blocks_per_seq,
    int64_t num_cache_blocks, float scale) {
    __shared__ float q_s[256];
    __shared__ float p_s[64];
    const int lane = threadIdx.x;
    const int h = blockIdx.x, t = blockIdx.y, seq = blockIdx.z;
    const int kvh = h / (Hq / Hkv);
    const int ctx = context_lens[seq];
    ET *orow = out + (((int64_t)seq * T + t) * Hq + h) * D;
    if (ctx <= 0) {
        for (int d = lane; d < D; d += 64) orow[d] = from_f32<ET>(0.0f);
        return;
    }
    const ET *qrow = q + (((int64_t)seq * T + t) * Hq + h) * D;
    for (int d = lane; d < D; d += 64) q_s[d] = to_f32<ET>(qrow[d]);
    __syncthreads();
    const int32_t *table = block_tables + (int64_t)seq * max_blocks_per_seq;
    const int limit = (T > 1) ? (ctx - T + t) : (ctx - 1);  // last visible key (paged.rs:357-381)
    float m = -INFINITY, l = 0.0f;
    float o[4] = {0.f, 0.f, 0.f, 0.f};  // dims lane, lane+64, lane+128, lane+192
    for (int base = 0; base <= limit; base += 64) {
        const int tok = base + lane;
        float s = -INFINITY;
        if (tok <= limit) {
            int64_t pb = table[tok / block_size];
            if (pb < 0) pb = 0;
            if (pb >= num_cache_blocks) pb = num_cache_blocks - 1;
            const ET *krow = k_cache + ((pb * block_size + tok % block_size) * Hkv + kvh) * D;
            float acc = 0.0f;
            for (int d = 0; d < D; ++d) acc += q_s[d] * to_f32<ET>(krow[d]);
            s = acc * scale;
        }
        float mx = s;
        for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
        const float m_new = fmaxf(m, mx);
        const float alpha = expf(m - m_new);
        const float pj = (tok <= limit) ? expf(s - m_new) : 0.0f;
        float ps = pj;
        for (int off = 32; off > 0; off >>= 1) ps += __shfl_xor(ps, off);
        l = l * alpha + ps;
        m = m_new;
        __syncthreads();
        p_s[lane] = pj;
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] *= alpha;
        const int n = (limit - base + 1) < 64 ? (limit - base + 1) : 64;
        for (int j = 0; j < n; ++j) {
            const int tk = base + j;
            int64_t pb = table[tk / block_size];
            if (pb < 0) pb = 0;
            if (pb >= num_cache_blocks) pb = num_cache_blocks - 1;
            const ET *vrow = v_cache + ((pb * block_size + tk % block_size) * Hkv + kvh) * D;
            const float pw = p_s[j];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int d = lane + 64 * i;
                if (d < D) o[i] += pw * to_f32<ET>(vrow[d]);
            }
        }
    }
    const float inv = l > 0.0f ? 1.0f / l : 0.0f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int d = lane + 64 * i;
        if (d < D) orow[d] = from_f32<ET>(o[i] * inv);
    }
}

template <int NQT, int HPB>
void launch_fast(const PagedAttnArgs &a, int splits, int bps, float *ws_o, float *ws_ml,
                 bool nt, hipStream_t stream) {
    const dim3 grid(splits, a.num_seqs, (a.num_kv_heads + HPB - 1) / HPB), block(64 * HPB);
    const float scale_log2e = a.scale * 1.4426950408889634f;
    auto q = static_cast<const __bf16 *>(a.q);
    auto kc = static_cast<const unsigned char *>(a.k_cache);
    auto vc = static_cast<const unsigned char *>(a.v_cache);
    auto o = static_cast<__bf16 *>(a.out);
    if (nt)
        paged_attn_kernel<NQT, HPB, true><<<grid, block, 0, stream>>>(
            q, kc, vc, a.block_tables, a.context_lens, o, ws_o, ws_ml, a.num_new_tokens,
            a.num_heads, a.num_kv_heads, a.max_blocks_per_seq, a.num_cache_blocks, scale_log2e,
            splits, bps);
    else
        paged_attn_kernel<NQT, HPB, false><<<grid, block, 0, stream>>>(
            q, kc, vc, a.block_tables, a.context_lens, o, ws_o, ws_ml, a.num_new_tokens,
            a.num_heads, a.num_kv_heads, a.max_blocks_per_seq, a.num_cache_blocks, scale_log2e,
            splits, bps);
}

}  // namespace

bool paged_attn_fast_path(DType dt, int num_seqs, int T, int Hq, int Hkv, int D, int block_size,
                          int max_blocks_per_seq) {
    if (dt != DType::BF16 || D != kD || block_size != kBS) return false;
    if (Hkv <= 0 || Hq % Hkv != 0 || T <= 0) return false;
    const int R = T * (Hq / Hkv);
    const int forced = launch_policy().paged_max_rows;
    if (forced > 0) return R <= (forced < kMaxFastRows ? forced : kMaxFastRows);
    if (R <= 16) return true;
    if (R > kMaxFastRows) return false;
    // Two or three query tiles run 4 waves per workgroup; that pays where the launch is bound by
    // HBM (measured at 32Q/8KV: 256 x 2048 and 128 x 8192 tokens of context and up) and on long
    // contexts, where split-K spreads even one sequence over the chip (1 x 32768, T = 8: 40 us
    // against 76 us).  Other launches are faster in the prefill kernel over the cache
    // (32 x 4096, T = 12: 135 -> 112 us).
    const int64_t max_ctx = (int64_t)max_blocks_per_seq * kBS;
    return max_ctx >= kMultiTileMinContext || num_seqs * max_ctx >= kMultiTileMinTokens;
}

int paged_attn_num_splits(int num_seqs, int num_kv_heads, int max_blocks_per_seq) {
    const int forced = launch_policy().paged_splits;
    // aim for >= kWaveTarget (sequence, kv head, split) waves: 8 per CU on 256 CUs is one
    // resident 8-wave workgroup per CU, which measured best (config 3 then needs no merge pass)
    const long kWaveTarget = launch_policy().paged_wave_target;
    const long waves = (long)num_seqs * num_kv_heads;
    // at least 16 blocks (256 tokens) per split; a launch that would stay under a quarter of the
    // wave target even then (a handful of sequences: latency, not throughput) goes down to 4 blocks
    int min_blocks = 16;
    if (waves * (max_blocks_per_seq / 16 > 0 ? max_blocks_per_seq / 16 : 1) < kWaveTarget / 4) min_blocks = 4;
    int max_splits = max_blocks_per_seq / min_blocks;
    if (max_splits < 1) max_splits = 1;
    if (max_splits > 64) max_splits = 64;  // the merge kernel holds one split's (m, l) per lane
    if (forced > 0) return forced < max_splits ? forced : max_splits;
    long splits = (kWaveTarget + waves - 1) / (waves > 0 ? waves : 1);
    if (splits < 1) splits = 1;
    if (splits > max_splits) splits = max_splits;
    return (int)splits;
}

namespace {
// 49..128 query rows per kv head: the two-workgroups-per-CU kernel (paged_attn_mid.hip) where it
// measured faster than the prefill kernel over the cache (profiles/r03_paged_mid.txt, 32Q/8KV):
//   from 1.5 (sequence, kv head) units per CU: up to 64 rows (48 x 2048 T = 16: 78 against 96 us);
//   from 3 units per CU: up to 96 rows (three of its four waves compute);
//   from 4 units per CU, i.e. two rounds of two workgroups: up to 128 rows when no context is longer
//   than 2048 tokens (with all four waves computing on longer walks the two kernels are level).
// Fewer units go to the prefill kernel, whose key split spreads them over the chip (16 x 2048: 38
// against 49 us).
bool use_mid(DType dt, const PagedAttnArgs &a) {
    const int mode = launch_policy().paged_mid;
    if (mode == 0 || launch_policy().paged_flash == 0 || !paged_attn_mid_supported(dt, a)) return false;
    if (mode > 0) return true;
    const int64_t units = (int64_t)a.num_seqs * a.num_kv_heads, cus = device_cus();
    const int rows = a.num_new_tokens * (a.num_heads / a.num_kv_heads);
    if (rows <= 64) return 2 * units >= 3 * cus;
    if (rows <= 96) return units >= 3 * cus;
    return units >= 4 * cus && (int64_t)a.max_blocks_per_seq * a.block_size <= 2048;
}
}  // namespace

PagedPath paged_attn_path(DType dt, const PagedAttnArgs &a) {
    if (launch_policy().paged_mid > 0 && use_mid(dt, a)) return PagedPath::Mid;  // forced: also for 33..48 rows
    if (paged_attn_fast_path(dt, a.num_seqs, a.num_new_tokens, a.num_heads, a.num_kv_heads, a.head_dim,
                             a.block_size, a.max_blocks_per_seq))
        return PagedPath::Decode;
    if (use_mid(dt, a)) return PagedPath::Mid;
    if (launch_policy().paged_flash != 0 && flash_attn_paged_supported(dt, a)) return PagedPath::K1Paged;
    return PagedPath::Generic;
}

// An upper bound over everything the shape alone does not say (cache size, pointer alignment): the
// launch may need less (the generic kernel needs none), never more.  The split factors depend on the
// CU count of the CURRENT device: query with the launch device current.
size_t paged_attn_workspace_bytes(int num_seqs, int T, int Hq, int Hkv, int D, int block_size,
                                  int max_blocks_per_seq) {
    if (!paged_attn_fast_path(DType::BF16, num_seqs, T, Hq, Hkv, D, block_size, max_blocks_per_seq)) {
        // more query rows per kv head than the decode kernel takes: the prefill kernel over the
        // cache, which splits the key walk of a small launch and needs room for the partials
        if (launch_policy().paged_flash == 0) return 0;  // generic kernel: no workspace
        if (D == kD && block_size == kBS && Hkv > 0 && Hq % Hkv == 0 && T > 0 && num_seqs > 0) {
            PagedAttnArgs a{};
            a.num_seqs = num_seqs;
            a.num_new_tokens = T;
            a.num_heads = Hq;
            a.num_kv_heads = Hkv;
            a.max_blocks_per_seq = max_blocks_per_seq;
            return flash_attn_paged_workspace_bytes(a);
        }
        return 0;
    }
    const int splits = paged_attn_num_splits(num_seqs, Hkv, max_blocks_per_seq);
    if (splits <= 1) return 0;
    const size_t rows = (size_t)num_seqs * T * Hq * splits;
    return rows * (kD + 2) * sizeof(float);
}

// ---------------------------------------------------------------------------
// Scratch of the entry point that allocates nothing (nvllm_paged_attn_bf16_noalloc): the split-K
// partials live in this file-scope device array instead of a caller workspace.  Every GPU the
// library runs on holds its own copy (kNoAllocScratchBytes, until the process ends) at its own
// address, so host code never uses the array's name as an address: the address is asked for once
// per device with hipGetSymbolAddress and only that pointer is handed to the kernels.
// ---------------------------------------------------------------------------
static __device__ __align__(256) unsigned char scratch[kNoAllocScratchBytes];

namespace {
// device address of `scratch` on the CURRENT device (looked up on the first call per device)
hipError_t noalloc_scratch(void **out) {
    constexpr int kMaxDevices = 64;
    static std::atomic<void *> ptrs[kMaxDevices];
    int dev = 0;
    hipError_t err = hipGetDevice(&dev);
    if (err != hipSuccess) return err;
    if (dev < 0 || dev >= kMaxDevices) return hipErrorInvalidDevice;
    void *p = ptrs[dev].load(std::memory_order_acquire);
    if (!p) {
        err = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (err != hipSuccess) return err;
        if (!p) return hipErrorInvalidSymbol;
        ptrs[dev].store(p, std::memory_order_release);
    }
    *out = p;
    return hipSuccess;
}
}  // namespace

// resolves the scratch address of the current device ahead of the first launch (nvllm_set_device
// calls it), so that no symbol lookup happens on a launch path -- e.g. under stream capture
hipError_t noalloc_prepare_device() {
    void *p = nullptr;
    return noalloc_scratch(&p);
}

hipError_t launch_paged_attn_noalloc(DType dt, const PagedAttnArgs &args, hipStream_t stream) {
    PagedAttnArgs a = args;
    a.workspace = nullptr;
    a.workspace_bytes = 0;
    const size_t need = paged_attn_launch_workspace_bytes(dt, a);
    if (need > kNoAllocScratchBytes) return hipErrorOutOfMemory;  // the C ABI rejects this before
    if (need > 0) {
        void *p = nullptr;
        const hipError_t err = noalloc_scratch(&p);
        if (err != hipSuccess) return err;
        a.workspace = p;  // every partial the merge pass reads is written by this launch first:
        a.workspace_bytes = kNoAllocScratchBytes;  // nothing in it has to start at zero
    }
    return launch_paged_attn(dt, a, stream);
}

size_t paged_attn_launch_workspace_bytes(DType dt, const PagedAttnArgs &a) {
    const PagedPath path = paged_attn_path(dt, a);
    if (dt != DType::BF16 || path == PagedPath::Generic || path == PagedPath::Mid) return 0;
    return paged_attn_workspace_bytes(a.num_seqs, a.num_new_tokens, a.num_heads, a.num_kv_heads, a.head_dim,
                                      a.block_size, a.max_blocks_per_seq);
}

hipError_t launch_paged_attn(DType dt, const PagedAttnArgs &a, hipStream_t stream) {
    if (a.num_seqs <= 0) return hipSuccess;
    if (launch_policy().paged_mid > 0 && use_mid(dt, a)) return launch_paged_attn_mid(a, stream);
    if (paged_attn_fast_path(dt, a.num_seqs, a.num_new_tokens, a.num_heads, a.num_kv_heads, a.head_dim,
                             a.block_size, a.max_blocks_per_seq)) {
        const int splits = paged_attn_num_splits(a.num_seqs, a.num_kv_heads, a.max_blocks_per_seq);
        const size_t rows = (size_t)a.num_seqs * a.num_new_tokens * a.num_heads;
        // the C ABI has checked the workspace against paged_attn_workspace_bytes (same policy, same
        // arithmetic); a direct caller that did not gets an error, never a silently different launch
        if (splits > 1 && (!a.workspace || a.workspace_bytes < rows * splits * (kD + 2) * sizeof(float) ||
                           (reinterpret_cast<uintptr_t>(a.workspace) & 15u) != 0))
            return hipErrorInvalidValue;
        int bps = (a.max_blocks_per_seq + splits - 1) / splits;
        bps = (bps + kTileBlocks - 1) / kTileBlocks * kTileBlocks;
        if (bps < kTileBlocks) bps = kTileBlocks;
        float *ws_o = static_cast<float *>(a.workspace);
        float *ws_ml = ws_o ? ws_o + rows * splits * kD : nullptr;
        // adaptive non-temporal K/V loads, unless the K/V the tables can reach fits the 256 MiB
        // Infinity Cache anyway (then a later layer / step may still find it there)
        const int64_t reach = (int64_t)a.num_seqs * a.max_blocks_per_seq * 2 * kBS * a.num_kv_heads * kRowBytes;
        const int nt_forced = launch_policy().paged_nt;
        const bool nt = nt_forced >= 0 ? nt_forced != 0 : reach > ((int64_t)256 << 20);
        const int R = a.num_new_tokens * (a.num_heads / a.num_kv_heads);
        // kv heads per workgroup: 8 waves sharing a block's rows is the throughput shape; a launch
        // of fewer workgroups than CUs spreads its heads over more of them instead
        int hpb = launch_policy().paged_hpb;
        if (hpb != 1 && hpb != 2 && hpb != 4 && hpb != 8) {
            hpb = a.num_kv_heads >= 8 ? 8 : (a.num_kv_heads >= 4 ? 4 : (a.num_kv_heads >= 2 ? 2 : 1));
            const int num_cus = device_cus();
            while (hpb > 1 && (long)splits * a.num_seqs * ((a.num_kv_heads + hpb - 1) / hpb) < num_cus) hpb /= 2;
        }
        // two or more query tiles need the 512-register budget of a 4-wave workgroup (with 8 waves
        // the kernel spilled and ran at half the bandwidth)
        if (R > 16 && hpb > 4) hpb = 4;
        const int nqt = (R + 15) / 16;
#define NVLLM_PAGED_CASE(NQT, HPB) \
    case NQT * 16 + HPB: launch_fast<NQT, HPB>(a, splits, bps, ws_o, ws_ml, nt, stream); break;
        switch (nqt * 16 + hpb) {
            NVLLM_PAGED_CASE(1, 8) NVLLM_PAGED_CASE(1, 4) NVLLM_PAGED_CASE(1, 2) NVLLM_PAGED_CASE(1, 1)
            NVLLM_PAGED_CASE(2, 4) NVLLM_PAGED_CASE(2, 2) NVLLM_PAGED_CASE(2, 1)
            NVLLM_PAGED_CASE(3, 4) NVLLM_PAGED_CASE(3, 2) NVLLM_PAGED_CASE(3, 1)
            default: return hipErrorInvalidValue;
        }
#undef NVLLM_PAGED_CASE
        hipError_t err = hipGetLastError();
        if (err != hipSuccess) return err;
        if (splits > 1) {
            const dim3 grid((unsigned)rows), block(256);
            paged_attn_merge_kernel<<<grid, block, 0, stream>>>(
                ws_o, ws_ml, static_cast<__bf16 *>(a.out), (int64_t)rows, splits);
            err = hipGetLastError();
        }
        return err;
    }
    if (use_mid(dt, a)) return launch_paged_attn_mid(a, stream);
    // more query rows per kv head (chunked prefill, long verify): the prefill kernel
    // reading K/V through the block table
    if (launch_policy().paged_flash != 0 && flash_attn_paged_supported(dt, a))
        return launch_flash_attn_paged(a, stream);
    // generic
    const dim3 grid(a.num_heads, a.num_new_tokens, a.num_seqs), block(64);
    if (dt == DType::BF16)
        paged_attn_generic_kernel<__bf16><<<grid, block, 0, stream>>>(
            static_cast<const __bf16 *>(a.q), static_cast<const __bf16 *>(a.k_cache),
            static_cast<const __bf16 *>(a.v_cache), a.block_tables, a.context_lens,
            static_cast<__bf16 *>(a.out), a.num_new_tokens, a.num_heads, a.num_kv_heads,
            a.head_dim, a.block_size, a.max_blocks_per_seq, a.num_cache_blocks, a.scale);
    else
        paged_attn_generic_kernel<float><<<grid, block, 0, stream>>>(
            static_cast<const float *>(a.q), static_cast<const float *>(a.k_cache),
            static_cast<const float *>(a.v_cache), a.block_tables, a.context_lens,
            static_cast<float *>(a.out), a.num_new_tokens, a.num_heads, a.num_kv_heads,
            a.head_dim, a.block_size, a.max_blocks_per_seq, a.num_cache_blocks, a.scale);
    return hipGetLastError();
}

// nvllm_debug_validate_tables: the attention kernels clamp an id outside the pool (or read it as
// zeros) and a context length outside its table row, so a wrong table gives plausible wrong output
// instead of a fault.  This pass counts what they would have had to clamp: counts[0] = ids < 0 or
// >= num_cache_blocks among the blocks a sequence's context length says it uses, counts[1] =
// context lengths < 0 or beyond max_blocks_per_seq * block_size.  One wave per sequence.
namespace {
__global__ __launch_bounds__(64) void validate_tables_kernel(const int32_t *__restrict__ block_tables,
                                                             const int32_t *__restrict__ context_lens,
                                                             int max_blocks_per_seq, int block_size,
                                                             int64_t num_cache_blocks, int32_t *__restrict__ counts) {
    const int seq = blockIdx.x;
    const int ctx = context_lens[seq];
    const int64_t cap = (int64_t)max_blocks_per_seq * block_size;
    const bool bad_len = ctx < 0 || ctx > cap;
    if (bad_len && threadIdx.x == 0) atomicAdd(counts + 1, 1);
    const int used = bad_len ? (ctx < 0 ? 0 : max_blocks_per_seq) : (ctx + block_size - 1) / block_size;
    int bad = 0;
    for (int i = threadIdx.x; i < used; i += 64) {
        const int64_t id = block_tables[(int64_t)seq * max_blocks_per_seq + i];
        bad += id < 0 || id >= num_cache_blocks;
    }
    for (int off = 32; off > 0; off >>= 1) bad += __shfl_xor(bad, off);
    if (bad && threadIdx.x == 0) atomicAdd(counts, bad);
}
}  // namespace

hipError_t launch_validate_tables(const int32_t *block_tables, const int32_t *context_lens, int num_seqs,
                                  int max_blocks_per_seq, int block_size, int64_t num_cache_blocks,
                                  int32_t *counts, hipStream_t stream) {
    hipError_t err = hipMemsetAsync(counts, 0, 2 * sizeof(int32_t), stream);
    if (err != hipSuccess) return err;
    validate_tables_kernel<<<dim3((unsigned)num_seqs), dim3(64), 0, stream>>>(
        block_tables, context_lens, max_blocks_per_seq, block_size, num_cache_blocks, counts);
    return hipGetLastError();
}

}  // namespace kernels
}  // namespace nvllm
