// K4 lab: mappings and cache policies for the KV scatter (reshape_and_cache), timed against a
// straight copy of the same bytes.  Row = Hkv*D*2 = 2 KiB (8 kv heads x 128, bf16), block = 16 slots.
// Not product code; the winner is carried into csrc/reshape_and_cache.hip by hand.
//   hipcc -O3 --offload-arch=gfx950 -o k4_variants k4_variants.hip && ./k4_variants
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <random>
#include <vector>

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
constexpr int kRowVec = 128;  // 16-byte vectors per 2 KiB row

#define CK(x)                                                                                \
    do {                                                                                     \
        hipError_t e_ = (x);                                                                 \
        if (e_ != hipSuccess) {                                                              \
            std::fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
            std::exit(1);                                                                    \
        }                                                                                    \
    } while (0)

template <bool NT>
__device__ __forceinline__ u32x4 ld(const u32x4 *p) {
    if constexpr (NT) return __builtin_nontemporal_load(p);
    else return *p;
}
template <bool NT>
__device__ __forceinline__ void st(u32x4 *p, u32x4 x) {
    if constexpr (NT) __builtin_nontemporal_store(x, p);
    else *p = x;
}

// V0: the product mapping (flat, one K and one V vector per thread)
template <bool NTL, bool NTS>
__global__ __launch_bounds__(256) void k4_flat(const u32x4 *__restrict__ key, const u32x4 *__restrict__ value,
                                               u32x4 *__restrict__ kc, u32x4 *__restrict__ vc,
                                               const int64_t *__restrict__ slots, int64_t ntok,
                                               int64_t num_slots) {
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (gid >= ntok * kRowVec) return;
    const int64_t tok = (unsigned)gid / (unsigned)kRowVec;
    const int idx = (int)(gid - tok * kRowVec);
    const u32x4 kx = ld<NTL>(key + gid), vx = ld<NTL>(value + gid);
    const int64_t slot = slots[tok];
    if (slot < 0 || slot >= num_slots) return;
    st<NTS>(kc + slot * kRowVec + idx, kx);
    st<NTS>(vc + slot * kRowVec + idx, vx);
}

// V1..: one wave moves TPW whole tokens; every load is issued before the first store; the slot
// is a scalar load.
template <int TPW, bool NTL, bool NTS>
__global__ __launch_bounds__(256) void k4_wave(const u32x4 *__restrict__ key, const u32x4 *__restrict__ value,
                                               u32x4 *__restrict__ kc, u32x4 *__restrict__ vc,
                                               const int64_t *__restrict__ slots, int64_t ntok,
                                               int64_t num_slots) {
    const int lane = threadIdx.x & 63;
    const int64_t w = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    const int64_t t0 = w * TPW;
    u32x4 k[TPW][2], v[TPW][2];
    int64_t s[TPW];
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
        const int64_t t = t0 + i;
        if (t < ntok) {
            s[i] = slots[t];
            k[i][0] = ld<NTL>(key + t * kRowVec + lane);
            k[i][1] = ld<NTL>(key + t * kRowVec + 64 + lane);
            v[i][0] = ld<NTL>(value + t * kRowVec + lane);
            v[i][1] = ld<NTL>(value + t * kRowVec + 64 + lane);
        } else {
            s[i] = -1;
        }
    }
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
        if (s[i] < 0 || s[i] >= num_slots) continue;
        st<NTS>(kc + s[i] * kRowVec + lane, k[i][0]);
        st<NTS>(kc + s[i] * kRowVec + 64 + lane, k[i][1]);
        st<NTS>(vc + s[i] * kRowVec + lane, v[i][0]);
        st<NTS>(vc + s[i] * kRowVec + 64 + lane, v[i][1]);
    }
}

// V5: resident waves walk the tokens with a stride; the next token's loads are in flight while
// this token's stores go out.
template <bool NTL, bool NTS>
__global__ __launch_bounds__(256) void k4_resident(const u32x4 *__restrict__ key, const u32x4 *__restrict__ value,
                                                   u32x4 *__restrict__ kc, u32x4 *__restrict__ vc,
                                                   const int64_t *__restrict__ slots, int64_t ntok,
                                                   int64_t num_slots) {
    const int lane = threadIdx.x & 63;
    const int64_t w = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    const int64_t stride = (int64_t)gridDim.x * 4;
    int64_t t = w;
    if (t >= ntok) return;
    u32x4 k0 = ld<NTL>(key + t * kRowVec + lane), k1 = ld<NTL>(key + t * kRowVec + 64 + lane);
    u32x4 v0 = ld<NTL>(value + t * kRowVec + lane), v1 = ld<NTL>(value + t * kRowVec + 64 + lane);
    int64_t s = slots[t];
    for (;;) {
        const int64_t tn = t + stride;
        u32x4 nk0, nk1, nv0, nv1;
        int64_t ns = -1;
        if (tn < ntok) {
            nk0 = ld<NTL>(key + tn * kRowVec + lane), nk1 = ld<NTL>(key + tn * kRowVec + 64 + lane);
            nv0 = ld<NTL>(value + tn * kRowVec + lane), nv1 = ld<NTL>(value + tn * kRowVec + 64 + lane);
            ns = slots[tn];
        }
        if (s >= 0 && s < num_slots) {
            st<NTS>(kc + s * kRowVec + lane, k0);
            st<NTS>(kc + s * kRowVec + 64 + lane, k1);
            st<NTS>(vc + s * kRowVec + lane, v0);
            st<NTS>(vc + s * kRowVec + 64 + lane, v1);
        }
        if (tn >= ntok) break;
        t = tn, s = ns, k0 = nk0, k1 = nk1, v0 = nv0, v1 = nv1;
    }
}

// the ceiling: the same bytes, destination in order
__global__ __launch_bounds__(256) void k4_copy(const u32x4 *__restrict__ key, const u32x4 *__restrict__ value,
                                               u32x4 *__restrict__ kc, u32x4 *__restrict__ vc, int64_t nvec) {
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (gid >= nvec) return;
    const u32x4 kx = __builtin_nontemporal_load(key + gid), vx = __builtin_nontemporal_load(value + gid);
    __builtin_nontemporal_store(kx, kc + gid);
    __builtin_nontemporal_store(vx, vc + gid);
}

struct Set {
    u32x4 *k, *v, *kc, *vc;
};

int main(int argc, char **argv) {
    const int bs = 16;
    std::vector<int64_t> sizes = {4096, 32768, 262144};
    if (argc > 2) {
        sizes.clear();
        for (int i = 2; i < argc; ++i) sizes.push_back(std::atoll(argv[i]));
    }
    for (int64_t n : sizes) {
        const int64_t nb = std::max<int64_t>(n / bs * 2, 4096), num_slots = nb * bs;
        const int nset = n >= 262144 ? 2 : 4;
        std::vector<Set> sets(nset);
        const size_t in_bytes = (size_t)n * kRowVec * 16, pool_bytes = (size_t)num_slots * kRowVec * 16;
        for (auto &s : sets) {
            CK(hipMalloc(&s.k, in_bytes)); CK(hipMalloc(&s.v, in_bytes));
            CK(hipMalloc(&s.kc, pool_bytes)); CK(hipMalloc(&s.vc, pool_bytes));
            CK(hipMemset(s.k, 0x11, in_bytes)); CK(hipMemset(s.v, 0x22, in_bytes));
            CK(hipMemset(s.kc, 0, pool_bytes)); CK(hipMemset(s.vc, 0, pool_bytes));
        }
        std::mt19937_64 rng(7);
        std::vector<int64_t> h_random(num_slots), h_blocks;
        std::iota(h_random.begin(), h_random.end(), 0);
        std::shuffle(h_random.begin(), h_random.end(), rng);
        h_random.resize(n);
        std::vector<int64_t> blk(nb);
        std::iota(blk.begin(), blk.end(), 0);
        std::shuffle(blk.begin(), blk.end(), rng);
        for (int64_t i = 0; i < n; ++i) h_blocks.push_back(blk[i / bs] * bs + i % bs);
        std::vector<int64_t> h_ident(n), h_sorted(h_random), h_blocks_asc;
        std::iota(h_ident.begin(), h_ident.end(), 0);
        std::sort(h_sorted.begin(), h_sorted.end());
        {
            std::vector<int64_t> b(blk.begin(), blk.begin() + (n + bs - 1) / bs);
            std::sort(b.begin(), b.end());
            for (int64_t i = 0; i < n; ++i) h_blocks_asc.push_back(b[i / bs] * bs + i % bs);
        }
        int64_t *d_ident, *d_sorted, *d_blocks_asc;
        CK(hipMalloc(&d_ident, n * 8)); CK(hipMalloc(&d_sorted, n * 8)); CK(hipMalloc(&d_blocks_asc, n * 8));
        CK(hipMemcpy(d_ident, h_ident.data(), n * 8, hipMemcpyHostToDevice));
        CK(hipMemcpy(d_sorted, h_sorted.data(), n * 8, hipMemcpyHostToDevice));
        CK(hipMemcpy(d_blocks_asc, h_blocks_asc.data(), n * 8, hipMemcpyHostToDevice));
        int64_t *d_random, *d_blocks;
        CK(hipMalloc(&d_random, n * 8)); CK(hipMalloc(&d_blocks, n * 8));
        CK(hipMemcpy(d_random, h_random.data(), n * 8, hipMemcpyHostToDevice));
        CK(hipMemcpy(d_blocks, h_blocks.data(), n * 8, hipMemcpyHostToDevice));
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        const double bytes = (double)n * (4.0 * kRowVec * 16 + 8);

        auto time = [&](const char *name, const char *kind, auto launch) {
            for (int i = 0; i < 6; ++i) launch(sets[i % nset]);
            CK(hipDeviceSynchronize());
            const int iters = 30;
            CK(hipEventRecord(e0));
            for (int i = 0; i < iters; ++i) launch(sets[i % nset]);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            CK(hipGetLastError());
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            const double us = ms * 1e3 / iters;
            std::printf("tokens %7lld  %-7s %-22s %9.2f us  %6.2f TB/s  frac %.3f\n", (long long)n, kind, name,
                        us, bytes / us * 1e-6, bytes / us * 1e-6 / 8.0);
            std::fflush(stdout);
        };
        const unsigned flat_grid = (unsigned)((n * kRowVec + 255) / 256);
        time("copy (ceiling)", "-", [&](Set &s) {
            k4_copy<<<flat_grid, 256>>>(s.k, s.v, s.kc, s.vc, n * kRowVec);
        });
        const bool quick = argc > 1;
        for (int pat = 0; pat < 5; ++pat) {
            const int64_t *pats[] = {d_random, d_blocks, d_ident, d_sorted, d_blocks_asc};
            const char *kinds[] = {"random", "blocks", "ident", "sorted", "blk-asc"};
            const int64_t *sl = pats[pat];
            const char *kind = kinds[pat];
#define FLAT(L, S)                                                                               \
    time("flat ntl" #L " nts" #S, kind, [&](Set &s) {                                           \
        k4_flat<L, S><<<flat_grid, 256>>>(s.k, s.v, s.kc, s.vc, sl, n, num_slots);               \
    })
#define WAVE(T, L, S)                                                                            \
    time("wave tpw" #T " ntl" #L " nts" #S, kind, [&](Set &s) {                                 \
        k4_wave<T, L, S><<<(unsigned)((n + 4 * T - 1) / (4 * T)), 256>>>(s.k, s.v, s.kc, s.vc, sl, n, num_slots); \
    })
#define RES(G, L, S)                                                                             \
    time("resident g" #G " ntl" #L " nts" #S, kind, [&](Set &s) {                               \
        k4_resident<L, S><<<G, 256>>>(s.k, s.v, s.kc, s.vc, sl, n, num_slots);                   \
    })
            FLAT(1, 1);
            WAVE(2, 1, 1);
            if (quick && pat < 2) {
                FLAT(0, 0);
                WAVE(1, 1, 1);
                WAVE(1, 0, 0);
                WAVE(2, 0, 0);
                WAVE(2, 0, 1);
            }
            if (quick || pat >= 2) continue;
            FLAT(1, 0);
            FLAT(0, 1);
            FLAT(0, 0);
            WAVE(1, 1, 1);
            WAVE(4, 1, 1);
            WAVE(1, 0, 0);
            WAVE(2, 0, 0);
            WAVE(2, 1, 0);
            WAVE(2, 0, 1);
            RES(512, 1, 1);
            RES(1024, 1, 1);
            RES(2048, 1, 1);
            RES(2048, 0, 0);
        }
        for (auto &s : sets) { CK(hipFree(s.k)); CK(hipFree(s.v)); CK(hipFree(s.kc)); CK(hipFree(s.vc)); }
        CK(hipFree(d_random)); CK(hipFree(d_blocks)); CK(hipFree(d_ident)); CK(hipFree(d_sorted)); CK(hipFree(d_blocks_asc));
    }
    return 0;
}
