// k1_lab -- A/B bench for variants of the K1 flash-attention prefill kernel (development tool,
// not part of the product library).  Every variant is one build of a flash_attn_mfma*.hip source
// whose launcher/supported symbols were renamed on the hipcc command line (build.sh); all variants
// run interleaved in ONE process on the same random data and are checked against the generic f32
// kernel of csrc/flash_attn.hip.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include "nvllm_kernels.hpp"

using nvllm::kernels::DType;
using nvllm::kernels::FlashAttnArgs;
using nvllm::kernels::PagedAttnArgs;

namespace nvllm {
namespace kernels {
#define LAB_VARIANT(name)                                                   \
    hipError_t launch_##name(const FlashAttnArgs &a, hipStream_t stream);  \
    hipError_t launch_paged_##name(const PagedAttnArgs &a, hipStream_t stream); \
    bool supported_##name(const FlashAttnArgs &a);
#include "variants.inc"
#undef LAB_VARIANT
// flash_attn.hip (generic reference) is built with its fast-path hooks renamed to these stubs
hipError_t lab_unused_launch(const FlashAttnArgs &, hipStream_t) { return hipErrorNotSupported; }
bool lab_unused_supported(const FlashAttnArgs &) { return false; }
}  // namespace kernels
}  // namespace nvllm

struct Variant {
    const char *name;
    hipError_t (*launch)(const FlashAttnArgs &, hipStream_t);
    bool (*supported)(const FlashAttnArgs &);
    hipError_t (*launch_paged)(const PagedAttnArgs &, hipStream_t);
};
static const Variant kVariants[] = {
#define LAB_VARIANT(name) {#name, nvllm::kernels::launch_##name, nvllm::kernels::supported_##name, nvllm::kernels::launch_paged_##name},
#include "variants.inc"
#undef LAB_VARIANT
};

#define CK(x)                                                                      \
    do {                                                                           \
        hipError_t e_ = (x);                                                       \
        if (e_ != hipSuccess) {                                                    \
            fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
            exit(2);                                                               \
        }                                                                          \
    } while (0)

static uint16_t f2bf(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    u += 0x7FFF + ((u >> 16) & 1);
    return (uint16_t)(u >> 16);
}
static float bf2f(uint16_t h) {
    uint32_t u = (uint32_t)h << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}

// LAB_PAGED=T: the paged variant (K1 over the cache): B sequences x S cached tokens, T new tokens
// each, identity block tables; timing and stamps only (correctness is the test suite's job)
static int paged_main(int N, int ctx, int Hq, int Hkv, int T, int rounds, int iters) {
    const int D = 128, bs = 16, nblk = ctx / bs;
    const size_t nq = (size_t)N * T * Hq * D, nc = (size_t)N * nblk * bs * Hkv * D;
    std::vector<uint16_t> hq(nq), hc(nc);
    std::mt19937 rng(99);
    std::normal_distribution<float> nd(0.f, 1.f);
    for (auto &x : hq) x = f2bf(nd(rng));
    for (size_t i = 0; i < nc; ++i) hc[i] = f2bf(nd(rng));
    std::vector<int32_t> tab((size_t)N * nblk), cl(N, ctx);
    for (int s = 0; s < N; ++s)
        for (int i = 0; i < nblk; ++i) tab[(size_t)s * nblk + i] = s * nblk + ((i * 37) % nblk);
    uint16_t *dq, *dk, *dv, *dout; int32_t *dtab, *dcl;
    CK(hipMalloc(&dq, nq * 2)); CK(hipMalloc(&dk, nc * 2)); CK(hipMalloc(&dv, nc * 2)); CK(hipMalloc(&dout, nq * 2));
    CK(hipMalloc(&dtab, tab.size() * 4)); CK(hipMalloc(&dcl, cl.size() * 4));
    CK(hipMemcpy(dq, hq.data(), nq * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dk, hc.data(), nc * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dv, hc.data(), nc * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dtab, tab.data(), tab.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dcl, cl.data(), cl.size() * 4, hipMemcpyHostToDevice));
    const size_t nwaves = (size_t)N * 64 * 4;  // grid.x <= 64 here
    unsigned long long *dd;
    CK(hipMalloc(&dd, nwaves * 128));
    PagedAttnArgs a{};
    a.q = dq; a.k_cache = dk; a.v_cache = dv; a.block_tables = dtab; a.context_lens = dcl; a.out = dout;
    a.workspace = nullptr; a.workspace_bytes = 0; a.num_seqs = N; a.num_new_tokens = T; a.num_heads = Hq;
    a.num_kv_heads = Hkv; a.head_dim = D; a.block_size = bs; a.max_blocks_per_seq = nblk;
    a.num_cache_blocks = (int64_t)N * nblk; a.scale = 1.0f / sqrtf((float)D);
    const double flops = 4.0 * N * T * Hq * (double)ctx * D;
    printf("paged shape N=%d ctx=%d T=%d Hq=%d Hkv=%d (%.2f GFLOP)\n", N, ctx, T, Hq, Hkv, flops / 1e9);
    const int nv = sizeof(kVariants) / sizeof(kVariants[0]);
    for (int vi = 0; vi < nv; ++vi) {
        const Variant &v = kVariants[vi];
        if (strncmp(v.name, "stamp", 5) != 0) continue;
        CK(hipMemset(dd, 0, nwaves * 128));
        PagedAttnArgs sa = a;
        sa.workspace = dd;  // stamp builds take the stamp buffer here
        for (int i = 0; i < 3; ++i) CK(v.launch_paged(sa, 0));
        CK(hipDeviceSynchronize());
        std::vector<unsigned long long> hd(nwaves * 16);
        CK(hipMemcpy(hd.data(), dd, nwaves * 128, hipMemcpyDeviceToHost));
        double sum[16] = {0}; size_t n = 0;
        for (size_t w = 0; w < nwaves; ++w) {
            if (hd[w * 16 + 15] == 0) continue;
            for (int i = 0; i < 16; ++i) sum[i] += (double)hd[w * 16 + i];
            ++n;
        }
        if (!n) continue;
        if (strstr(v.name, "rt")) {  // wall-clock stamps (10 ns ticks) in slots 4 / 9
            unsigned long long t0 = ~0ull, t1 = 0;
            for (size_t w = 0; w < nwaves; ++w)
                if (hd[w * 16 + 15]) { t0 = std::min(t0, hd[w * 16 + 4]); t1 = std::max(t1, hd[w * 16 + 9]); }
            double dur = 0; int late[6] = {0};
            for (size_t w = 0; w < nwaves; ++w) {
                if (!hd[w * 16 + 15]) continue;
                dur += (double)(hd[w * 16 + 9] - hd[w * 16 + 4]);
                const double off = (double)(hd[w * 16 + 4] - t0) * 0.01;
                ++late[off < 5 ? 0 : off < 20 ? 1 : off < 40 ? 2 : off < 60 ? 3 : off < 100 ? 4 : 5];
            }
            printf("%-14s wall: first start .. last end %.1f us, mean wave %.1f us; start offsets <5us:%d <20:%d <40:%d <60:%d <100:%d more:%d\n",
                   v.name, (double)(t1 - t0) * 0.01, dur / n * 0.01, late[0], late[1], late[2], late[3], late[4], late[5]);
            sum[4] = sum[9] = 0;
        }
        printf("%-14s stamps over %zu waves: steps/wave %.1f, c12=%.1f c13=%.1f s14=%.0f\n", v.name, n, sum[15] / n, sum[12] / n, sum[13] / n, sum[14] / n);
        printf("%-14s   cycles per wave: ", v.name);
        double tot = 0;
        for (int i = 0; i < 12; ++i) { printf("s%d=%.0f ", i, sum[i] / n); tot += sum[i]; }
        printf("total=%.0f\n", tot / n);
    }
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int vi = 0; vi < nv; ++vi) {
        const Variant &v = kVariants[vi];
        PagedAttnArgs sa = a;
        if (strncmp(v.name, "stamp", 5) == 0) sa.workspace = dd;
        std::vector<double> t;
        for (int r = 0; r < rounds + 1; ++r) {
            CK(hipEventRecord(e0, 0));
            for (int i = 0; i < iters; ++i) CK(v.launch_paged(sa, 0));
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (r > 0) t.push_back(ms / iters);
        }
        std::sort(t.begin(), t.end());
        printf("%-14s median %8.1f us  %7.1f TF\n", v.name, t[t.size() / 2] * 1e3, flops / (t[t.size() / 2] * 1e-3) / 1e12);
    }
    return 0;
}

int main(int argc, char **argv) {
    int B = 1, S = 4096, Hq = 16, Hkv = 8, causal = 1, rounds = 5, iters = 20, check = 1, spike = 0;
    for (int i = 1; i < argc; ++i) {
        auto eq = strchr(argv[i], '=');
        if (!eq) continue;
        std::string k(argv[i], eq - argv[i]);
        int v = atoi(eq + 1);
        if (k == "B") B = v; else if (k == "S") S = v; else if (k == "Hq") Hq = v;
        else if (k == "Hkv") Hkv = v; else if (k == "causal") causal = v;
        else if (k == "rounds") rounds = v; else if (k == "iters") iters = v;
        else if (k == "check") check = v; else if (k == "spike") spike = v;
    }
    const int D = 128;
    if (getenv("LAB_PAGED")) return paged_main(B, S, Hq, Hkv, atoi(getenv("LAB_PAGED")), rounds, iters);
    const size_t nq = (size_t)B * S * Hq * D, nk = (size_t)B * S * Hkv * D;
    std::vector<uint16_t> hq(nq), hk(nk), hv(nk);
    std::mt19937 rng(1234);
    std::normal_distribution<float> nd(0.f, 1.f);
    for (auto &x : hq) x = f2bf(nd(rng));
    for (auto &x : hk) x = f2bf(nd(rng));
    for (auto &x : hv) x = f2bf(nd(rng));
    if (spike) {
        // force the deferred-rescale branch: a few K rows strongly aligned with every Q row of
        // head 0 (score jump >> threshold) at chosen key positions
        for (int b = 0; b < B; ++b)
            for (int key : {70, S / 2 + 3, S - 5}) {
                if (key >= S) continue;
                for (int d = 0; d < D; ++d) {
                    hk[(((size_t)b * S + key) * Hkv + 0) * D + d] = f2bf(d < 64 ? 3.0f : -3.0f);
                }
            }
        for (int b = 0; b < B; ++b)
            for (int s = 0; s < S; s += 3)
                for (int d = 0; d < D; ++d)
                    hq[(((size_t)b * S + s) * Hq + 0) * D + d] = f2bf(d < 64 ? 1.5f : -1.5f);
    }
    uint16_t *dq, *dk, *dv, *dout;
    CK(hipMalloc(&dq, nq * 2)); CK(hipMalloc(&dk, nk * 2)); CK(hipMalloc(&dv, nk * 2));
    CK(hipMalloc(&dout, nq * 2));
    CK(hipMemcpy(dq, hq.data(), nq * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dk, hk.data(), nk * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dv, hv.data(), nk * 2, hipMemcpyHostToDevice));
    FlashAttnArgs a{};
    a.q = dq; a.k = dk; a.v = dv; a.o = dout; a.lse = nullptr;
    a.batch = B; a.seq_q = S; a.seq_kv = S; a.num_heads = Hq; a.num_kv_heads = Hkv; a.head_dim = D;
    a.softmax_scale = 1.0f / sqrtf((float)D); a.causal = causal;

    std::vector<float> ref;
    if (check) {
        // reference: generic kernel, f32 in / f32 out, on the same (bf16-exact) values
        std::vector<float> fq(nq), fk(nk), fv(nk);
        for (size_t i = 0; i < nq; ++i) fq[i] = bf2f(hq[i]);
        for (size_t i = 0; i < nk; ++i) { fk[i] = bf2f(hk[i]); fv[i] = bf2f(hv[i]); }
        float *rq, *rk, *rv, *ro;
        CK(hipMalloc(&rq, nq * 4)); CK(hipMalloc(&rk, nk * 4)); CK(hipMalloc(&rv, nk * 4));
        CK(hipMalloc(&ro, nq * 4));
        CK(hipMemcpy(rq, fq.data(), nq * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(rk, fk.data(), nk * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(rv, fv.data(), nk * 4, hipMemcpyHostToDevice));
        FlashAttnArgs r = a;
        r.q = rq; r.k = rk; r.v = rv; r.o = ro;
        CK(nvllm::kernels::launch_flash_attn(DType::F32, r, 0));
        CK(hipDeviceSynchronize());
        ref.resize(nq);
        CK(hipMemcpy(ref.data(), ro, nq * 4, hipMemcpyDeviceToHost));
        hipFree(rq); hipFree(rk); hipFree(rv); hipFree(ro);
    }
    const double flops = 4.0 * B * Hq * (double)S * S * D * (causal ? 0.5 : 1.0);
    const int nv = sizeof(kVariants) / sizeof(kVariants[0]);
    std::vector<std::vector<double>> times(nv);
    std::vector<uint16_t> hout(nq);
    printf("shape B=%d S=%d Hq=%d Hkv=%d causal=%d spike=%d  (%.2f GFLOP)\n", B, S, Hq, Hkv, causal, spike, flops / 1e9);
    for (int vi = 0; vi < nv; ++vi) {
        const Variant &v = kVariants[vi];
        if (!v.supported(a)) { printf("%-14s unsupported\n", v.name); continue; }
        CK(hipMemset(dout, 0xFF, nq * 2));
        CK(v.launch(a, 0));
        CK(hipDeviceSynchronize());
        if (check) {
            CK(hipMemcpy(hout.data(), dout, nq * 2, hipMemcpyDeviceToHost));
            double mx = 0, sum = 0; size_t bad = 0, worst = 0;
            for (size_t i = 0; i < nq; ++i) {
                const float g = bf2f(hout[i]);
                double e = std::fabs((double)g - ref[i]);
                if (!(e == e)) { e = 1e9; }
                if (e > mx) { mx = e; worst = i; }
                sum += e;
                if (e > 2e-2) ++bad;
            }
            if (getenv("LAB_ROWS")) {
                printf("%-14s per 32-row block max err (b=0,h=0):", v.name);
                for (int r0 = 0; r0 < S; r0 += 32) {
                    double m = 0;
                    for (int r = r0; r < r0 + 32 && r < S; ++r)
                        for (int d = 0; d < D; ++d) {
                            const size_t i = ((size_t)r * Hq + 0) * D + d;
                            const double e = std::fabs((double)bf2f(hout[i]) - ref[i]);
                            if (e > m || !(e == e)) m = e;
                        }
                    printf(" %.1e", m);
                }
                printf("\n");
            }
            printf("%-14s check max_abs=%.3e mean_abs=%.3e bad=%zu worst@(b,s,h,d)=(%zu,%zu,%zu,%zu) %s\n",
                   v.name, mx, sum / nq, bad, worst / ((size_t)S * Hq * D), (worst / ((size_t)Hq * D)) % S,
                   (worst / D) % Hq, worst % D, (mx < 2e-2 && sum / nq < 1e-3) ? "OK" : "FAIL");
        }
    }
    // stamped diagnostic variants (name starts with "stamp", built with -DNVLLM_K1_STAMPS) write
    // per-wave segment cycle sums through the lse pointer: 8 x u64 per wave
    for (int vi = 0; vi < nv; ++vi) {
        const Variant &v = kVariants[vi];
        if (strncmp(v.name, "stamp", 5) != 0 || !v.supported(a)) continue;
        const size_t nwaves = (size_t)B * ((S + 127) / 128) * Hq * 8;  // upper bound (<= 8 waves/WG)
        unsigned long long *dd;
        CK(hipMalloc(&dd, nwaves * 128));
        CK(hipMemset(dd, 0, nwaves * 128));
        FlashAttnArgs sa = a;
        sa.lse = reinterpret_cast<float *>(dd);
        for (int i = 0; i < 3; ++i) CK(v.launch(sa, 0));
        CK(hipDeviceSynchronize());
        std::vector<unsigned long long> hd(nwaves * 16);
        CK(hipMemcpy(hd.data(), dd, nwaves * 128, hipMemcpyDeviceToHost));
        // slots 0..11 = cycle sums of segments, 12..14 = counters, 15 = steps
        double sum[16] = {0}; size_t n = 0; double mx_total = 0; size_t mx_i = 0;
        for (size_t w = 0; w < nwaves; ++w) {
            if (hd[w * 16 + 15] == 0) continue;
            double tot = 0;
            for (int i = 0; i < 12; ++i) tot += (double)hd[w * 16 + i];
            if (tot > mx_total) { mx_total = tot; mx_i = w; }
            for (int i = 0; i < 16; ++i) sum[i] += (double)hd[w * 16 + i];
            ++n;
        }
        printf("%-14s stamps over %zu waves: steps/wave %.1f, counters c12=%.1f c13=%.1f c14=%.1f per wave\n", v.name, n,
               sum[15] / n, sum[12] / n, sum[13] / n, sum[14] / n);
        printf("%-14s   cycles per wave: ", v.name);
        double tot = 0;
        for (int i = 0; i < 12; ++i) { printf("s%d=%.0f ", i, sum[i] / n); tot += sum[i]; }
        printf("total=%.0f\n", tot / n);
        printf("%-14s   longest wave %zu (steps %llu, c12=%llu c13=%llu c14=%llu): ", v.name, mx_i, hd[mx_i * 16 + 15],
               hd[mx_i * 16 + 12], hd[mx_i * 16 + 13], hd[mx_i * 16 + 14]);
        for (int i = 0; i < 12; ++i) printf("s%d=%.0f ", i, (double)hd[mx_i * 16 + i]);
        printf("total=%.0f cycles\n", mx_total);
        hipFree(dd);
    }
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int r = 0; r < rounds + 1; ++r)
        for (int vi = 0; vi < nv; ++vi) {
            const Variant &v = kVariants[vi];
            if (!v.supported(a)) continue;
            CK(hipEventRecord(e0, 0));
            for (int i = 0; i < iters; ++i) CK(v.launch(a, 0));
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (r > 0) times[vi].push_back(ms / iters);  // round 0 is warm-up
        }
    for (int vi = 0; vi < nv; ++vi) {
        auto &t = times[vi];
        if (t.empty()) continue;
        std::sort(t.begin(), t.end());
        const double med = t[t.size() / 2], mn = t[0];
        printf("%-14s median %8.1f us  %7.1f TF   best %8.1f us  %7.1f TF\n", kVariants[vi].name, med * 1e3,
               flops / (med * 1e-3) / 1e12, mn * 1e3, flops / (mn * 1e-3) / 1e12);
    }
    return 0;
}
