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

namespace nvllm {
namespace kernels {
#define LAB_VARIANT(name)                                                   \
    hipError_t launch_##name(const FlashAttnArgs &a, hipStream_t stream);  \
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
};
static const Variant kVariants[] = {
#define LAB_VARIANT(name) {#name, nvllm::kernels::launch_##name, nvllm::kernels::supported_##name},
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
