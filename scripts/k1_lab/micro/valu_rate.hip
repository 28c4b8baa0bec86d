// Micro-benchmark: issue cycles per VALU instruction of the softmax path (one wave per SIMD), alone
// and interleaved one group per v_mfma_f32_32x32x16_bf16.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

// MODE: 0 v_exp_f32, 1 v_fma_f32, 2 v_pk_fma_f32, 3 v_pk_add_f32, 4 v_lshl_add_u32, 5 v_cvt_pk_bf16_f32,
//       6 v_max3_f32, 7 v_pk_mul_f32, 8 v_ldexp_f32
// WITH_MFMA: one MFMA in front of every group of G instructions
template <int MODE> __device__ __forceinline__ void op(float &x0, float &x1, f32x2 &p, unsigned &u, float c) {
    if (MODE == 0) asm volatile("v_exp_f32 %0, %1" : "=v"(x0) : "v"(x1));
    if (MODE == 1) asm volatile("v_fma_f32 %0, %1, %2, %1" : "=v"(x0) : "v"(x1), "v"(c));
    if (MODE == 2) asm volatile("v_pk_fma_f32 %0, %1, %1, %1" : "=v"(p) : "v"(p));
    if (MODE == 3) asm volatile("v_pk_add_f32 %0, %1, %1" : "=v"(p) : "v"(p));
    if (MODE == 4) asm volatile("v_lshl_add_u32 %0, %1, 23, %2" : "=v"(u) : "v"(x1), "v"(x0));
    if (MODE == 5) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(u) : "v"(x0), "v"(x1));
    if (MODE == 6) asm volatile("v_max3_f32 %0, %1, %2, %3" : "=v"(x0) : "v"(x1), "v"(c), "v"(x0));
    if (MODE == 7) asm volatile("v_pk_mul_f32 %0, %1, %1" : "=v"(p) : "v"(p));
    if (MODE == 8) asm volatile("v_ldexp_f32 %0, %1, %2" : "=v"(x0) : "v"(x1), "v"(u));
}

template <int MODE, int G, bool WITH_MFMA>
__global__ __launch_bounds__(256, 1) void k(const u32x4 *in, float *out, unsigned long long *cyc, int iters) {
    const int tid = threadIdx.x;
    u32x4 a = in[tid], b = in[tid + 256];
    f32x16 s0 = {};
    float x[8];
    f32x2 p[8];
    unsigned u[8];
    for (int i = 0; i < 8; ++i) { x[i] = __builtin_bit_cast(float, a[i & 3]) * 1e-9f; p[i] = f32x2{x[i], x[i]}; u[i] = i; }
    const float c = x[0] + 1.0f;
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            if (WITH_MFMA) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(s0) : "v"(a), "v"(b));
#pragma unroll
            for (int g = 0; g < G; ++g) op<MODE>(x[(g * 2) & 7], x[(g * 2 + 1) & 7], p[g & 7], u[g & 7], c);
        }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    float acc = s0[0];
    for (int i = 0; i < 8; ++i) acc += x[i] + p[i][0] + p[i][1] + (float)u[i];
    out[blockIdx.x * 256 + tid] = acc;
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE, int G, bool W> int run(const char *name, const u32x4 *in, float *out, unsigned long long *cyc) {
    const int iters = 2000;
    k<MODE, G, W><<<256, 256>>>(in, out, cyc, iters);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    k<MODE, G, W><<<256, 256>>>(in, out, cyc, iters);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long h[256]; CK(hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost));
    double s = 0; for (int i = 0; i < 256; ++i) s += (double)h[i];
    const double groups = 16.0 * iters;
    // s_memtime counts at a fixed 100 MHz; report wall ns per group and per instruction instead
    printf("%-28s G=%2d mfma=%d: %.2f ns per group, %.2f ns per instr (memtime ticks/group %.2f)\n", name, G, (int)W,
           ms * 1e6 / groups, W ? 0.0 : ms * 1e6 / groups / G, s / 256 / groups);
    return 0;
}

int main() {
    u32x4 *in; float *out; unsigned long long *cyc;
    CK(hipMalloc(&in, 8192 * 16)); CK(hipMemset(in, 0x3c, 8192 * 16));
    CK(hipMalloc(&out, 256 * 256 * 4)); CK(hipMalloc(&cyc, 256 * 8));
    run<0, 8, false>("v_exp_f32", in, out, cyc);
    run<1, 8, false>("v_fma_f32", in, out, cyc);
    run<2, 8, false>("v_pk_fma_f32", in, out, cyc);
    run<3, 8, false>("v_pk_add_f32", in, out, cyc);
    run<4, 8, false>("v_lshl_add_u32", in, out, cyc);
    run<5, 8, false>("v_cvt_pk_bf16_f32", in, out, cyc);
    run<6, 8, false>("v_max3_f32", in, out, cyc);
    run<7, 8, false>("v_pk_mul_f32", in, out, cyc);
    run<8, 8, false>("v_ldexp_f32", in, out, cyc);
    run<1, 0, true>("mfma alone", in, out, cyc);
    run<0, 1, true>("mfma + 1 exp", in, out, cyc);
    run<0, 2, true>("mfma + 2 exp", in, out, cyc);
    run<1, 4, true>("mfma + 4 fma", in, out, cyc);
    run<1, 6, true>("mfma + 6 fma", in, out, cyc);
    run<1, 8, true>("mfma + 8 fma", in, out, cyc);
    run<2, 4, true>("mfma + 4 pk_fma", in, out, cyc);
    run<2, 6, true>("mfma + 6 pk_fma", in, out, cyc);
    run<2, 8, true>("mfma + 8 pk_fma", in, out, cyc);
    return 0;
}
