// Micro-benchmark: cycles per MFMA gap for filler mixes, one wave per SIMD, 256 workgroups.
// Every gap is ONE asm statement (MFMA + fillers), so hipcc adds nothing between the instructions.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

#define EXP(d, s) "v_exp_f32 %" #d ", %" #s "\n\t"
#define FMA(d, s) "v_fma_f32 %" #d ", %" #s ", %20, %21\n\t"
#define CVT(d, s, t) "v_cvt_pk_bf16_f32 %" #d ", %" #s ", %" #t "\n\t"
#define MAX3(d, s, t) "v_max3_f32 %" #d ", %" #d ", %" #s ", %" #t "\n\t"
#define MFMA "v_mfma_f32_32x32x16_bf16 %16, %17, %18, %16\n\t"
#define MFMA2 "v_mfma_f32_32x32x16_bf16 %19, %17, %18, %19\n\t"
#define MFMA_ACC "v_mfma_f32_32x32x16_bf16 a[0:15], %17, %18, a[0:15]\n\t"
#define MFMA_ACC2 "v_mfma_f32_32x32x16_bf16 a[16:31], %17, %18, a[16:31]\n\t"
#define MFMA_QB "v_mfma_f32_32x32x16_bf16 %16, %17, a[32:35], %16\n\t"
#define LDR "ds_read_b128 %22, %24\n\t"
#define LDTR "ds_read_b64_tr_b16 %23, %24\n\t"
#define NOP "s_nop 0\n\t"

// operands: 0..7 outputs x0..x7 (float), 8..15 inputs y0..y7, 16 acc s0, 17 a, 18 b, 19 acc s1, 20 c, 21 c2, 22 ld4, 23 ld2, 24 addr
#define GAP(body)                                                                                          \
    asm volatile(body : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), \
                 "+v"(x[7]), "+v"(y[0]), "+v"(y[1]), "+v"(y[2]), "+v"(y[3]), "+v"(y[4]), "+v"(y[5]), "+v"(y[6]), \
                 "+v"(y[7]), "+v"(s0), "+v"(a), "+v"(b), "+v"(s1), "+v"(c), "+v"(c2), "=v"(ld4), "=v"(ld2) : "v"(addr) : "memory", "a0","a1","a2","a3","a4","a5","a6","a7","a8","a9","a10","a11","a12","a13","a14","a15","a16","a17","a18","a19","a20","a21","a22","a23","a24","a25","a26","a27","a28","a29","a30","a31","a32","a33","a34","a35")

template <int MODE>
__global__ __launch_bounds__(256, 1) void k(const u32x4 *in, float *out, unsigned long long *cyc, int iters) {
    __shared__ unsigned char lds[32768];
    const int tid = threadIdx.x;
    u32x4 a = in[tid], b = in[tid + 256], ld4 = {};
    unsigned long long ld2 = 0;
    f32x16 s0 = {}, s1 = {};
    float x[8], y[8];
    for (int i = 0; i < 8; ++i) { x[i] = __builtin_bit_cast(float, a[i & 3]) * 1e-9f; y[i] = x[i] + i; }
    float c = x[0] + 1.0f, c2 = 0.5f;
    for (int i = tid; i < 8192; i += 256) ((unsigned *)lds)[i] = i;
    __syncthreads();
    unsigned addr = (tid & 63) * 16 + (tid >> 6) * 4096;
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            if (MODE == 0) GAP(MFMA);
            if (MODE == 1) GAP(MFMA EXP(0, 8));
            if (MODE == 2) GAP(MFMA EXP(0, 8) EXP(1, 9));
            if (MODE == 3) GAP(MFMA EXP(0, 8) EXP(1, 9) EXP(2, 10));
            if (MODE == 4) GAP(MFMA FMA(0, 8) FMA(1, 9) FMA(2, 10) FMA(3, 11));
            if (MODE == 5) GAP(MFMA FMA(0, 8) FMA(1, 9) FMA(2, 10) FMA(3, 11) FMA(4, 12) FMA(5, 13));
            if (MODE == 6) GAP(MFMA EXP(0, 8) FMA(1, 9) CVT(2, 10, 11));
            if (MODE == 7) GAP(MFMA EXP(0, 8) FMA(1, 9) EXP(2, 10) FMA(3, 11) CVT(4, 12, 13));
            if (MODE == 8) GAP(MFMA MAX3(0, 8, 9) MAX3(1, 10, 11) MAX3(2, 12, 13) MAX3(3, 14, 15));
            if (MODE == 9) GAP(MFMA LDR EXP(0, 8) FMA(1, 9) "s_waitcnt lgkmcnt(0)\n\t");
            if (MODE == 10) GAP(MFMA LDTR LDTR EXP(0, 8) FMA(1, 9) "s_waitcnt lgkmcnt(0)\n\t");
            if (MODE == 11) GAP(MFMA EXP(0, 8) EXP(1, 9) FMA(2, 10) FMA(3, 11) CVT(4, 12, 13) LDR "s_waitcnt lgkmcnt(0)\n\t");
            if (MODE == 12) GAP(MFMA CVT(0, 8, 9) CVT(1, 10, 11) CVT(2, 12, 13) CVT(3, 14, 15));
            if (MODE == 13) GAP(MFMA NOP NOP NOP NOP);
            if (MODE == 14) GAP(MFMA EXP(0, 8) FMA(8, 9));   // fma feeds the NEXT gap's exp (dependent, distance 1 gap)
            if (MODE == 15) GAP(MFMA FMA(8, 9) EXP(0, 8));   // exp right behind the fma it depends on
            if (MODE == 16) GAP(MFMA MFMA2);                 // two accumulators
            if (MODE == 17) GAP(MFMA EXP(0, 8) FMA(1, 9) MFMA2 EXP(2, 10) FMA(3, 11));
            if (MODE == 18) GAP(MFMA EXP(0, 8) FMA(1, 9) FMA(2, 10) CVT(3, 11, 12) MAX3(4, 13, 14));
            if (MODE == 19) GAP(MFMA EXP(0, 8) FMA(1, 9) FMA(2, 10) FMA(3, 11) FMA(4, 12));
            if (MODE == 20) GAP(MFMA EXP(0, 8) FMA(1, 9) CVT(2, 10, 11) LDTR LDTR);
            if (MODE == 21) GAP(MFMA EXP(0, 8) FMA(1, 9) CVT(2, 10, 11) LDR);
            if (MODE == 22) GAP(MFMA EXP(0, 8) FMA(1, 9) LDR LDR);
            if (MODE == 23) GAP(MFMA EXP(0, 8) FMA(1, 9) "ds_write_b128 %24, %17\n\t");
            if (MODE == 24) GAP(MFMA EXP(0, 8) FMA(1, 9) CVT(2, 10, 11) FMA(3, 12) MAX3(4, 13, 14) MAX3(5, 14, 15));
            if (MODE == 25) GAP(MFMA EXP(0, 8) FMA(1, 9) CVT(2, 10, 11) FMA(3, 12));
            if (MODE == 26) GAP(MFMA FMA(1, 9) CVT(2, 10, 11) FMA(3, 12) MAX3(4, 13, 14) MAX3(5, 14, 15));
            if (MODE == 27) GAP(MFMA LDTR LDTR LDTR LDTR);
            if (MODE == 28) GAP(MFMA EXP(0, 8) LDTR LDTR LDTR LDTR);
            if (MODE == 29) GAP(MFMA EXP(0, 8) FMA(1, 9) "ds_write_b64 %24, %23\n\t" "ds_write_b64 %24, %23 offset:8\n\t");
            if (MODE == 30) GAP(MFMA "v_exp_f32 %0, %8\n\tv_fma_f32 %1, %9, %20, %21\n\tv_exp_f32 %2, %10\n\t");
            if (MODE == 31) GAP(MFMA "v_exp_f32 %0, %8\n\tv_fma_f32 %1, %9, %20, %21\n\tv_fma_f32 %3, %11, %20, %21\n\tv_cvt_pk_bf16_f32 %4, %12, %13\n\tv_exp_f32 %2, %10\n\t");
            if (MODE == 32) GAP(MFMA_ACC);
            if (MODE == 33) GAP(MFMA_ACC EXP(0, 8) FMA(1, 9) CVT(2, 10, 11));
            if (MODE == 34) GAP(MFMA_ACC EXP(0, 8) FMA(1, 9) CVT(2, 10, 11) FMA(3, 12));
            if (MODE == 35) GAP(MFMA_ACC MAX3(0, 8, 9) MAX3(1, 10, 11) MAX3(2, 12, 13) MAX3(3, 14, 15));
            if (MODE == 36) GAP(MFMA_QB EXP(0, 8) FMA(1, 9) CVT(2, 10, 11));
            if (MODE == 37) { GAP(MFMA_ACC EXP(0, 8) FMA(1, 9) CVT(2, 10, 11)); GAP(MFMA_ACC2 EXP(3, 12) FMA(4, 13)); }
            if (MODE == 38) GAP(MFMA_ACC EXP(0, 8) FMA(1, 9) EXP(2, 10) FMA(3, 11) CVT(4, 12, 13));
            if (MODE == 39) GAP(MFMA_ACC FMA(0, 8) FMA(1, 9) FMA(2, 10) FMA(3, 11) FMA(4, 12) FMA(5, 13));
            if (m == 7 && MODE >= 20 && MODE <= 29) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    float acc = s0[0] + s1[3] + (float)ld4[0] + (float)ld2;
    for (int i = 0; i < 8; ++i) acc += x[i] + y[i];
    out[blockIdx.x * 256 + tid] = acc;
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE> int run(const char *name, const u32x4 *in, float *out, unsigned long long *cyc) {
    const int iters = 2000;
    k<MODE><<<256, 256>>>(in, out, cyc, iters);
    CK(hipDeviceSynchronize());
    k<MODE><<<256, 256>>>(in, out, cyc, iters);
    CK(hipDeviceSynchronize());
    unsigned long long h[256]; CK(hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost));
    double s = 0; for (int i = 0; i < 256; ++i) s += (double)h[i];
    printf("%-58s %.2f cycles per gap\n", name, s / 256 / (8.0 * iters));
    return 0;
}

int main() {
    u32x4 *in; float *out; unsigned long long *cyc;
    CK(hipMalloc(&in, 8192 * 16)); CK(hipMemset(in, 0x3c, 8192 * 16));
    CK(hipMalloc(&out, 256 * 256 * 4)); CK(hipMalloc(&cyc, 256 * 8));
    run<0>("mfma", in, out, cyc);
    run<1>("mfma + exp", in, out, cyc);
    run<2>("mfma + 2 exp", in, out, cyc);
    run<3>("mfma + 3 exp", in, out, cyc);
    run<4>("mfma + 4 fma", in, out, cyc);
    run<5>("mfma + 6 fma", in, out, cyc);
    run<6>("mfma + exp fma cvt", in, out, cyc);
    run<7>("mfma + exp fma exp fma cvt", in, out, cyc);
    run<8>("mfma + 4 max3", in, out, cyc);
    run<9>("mfma + ds_read_b128 exp fma wait", in, out, cyc);
    run<10>("mfma + 2 ds_read_tr exp fma wait", in, out, cyc);
    run<11>("mfma + 2 exp 2 fma cvt ds_read_b128 wait", in, out, cyc);
    run<12>("mfma + 4 cvt", in, out, cyc);
    run<13>("mfma + 4 s_nop 0", in, out, cyc);
    run<14>("mfma + exp fma(dep next gap)", in, out, cyc);
    run<15>("mfma + fma exp(dep immediate)", in, out, cyc);
    run<16>("2 mfma (two accumulators) per gap", in, out, cyc);
    run<17>("mfma exp fma mfma exp fma (per pair)", in, out, cyc);
    run<18>("mfma + exp fma fma cvt max3", in, out, cyc);
    run<19>("mfma + exp 4 fma", in, out, cyc);
    run<20>("mfma + exp fma cvt 2 ds_read_tr (no wait)", in, out, cyc);
    run<21>("mfma + exp fma cvt ds_read_b128 (no wait)", in, out, cyc);
    run<22>("mfma + exp fma 2 ds_read_b128 (no wait)", in, out, cyc);
    run<23>("mfma + exp fma ds_write_b128", in, out, cyc);
    run<24>("mfma + exp fma cvt fma max3 max3 (6 fillers)", in, out, cyc);
    run<25>("mfma + exp fma cvt fma (4 fillers)", in, out, cyc);
    run<26>("mfma + fma cvt fma max3 max3 (5, no exp)", in, out, cyc);
    run<27>("mfma + 4 ds_read_tr", in, out, cyc);
    run<28>("mfma + exp 4 ds_read_tr", in, out, cyc);
    run<29>("mfma + exp fma 2 ds_write_b64", in, out, cyc);
    run<30>("mfma + exp fma exp", in, out, cyc);
    run<31>("mfma + exp fma fma cvt exp", in, out, cyc);
    run<32>("mfma(acc dst)", in, out, cyc);
    run<33>("mfma(acc dst) + exp fma cvt", in, out, cyc);
    run<34>("mfma(acc dst) + exp fma cvt fma", in, out, cyc);
    run<35>("mfma(acc dst) + 4 max3", in, out, cyc);
    run<36>("mfma(B from acc) + exp fma cvt", in, out, cyc);
    run<37>("2 x mfma(acc dst, alternating) + 5 fillers (per 2 gaps)", in, out, cyc);
    run<38>("mfma(acc dst) + exp fma exp fma cvt", in, out, cyc);
    run<39>("mfma(acc dst) + 6 fma", in, out, cyc);
    return 0;
}
