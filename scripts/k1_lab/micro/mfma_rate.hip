// Micro-benchmark: cycles per v_mfma_f32_32x32x16_bf16 for the operand placements and feeding
// patterns the K1 kernel uses (one wave per SIMD, 256-thread workgroups, one per CU).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
#define AG "a0","a1","a2","a3","a4","a5","a6","a7","a8","a9","a10","a11","a12","a13","a14","a15","a16","a17","a18","a19","a20","a21","a22","a23","a24","a25","a26","a27","a28","a29","a30","a31","a32","a33","a34","a35"

template <int MODE>
__global__ __launch_bounds__(256, 1) void k(const u32x4 *in, float *out, unsigned long long *cyc, int iters) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[65536];
    const int tid = threadIdx.x, lane = tid & 63;
    u32x4 a = in[tid], b = in[tid + 256];
    for (int i = tid; i < 4096; i += 256) reinterpret_cast<u32x4 *>(lds)[i] = in[i & 511];
    __syncthreads();
    f32x16 s0 = {}, s1 = {};
    asm volatile("" ::: AG);
    for (int r = 0; r < 32; ++r) asm volatile("v_accvgpr_write_b32 a0, 0");
#define ZA(n) asm volatile("v_accvgpr_write_b32 a" #n ", %0" ::"v"(a[0]));
    ZA(0) ZA(1) ZA(2) ZA(3) ZA(4) ZA(5) ZA(6) ZA(7) ZA(8) ZA(9) ZA(10) ZA(11) ZA(12) ZA(13) ZA(14) ZA(15)
    ZA(16) ZA(17) ZA(18) ZA(19) ZA(20) ZA(21) ZA(22) ZA(23) ZA(24) ZA(25) ZA(26) ZA(27) ZA(28) ZA(29) ZA(30) ZA(31)
    ZA(32) ZA(33) ZA(34) ZA(35)
    const unsigned char *lp = lds + (lane & 31) * 272 + 16 * (lane >> 5);
    u32x4 fr[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) fr[j] = *reinterpret_cast<const u32x4 *>(lp + 32 * j);
    float ps = 0.f, ps2 = 0.f, c0 = __builtin_bit_cast(float, a[1]) * 1e-3f, c1 = 0.25f;
    float f4[6] = {c0, c1, c0 + 1, c1 + 1, c0 + 2, c1 + 2};
    unsigned pw = 0;
    s1 = f32x16{1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16};
    auto pin1 = [](float &x) { asm volatile("" : "+v"(x)); };
    auto pin2 = [](float &x, unsigned &y) { asm volatile("" : "+v"(x), "+v"(y)); };
    auto pk = [](float lo, float hi) { typedef __attribute__((ext_vector_type(2))) __bf16 b2; b2 r; r[0] = (__bf16)lo; r[1] = (__bf16)hi; return __builtin_bit_cast(unsigned, r); };
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    constexpr int UN = MODE == 22 ? 16 : MODE == 23 ? 64 : 1;
#pragma unroll UN
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (MODE == 0) {  // all VGPR, register operands
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(s0) : "v"(a), "v"(b));
            } else if (MODE == 1) {  // B from AGPR
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, a[32:35], %0" : "+v"(s0) : "v"(a));
            } else if (MODE == 2) {  // C/D in AGPR
                asm volatile("v_mfma_f32_32x32x16_bf16 a[0:15], %0, %1, a[0:15]" : : "v"(a), "v"(b));
            } else if (MODE == 3) {  // A from LDS ring (prefetch 7), B AGPR, D VGPR  (the QK slot)
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, a[32:35], %0" : "+v"(s0) : "v"(fr[i & 7]));
                fr[(i + 7) & 7] = *reinterpret_cast<const u32x4 *>(lp + 32 * ((i + 7) & 7) + 8704 * ((it + i) & 1));
            } else if (MODE == 4) {  // A from LDS ring, all VGPR
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(s0) : "v"(fr[i & 7]), "v"(b));
                fr[(i + 7) & 7] = *reinterpret_cast<const u32x4 *>(lp + 32 * ((i + 7) & 7) + 8704 * ((it + i) & 1));
            } else if (MODE == 5) {  // two independent accumulators alternating, all VGPR
                if (i & 1) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(s0) : "v"(a), "v"(b));
                else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(s1) : "v"(a), "v"(b));
            } else if (MODE >= 7 && MODE <= 12) {
                // A from LDS ring, B AGPR, D VGPR + VALU fillers on other registers
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, a[32:35], %0" : "+v"(s0) : "v"(fr[i & 7]));
                fr[(i + 7) & 7] = *reinterpret_cast<const u32x4 *>(lp + 32 * ((i + 7) & 7) + 8704 * ((it + i) & 1));
                if (MODE == 7) {          // 1 element, dependent chain fma -> exp -> add
                    float p0 = __builtin_amdgcn_exp2f(__builtin_fmaf(s1[i], c0, -c1)); ps += p0; pin1(ps);
                } else if (MODE == 8) {   // 2 elements, dependent chains, + cvt
                    float p0 = __builtin_amdgcn_exp2f(__builtin_fmaf(s1[i], c0, -c1));
                    float p1 = __builtin_amdgcn_exp2f(__builtin_fmaf(s1[(i + 5) & 15], c0, -c1));
                    ps += p0; ps += p1; pw = pk(p0, p1); pin2(ps, pw);
                } else if (MODE == 9) {   // 2 elements, fma's first then exps then adds (less dependent stalls)
                    float f0 = __builtin_fmaf(s1[i], c0, -c1), f1 = __builtin_fmaf(s1[(i + 5) & 15], c0, -c1);
                    pin1(f0); pin1(f1);
                    float p0 = __builtin_amdgcn_exp2f(f0), p1 = __builtin_amdgcn_exp2f(f1);
                    pin1(p0); pin1(p1);
                    ps += p0; ps2 += p1; pw = pk(p0, p1); pin2(ps, pw); pin1(ps2);
                } else if (MODE == 10) {  // 4 plain fma fillers (independent)
                    f4[0] = __builtin_fmaf(f4[0], c0, c1); f4[1] = __builtin_fmaf(f4[1], c0, c1);
                    f4[2] = __builtin_fmaf(f4[2], c0, c1); f4[3] = __builtin_fmaf(f4[3], c0, c1);
                    pin1(f4[0]); pin1(f4[1]); pin1(f4[2]); pin1(f4[3]);
                } else if (MODE == 11) {  // 6 plain fma fillers, one pin
                    for (int q = 0; q < 6; ++q) f4[q] = __builtin_fmaf(f4[q], c0, c1);
                    asm volatile("" : "+v"(f4[0]), "+v"(f4[1]), "+v"(f4[2]), "+v"(f4[3]), "+v"(f4[4]), "+v"(f4[5]));
                } else if (MODE == 12) {  // 2 exps + 2 fma, independent, one pin
                    f4[0] = __builtin_amdgcn_exp2f(f4[0]); f4[1] = __builtin_amdgcn_exp2f(f4[1]);
                    f4[2] = __builtin_fmaf(f4[2], c0, c1); f4[3] = __builtin_fmaf(f4[3], c0, c1);
                    asm volatile("" : "+v"(f4[0]), "+v"(f4[1]), "+v"(f4[2]), "+v"(f4[3]));
                }
            } else if (MODE >= 13 && MODE <= 18) {
                // C/D in AGPR (the P.V slot)
                if (MODE == 13 || MODE == 16) {       // register operands
                    asm volatile("v_mfma_f32_32x32x16_bf16 a[0:15], %0, %1, a[0:15]" : : "v"(a), "v"(b));
                } else {                // A from LDS by two transposed reads, 7 ahead
                    asm volatile("v_mfma_f32_32x32x16_bf16 a[0:15], %0, %1, a[0:15]" : : "v"(fr[i & 7]), "v"(b));
                    typedef __attribute__((ext_vector_type(4))) short s4; typedef __attribute__((ext_vector_type(8))) short s8;
                    typedef __attribute__((address_space(3))) s4 lds_s4;
                    const int c16 = lane & 15, qq = c16 >> 2, pp = c16 & 3, g16 = lane >> 4, hh = lane >> 5;
                    const int vrow = 4 * hh + qq, vsw = 4 * (vrow & 3), vlo = 2 * (g16 & 1) + (pp >> 1);
                    const unsigned char *p0 = lds + vrow * 256 + 8 * (pp & 1) + 16 * ((4 * (((i + 7) >> 2) & 3) + vlo) ^ vsw) + 4096 * ((i + 7) & 3);
                    s4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4 *)p0);
                    s4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4 *)(p0 + 2048));
                    s8 va = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
                    fr[(i + 7) & 7] = __builtin_bit_cast(u32x4, va);
                }
                if (MODE == 13 || MODE == 14) {       // 6 fma fillers
                    for (int q = 0; q < 6; ++q) f4[q] = __builtin_fmaf(f4[q], c0, c1);
                    asm volatile("" : "+v"(f4[0]), "+v"(f4[1]), "+v"(f4[2]), "+v"(f4[3]), "+v"(f4[4]), "+v"(f4[5]));
                } else if (MODE == 17) {              // 1 element: fma, exp + half a cvt (approx)
                    f4[0] = __builtin_amdgcn_exp2f(__builtin_fmaf(f4[0], c0, c1));
                    if (i & 1) pw = pk(f4[0], f4[1]);
                    asm volatile("" : "+v"(f4[0]), "+v"(pw));
                } else if (MODE == 18) {              // 3 fma fillers
                    for (int q = 0; q < 3; ++q) f4[q] = __builtin_fmaf(f4[q], c0, c1);
                    asm volatile("" : "+v"(f4[0]), "+v"(f4[1]), "+v"(f4[2]));
                }
            } else if (MODE == 19) {  // register operands, D VGPR, 6 fma
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(s0) : "v"(a), "v"(b));
                for (int q = 0; q < 6; ++q) f4[q] = __builtin_fmaf(f4[q], c0, c1);
                asm volatile("" : "+v"(f4[0]), "+v"(f4[1]), "+v"(f4[2]), "+v"(f4[3]), "+v"(f4[4]), "+v"(f4[5]));
            } else if (MODE >= 20 && MODE <= 23) {  // b128 LDS-fed, acc AGPR, 6 fma / 3 fma
                asm volatile("v_mfma_f32_32x32x16_bf16 a[0:15], %0, %1, a[0:15]" : : "v"(fr[i & 7]), "v"(b));
                fr[(i + 7) & 7] = *reinterpret_cast<const u32x4 *>(lp + 32 * ((i + 7) & 7) + 8704 * ((it + i) & 1));
                for (int q = 0; q < (MODE == 20 ? 6 : 3); ++q) f4[q] = __builtin_fmaf(f4[q], c0, c1);
                asm volatile("" : "+v"(f4[0]), "+v"(f4[1]), "+v"(f4[2]), "+v"(f4[3]), "+v"(f4[4]), "+v"(f4[5]));
            } else if (MODE == 6) {  // builtin, compiler-scheduled
                s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(__attribute__((ext_vector_type(8))) __bf16, a),
                                                              __builtin_bit_cast(__attribute__((ext_vector_type(8))) __bf16, b), s0, 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    float acc = 0;
    for (int r = 0; r < 16; ++r) acc += s0[r] + s1[r];
    float x;
    asm volatile("v_accvgpr_read_b32 %0, a3" : "=v"(x));
    out[blockIdx.x * 256 + tid] = acc + x + ps + ps2 + f4[0] + f4[1] + f4[2] + f4[3] + f4[4] + f4[5] + __builtin_bit_cast(float, pw) + __builtin_bit_cast(float, fr[3][1]);
    if (lane == 0) cyc[blockIdx.x * 4 + (tid >> 6)] = t1 - t0;
}

template <int MODE> int run(const char *name, const u32x4 *din, float *dout, unsigned long long *dc, int grid) {
    const int iters = 2000;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    k<MODE><<<grid, 256>>>(din, dout, dc, iters);
    CK(hipEventRecord(e0));
    k<MODE><<<grid, 256>>>(din, dout, dc, iters);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> hc(grid * 4);
    CK(hipMemcpy(hc.data(), dc, hc.size() * 8, hipMemcpyDeviceToHost));
    double s = 0; for (auto c : hc) s += (double)c;
    const double cyc = s / hc.size() / (iters * 16.0);
    printf("%-34s grid %4d: %.1f cycles/MFMA  (%.3f ms, %.0f TF, clock %.2f GHz)\n", name, grid, cyc, ms,
           grid * 4.0 * iters * 16 * 32768 / (ms * 1e-3) / 1e12, s / hc.size() / (ms * 1e-3) / 1e9);
    return 0;
}
int main() {
    std::vector<unsigned> h(4096 * 4);
    for (size_t i = 0; i < h.size(); ++i) h[i] = 0x3f803f80u ^ (unsigned)((i * 2654435761u) & 0x007f007f);
    u32x4 *din; float *dout; unsigned long long *dc;
    CK(hipMalloc(&din, h.size() * 4)); CK(hipMalloc(&dout, 1024 * 256 * 4)); CK(hipMalloc(&dc, 1024 * 4 * 8));
    CK(hipMemcpy(din, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    for (int grid : {256}) {
        if (run<0>("all VGPR", din, dout, dc, grid)) return 1;
        if (run<1>("B in AGPR", din, dout, dc, grid)) return 1;
        if (run<2>("C/D in AGPR", din, dout, dc, grid)) return 1;
        if (run<3>("A from LDS ring, B AGPR, D VGPR", din, dout, dc, grid)) return 1;
        if (run<4>("A from LDS ring, all VGPR", din, dout, dc, grid)) return 1;
        if (run<5>("two accumulators, all VGPR", din, dout, dc, grid)) return 1;
        if (run<6>("builtin", din, dout, dc, grid)) return 1;
        if (run<7>("LDS-fed + 1 elem (fma>exp>add)", din, dout, dc, grid)) return 1;
        if (run<8>("LDS-fed + 2 elem dependent + cvt", din, dout, dc, grid)) return 1;
        if (run<9>("LDS-fed + 2 elem staged + cvt", din, dout, dc, grid)) return 1;
        if (run<10>("LDS-fed + 4 fma, 4 pins", din, dout, dc, grid)) return 1;
        if (run<11>("LDS-fed + 6 fma, 1 pin", din, dout, dc, grid)) return 1;
        if (run<12>("LDS-fed + 2 exp + 2 fma, 1 pin", din, dout, dc, grid)) return 1;
        if (run<19>("regs, D VGPR + 6 fma", din, dout, dc, grid)) return 1;
        if (run<20>("b128 LDS-fed, acc AGPR + 6 fma", din, dout, dc, grid)) return 1;
        if (run<21>("b128 LDS-fed, acc AGPR + 3 fma", din, dout, dc, grid)) return 1;
        if (run<22>("  same, body unrolled to 256 slices", din, dout, dc, grid)) return 1;
        if (run<23>("  same, body unrolled to 1024 slices", din, dout, dc, grid)) return 1;
        if (run<13>("acc AGPR, regs + 6 fma", din, dout, dc, grid)) return 1;
        if (run<14>("acc AGPR, tr-read fed + 6 fma", din, dout, dc, grid)) return 1;
        if (run<15>("acc AGPR, tr-read fed", din, dout, dc, grid)) return 1;
        if (run<16>("acc AGPR, regs", din, dout, dc, grid)) return 1;
        if (run<17>("acc AGPR, tr-read fed + 1 elem", din, dout, dc, grid)) return 1;
        if (run<18>("acc AGPR, tr-read fed + 3 fma", din, dout, dc, grid)) return 1;
    }
    return 0;
}
