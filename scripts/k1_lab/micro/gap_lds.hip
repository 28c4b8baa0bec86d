// Micro-benchmark: cost of LDS fragment reads (distinct destination registers, counted waits) and
// staging stores in an MFMA gap, one wave per SIMD.
#include <hip/hip_runtime.h>
#ifndef UNROLL
#define UNROLL 2
#endif
#include <cstdio>
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
#define MFMA "v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n\t"
#define G(body) asm volatile(body : "+v"(s0), "+v"(a), "+v"(b), "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3), "+v"(t0), "+v"(t1), "+v"(t2), "+v"(t3), "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(y0), "+v"(y1), "+v"(y2), "+v"(y3) : "v"(addr), "v"(c) : "memory", "vcc", "scc", "s4", "s5")
// 3..6 q (b128 dests), 7..10 t (b64 dests), 11..14 x, 15..18 y, 19 addr, 20 c
template <int MODE>
__global__ __launch_bounds__(256, 1) void k(const u32x4 *in, float *out, unsigned long long *cyc, int iters) {
    __shared__ unsigned char lds[65536];
    const int tid = threadIdx.x;
    u32x4 a = in[tid], b = in[tid + 256], q0 = {}, q1 = {}, q2 = {}, q3 = {};
    u32x2 t0 = {}, t1 = {}, t2 = {}, t3 = {};
    f32x16 s0 = {};
    float x0 = 1e-3f * tid, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, y0 = x0, y1 = x1, y2 = x2, y3 = x3, c = 0.5f;
    for (int i = tid; i < 16384; i += 256) ((unsigned *)lds)[i] = i;
    __syncthreads();
    unsigned addr = (tid & 63) * 16 + (tid >> 6) * 8192;
    unsigned long long c0, c1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(c0)::"memory");
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < UNROLL; ++m) {
            // 4 gaps per group; reads of 4 distinct registers, waited at the end of the group
            if (MODE == 0) { G(MFMA); G(MFMA); G(MFMA); G(MFMA); }
            if (MODE == 1) { G(MFMA "ds_read_b128 %3, %19\n\t"); G(MFMA "ds_read_b128 %4, %19 offset:1024\n\t"); G(MFMA "ds_read_b128 %5, %19 offset:2048\n\t"); G(MFMA "ds_read_b128 %6, %19 offset:3072\n\ts_waitcnt lgkmcnt(2)\n\t"); }
            if (MODE == 2) { G(MFMA "ds_read_b128 %3, %19\n\tds_read_b128 %4, %19 offset:1024\n\t"); G(MFMA "ds_read_b128 %5, %19 offset:2048\n\tds_read_b128 %6, %19 offset:3072\n\t"); G(MFMA); G(MFMA "s_waitcnt lgkmcnt(0)\n\t"); }
            if (MODE == 3) { G(MFMA "ds_read_b64_tr_b16 %7, %19\n\tds_read_b64_tr_b16 %8, %19 offset:1024\n\t"); G(MFMA "ds_read_b64_tr_b16 %9, %19 offset:2048\n\tds_read_b64_tr_b16 %10, %19 offset:3072\n\t"); G(MFMA); G(MFMA "s_waitcnt lgkmcnt(0)\n\t"); }
            if (MODE == 4) { G(MFMA "ds_read_b128 %3, %19\n\tv_exp_f32 %11, %15\n\tv_fma_f32 %12, %16, %20, %20\n\t"); G(MFMA "ds_read_b128 %4, %19 offset:1024\n\tv_exp_f32 %13, %17\n\tv_fma_f32 %14, %18, %20, %20\n\t"); G(MFMA "ds_read_b128 %5, %19 offset:2048\n\tv_exp_f32 %11, %15\n\tv_fma_f32 %12, %16, %20, %20\n\t"); G(MFMA "ds_read_b128 %6, %19 offset:3072\n\tv_exp_f32 %13, %17\n\tv_fma_f32 %14, %18, %20, %20\n\ts_waitcnt lgkmcnt(2)\n\t"); }
            if (MODE == 5) { G(MFMA "v_exp_f32 %11, %15\n\tv_fma_f32 %12, %16, %20, %20\n\tv_cvt_pk_bf16_f32 %13, %17, %18\n\t"); G(MFMA "v_exp_f32 %11, %15\n\tv_fma_f32 %12, %16, %20, %20\n\t"); G(MFMA "ds_write_b128 %19, %3 offset:32768\n\tv_exp_f32 %11, %15\n\tv_fma_f32 %12, %16, %20, %20\n\t"); G(MFMA "v_exp_f32 %11, %15\n\tv_fma_f32 %12, %16, %20, %20\n\t"); }
            if (MODE == 6) { G(MFMA "ds_read_b64_tr_b16 %7, %19\n\tds_read_b64_tr_b16 %8, %19 offset:1024\n\tv_exp_f32 %11, %15\n\tv_fma_f32 %12, %16, %20, %20\n\t"); G(MFMA "ds_read_b64_tr_b16 %9, %19 offset:2048\n\tds_read_b64_tr_b16 %10, %19 offset:3072\n\tv_exp_f32 %13, %17\n\tv_fma_f32 %14, %18, %20, %20\n\t"); G(MFMA "v_exp_f32 %11, %15\n\tv_fma_f32 %12, %16, %20, %20\n\tv_cvt_pk_bf16_f32 %13, %17, %18\n\t"); G(MFMA "v_exp_f32 %11, %15\n\tv_fma_f32 %12, %16, %20, %20\n\ts_waitcnt lgkmcnt(0)\n\t"); }
            if (MODE == 7) { G(MFMA "s_nop 0\n\t"); G(MFMA "s_nop 0\n\ts_nop 0\n\t"); G(MFMA "s_nop 0\n\ts_nop 0\n\ts_nop 0\n\t"); G(MFMA "s_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\t"); }
            if (MODE == 8) { G(MFMA "v_cmp_gt_f32 vcc, %11, %15\n\ts_cmp_lg_u64 vcc, 0\n\ts_cselect_b64 s[4:5], -1, 0\n\t"); G(MFMA "v_exp_f32 %11, %15\n\t"); G(MFMA "v_exp_f32 %12, %16\n\t"); G(MFMA "v_exp_f32 %13, %17\n\t"); }
            if (MODE == 13) { G(MFMA "v_max3_f32 %11, %11, %15, %16\n\tv_max3_f32 %11, %11, %17, %18\n\tv_max3_f32 %11, %11, %15, %16\n\tv_max3_f32 %11, %11, %17, %18\n\t"); G(MFMA "v_max3_f32 %11, %11, %15, %16\n\tv_max3_f32 %11, %11, %17, %18\n\tv_max3_f32 %11, %11, %15, %16\n\tv_max3_f32 %11, %11, %17, %18\n\t"); G(MFMA "v_max3_f32 %11, %11, %15, %16\n\tv_max3_f32 %11, %11, %17, %18\n\tv_max3_f32 %11, %11, %15, %16\n\tv_max3_f32 %11, %11, %17, %18\n\t"); G(MFMA "v_max3_f32 %11, %11, %15, %16\n\tv_max3_f32 %11, %11, %17, %18\n\tv_max3_f32 %11, %11, %15, %16\n\tv_max3_f32 %11, %11, %17, %18\n\t"); }
            if (MODE == 14) { G(MFMA "v_max3_f32 %11, %11, %15, %16\n\tv_max3_f32 %12, %12, %17, %18\n\tv_max3_f32 %11, %11, %15, %16\n\tv_max3_f32 %12, %12, %17, %18\n\t"); G(MFMA "v_max3_f32 %11, %11, %15, %16\n\tv_max3_f32 %12, %12, %17, %18\n\tv_max3_f32 %11, %11, %15, %16\n\tv_max3_f32 %12, %12, %17, %18\n\t"); G(MFMA "v_max3_f32 %11, %11, %15, %16\n\tv_max3_f32 %12, %12, %17, %18\n\tv_max3_f32 %11, %11, %15, %16\n\tv_max3_f32 %12, %12, %17, %18\n\t"); G(MFMA "v_max3_f32 %11, %11, %15, %16\n\tv_max3_f32 %12, %12, %17, %18\n\tv_max3_f32 %11, %11, %15, %16\n\tv_max3_f32 %12, %12, %17, %18\n\t"); }
            if (MODE == 15) { G(MFMA "v_max3_f32 %11, %11, %15, %16\n\ts_nop 0\n\tv_max3_f32 %11, %11, %17, %18\n\ts_nop 0\n\t"); G(MFMA "v_max3_f32 %11, %11, %15, %16\n\ts_nop 0\n\tv_max3_f32 %11, %11, %17, %18\n\ts_nop 0\n\t"); G(MFMA "v_max3_f32 %11, %11, %15, %16\n\ts_nop 0\n\tv_max3_f32 %11, %11, %17, %18\n\ts_nop 0\n\t"); G(MFMA "v_max3_f32 %11, %11, %15, %16\n\ts_nop 0\n\tv_max3_f32 %11, %11, %17, %18\n\ts_nop 0\n\t"); }
            if (MODE == 16) { G(MFMA "v_max3_f32 %11, %11, %15, %16\n\tv_max3_f32 %12, %12, %17, %18\n\tv_cmp_gt_f32 vcc, %11, %12\n\ts_cbranch_vccnz 1f\n\t1:\n\t"); G(MFMA "v_exp_f32 %13, %15\n\tv_fma_f32 %14, %16, %20, %20\n\t"); G(MFMA "v_exp_f32 %13, %17\n\tv_fma_f32 %14, %16, %20, %20\n\t"); G(MFMA "v_exp_f32 %13, %17\n\tv_fma_f32 %14, %16, %20, %20\n\t"); }
            if (MODE == 9) { G(MFMA "v_exp_f32 %11, %15\n\tv_cmp_gt_f32 vcc, %12, %16\n\ts_cbranch_vccnz 1f\n\t1:\n\tv_fma_f32 %13, %16, %20, %20\n\t"); G(MFMA "v_exp_f32 %11, %15\n\tv_fma_f32 %12, %16, %20, %20\n\t"); G(MFMA "v_exp_f32 %12, %16\n\tv_fma_f32 %12, %16, %20, %20\n\t"); G(MFMA "v_exp_f32 %13, %17\n\tv_fma_f32 %12, %16, %20, %20\n\t"); }
            if (MODE == 10) { G(MFMA "v_exp_f32 %11, %15\n\tv_cmp_gt_f32 vcc, %12, %16\n\ts_cbranch_vccz 1f\n\t1:\n\tv_fma_f32 %13, %16, %20, %20\n\t"); G(MFMA "v_exp_f32 %11, %15\n\tv_fma_f32 %12, %16, %20, %20\n\t"); G(MFMA "v_exp_f32 %12, %16\n\tv_fma_f32 %12, %16, %20, %20\n\t"); G(MFMA "v_exp_f32 %13, %17\n\tv_fma_f32 %12, %16, %20, %20\n\t"); }
            if (MODE == 11) { G(MFMA "v_exp_f32 %11, %15\n\tv_cmp_gt_f32 vcc, %12, %16\n\tv_fma_f32 %13, %16, %20, %20\n\t"); G(MFMA "v_exp_f32 %11, %15\n\tv_fma_f32 %12, %16, %20, %20\n\t"); G(MFMA "v_exp_f32 %12, %16\n\tv_fma_f32 %12, %16, %20, %20\n\t"); G(MFMA "v_exp_f32 %13, %17\n\tv_fma_f32 %12, %16, %20, %20\n\t"); }
            if (MODE == 12) { G(MFMA "v_exp_f32 %11, %15\n\tv_cmp_gt_f32 vcc, %12, %16\n\ts_cmp_lg_u64 vcc, 0\n\ts_cbranch_scc1 1f\n\t1:\n\tv_fma_f32 %13, %16, %20, %20\n\t"); G(MFMA "v_exp_f32 %11, %15\n\tv_fma_f32 %12, %16, %20, %20\n\t"); G(MFMA "v_exp_f32 %12, %16\n\tv_fma_f32 %12, %16, %20, %20\n\t"); G(MFMA "v_exp_f32 %13, %17\n\tv_fma_f32 %12, %16, %20, %20\n\t"); }
        }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(c1)::"memory");
    float acc = s0[0] + (float)(q0[0] + q1[1] + q2[2] + q3[3] + t0[0] + t1[1] + t2[0] + t3[1]) + x0 + x1 + x2 + x3 + y0 + y1 + y2 + y3;
    out[blockIdx.x * 256 + tid] = acc;
    if (tid == 0) cyc[blockIdx.x] = c1 - c0;
}
template <int MODE> int run(const char *name, const u32x4 *in, float *out, unsigned long long *cyc) {
    const int iters = 8000 / UNROLL;
    k<MODE><<<256, 256>>>(in, out, cyc, iters);
    CK(hipDeviceSynchronize());
    k<MODE><<<256, 256>>>(in, out, cyc, iters);
    CK(hipDeviceSynchronize());
    unsigned long long h[256]; CK(hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost));
    double s = 0; for (int i = 0; i < 256; ++i) s += (double)h[i];
    printf("%-66s %.2f cycles per gap\n", name, s / 256 / (4.0 * UNROLL * iters));
    return 0;
}
int main() {
    u32x4 *in; float *out; unsigned long long *cyc;
    CK(hipMalloc(&in, 8192 * 16)); CK(hipMemset(in, 0x3c, 8192 * 16));
    CK(hipMalloc(&out, 256 * 256 * 4)); CK(hipMalloc(&cyc, 256 * 8));
    run<0>("mfma x4", in, out, cyc);
    run<1>("1 ds_read_b128 per gap, wait(2) per 4 gaps", in, out, cyc);
    run<2>("2+2 ds_read_b128 in gaps 0,1, wait(0) at gap 3", in, out, cyc);
    run<3>("2+2 ds_read_tr in gaps 0,1, wait(0) at gap 3", in, out, cyc);
    run<4>("1 ds_read_b128 + exp fma per gap, wait(2) per 4 gaps", in, out, cyc);
    run<5>("exp fma (cvt) per gap, one ds_write_b128 per 4 gaps", in, out, cyc);
    run<6>("2+2 ds_read_tr + exp fma (cvt) per gap, wait(0) per 4 gaps", in, out, cyc);
    run<7>("s_nop 0: 1, 2, 3, 6 per gap (avg 3)", in, out, cyc);
    run<8>("v_cmp + s_cmp + s_cselect in 1 of 4 gaps, exp in the others", in, out, cyc);
    run<13>("4 max3 per gap, ONE dependent chain", in, out, cyc);
    run<14>("4 max3 per gap, two alternating chains", in, out, cyc);
    run<15>("2 max3 + 2 s_nop per gap, one chain", in, out, cyc);
    run<16>("2 max3 -> v_cmp -> s_cbranch in 1 of 4 gaps, exp fma in the others", in, out, cyc);
    run<11>("exp fma per gap + v_cmp in 1 of 4 (no branch)", in, out, cyc);
    run<9>("exp fma per gap + v_cmp, s_cbranch_vccnz NOT taken in 1 of 4", in, out, cyc);
    run<10>("exp fma per gap + v_cmp, s_cbranch_vccz TAKEN (to next instr) in 1 of 4", in, out, cyc);
    run<12>("exp fma per gap + v_cmp, s_cmp, s_cbranch_scc1 not taken in 1 of 4", in, out, cyc);
    return 0;
}
