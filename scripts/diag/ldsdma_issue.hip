// Lab: what does ISSUING an LDS-DMA request cost the wave?  4 waves per workgroup, one workgroup per CU,
// every wave issues 8 x 1 KiB per round from an L2-resident buffer and stamps s_memtime around the issue
// (not the landing).  Variants: A = s_mov m0 per piece (K1's form), B = one m0 per four pieces with
// immediate offsets (mid kernel), C = global_load_lds_dwordx4, D = buffer_load_dwordx4 into registers,
// E = buffer_load_dword ... lds (256 B per instruction, 4x as many).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
template <int V>
__global__ __launch_bounds__(256, 1) void probe(const unsigned char *src, unsigned long long *out, int rounds, int waves_active) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[64 * 1024];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned lds_base = (unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned char *)lds;
    const unsigned char *mine = src + ((size_t)blockIdx.x * 4 + wave) * 65536;
    const u32x4 rs = {(unsigned)(unsigned long long)mine, (unsigned)((unsigned long long)mine >> 32) & 0xffffu, 65536u, 0x00020000u};
    unsigned off[8];
    for (int j = 0; j < 8; ++j) off[j] = 16u * lane + 1024u * j;
    unsigned long long sum = 0;
    u32x4 sink = {0, 0, 0, 0};
    if (wave < waves_active)
    for (int r = 0; r < rounds; ++r) {
        unsigned long long t0, t1;
        const unsigned dst = lds_base + wave * 16384u + (r & 1) * 8192u;
        const unsigned ro = (unsigned)(r & 7) * 8192u;
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
        if constexpr (V == 0) {
#pragma unroll
            for (int j = 0; j < 8; ++j)
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" : : "s"(dst + 1024u * j), "v"(off[j] + ro), "s"(rs) : "memory");
        } else if constexpr (V == 1) {
#pragma unroll
            for (int g = 0; g < 2; ++g)
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\tbuffer_load_dwordx4 %1, %2, 0 offen offset:1024 lds\n\t"
                             "buffer_load_dwordx4 %1, %2, 0 offen offset:2048 lds\n\tbuffer_load_dwordx4 %1, %2, 0 offen offset:3072 lds"
                             : : "s"(dst + 4096u * g), "v"(off[4 * g] + ro), "s"(rs) : "memory");
        } else if constexpr (V == 2) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const unsigned char *p = mine + off[j] + ro;
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" : : "s"(dst + 1024u * j), "v"(p) : "memory");
            }
        } else if constexpr (V == 3) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                u32x4 x;
                asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(x) : "v"(off[j] + ro), "s"(rs) : "memory");
                asm volatile("" :: "v"(x));
            }
        } else {
#pragma unroll
            for (int j = 0; j < 32; ++j)
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dword %1, %2, 0 offen lds" : : "s"(dst + 256u * j), "v"(4u * lane + 256u * j + ro), "s"(rs) : "memory");
        }
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        sum += t1 - t0;
    }
    if (lane == 0) out[blockIdx.x * 4 + wave] = sum + sink[0];
}
template <int V>
void run(const char *name, const unsigned char *src, unsigned long long *out, int waves_active) {
    const int rounds = 2000;
    probe<V><<<256, 256>>>(src, out, rounds, waves_active);
    probe<V><<<256, 256>>>(src, out, rounds, waves_active);
    if (hipDeviceSynchronize() != hipSuccess) { printf("%s failed\n", name); return; }
    std::vector<unsigned long long> h(1024);
    hipMemcpy(h.data(), out, 8192, hipMemcpyDeviceToHost);
    double s = 0; int n = 0;
    for (int b = 0; b < 256; ++b) for (int w = 0; w < waves_active; ++w) { s += h[b * 4 + w]; ++n; }
    printf("%-48s waves/CU %d: %6.1f cycles to issue 8 KiB (%5.1f per 1 KiB)\n", name, waves_active, s / n / rounds, s / n / rounds / 8);
}
int main() {
    unsigned char *src; unsigned long long *out;
    hipMalloc(&src, (size_t)1024 * 65536); hipMemset(src, 1, (size_t)1024 * 65536); hipMalloc(&out, 8192);
    for (int wa : {1, 4}) {
        run<0>("A buffer_load_dwordx4 lds, m0 per piece", src, out, wa);
        run<1>("B buffer_load_dwordx4 lds, m0 per 4 pieces", src, out, wa);
        run<2>("C global_load_lds_dwordx4, m0 per piece", src, out, wa);
        run<3>("D buffer_load_dwordx4 into registers", src, out, wa);
        run<4>("E buffer_load_dword lds (32 x 256 B)", src, out, wa);
    }
    return 0;
}
