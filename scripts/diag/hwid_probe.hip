// Lab: where do the waves of two co-resident 4-wave workgroups land?  Each wave records HW_REG_HW_ID
// and its XCC id; the workgroups use 80 KiB of LDS (two per CU) and spin for a while so that they overlap.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>
#include <tuple>
__global__ __launch_bounds__(256, 2) void probe(unsigned *out, int spin) {
    __shared__ unsigned char lds[80 * 1024];
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    volatile unsigned char *l = lds;
    unsigned acc = 0;
    for (int i = 0; i < spin; ++i) { l[(threadIdx.x * 64 + i) % (80 * 1024)] = (unsigned char)i; acc += l[(i * 977) % (80 * 1024)]; }
    if ((threadIdx.x & 63) == 0) {
        const int w = threadIdx.x >> 6;
        out[(blockIdx.x * 4 + w) * 3 + 0] = hw;
        out[(blockIdx.x * 4 + w) * 3 + 1] = xcc;
        out[(blockIdx.x * 4 + w) * 3 + 2] = acc;
    }
}
int main() {
    const int nwg = 2048;
    unsigned *d; hipMalloc(&d, nwg * 4 * 3 * 4);
    probe<<<nwg, 256>>>(d, 20000);
    if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
    std::vector<unsigned> h(nwg * 4 * 3);
    hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
    // per workgroup: simd ids, wave slot ids, tg id of its 4 waves
    std::map<std::tuple<int,int,int,int,int>, int> pat;
    for (int b = 0; b < nwg; ++b) {
        int simd[4], slot[4], tg[4];
        for (int w = 0; w < 4; ++w) { unsigned x = h[(b * 4 + w) * 3]; simd[w] = (x >> 4) & 3; slot[w] = x & 15; tg[w] = (x >> 16) & 15; }
        if (b < 24 || (b >= 512 && b < 536))
            printf("wg %4d xcc %u cu %2u se %u | simd %d %d %d %d | slot %d %d %d %d | tg %d %d %d %d | raw %08x\n", b, h[(b * 4) * 3 + 1] & 15,
                   (h[(b * 4) * 3] >> 8) & 15, (h[(b * 4) * 3] >> 13) & 7, simd[0], simd[1], simd[2], simd[3], slot[0], slot[1], slot[2], slot[3], tg[0], tg[1], tg[2], tg[3], h[(b*4)*3]);
        pat[{simd[0], simd[1], simd[2], simd[3], 0}]++;
    }
    for (auto &kv : pat) printf("simd pattern %d %d %d %d : %d workgroups\n", std::get<0>(kv.first), std::get<1>(kv.first), std::get<2>(kv.first), std::get<3>(kv.first), kv.second);
    std::map<int,int> tgs, slots;
    for (int b = 0; b < nwg; ++b) { tgs[(h[(b*4)*3] >> 16) & 15]++; slots[h[(b*4)*3] & 15]++; }
    for (auto &kv : tgs) printf("tg id %d : %d workgroups\n", kv.first, kv.second);
    for (auto &kv : slots) printf("wave-0 slot %d : %d workgroups\n", kv.first, kv.second);
    return 0;
}
