// Read-only HBM bandwidth ceiling on this part: what a kernel that only streams 16-byte loads reaches,
// for several loads-in-flight per thread and grid sizes.  The decode kernel's 6.69 TB/s is judged
// against this, not only against the 8 TB/s on the data sheet.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int U, bool NT>
__global__ __launch_bounds__(256) void read_kernel(const u32x4 *__restrict__ p, size_t n_vec, unsigned *out) {
    // block b reads U*256 consecutive vectors per trip, grid-strided
    u32x4 acc = {0u, 0u, 0u, 0u};
    for (size_t base = (size_t)blockIdx.x * U * 256; base < n_vec; base += (size_t)gridDim.x * U * 256) {
        u32x4 x[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t i = base + (size_t)u * 256 + threadIdx.x;
            x[u] = i < n_vec ? (NT ? __builtin_nontemporal_load(p + i) : p[i]) : u32x4{0u, 0u, 0u, 0u};
        }
#pragma unroll
        for (int u = 0; u < U; ++u) acc ^= x[u];
    }
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) out[0] = 1;  // practically never
}

template <int U, bool NT> void run(const char *name, const u32x4 *p, size_t n_vec, unsigned *out, int grid) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 2; ++i) read_kernel<U, NT><<<grid, 256>>>(p, n_vec, out);
    CK(hipEventRecord(a));
    const int iters = 5;
    for (int i = 0; i < iters; ++i) read_kernel<U, NT><<<grid, 256>>>(p, n_vec, out);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    printf("%-10s grid %6d: %.1f GB/s\n", name, grid, n_vec * 16.0 * iters / (ms * 1e-3) / 1e9);
    fflush(stdout);
}

int main() {
    const size_t bytes = (size_t)8 << 30;  // 8 GiB >> 256 MiB Infinity Cache
    const size_t n_vec = bytes / 16;
    u32x4 *p; unsigned *out;
    CK(hipMalloc(&p, bytes)); CK(hipMalloc(&out, 4));
    CK(hipMemset(p, 1, bytes));
    for (int grid : {1024, 2048, 4096, 16384, 65536}) {
        run<1, false>("u1", p, n_vec, out, grid);
        run<4, false>("u4", p, n_vec, out, grid);
        run<4, true>("u4_nt", p, n_vec, out, grid);
        run<8, true>("u8_nt", p, n_vec, out, grid);
    }
    return 0;
}
