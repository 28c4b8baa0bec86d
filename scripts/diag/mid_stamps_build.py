"""Lab: build scripts/diag/libnvllm_attn_midstamps.so = the product objects with paged_attn_mid.hip patched for
cycle stamps (s_memtime sums per wave: issue / Q.K^T / P.V+softmax / wait / prologue / epilogue / barrier) and an
extra export nvllm_lab_mid_stamps.  The product source carries none of this; run nano-vllm-rs_amd/build.py first."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
C = os.path.join(ROOT, 'nano-vllm-rs_amd', 'csrc')
s = open(os.path.join(C, 'paged_attn_mid.hip')).read()
def rep(old, new, count=1):
    global s
    assert old in s, old[:60]
    s = s.replace(old, new, count)
rep('// NT = K/V requests carry the non-temporal hint', '''__device__ unsigned long long mid_stamps[8192 * 4 * 8];
#define MID_STAMP(i)                                                                  \\
    do {                                                                              \\
        unsigned long long t_;                                                        \\
        __builtin_amdgcn_sched_barrier(0);                                            \\
        asm volatile("s_memtime %0\\n\\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");    \\
        __builtin_amdgcn_sched_barrier(0);                                            \\
        seg[i] += t_ - t_prev;                                                        \\
        t_prev = t_;                                                                  \\
    } while (0)
// NT = K/V requests carry the non-temporal hint''')
rep('''    const int tid = threadIdx.x;
    const int lane = tid & 63;''', '''    unsigned long long seg[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t_prev;
    asm volatile("s_memtime %0\\n\\ts_waitcnt lgkmcnt(0)" : "=s"(t_prev)::"memory");
    const int tid = threadIdx.x;
    const int lane = tid & 63;''')
rep('''        wait_all_but_last_group();  // K(0) is in
        __syncthreads();''', '''        wait_all_but_last_group();  // K(0) is in
        __syncthreads();
        MID_STAMP(4);''')
rep('''            if (wave_live) {
                f32x16 sa, sb;
                qk_tile(''', '''            MID_STAMP(0);
            if (wave_live) {
                f32x16 sa, sb;
                qk_tile(''')
rep('''                mask_tile(t, sa, sb);
                // d tile 0''', '''                MID_STAMP(1);
                mask_tile(t, sa, sb);
                // d tile 0''')
rep('''                for (int sp = 0; sp < 4; ++sp) p[sp] = pn[sp];
            }
            wait_all_but_last_group();
            __syncthreads();
        }''', '''                for (int sp = 0; sp < 4; ++sp) p[sp] = pn[sp];
                MID_STAMP(2);
            }
            wait_all_but_last_group();
            MID_STAMP(3);
            __syncthreads();
            MID_STAMP(6);
        }''')
rep('''    if (!wave_live) return;
    float l = l_run + __shfl_xor(l_run, 32);''', '''    {
        MID_STAMP(5);
        if (lane == 0 && blockIdx.x < 8192) {
            seg[7] = (unsigned long long)rblk | ((unsigned long long)wave_live << 8);
            for (int i = 0; i < 8; ++i) mid_stamps[((size_t)blockIdx.x * 4 + wave) * 8 + i] = seg[i];
        }
    }
    if (!wave_live) return;
    float l = l_run + __shfl_xor(l_run, 32);''')
rep('''}  // namespace kernels
}  // namespace nvllm''', '''}  // namespace kernels
}  // namespace nvllm
extern "C" int nvllm_lab_mid_stamps(unsigned long long *host, size_t n) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(nvllm::kernels::mid_stamps), n * sizeof(unsigned long long));
}''')
tmp = '/tmp/paged_attn_mid_stamps.hip'
open(tmp, 'w').write(s)
hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
subprocess.check_call([hipcc, '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-I' + C, '-I' + os.path.join(ROOT, 'include'),
                       '-x', 'hip', '-c', tmp, '-o', '/tmp/paged_attn_mid_stamps.o'])
objs = [os.path.join(C, '_obj', o) for o in sorted(os.listdir(os.path.join(C, '_obj'))) if o.endswith('.o') and not o.startswith('paged_attn_mid')]
out = os.path.join(ROOT, 'scripts', 'diag', 'libnvllm_attn_midstamps.so')
subprocess.check_call([hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', out] + objs + ['/tmp/paged_attn_mid_stamps.o'])
print(out)
