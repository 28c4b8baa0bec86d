"""Fixed per-workgroup cost of the paged prefill kernel: time vs context at one workgroup per CU."""
import math, os, sys
import torch
sys.path.insert(0, '.')
os.environ['NVLLM_K1_KSPLIT'] = '1'
import nano_vllm_rs_amd as nv
dev = torch.device('cuda:0')
Hq, Hkv, D, bs, N, T = 32, 8, 128, 16, 8, 256
def timeit(fn, iters=30, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3
for ctx in (256, 512, 1024, 2048, 4096, 8192):
    nblk = ctx // bs
    nb = N * nblk
    kc = torch.randn((nb, bs, Hkv, D), device=dev, dtype=torch.bfloat16)
    vc = torch.randn((nb, bs, Hkv, D), device=dev, dtype=torch.bfloat16)
    q = torch.randn((N, T, Hq, D), device=dev, dtype=torch.bfloat16)
    bt = torch.randperm(nb).to(torch.int32).reshape(N, nblk).to(dev)
    cl = torch.full((N,), ctx, dtype=torch.int32, device=dev)
    t = timeit(lambda: nv.paged_attention_batched(q, kc, vc, bt, cl, 1 / math.sqrt(D)))
    print('ctx', ctx, 'steps', ctx // 64, 'us', round(t, 1), flush=True)
