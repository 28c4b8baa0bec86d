import math, sys, torch
sys.path.insert(0, '.')
import nano_vllm_rs_amd as nv
dev = torch.device('cuda:0')
Hq, Hkv, D, bs = 32, 8, 128, 16
L = nv.capi.lib()
def run(N, ctx, T, same, policy=()):
    nblk = ctx // bs
    nb = N * nblk
    kc = torch.randn((nb, bs, Hkv, D), device=dev, dtype=torch.bfloat16)
    vc = torch.randn((nb, bs, Hkv, D), device=dev, dtype=torch.bfloat16)
    bt = torch.randperm(nb).to(torch.int32).reshape(N, nblk).to(dev)
    if same: bt = bt[:1].repeat(N, 1).contiguous()
    cl = torch.full((N,), ctx, dtype=torch.int32, device=dev)
    q = torch.randn((N, T, Hq, D), device=dev, dtype=torch.bfloat16)
    L.nvllm_debug_reset_policy()
    for k, v in policy: L.nvllm_debug_set_policy(k.encode(), v)
    fn = lambda: nv.paged_attention_batched(q, kc, vc, bt, cl, 1 / math.sqrt(D))
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20): fn()
    b.record(); torch.cuda.synchronize()
    t = a.elapsed_time(b) / 20 * 1e3
    fl = 4.0 * N * T * Hq * ctx * D
    print('N %4d ctx %5d T %4d same=%d %-40s %8.1f us %6.0f TF' % (N, ctx, T, same, str(policy), t, fl / t / 1e6), flush=True)
for ctx in (512, 1024, 2048, 4096, 8192):
    run(32, ctx, 64, 1)
for ctx in (1024, 2048, 4096):
    run(16, ctx, 128, 1)
run(32, 2048, 64, 1, (("NVLLM_K1_PAIRING", 0),))
run(32, 2048, 60, 1)
run(32, 2040, 64, 1)
