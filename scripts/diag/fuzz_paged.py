"""Randomised shapes through nvllm_paged_attn_bf16 (decode kernel / prefill kernel over the cache,
whichever the launcher picks) against the generic kernel on the same values in f32."""
import math, sys
import numpy as np
import torch
sys.path.insert(0, '.')
import nano_vllm_rs_amd as nv
dev = torch.device('cuda:0')
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
D, bs = 128, 16
worst = 0.0
for case in range(int(sys.argv[2]) if len(sys.argv) > 2 else 80):
    Hkv = int(rng.choice([1, 2, 4, 8]))
    G = int(rng.choice([1, 2, 3, 4, 8, 16]))
    Hq = Hkv * G
    N = int(rng.integers(1, 7))
    T = int(rng.choice([1, 2, 5, 8, 9, 12, 13, 17, 31, 32, 33, 63, 64, 65, 100, 128, 129, 257, 300]))
    ctx = [int(T + rng.integers(0, 700)) for _ in range(N)]
    max_blocks = max(-(-c // bs) for c in ctx) + int(rng.integers(0, 3))
    nb = N * max_blocks + 3
    g = torch.Generator(device='cpu').manual_seed(case)
    kc = torch.randn((nb, bs, Hkv, D), generator=g).to(dev, torch.bfloat16)
    vc = torch.randn((nb, bs, Hkv, D), generator=g).to(dev, torch.bfloat16)
    q = torch.randn((N, T, Hq, D), generator=g).to(dev, torch.bfloat16)
    bt = torch.randperm(nb, generator=g)[:N * max_blocks].to(torch.int32).reshape(N, max_blocks).to(dev)
    cl = torch.tensor(ctx, dtype=torch.int32, device=dev)
    scale = 1 / math.sqrt(D)
    got = nv.paged_attention_batched(q, kc, vc, bt, cl, scale).float()
    want = nv.paged_attention_batched(q.float(), kc.float(), vc.float(), bt, cl, scale)
    torch.cuda.synchronize()
    err = (got - want).abs()
    mx, mean = float(err.max()), float(err.mean())
    worst = max(worst, mx)
    ok = mx < 2e-2 and mean < 1e-3 and bool(torch.isfinite(got).all())
    print('case', case, 'N', N, 'T', T, 'Hq', Hq, 'Hkv', Hkv, 'ctx', ctx, 'max_abs %.2e mean %.2e' % (mx, mean), 'OK' if ok else 'FAIL', flush=True)
    if not ok:
        sys.exit(1)
print('all ok, worst max_abs %.3e' % worst)
