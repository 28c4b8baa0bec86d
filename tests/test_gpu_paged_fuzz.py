"""Randomised launch shapes through nvllm_paged_attn_bf16 -- the launcher picks the decode kernel
(1..3 query tiles, 1..8 kv heads per workgroup, split-K or not) or the prefill kernel over the
cache (head group packed or not, key split or not) -- against the generic kernel run on the same
values in f32 (itself pinned to the oracle in test_gpu_paged_attention.py).  Same bar as
everywhere: 2e-2 max-abs, 1e-3 mean-abs."""
import math

import numpy as np
import pytest
import torch

import nano_vllm_rs_amd as nv

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed,ksplit", [(11, None), (12, None), (13, "3")])
def test_random_shapes_match_the_generic_kernel(dev, policy, seed, ksplit):
    if ksplit:
        policy.setenv("NVLLM_K1_KSPLIT", ksplit)
    rng = np.random.default_rng(seed)
    D, bs = 128, 16
    for case in range(30):
        Hkv = int(rng.choice([1, 2, 4, 8]))
        G = int(rng.choice([1, 2, 3, 4, 8, 16]))
        Hq = Hkv * G
        N = int(rng.integers(1, 7))
        T = int(rng.choice([1, 2, 5, 8, 9, 12, 13, 17, 31, 32, 33, 63, 64, 65, 100, 128, 129, 257, 300]))
        ctx = [int(T + rng.integers(0, 700)) for _ in range(N)]
        max_blocks = max(-(-c // bs) for c in ctx) + int(rng.integers(0, 3))
        nb = N * max_blocks + 3
        g = torch.Generator(device="cpu").manual_seed(1000 * seed + case)
        kc = torch.randn((nb, bs, Hkv, D), generator=g).to(dev, torch.bfloat16)
        vc = torch.randn((nb, bs, Hkv, D), generator=g).to(dev, torch.bfloat16)
        q = torch.randn((N, T, Hq, D), generator=g).to(dev, torch.bfloat16)
        bt = torch.randperm(nb, generator=g)[:N * max_blocks].to(torch.int32).reshape(N, max_blocks).to(dev)
        cl = torch.tensor(ctx, dtype=torch.int32, device=dev)
        scale = 1 / math.sqrt(D)
        got = nv.paged_attention_batched(q, kc, vc, bt, cl, scale).float()
        want = nv.paged_attention_batched(q.float(), kc.float(), vc.float(), bt, cl, scale)
        err = (got - want).abs()
        what = dict(case=case, N=N, T=T, Hq=Hq, Hkv=Hkv, ctx=ctx)
        assert bool(torch.isfinite(got).all()), what
        assert float(err.max()) < 2e-2 and float(err.mean()) < 1e-3, (what, float(err.max()), float(err.mean()))
