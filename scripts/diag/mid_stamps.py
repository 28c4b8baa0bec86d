"""Lab: cycle stamps of the mid kernel (a library built with -DNVLLM_MID_STAMPS, scripts/diag/libnvllm_attn_midstamps.so):
where a wave's time goes per step -- issuing requests, Q.K^T, P.V + softmax, the counted wait, the barrier."""
import ctypes, math, os, sys
import numpy as np
import torch
sys.path.insert(0, '.')
import nano_vllm_rs_amd.capi as capi
capi.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'libnvllm_attn_midstamps.so')
import nano_vllm_rs_amd as nv
dev = torch.device('cuda:0')
Hq, Hkv, D, bs = 32, 8, 128, 16
N, ctx = int(os.environ.get('SWEEP_N', 256)), int(os.environ.get('SWEEP_CTX', 2048))
nblk = ctx // bs
nb = N * nblk
kc = torch.randn((nb, bs, Hkv, D), device=dev, dtype=torch.bfloat16)
vc = torch.randn((nb, bs, Hkv, D), device=dev, dtype=torch.bfloat16)
lib = ctypes.CDLL(capi.LIB_PATH)
lib.nvllm_lab_mid_stamps.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
names = ['issue', 'qk', 'pv+softmax', 'wait', 'prologue', 'epilogue', 'barrier']
for hot in (0, 256, 32):
    bt = torch.randperm(nb).to(torch.int32).reshape(N, nblk).to(dev)
    if hot:
        bt = bt % hot
    cl = torch.full((N,), ctx, dtype=torch.int32, device=dev)
    for T in [int(x) for x in os.environ.get('SWEEP_T', '16,32').split(',')]:
        q = torch.randn((N, T, Hq, D), device=dev, dtype=torch.bfloat16)
        for _ in range(20):
            nv.paged_attention_batched(q, kc, vc, bt, cl, 1 / math.sqrt(D))
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10):
            nv.paged_attention_batched(q, kc, vc, bt, cl, 1 / math.sqrt(D))
        b.record(); torch.cuda.synchronize()
        us = a.elapsed_time(b) / 10 * 1e3
        nw = N * Hkv * 4
        host = np.zeros(nw * 8, dtype=np.uint64)
        rc = lib.nvllm_lab_mid_stamps(host.ctypes.data_as(ctypes.c_void_p), host.size)
        assert rc == 0, rc
        s = host.reshape(nw, 8).astype(np.float64)
        live = (s[:, 7].astype(np.int64) >> 8) & 1
        nt = ctx // 64
        print('hot_blocks', hot, 'T', T, 'us', round(us, 1), 'steps', nt, flush=True)
        for tag, sel in (('computing waves', live == 1), ('fetch-only waves', live == 0)):
            if sel.sum() == 0:
                continue
            m = s[sel].mean(axis=0)
            tot = m[:7].sum()
            print('  %-16s n=%5d  total %7.0f cycles/wave | per step: ' % (tag, sel.sum(), tot) +
                  '  '.join('%s %.0f' % (names[i], m[i] / nt) for i in (0, 1, 2, 3, 6)) +
                  ' | prologue %.0f  epilogue %.0f' % (m[4], m[5]), flush=True)
