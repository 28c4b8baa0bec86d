"""BASELINE config 1 as SURVEY 8d defines it ("plumbing"): the reference's greedy loop
(src/engine/llm.rs:351-478) with the Qwen3-0.6B attention geometry -- 28 layers, 16 Q / 8 KV heads,
head_dim 128, block_size 16 -- one sequence, a 128-token prompt, 32 greedy decode steps, seeded
N(0,1) Q/K/V (seed 1234) standing in for the projections (no weights offline).  The scheduler
drives: admission allocates blocks 0..7, the first decode step grows the table by block 8 at token
129 (batch.rs:266-286); BatchMetadata turns each step into slot_mapping / block_tables /
context_lens; K4 writes the step's K/V into every layer's cache; attention runs on the GPU (prefill:
the flash kernel AND the paged kernel over the just-written cache; decode: paged decode).  Checked
per layer and step against the CPU oracle run the reference's way: write_kv_to_cache on its own
cache, flash_attention_cpu for the prompt, and for decode the contiguous SDPA over the K/V history
(paged == contiguous: docs/paged_attention.md:328-357) as well as oracle.paged_attention.  At the
end every layer's cache must equal the oracle's bit for bit."""
import math

import numpy as np
import pytest
import torch

import nano_vllm_rs_amd as nv
import oracle
from helpers import assert_attention_close, bf16_randn

pytestmark = pytest.mark.gpu


def test_config1_greedy_decode_plumbing(dev):
    L, Hq, Hkv, D, bs = 28, 16, 8, 128, 16
    prompt_len, steps, nb, max_blocks = 128, 32, 1024, 16        # EngineConfig: 1024 blocks (config.rs:27-40)
    scale = 1.0 / math.sqrt(D)
    rng = np.random.default_rng(1234)
    sched = nv.Scheduler(nv.SchedulerConfig(), bs, nb)            # the reference's defaults
    prompt = rng.integers(0, 32000, size=prompt_len).tolist()
    sched.add_sequence(1, prompt)
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev, torch.bfloat16)
    cache = nv.KVCache(nv.KVCacheConfig(nb, bs, Hkv, D, L, torch.bfloat16), dev)
    okc = [np.zeros((nb, bs, Hkv, D), np.float32) for _ in range(L)]
    ovc = [np.zeros((nb, bs, Hkv, D), np.float32) for _ in range(L)]
    hist_k = [np.zeros((0, Hkv, D), np.float32) for _ in range(L)]   # contiguous history per layer
    hist_v = [np.zeros((0, Hkv, D), np.float32) for _ in range(L)]

    def greedy_token(vec):
        """stand-in for lm_head + argmax: the index of the largest attention output of the last layer"""
        return int(np.argmax(vec))

    def layer_step(layer, batch, T, ctx):
        bt, cl, sm, qs, ps = batch.upload(dev)
        slots = batch.slot_mapping()
        q, k, v = bf16_randn(rng, (T, Hq, D)), bf16_randn(rng, (T, Hkv, D)), bf16_randn(rng, (T, Hkv, D))
        lay = cache.layer(layer)
        nv.reshape_and_cache(d(k), d(v), lay.key_cache(), lay.value_cache(), sm)
        got = nv.paged_attention_batched(d(q)[None], lay.key_cache(), lay.value_cache(), bt, cl, scale)
        got = got.float().cpu().numpy()
        okc[layer], ovc[layer] = oracle.write_kv_to_cache(k[None], v[None], okc[layer], ovc[layer], slots, bs)
        hist_k[layer] = np.concatenate([hist_k[layer], k], 0)
        hist_v[layer] = np.concatenate([hist_v[layer], v], 0)
        ids = sched.get_sequence(1).block_ids()
        want = oracle.paged_attention(q[None], okc[layer], ovc[layer], ids, bs, ctx, Hq, Hkv, D, scale)
        assert_attention_close(got, want)
        if T > 1:      # prompt: the wired path of the reference is flash_attention over contiguous K/V
            fl = nv.flash_attention(d(q)[None], d(k)[None], d(v)[None], nv.FlashAttentionConfig.new(D, True))
            wf = oracle.flash_attention_cpu(q[None], k[None], v[None], True)
            assert_attention_close(fl.float().cpu().numpy(), wf)
            assert_attention_close(got.reshape(1, T, Hq, D), wf)               # paged chunk == flash
        else:          # decode: paged == SDPA over the contiguous history (no mask: one query row)
            kf = np.repeat(hist_k[layer], Hq // Hkv, axis=1).astype(np.float64)    # [ctx, Hq, D]
            vf = np.repeat(hist_v[layer], Hq // Hkv, axis=1).astype(np.float64)
            s = np.einsum("hd,shd->hs", q[0].astype(np.float64), kf) * scale
            p = np.exp(s - s.max(-1, keepdims=True))
            p /= p.sum(-1, keepdims=True)
            sd = np.einsum("hs,shd->hd", p, vf).reshape(1, 1, Hq * D)
            assert np.abs(want - sd).max() < 1e-4                              # oracle: paged == contiguous
            assert_attention_close(got, sd)
        return got, want

    # ---- prefill step
    out = sched.schedule()
    assert out.prefill_sequences == [1] and out.prefill_chunk_sizes[1] == prompt_len and not out.decode_sequences
    assert sched.get_sequence(1).block_ids() == list(range(8))                 # ascending first fit
    batch, ids = sched.prefill_batch(prompt_len, max_blocks)
    assert ids == [1] and batch.slot_mapping().tolist() == list(range(prompt_len))
    assert batch.slot_mapping().tolist() == oracle.get_slot_mapping(list(range(8)), bs, prompt_len).tolist()
    for layer in range(L):
        got, want = layer_step(layer, batch, prompt_len, prompt_len)
    sched.mark_prefilled(1, prompt_len)
    tok = greedy_token(want[0, -1])
    sched.append_token(1, tok)
    tokens = [tok]

    # ---- 32 greedy decode steps
    for step in range(steps):
        out = sched.schedule()
        assert out.decode_sequences == [1] and not out.prefill_sequences
        total = prompt_len + step + 1
        seq = sched.get_sequence(1)
        assert seq.total_len() == total
        assert seq.block_ids() == list(range(oracle.compute_num_blocks(total, bs)))
        if step == 0:
            assert seq.block_ids()[-1] == 8                                    # block 8 at token 129
        batch = sched.decode_batch(max_blocks)
        pos = total - 1
        assert batch.slot_mapping().tolist() == [seq.block_ids()[pos // bs] * bs + pos % bs]
        assert batch.context_lens().tolist() == [total] and batch.positions().tolist() == [pos]
        for layer in range(L):
            got, want = layer_step(layer, batch, 1, total)
        tok = greedy_token(want[0, 0])
        gap = np.sort(want[0, 0])[-1] - np.sort(want[0, 0])[-2]
        if gap > 4e-2:                                                         # beyond the tolerance band
            assert greedy_token(got[0, 0]) == tok
        sched.append_token(1, tok)
        tokens.append(tok)
    assert len(tokens) == steps + 1 and sched.get_sequence(1).output_len() == steps + 1
    # blocks grow at schedule() time (batch.rs:266-286): the token appended last has no block yet
    assert sched.num_used_blocks() == oracle.compute_num_blocks(prompt_len + steps, bs)

    # ---- cache contents: every layer bit-exact (K4 moves bytes; values are bf16 already)
    torch.cuda.synchronize()
    for layer in range(L):
        lay = cache.layer(layer)
        assert np.array_equal(lay.key_cache().float().cpu().numpy(), okc[layer])
        assert np.array_equal(lay.value_cache().float().cpu().numpy(), ovc[layer])
