"""The whole caller side driving the kernels (SURVEY 8f-1 + 8f-2 + 8f-4): a Scheduler with chunked
prefill (rescheduled chunks), prefix caching and decode produces SchedulerOutputs; those become
BatchMetadata; K5 normalises + rotates Q/K and writes K/V into the paged cache; paged attention
(K1 paged variant for the chunks, K2 for decode) attends.  Checked every step, per sequence,
against the oracle chain run the reference's way: qk_norm_rope -> write_kv_to_cache ->
paged_attention on the oracle's own cache.  Sequences that share a cached prefix skip its
prefill and must still get the oracle's full-context answer."""
import math

import numpy as np
import pytest
import torch

import nano_vllm_rs_amd as nv
import oracle
from helpers import assert_attention_close, bf16_randn

pytestmark = pytest.mark.gpu


def test_scheduler_driven_prefill_and_decode(dev):
    Hq, Hkv, D, bs, nb, max_blocks, eps = 32, 8, 128, 16, 128, 16, 1e-6
    scale = 1.0 / math.sqrt(D)
    rng = np.random.default_rng(21)
    cfg = nv.SchedulerConfig(max_num_seqs=8, max_prefill_tokens=160, enable_chunked_prefill=True,
                             chunk_size=48, reschedule_partial_prefill=True, enable_prefix_caching=True)
    sched = nv.Scheduler(cfg, bs, nb)
    shared = rng.integers(0, 5000, size=64).tolist()              # 4 full blocks
    prompts = {1: shared + rng.integers(0, 5000, size=37).tolist(),
               2: rng.integers(0, 5000, size=150).tolist(),
               3: rng.integers(0, 5000, size=9).tolist()}
    late = {4: shared + rng.integers(0, 5000, size=20).tolist(),  # arrives after seq 1 is prefilled
            5: list(shared)}                                       # nothing but the shared prefix
    for sid, p in prompts.items():
        sched.add_sequence(sid, p)
    prompts.update(late)

    rot = nv.RotaryEmbedding(D, 1024, 1e6, torch.bfloat16, dev)
    cos, sin = rot.cos_cache().float().cpu().numpy(), rot.sin_cache().float().cpu().numpy()
    qw = oracle.bf16_round(1 + 0.1 * rng.standard_normal(D).astype(np.float32))
    kw = oracle.bf16_round(1 + 0.1 * rng.standard_normal(D).astype(np.float32))
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev, torch.bfloat16)
    kc = torch.zeros((nb, bs, Hkv, D), dtype=torch.bfloat16, device=dev)
    vc = torch.zeros_like(kc)
    okc = np.zeros((nb, bs, Hkv, D), np.float32)
    ovc = np.zeros_like(okc)

    # "model": the q/k/v projections of token t at position p depend only on (t, p), so sequences
    # sharing a prefix produce identical K/V for it, as a real model does
    def qkv(tokens, positions):
        q = np.empty((len(tokens), Hq, D), np.float32)
        k = np.empty((len(tokens), Hkv, D), np.float32)
        v = np.empty((len(tokens), Hkv, D), np.float32)
        for i, (t, p) in enumerate(zip(tokens, positions)):
            g = np.random.default_rng(t * 100003 + p)
            q[i], k[i], v[i] = bf16_randn(g, (Hq, D)), bf16_randn(g, (Hkv, D)), bf16_randn(g, (Hkv, D))
        return q, k, v

    def all_tokens(sid):
        return prompts[sid] + sched.get_sequence(sid).output_token_ids()

    def run_launch(batch, ids, T):
        nonlocal okc, ovc
        bt, cl, sm, qs, ps = batch.upload(dev)
        pos, ctxs = batch.positions(), batch.context_lens()
        toks = [all_tokens(sid)[c - T:c] for sid, c in zip(ids, ctxs)]
        q, k, v = qkv([t for ts in toks for t in ts], pos)
        qd = nv.qk_norm_rope_cache(d(q), d(k), d(v), d(qw), d(kw), eps, rot, ps, kc, vc, sm)
        got = nv.paged_attention_batched(qd.reshape(len(ids), T, Hq, D), kc, vc, bt, cl, scale)
        got = got.float().cpu().numpy()
        slots = batch.slot_mapping()
        for i, sid in enumerate(ids):
            sl = slice(i * T, (i + 1) * T)
            wq = np.empty((T, Hq, D), np.float32)
            wk = np.empty((T, Hkv, D), np.float32)
            for j in range(T):
                a, b = oracle.qk_norm_rope(q[None, i * T + j:i * T + j + 1], k[None, i * T + j:i * T + j + 1],
                                           qw, kw, eps, cos, sin, int(pos[i * T + j]))
                wq[j], wk[j] = a[0, 0], b[0, 0]
            okc, ovc = oracle.write_kv_to_cache(oracle.bf16_round(wk)[None], v[None, sl], okc, ovc, slots[sl], bs)
            want = oracle.paged_attention(oracle.bf16_round(wq)[None], okc, ovc,
                                          sched.get_sequence(sid).block_ids(), bs, int(ctxs[i]), Hq, Hkv, D, scale)
            assert_attention_close(got[i:i + 1], want)

    saw_flash_chunk = saw_shared = False
    for step in range(14):
        if step == 4:
            for sid in late:
                sched.add_sequence(sid, prompts[sid])
        out = sched.schedule()
        for T in sorted(set(out.prefill_chunk_sizes.values())):
            batch, ids = sched.prefill_batch(T, max_blocks)
            saw_flash_chunk |= T * (Hq // Hkv) > 32
            run_launch(batch, ids, T)
        if out.decode_sequences:
            # decode feeds the token sampled last step; the first one comes from the prefill
            batch = sched.decode_batch(max_blocks)
            run_launch(batch, out.decode_sequences, 1)
        for sid in out.prefill_sequences:
            sched.mark_prefilled(sid, out.prefill_chunk_sizes[sid])
            if sched.get_sequence(sid).is_prefill_complete():
                sched.append_token(sid, int(rng.integers(0, 5000)))     # "sampled" from the last prompt position
        for sid in out.decode_sequences:
            sched.append_token(sid, int(rng.integers(0, 5000)))
        if step >= 5:
            s1, s4, s5 = (sched.get_sequence(i).block_ids() for i in (1, 4, 5))
            if s4 and s5:
                saw_shared = s4[:4] == s1[:4] and s5[:4] == s1[:4]
    assert saw_flash_chunk and saw_shared
    for sid in prompts:
        q = sched.get_sequence(sid)
        assert q.is_prefill_complete() and q.output_len() >= 3
    # the caches hold exactly what the oracle wrote (same rounding of the same values, +-1 ulp)
    diff = np.abs(kc.float().cpu().numpy() - oracle.bf16_round(okc))
    assert (diff <= 2.0 ** -7 * np.abs(okc) + 1e-6).all()
    assert np.array_equal(vc.float().cpu().numpy(), ovc)
