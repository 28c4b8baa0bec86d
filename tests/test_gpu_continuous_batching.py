"""End-to-end plumbing on the GPU: the scheduler-shaped flow the drop-in is
for -- allocate blocks, build the step's flat metadata, write the new tokens' K/V
(reshape_and_cache), attend over the paged cache -- for a continuous batch that mixes chunked
prefill, decode and speculative verification, checked step by step against the CPU oracle run the
reference's way (one write_kv_to_cache + paged_attention per sequence).  Rejected draft tokens
are rolled back by truncating context_len; their stale cache slots must never be seen again."""
import math

import numpy as np
import pytest
import torch

import nano_vllm_rs_amd as nv
import oracle
from helpers import assert_attention_close, bf16_randn

pytestmark = pytest.mark.gpu


def test_mixed_prefill_decode_verify_with_rollback(dev):
    Hq, Hkv, D, bs = 16, 8, 128, 16
    nb, max_blocks = 256, 24
    scale = 1.0 / math.sqrt(D)
    rng = np.random.default_rng(7)
    mgr = nv.BlockManager(nb, bs)
    kc = torch.full((nb, bs, Hkv, D), float("nan"), dtype=torch.bfloat16, device=dev)
    vc = torch.full_like(kc, float("nan"))
    okc = np.full((nb, bs, Hkv, D), np.nan, np.float32)   # oracle's cache
    ovc = np.full_like(okc, np.nan)

    nseq = 5
    tables = [nv.BlockTable(bs) for _ in range(nseq)]
    ctx = [0] * nseq                       # tokens in the cache per sequence
    # per step: new tokens per sequence (0 = not scheduled) and how many of them survive
    # (speculative verification keeps a prefix of the T new tokens)
    plan = [
        ([8, 5, 0, 3, 8], None),           # chunked prefill, first chunks (T <= 8 per launch)
        ([8, 1, 7, 1, 2], None),           # more prefill, two sequences already decoding
        ([1, 1, 1, 1, 1], None),           # plain decode step
        ([5, 5, 5, 5, 5], [2, 5, 0, 1, 3]),  # verify K+1 = 5 drafted tokens, keep a prefix
        ([1, 1, 1, 1, 1], None),           # decode after the rollback: stale slots get rewritten
        ([4, 0, 4, 4, 0], [4, 0, 1, 2, 0]),
        ([1, 1, 1, 1, 1], None),
    ]
    batch = nv.BatchMetadata(max_blocks)
    for new_counts, keep in plan:
        sched = [i for i in range(nseq) if new_counts[i] > 0]
        T = max(new_counts[i] for i in sched)
        # the batched kernel takes one T for the launch: group sequences by their T
        for tcur in sorted({new_counts[i] for i in sched}):
            group = [i for i in sched if new_counts[i] == tcur]
            batch.clear()
            k_new = bf16_randn(rng, (len(group), tcur, Hkv, D))
            v_new = bf16_randn(rng, (len(group), tcur, Hkv, D))
            q_new = bf16_randn(rng, (len(group), tcur, Hq, D))
            for gi, i in enumerate(group):
                total = ctx[i] + tcur
                while tables[i].num_blocks() < nv.compute_num_blocks(total, bs):
                    tables[i].append_block(mgr.allocate())
                batch.add_sequence(tables[i], total, tcur)
            bt, cl, sm, qs, _ps = batch.upload(dev)
            assert qs.cpu().tolist() == [tcur * j for j in range(len(group) + 1)]
            to = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev, torch.bfloat16)
            nv.reshape_and_cache(to(k_new.reshape(-1, Hkv, D)), to(v_new.reshape(-1, Hkv, D)), kc, vc, sm)
            got = nv.paged_attention_batched(to(q_new), kc, vc, bt, cl, scale).float().cpu().numpy()
            # the reference's way, per sequence, on the oracle's own cache
            for gi, i in enumerate(group):
                total = ctx[i] + tcur
                slots = tables[i].get_slot_mapping(total)[ctx[i]:]
                okc, ovc = oracle.write_kv_to_cache(k_new[gi:gi + 1], v_new[gi:gi + 1], okc, ovc, slots, bs)
                want = oracle.paged_attention(q_new[gi:gi + 1], np.nan_to_num(okc), np.nan_to_num(ovc),
                                              tables[i].get_physical_block_ids(), bs, total, Hq, Hkv,
                                              D, scale)
                assert_attention_close(got[gi:gi + 1], want)
                ctx[i] = total if keep is None else ctx[i] + keep[i]   # rollback = shorter context
        # cache contents written so far are bit-exact (stale slots of rejected tokens included:
        # both sides wrote the same bytes there)
        live = ~np.isnan(okc)
        assert np.array_equal(kc.float().cpu().numpy()[live], okc[live])
        assert np.array_equal(vc.float().cpu().numpy()[live], ovc[live])
    assert all(c > 0 for c in ctx)
