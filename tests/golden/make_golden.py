#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ (run from the repo root).

The reference (Rust + candle) cannot run in the build image and holds NO golden vectors of
its own (its tests use unseeded randn), so these vectors are produced by the CPU oracle
(oracle/nvllm_oracle.c, a restatement of the reference pinned to the reference's own test
assertions in tests/test_oracle_pins.py).  Inputs are seeded N(0,1) rounded to bf16 and stored
as raw bf16 bit patterns (uint16); expected outputs are float32.  The cases are the list in
SURVEY.md 8(c), at sizes small enough to commit.
"""
import json
import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle  # noqa: E402
from helpers import bf16_randn, hashed_bf16, make_paged_case  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def bits(a):
    return (np.ascontiguousarray(a, dtype=np.float32).view(np.uint32) >> 16).astype(np.uint16)


def flash_cases():
    cases = {}
    specs = [("mha_16x4x32", (1, 16, 4, 4, 32), True), ("gqa_16x8x2x32", (1, 16, 8, 2, 32), True),
             ("b2_128x4x64", (2, 128, 4, 4, 64), True), ("qwen_64x16x8x128", (1, 64, 16, 8, 128), True),
             ("noncausal_16x2x32", (1, 16, 2, 2, 32), False), ("ragged_77x8x2x128", (1, 77, 8, 2, 128), True)]
    for i, (name, (B, S, Hq, Hkv, D), causal) in enumerate(specs):
        rng = np.random.default_rng(100 + i)
        q, k, v = bf16_randn(rng, (B, S, Hq, D)), bf16_randn(rng, (B, S, Hkv, D)), bf16_randn(rng, (B, S, Hkv, D))
        out = oracle.flash_attention_cpu(q, k, v, causal)
        cases[name] = dict(q=bits(q), k=bits(k), v=bits(v), out=out, causal=np.array(causal))
    return cases


def paged_cases():
    cases = {}
    rng = np.random.default_rng(200)
    # bs 4, table [0,2,5], ctx 10, T=1  (tests/paged_attention_test.rs:266-301 geometry)
    for name, table, ctx, T in [("bs4_t025_ctx10_T1", [0, 2, 5], 10, 1), ("bs4_t012_ctx12_T4", [0, 1, 2], 12, 4)]:
        kc, vc = bf16_randn(rng, (8, 4, 2, 8)), bf16_randn(rng, (8, 4, 2, 8))
        q = bf16_randn(rng, (1, T, 4, 8))
        out = oracle.paged_attention(q, kc, vc, table, 4, ctx, 4, 2, 8, 1 / math.sqrt(8))
        cases[name] = dict(q=bits(q), kc=bits(kc), vc=bits(vc), table=np.array(table, np.int32),
                           ctx=np.array(ctx), out=out)
    # Qwen3-4B head shape, one sequence, ragged context
    kc, vc, tables = make_paged_case(rng, [301], 24, 16, 8, 128, poison=0.0)
    q = bf16_randn(rng, (1, 1, 32, 128))
    out = oracle.paged_attention(q, kc, vc, tables[0], 16, 301, 32, 8, 128, 1 / math.sqrt(128))
    cases["qwen4b_ctx301_T1"] = dict(q=bits(q), kc=bits(kc), vc=bits(vc),
                                     table=np.array(tables[0], np.int32), ctx=np.array(301), out=out)
    # two sequences sharing their first 3 blocks, T=4 (speculative verify)
    kc, vc, tables = make_paged_case(rng, [100, 83], 24, 16, 8, 128, shared_prefix_blocks=3, poison=0.0)
    q = bf16_randn(rng, (2, 4, 32, 128))
    outs = [oracle.paged_attention(q[i:i + 1], kc, vc, tables[i], 16, c, 32, 8, 128, 1 / math.sqrt(128))
            for i, c in enumerate([100, 83])]
    tab = np.zeros((2, 7), np.int32)
    for i, t in enumerate(tables):
        tab[i, :len(t)] = t
    cases["shared_prefix_T4"] = dict(q=bits(q), kc=bits(kc), vc=bits(vc), table=tab,
                                     ctx=np.array([100, 83]), out=np.concatenate(outs, 0))
    return cases


def paged_hashed_cases():
    """block 16, 32 Q / 8 KV heads, D 128, ONE sequence of ctx 2048, T = 1 (one sequence
    of BASELINE config 3).  Its 2 x 4 MiB of K/V are not committed: inputs come from
    helpers.hashed_bf16 (integer hash of the element index; stream ids below), only the table and
    the oracle's output are stored."""
    nb, bs, Hq, Hkv, D, ctx = 160, 16, 32, 8, 128, 2048
    rng = np.random.default_rng(201)
    table = rng.permutation(nb)[:ctx // bs].astype(np.int32)
    kc = hashed_bf16((nb, bs, Hkv, D), 1)
    vc = hashed_bf16((nb, bs, Hkv, D), 2)
    q = hashed_bf16((1, 1, Hq, D), 3)
    out = oracle.paged_attention(q, kc, vc, table.tolist(), bs, ctx, Hq, Hkv, D, 1 / math.sqrt(D))
    return {"qwen4b_ctx2048_T1": dict(streams=np.array([1, 2, 3]), shape=np.array([nb, bs, Hq, Hkv, D]),
                                      table=table, ctx=np.array(ctx), out=out)}


def kv_write_cases():
    cases = {}
    rng = np.random.default_rng(300)
    for name, slots, nb, bs, Hkv, D in [("slots_0_1_4", [0, 1, 4], 2, 4, 2, 4),
                                        ("duplicates", [3, 5, 3, 7, 5, 5, 0, 3, 15], 4, 4, 2, 8),
                                        ("table_5_12_len20", oracle.get_slot_mapping([5, 12], 16, 20).tolist(), 14, 16, 8, 128)]:
        S = len(slots)
        key, val = bf16_randn(rng, (1, S, Hkv, D)), bf16_randn(rng, (1, S, Hkv, D))
        kc0, vc0 = bf16_randn(rng, (nb, bs, Hkv, D)), bf16_randn(rng, (nb, bs, Hkv, D))
        kc, vc = oracle.write_kv_to_cache(key, val, kc0, vc0, slots, bs)
        cases[name] = dict(key=bits(key), value=bits(val), kc0=bits(kc0), vc0=bits(vc0),
                           slots=np.array(slots, np.int64), kc=bits(kc), vc=bits(vc))
    return cases


def allocator_traces():
    """Operation traces with the ids/return values the reference semantics produce."""
    traces = []
    rng = np.random.default_rng(400)
    for n in (4, 16):
        m = oracle.BlockManagerOracle(n, 16)
        ops, live = [], []
        for _ in range(120):
            op = int(rng.integers(0, 5))
            if op == 0:
                try:
                    r = m.allocate(); live.append(r)
                except oracle.OutOfBlocks:
                    r = "OutOfBlocks"
                ops.append(["allocate", None, r])
            elif op == 1 and live:
                x = live[int(rng.integers(0, len(live)))]
                ops.append(["free", x, m.free(x)])
            elif op == 2:
                h = int(rng.integers(0, 4))
                try:
                    r = list(m.allocate_with_prefix(h)); live.append(r[0])
                except oracle.OutOfBlocks:
                    r = "OutOfBlocks"
                ops.append(["allocate_with_prefix", h, r])
            elif op == 3 and live:
                x = live[int(rng.integers(0, len(live)))]
                ops.append(["increment_ref", x, m.increment_ref(x)])
            elif op == 4:
                ops.append(["counts", None, [m.num_free_blocks(), m.num_used_blocks(), m.num_cached_prefixes()]])
        traces.append(dict(num_blocks=n, ops=ops))
    hashes = [dict(tokens=[1, 2, 3, 4], parent=None, hash=oracle.hash_token_block([1, 2, 3, 4])),
              dict(tokens=[1, 2, 3, 4], parent=999, hash=oracle.hash_token_block([1, 2, 3, 4], 999)),
              dict(tokens=list(range(16)), parent=None, hash=oracle.hash_token_block(list(range(16)))),
              dict(tokens=[], parent=None, hash=oracle.hash_token_block([]))]
    return dict(traces=traces, hashes=hashes,
                note="hash values follow SipHash-1-3 with zero keys (Rust DefaultHasher); the "
                     "reference pins only equality/inequality, so values are best-effort")


def norm_rope_cases():
    """K5: per-head RMSNorm + RoPE on q/k, then K/V into their slots.  bf16 inputs,
    weights and cos/sin tables (the reference holds them in the model dtype); expected values are
    the oracle's f32 results before the final rounding."""
    cases = {}
    rng = np.random.default_rng(500)
    for name, n, Hq, Hkv, D, nb, bs, max_pos, theta in [("d128_gqa", 19, 8, 2, 128, 4, 16, 256, 1e6),
                                                        ("d64_mha", 7, 4, 4, 64, 2, 8, 64, 1e4),
                                                        # positions near 32k: where rope.rs:55's f32
                                                        # division shows (a f64 one moves cos/sin by 4e-4)
                                                        ("d128_longpos", 11, 4, 2, 128, 2, 16, 32768, 1e6)]:
        q, k, v = bf16_randn(rng, (n, Hq, D)), bf16_randn(rng, (n, Hkv, D)), bf16_randn(rng, (n, Hkv, D))
        qw = oracle.bf16_round(1.0 + 0.25 * bf16_randn(rng, (D,)))
        kw = oracle.bf16_round(1.0 + 0.25 * bf16_randn(rng, (D,)))
        pos = rng.integers(max_pos - max_pos // 8 if max_pos > 4096 else 0, max_pos, size=n).astype(np.int32)
        slots = rng.permutation(nb * bs)[:n].astype(np.int64)
        cos, sin = oracle.rope_cache(D, max_pos, theta)
        cos, sin = oracle.bf16_round(cos), oracle.bf16_round(sin)
        qo, ko = np.empty_like(q), np.empty_like(k)
        for i in range(n):
            a, b = oracle.qk_norm_rope(q[None, i:i + 1], k[None, i:i + 1], qw, kw, 1e-6, cos, sin, int(pos[i]))
            qo[i], ko[i] = a[0, 0], b[0, 0]
        kc0, vc0 = bf16_randn(rng, (nb, bs, Hkv, D)), bf16_randn(rng, (nb, bs, Hkv, D))
        kc, vc = oracle.write_kv_to_cache(ko[None], v[None], kc0, vc0, slots, bs)
        cases[name] = dict(q=bits(q), k=bits(k), v=bits(v), qw=bits(qw), kw=bits(kw), pos=pos, slots=slots,
                           theta=np.float64(theta), max_pos=np.int64(max_pos), eps=np.float64(1e-6),
                           kc0=bits(kc0), vc0=bits(vc0), q_out=qo, k_out=ko, kc=kc, vc=bits(vc))
    return cases


def scheduler_traces():
    """Operation traces of the scheduler with every decision the reference's
    semantics produce: sequence ids per list, chunk sizes, token counts, queue sizes, block ids."""
    traces = []
    for seed in (0, 1, 2, 3):
        rng = np.random.default_rng(600 + seed)
        cfg = dict(max_num_seqs=int(rng.integers(2, 9)), max_prefill_tokens=int(rng.integers(20, 200)),
                   enable_chunked_prefill=bool(seed & 1), chunk_size=int(rng.integers(8, 64)),
                   enable_priority=True, enable_preemption=bool(seed & 2))
        nb = int(rng.integers(6, 40))
        s = oracle.SchedulerOracle(block_size=16, num_blocks=nb, **cfg)
        ops, next_id, live = [], 1, []
        for _ in range(100):
            r = rng.random()
            if r < 0.35:
                prompt = rng.integers(0, 1000, size=int(rng.integers(1, 90))).tolist()
                prio = int(rng.integers(-3, 4))
                s.add_sequence(oracle.SequenceOracle(next_id, prompt, prio))
                ops.append(["add", next_id, prompt, prio])
                live.append(next_id)
                next_id += 1
            elif r < 0.75:
                o = s.schedule()
                for sid in o.prefill_sequences:
                    s.mark_prefilled(sid, o.prefill_chunk_sizes[sid])
                for sid in o.decode_sequences:
                    s.append_token(sid, 7)
                ops.append(["schedule", dict(prefill=list(o.prefill_sequences), decode=list(o.decode_sequences),
                                             preempted=list(o.preempted_sequences),
                                             chunks={str(k): int(v) for k, v in o.prefill_chunk_sizes.items()},
                                             tokens=[int(o.num_prefill_tokens), int(o.num_decode_tokens)])])
            elif r < 0.87 and live:
                sid = live[int(rng.integers(len(live)))]
                s.finish_sequence(sid, 1)
                ops.append(["finish", sid])
            elif live:
                sid = live.pop(int(rng.integers(len(live))))
                ops.append(["remove", sid, s.remove_sequence(sid) is not None])
            else:
                continue
            ops[-1].append(dict(queues=[s.num_waiting(), s.num_running(), s.num_swapped()],
                                used=s.bm.num_used_blocks(), running=list(s.running_ids),
                                blocks={str(i): list(s.sequences[i].block_ids) for i in live if i in s.sequences}))
        traces.append(dict(config=cfg, block_size=16, num_blocks=nb, ops=ops))
    return dict(traces=traces, finish_reason="1 = MaxTokens; every schedule is followed by the engine's "
                "step: mark_prefilled(chunk) per prefill id, append_token(7) per decode id")


def main():
    np.savez_compressed(os.path.join(OUT, "flash.npz"),
                        **{"%s/%s" % (c, k): v for c, d in flash_cases().items() for k, v in d.items()})
    np.savez_compressed(os.path.join(OUT, "paged.npz"),
                        **{"%s/%s" % (c, k): v for c, d in paged_cases().items() for k, v in d.items()})
    np.savez_compressed(os.path.join(OUT, "paged_hashed.npz"),
                        **{"%s/%s" % (c, k): v for c, d in paged_hashed_cases().items() for k, v in d.items()})
    np.savez_compressed(os.path.join(OUT, "kv_write.npz"),
                        **{"%s/%s" % (c, k): v for c, d in kv_write_cases().items() for k, v in d.items()})
    np.savez_compressed(os.path.join(OUT, "norm_rope.npz"),
                        **{"%s/%s" % (c, k): v for c, d in norm_rope_cases().items() for k, v in d.items()})
    with open(os.path.join(OUT, "allocator.json"), "w") as f:
        json.dump(allocator_traces(), f)
    with open(os.path.join(OUT, "scheduler.json"), "w") as f:
        json.dump(scheduler_traces(), f)
    for n in ("flash.npz", "paged.npz", "paged_hashed.npz", "kv_write.npz", "norm_rope.npz", "allocator.json", "scheduler.json"):
        print(n, os.path.getsize(os.path.join(OUT, n)))


if __name__ == "__main__":
    main()
