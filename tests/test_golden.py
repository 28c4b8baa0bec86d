"""Committed golden fixtures (tests/golden/, made by tests/golden/make_golden.py from the CPU
oracle -- the reference holds no vectors of its own and cannot run here).
CPU part: the oracle still reproduces them; the product allocator/hash replay the traces
bit-exactly.  GPU part (marked gpu): the HIP kernels reproduce them through the C ABI."""
import json
import math
import os

import numpy as np
import pytest
import torch

import nano_vllm_rs_amd as nv
import oracle
from helpers import assert_attention_close, hashed_bf16

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _load(name):
    z = np.load(os.path.join(G, name), allow_pickle=False)
    cases = {}
    for key in z.files:
        c, f = key.split("/")
        cases.setdefault(c, {})[f] = z[key]
    return cases


def _f32(bits):
    return (bits.astype(np.uint32) << 16).view(np.float32)


def _bf16_dev(bits, dev):
    return torch.from_numpy(bits.view(np.int16).copy()).to(dev).view(torch.bfloat16)


def _bits_host(t):
    return t.view(torch.int16).cpu().numpy().view(np.uint16)


# ---------------------------------------------------------------- CPU: oracle + host logic
def test_oracle_reproduces_flash_goldens():
    for name, c in _load("flash.npz").items():
        out = oracle.flash_attention_cpu(_f32(c["q"]), _f32(c["k"]), _f32(c["v"]), bool(c["causal"]))
        assert np.abs(out - c["out"]).max() < 1e-5, name


def test_oracle_reproduces_paged_goldens():
    for name, c in _load("paged.npz").items():
        q, kc, vc = _f32(c["q"]), _f32(c["kc"]), _f32(c["vc"])
        nb, bs, Hkv, D = kc.shape
        tabs = np.atleast_2d(c["table"])
        ctxs = np.atleast_1d(c["ctx"])
        for i in range(q.shape[0]):
            n = oracle.compute_num_blocks(int(ctxs[i]), bs)
            out = oracle.paged_attention(q[i:i + 1], kc, vc, tabs[i][:n], bs, int(ctxs[i]),
                                         q.shape[2], Hkv, D, 1 / math.sqrt(D))
            assert np.abs(out - c["out"][i:i + 1]).max() < 1e-5, name


def _hashed_case(c):
    nb, bs, Hq, Hkv, D = (int(x) for x in c["shape"])
    sk, sv, sq = (int(x) for x in c["streams"])
    return (hashed_bf16((1, 1, Hq, D), sq), hashed_bf16((nb, bs, Hkv, D), sk),
            hashed_bf16((nb, bs, Hkv, D), sv), bs, Hq, Hkv, D)


def test_oracle_reproduces_hashed_paged_golden():
    # SURVEY 8c's ctx-2048 vector (one sequence of config 3); inputs regenerated from stream ids
    for name, c in _load("paged_hashed.npz").items():
        q, kc, vc, bs, Hq, Hkv, D = _hashed_case(c)
        assert abs(float(kc.std()) - 1.0) < 0.02 and abs(float(kc.mean())) < 0.01
        out = oracle.paged_attention(q, kc, vc, c["table"].tolist(), bs, int(c["ctx"]), Hq, Hkv, D,
                                     1 / math.sqrt(D))
        assert np.abs(out - c["out"]).max() < 1e-5, name


def test_oracle_reproduces_kv_write_goldens():
    for name, c in _load("kv_write.npz").items():
        bs = c["kc0"].shape[1]
        kc, vc = oracle.write_kv_to_cache(_f32(c["key"]), _f32(c["value"]), _f32(c["kc0"]),
                                          _f32(c["vc0"]), c["slots"], bs)
        assert np.array_equal(kc, _f32(c["kc"])) and np.array_equal(vc, _f32(c["vc"])), name


def test_product_allocator_replays_traces_bit_exact():
    data = json.load(open(os.path.join(G, "allocator.json")))
    for tr in data["traces"]:
        m = nv.BlockManager(tr["num_blocks"], 16)
        for op, arg, want in tr["ops"]:
            if op == "allocate":
                try:
                    got = m.allocate()
                except nv.OutOfBlocksError:
                    got = "OutOfBlocks"
            elif op == "free":
                got = m.free(arg)
            elif op == "allocate_with_prefix":
                try:
                    got = list(m.allocate_with_prefix(arg))
                except nv.OutOfBlocksError:
                    got = "OutOfBlocks"
            elif op == "increment_ref":
                got = m.increment_ref(arg)
            else:
                got = [m.num_free_blocks(), m.num_used_blocks(), m.num_cached_prefixes()]
            assert got == want, (op, arg, got, want)
    for h in data["hashes"]:
        assert nv.hash_token_block(h["tokens"], h["parent"]) == h["hash"]
        assert oracle.hash_token_block(h["tokens"], h["parent"]) == h["hash"]


def _norm_rope_tables(c):
    cos, sin = oracle.rope_cache(c["q"].shape[-1], int(c["max_pos"]), float(c["theta"]))
    return oracle.bf16_round(cos), oracle.bf16_round(sin)


def test_oracle_reproduces_norm_rope_goldens():
    for name, c in _load("norm_rope.npz").items():
        q, k, v = _f32(c["q"]), _f32(c["k"]), _f32(c["v"])
        cos, sin = _norm_rope_tables(c)
        for i in range(q.shape[0]):
            a, b = oracle.qk_norm_rope(q[None, i:i + 1], k[None, i:i + 1], _f32(c["qw"]), _f32(c["kw"]),
                                       float(c["eps"]), cos, sin, int(c["pos"][i]))
            assert np.array_equal(a[0, 0], c["q_out"][i]) and np.array_equal(b[0, 0], c["k_out"][i]), name
        bs = c["kc0"].shape[1]
        kc, vc = oracle.write_kv_to_cache(c["k_out"][None], v[None], _f32(c["kc0"]), _f32(c["vc0"]),
                                          c["slots"], bs)
        assert np.array_equal(kc, c["kc"]) and np.array_equal(vc, _f32(c["vc"])), name


def _replay_scheduler(tr, make, add, seq_blocks, running, used, remove):
    s = make(tr)
    live = []
    for op in tr["ops"]:
        kind, state = op[0], op[-1]
        if kind == "add":
            add(s, op[1], op[2], op[3])
            live.append(op[1])
        elif kind == "schedule":
            o, want = s.schedule(), op[1]
            assert list(o.prefill_sequences) == want["prefill"]
            assert list(o.decode_sequences) == want["decode"]
            assert list(o.preempted_sequences) == want["preempted"]
            assert {str(k): int(v) for k, v in o.prefill_chunk_sizes.items()} == want["chunks"]
            assert [o.num_prefill_tokens, o.num_decode_tokens] == want["tokens"]
            for sid in o.prefill_sequences:
                s.mark_prefilled(sid, o.prefill_chunk_sizes[sid])
            for sid in o.decode_sequences:
                s.append_token(sid, 7)
        elif kind == "finish":
            s.finish_sequence(op[1], 1)
        else:
            assert remove(s, op[1]) == op[2]
            live.remove(op[1])
        assert [s.num_waiting(), s.num_running(), s.num_swapped()] == state["queues"]
        assert used(s) == state["used"] and list(running(s)) == state["running"]
        for sid, blocks in state["blocks"].items():
            assert seq_blocks(s, int(sid)) == blocks, (kind, sid)


def test_oracle_and_product_scheduler_replay_traces_bit_exact():
    data = json.load(open(os.path.join(G, "scheduler.json")))
    for tr in data["traces"]:
        _replay_scheduler(
            tr, lambda t: oracle.SchedulerOracle(block_size=t["block_size"], num_blocks=t["num_blocks"], **t["config"]),
            lambda s, i, p, pr: s.add_sequence(oracle.SequenceOracle(i, p, pr)),
            lambda s, i: list(s.sequences[i].block_ids), lambda s: s.running_ids,
            lambda s: s.bm.num_used_blocks(), lambda s, i: s.remove_sequence(i) is not None)
        _replay_scheduler(
            tr, lambda t: nv.Scheduler(nv.SchedulerConfig(**t["config"]), t["block_size"], t["num_blocks"]),
            lambda s, i, p, pr: s.add_sequence(i, p, pr),
            lambda s, i: list(s.get_sequence(i).block_ids()), lambda s: s.running_sequence_ids(),
            lambda s: s.num_used_blocks(), lambda s, i: bool(s.remove_sequence(i)))


# ---------------------------------------------------------------- GPU: kernels via the C ABI
@pytest.mark.gpu
def test_gpu_flash_goldens(dev):
    for name, c in _load("flash.npz").items():
        D = c["q"].shape[-1]
        out = nv.flash_attention(_bf16_dev(c["q"], dev), _bf16_dev(c["k"], dev), _bf16_dev(c["v"], dev),
                                 nv.FlashAttentionConfig.new(D, bool(c["causal"])))
        assert_attention_close(out.float().cpu().numpy(), c["out"])


@pytest.mark.gpu
def test_gpu_paged_goldens(dev):
    for name, c in _load("paged.npz").items():
        kc, vc, q = _bf16_dev(c["kc"], dev), _bf16_dev(c["vc"], dev), _bf16_dev(c["q"], dev)
        D = q.shape[-1]
        tabs = torch.from_numpy(np.atleast_2d(c["table"]).astype(np.int32)).to(dev)
        ctxs = torch.from_numpy(np.atleast_1d(c["ctx"]).astype(np.int32)).to(dev)
        out = nv.paged_attention_batched(q, kc, vc, tabs, ctxs, 1 / math.sqrt(D))
        assert_attention_close(out.float().cpu().numpy(), c["out"])


@pytest.mark.gpu
def test_gpu_hashed_paged_golden(dev):
    """ctx 2048, 32Q/8KV, D128, block 16, T=1 against the committed oracle output; blocks outside
    the table are NaN so a wrong block id cannot pass."""
    for name, c in _load("paged_hashed.npz").items():
        q, kc, vc, bs, Hq, Hkv, D = _hashed_case(c)
        dead = np.ones(kc.shape[0], bool)
        dead[c["table"]] = False
        kc[dead] = np.nan
        vc[dead] = np.nan
        to = lambda a: torch.from_numpy(a).to(dev, torch.bfloat16)
        out = nv.paged_attention_batched(to(q), to(kc), to(vc), torch.from_numpy(c["table"][None]).to(dev),
                                         torch.tensor([int(c["ctx"])], dtype=torch.int32, device=dev),
                                         1 / math.sqrt(D))
        assert_attention_close(out.float().cpu().numpy(), c["out"])


@pytest.mark.gpu
def test_gpu_kv_write_goldens_bit_exact(dev):
    for name, c in _load("kv_write.npz").items():
        bs = c["kc0"].shape[1]
        kc, vc = _bf16_dev(c["kc0"], dev), _bf16_dev(c["vc0"], dev)
        nv.write_kv_to_cache(_bf16_dev(c["key"], dev), _bf16_dev(c["value"], dev), kc, vc,
                             c["slots"].tolist(), bs)
        torch.cuda.synchronize()
        assert np.array_equal(_bits_host(kc), c["kc"]) and np.array_equal(_bits_host(vc), c["vc"]), name


@pytest.mark.gpu
def test_gpu_norm_rope_goldens(dev):
    """K5 through the C ABI: q/k within one bf16 ulp of the golden f32 values (the kernel rounds
    once), V and untouched slots of both caches bit-exact."""
    for name, c in _load("norm_rope.npz").items():
        D = c["q"].shape[-1]
        rot = nv.RotaryEmbedding(D, int(c["max_pos"]), float(c["theta"]), torch.bfloat16, dev)
        kc, vc = _bf16_dev(c["kc0"], dev), _bf16_dev(c["vc0"], dev)
        k_out = torch.empty(c["k"].shape, dtype=torch.bfloat16, device=dev)
        q_out = nv.qk_norm_rope_cache(_bf16_dev(c["q"], dev), _bf16_dev(c["k"], dev), _bf16_dev(c["v"], dev),
                                      _bf16_dev(c["qw"], dev), _bf16_dev(c["kw"], dev), float(c["eps"]), rot,
                                      torch.from_numpy(c["pos"]).to(dev), kc, vc,
                                      torch.from_numpy(c["slots"]).to(dev), k_out=k_out)
        torch.cuda.synchronize()
        for got, want in ((q_out, c["q_out"]), (k_out, c["k_out"]), (kc, c["kc"])):
            got = got.float().cpu().numpy()
            assert np.all(np.abs(got - want) <= (2.0 ** -7) * np.abs(want) + 1e-6), name
        assert np.array_equal(_bits_host(vc), c["vc"]), name
        bs = c["kc0"].shape[1]
        untouched = np.ones(c["kc0"].shape[0] * bs, bool)
        untouched[c["slots"]] = False
        kcb = _bits_host(kc).reshape(-1, *c["kc0"].shape[2:])
        assert np.array_equal(kcb[untouched], c["kc0"].reshape(kcb.shape)[untouched]), name
