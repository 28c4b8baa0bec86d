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
from helpers import assert_attention_close

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
def test_gpu_kv_write_goldens_bit_exact(dev):
    for name, c in _load("kv_write.npz").items():
        bs = c["kc0"].shape[1]
        kc, vc = _bf16_dev(c["kc0"], dev), _bf16_dev(c["vc0"], dev)
        nv.write_kv_to_cache(_bf16_dev(c["key"], dev), _bf16_dev(c["value"], dev), kc, vc,
                             c["slots"].tolist(), bs)
        torch.cuda.synchronize()
        assert np.array_equal(_bits_host(kc), c["kc"]) and np.array_equal(_bits_host(vc), c["vc"]), name
