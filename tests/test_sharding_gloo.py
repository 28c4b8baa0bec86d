"""N > 1 path on CPU: world_size-2 gloo processes.  The partitioning and the collectives are
the product code (nano-vllm-rs_amd/sharding.py); the per-shard attention is computed by the CPU
oracle (allowed in tests), so the assembled multi-rank result must equal the unsharded oracle
result exactly.  GPU ranks run the same code with backend "nccl" (RCCL)."""
import math
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import nano_vllm_rs_amd as nv
import oracle
from helpers import bf16_randn, make_paged_case, oracle_paged_batch


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _case():
    rng = np.random.default_rng(42)
    Hq, Hkv, D, bs, T = 8, 4, 16, 4, 2
    ctx_lens = [5, 9, 16, 3, 12, 7, 20]
    kc, vc, tables = make_paged_case(rng, ctx_lens, 64, bs, Hkv, D, poison=0.0)
    q = bf16_randn(rng, (len(ctx_lens), T, Hq, D))
    return q, kc, vc, tables, ctx_lens, bs, 1 / math.sqrt(D)


def _worker(rank, world, port, mode, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        q, kc, vc, tables, ctx_lens, bs, scale = _case()
        N, T, Hq, D = q.shape
        Hkv = kc.shape[2]
        sh = nv.sharding
        if mode == "sequence":
            b, e = sh.sequence_range(N, world, rank)
            local = oracle_paged_batch(q[b:e], kc, vc, tables[b:e], ctx_lens[b:e], bs, scale)
            full = sh.gather_sequences(torch.from_numpy(local), N)
        else:
            ql = sh.shard_query_heads(torch.from_numpy(q), Hkv, world, rank).numpy()
            kl = sh.shard_cache_heads(torch.from_numpy(kc), world, rank).numpy()
            vl = sh.shard_cache_heads(torch.from_numpy(vc), world, rank).numpy()
            local = oracle_paged_batch(ql, kl, vl, tables, ctx_lens, bs, scale)
            full = sh.all_gather_heads(torch.from_numpy(local), D)
        if rank == 0:
            ret["out"] = full.numpy().copy()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["sequence", "kv_head"])
def test_two_rank_result_equals_unsharded(mode):
    world, port = 2, _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, mode, ret), nprocs=world, join=True)
    q, kc, vc, tables, ctx_lens, bs, scale = _case()
    want = oracle_paged_batch(q, kc, vc, tables, ctx_lens, bs, scale)
    assert np.array_equal(ret["out"], want)


def test_ranges_partition_exactly():
    for n in (1, 7, 256, 1024):
        for world in (1, 2, 3, 4, 8):
            spans = [nv.sharding.sequence_range(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            assert max(e - b for b, e in spans) - min(e - b for b, e in spans) <= 1
    assert nv.sharding.kv_head_range(8, 4, 3) == (6, 8)
    assert nv.sharding.q_head_range(32, 8, 4, 3) == (24, 32)
    with pytest.raises(ValueError):
        nv.sharding.kv_head_range(8, 3, 0)
