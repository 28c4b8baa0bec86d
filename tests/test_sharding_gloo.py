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


def test_prefix_sharding_colocates_shared_first_blocks():
    """sequences that share their first block (the key allocate_with_prefix looks up,
    block_manager.rs:337-347) go to ONE rank, so the shared blocks exist on one GPU only: with a
    BlockManager per rank, the placement costs exactly one copy of the prefix in the whole job,
    against one per rank under contiguous ranges."""
    sh = nv.sharding
    bs, world = 16, 4
    prompts = []
    for fam in range(6):                                  # six prompt families of 5 sequences each
        head = np.arange(fam * 100, fam * 100 + 2 * bs, dtype=np.uint32)
        for j in range(5):
            prompts.append(np.concatenate([head, np.arange(7 + j, dtype=np.uint32) + 9000 * (j + 1)]))
    prompts.append(np.arange(5, dtype=np.uint32))         # shorter than a block: no hash
    rng = np.random.default_rng(0)
    order = rng.permutation(len(prompts))
    prompts = [prompts[i] for i in order]
    hashes = [nv.hash_token_block(p[:bs], None) if len(p) >= bs else None for p in prompts]

    ranks = [sh.shard_by_prefix(h, world, seq_id=i) for i, h in enumerate(hashes)]
    assert all(0 <= r < world for r in ranks)
    by_hash = {}
    for h, r in zip(hashes, ranks):
        if h is not None:
            by_hash.setdefault(h, set()).add(r)
    assert len(by_hash) == 6 and all(len(v) == 1 for v in by_hash.values())
    assert sh.shard_by_prefix(None, world, seq_id=9) == 1

    parts = sh.partition_by_prefix(hashes, world)
    assert sorted(i for p in parts for i in p) == list(range(len(prompts)))
    for h in by_hash:
        owners = {r for r, p in enumerate(parts) for i in p if hashes[i] == h}
        assert len(owners) == 1
    sizes = [len(p) for p in parts]
    assert max(sizes) - min(sizes) <= 5                   # at most the largest group apart
    assert parts == sh.partition_by_prefix(hashes, world)  # deterministic

    def prefix_block_copies(assignment):
        """blocks spent on the two shared head blocks over all ranks, one BlockManager per rank"""
        copies = 0
        for idxs in assignment:
            mgr = nv.BlockManager(4096, bs)
            for i in idxs:
                p, parent = prompts[i], None
                for b in range(len(p) // bs):
                    parent = nv.hash_token_block(p[b * bs:(b + 1) * bs], parent)
                    _, cached = mgr.allocate_with_prefix(parent)
                    copies += 0 if cached else 1
        return copies

    contiguous = [list(range(*sh.sequence_range(len(prompts), world, r))) for r in range(world)]
    assert prefix_block_copies(parts) == 6 * 2            # every family's two blocks exactly once
    assert prefix_block_copies(contiguous) > 6 * 2
    with pytest.raises(ValueError):
        sh.shard_by_prefix(1, 0)
