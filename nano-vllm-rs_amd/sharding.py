"""Multi-GPU partitioning of the paged-attention path (no counterpart in the
reference, which is single-device).

One process per GPU (torch.distributed; backend "nccl" is RCCL over xGMI on ROCm, "gloo" in the
CPU tests).  Two modes:

  * sequence-sharded (primary): rank r owns a contiguous range of sequences together with their
    block pool, block tables and cache slab.  Sequences are independent units of
    `paged_attention` (reference src/attention/paged.rs:117-192 touches one table and its
    blocks), so the attention path needs NO collective; `gather_sequences` exists only for
    callers that want the outputs replicated.
  * kv-head-sharded: rank r holds KV heads [r*Hkv/n, (r+1)*Hkv/n) of every block (cache
    [num_blocks, bs, Hkv/n, D]) and computes the matching Hq/n query heads of ALL sequences;
    `all_gather_heads` then assembles [num_seqs, T, Hq*D] with ONE all-gather per layer.  The
    per-rank message is num_seqs*T*(Hq/n)*D*2 bytes (256 KiB at config 3, n = 8): on the 7-link
    xGMI mesh that is latency-bound, so it is a single collective, never a ring of small ones.

Prefix-shared sequences (BASELINE config 4) should live on ONE rank, or every rank that holds
one of them caches a copy of the shared blocks: `shard_by_prefix` / `partition_by_prefix` place a
sequence by the prefix hash of its first block, the value `BlockManager::allocate_with_prefix`
(reference src/core/block_manager.rs:337-347) keys its prefix cache on.
"""
import torch
import torch.distributed as dist

_MASK64 = (1 << 64) - 1


def _mix64(x):
    """splitmix64 finaliser: spreads a 64-bit hash over its low bits before the modulo."""
    x &= _MASK64
    x ^= x >> 30
    x = (x * 0xBF58476D1CE4E5B9) & _MASK64
    x ^= x >> 27
    x = (x * 0x94D049BB133111EB) & _MASK64
    return x ^ (x >> 31)


def shard_by_prefix(first_block_hash, world, seq_id=0):
    """Rank that owns a sequence, from the prefix hash of its FIRST full block
    (`hash_token_block(tokens[:block_size], None)`, block.rs:43-57): every sequence that starts
    with the same block lands on the same rank, so the blocks they share through
    `allocate_with_prefix` (block_manager.rs:337-347) exist on one GPU only.  A sequence shorter
    than one block has no hash (None) and nothing to share: it goes by `seq_id`.  Stateless (an
    admission-time rule); `partition_by_prefix` balances a known batch."""
    if world <= 0:
        raise ValueError("world must be positive")
    if first_block_hash is None:
        return int(seq_id) % world
    return _mix64(int(first_block_hash)) % world


def partition_by_prefix(first_block_hashes, world):
    """Balanced placement of a known batch: sequences are grouped by first-block hash (None =
    a group of its own), the groups are dealt largest first to the least loaded rank (ties: lowest
    rank).  Returns `world` lists of sequence indices, each in ascending order.  Co-location is
    kept (a group is never split); the load differs by at most the largest group."""
    if world <= 0:
        raise ValueError("world must be positive")
    groups, order = {}, []
    for i, h in enumerate(first_block_hashes):
        key = ("seq", i) if h is None else ("hash", int(h))
        if key not in groups:
            groups[key] = []
            order.append(key)
        groups[key].append(i)
    order.sort(key=lambda k: (-len(groups[k]), groups[k][0]))
    ranks, load = [[] for _ in range(world)], [0] * world
    for key in order:
        r = min(range(world), key=lambda j: (load[j], j))
        ranks[r].extend(groups[key])
        load[r] += len(groups[key])
    return [sorted(r) for r in ranks]


def sequence_range(num_seqs, world, rank):
    """Contiguous, balanced [begin, end) of the sequences rank `rank` owns."""
    base, rem = divmod(num_seqs, world)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def kv_head_range(num_kv_heads, world, rank):
    """KV heads [begin, end) of rank `rank`; world must divide num_kv_heads."""
    if num_kv_heads % world != 0:
        raise ValueError("kv-head sharding needs world (%d) to divide num_kv_heads (%d)"
                         % (world, num_kv_heads))
    per = num_kv_heads // world
    return rank * per, (rank + 1) * per


def q_head_range(num_heads, num_kv_heads, world, rank):
    """Query heads that read this rank's KV heads (GQA: q head h -> kv head h // (Hq/Hkv))."""
    g = num_heads // num_kv_heads
    b, e = kv_head_range(num_kv_heads, world, rank)
    return b * g, e * g


def shard_query_heads(query, num_kv_heads, world, rank):
    """[N, T, Hq, D] -> this rank's [N, T, Hq/n, D] (contiguous copy)."""
    b, e = q_head_range(query.shape[2], num_kv_heads, world, rank)
    return query[:, :, b:e, :].contiguous()


def shard_cache_heads(cache, world, rank):
    """[num_blocks, bs, Hkv, D] -> this rank's [num_blocks, bs, Hkv/n, D] (contiguous copy)."""
    b, e = kv_head_range(cache.shape[2], world, rank)
    return cache[:, :, b:e, :].contiguous()


def assemble_heads(gathered, world):
    """Rank-major concatenation [world * N, T, (Hq/n)*D] (what the all-gather delivers) ->
    [N, T, Hq*D]: rank r's heads land at columns [r*(Hq/n)*D, (r+1)*(Hq/n)*D)."""
    WN, T, HD = gathered.shape
    N = WN // world
    return gathered.view(world, N, T, HD).permute(1, 2, 0, 3).reshape(N, T, world * HD)


def all_gather_heads(out_local, head_dim, group=None):
    """out_local [N, T, (Hq/n)*D] on every rank -> [N, T, Hq*D] on every rank.
    One all_gather_into_tensor (RCCL ncclAllGather) + a local permute of the small result."""
    world = dist.get_world_size(group)
    if world == 1:
        return out_local
    N, T, HD = out_local.shape
    # output is the rank-major concatenation along dim 0 (the form every backend accepts)
    gathered = torch.empty((world * N, T, HD), dtype=out_local.dtype, device=out_local.device)
    dist.all_gather_into_tensor(gathered, out_local.contiguous(), group=group)
    return assemble_heads(gathered, world)


def gather_sequences(out_local, num_seqs, group=None):
    """Optional: replicate sequence-sharded outputs.  out_local [n_local, T, Hq*D] ->
    [num_seqs, T, Hq*D] (ranks may own different counts)."""
    world = dist.get_world_size(group)
    if world == 1:
        return out_local
    rank = dist.get_rank(group)
    sizes = [sequence_range(num_seqs, world, r) for r in range(world)]
    maxn = max(e - b for b, e in sizes)
    pad = torch.zeros((maxn,) + tuple(out_local.shape[1:]), dtype=out_local.dtype,
                      device=out_local.device)
    b, e = sizes[rank]
    pad[: e - b] = out_local
    gathered = torch.empty((world * maxn,) + tuple(pad.shape[1:]), dtype=pad.dtype,
                           device=pad.device)
    dist.all_gather_into_tensor(gathered, pad, group=group)
    gathered = gathered.view((world, maxn) + tuple(pad.shape[1:]))
    return torch.cat([gathered[r, : sizes[r][1] - sizes[r][0]] for r in range(world)], dim=0)
