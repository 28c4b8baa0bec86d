#!/usr/bin/env python3
"""bench.py -- throughput of the attention hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload decode|shared|long|verify|prefill|kvwrite|producer]
                    [--parallelism sequences|heads]

`python bench.py --gpus N` with N > 1 starts N fresh rank processes itself (through
torch.distributed.run on 127.0.0.1; the parent touches no GPU API); started under
torch.distributed.run by someone else (WORLD_SIZE set) it is one of the ranks.

Headline (default, BASELINE.json `metric`): paged-decode attention GB/s on
BASELINE config 3 -- 256 sequences x 2048-token contexts, 32 Q / 8 KV heads,
head_dim 128, block_size 16, bf16, physical blocks a seeded random permutation
of a 40,960-block pool.  A "step" is ONE launch of the paged-attention hot path
over the whole batch for one layer; steps rotate over several disjoint layer
caches so nothing is re-read out of L2 / Infinity Cache.  Inputs are resident
in HBM before the timed region.

For N > 1 (launched by torch.distributed.run, one rank per GPU) the sequences
are sharded: every rank owns its own 256 sequences, block pool and cache -- no
data-path collective ("weak" scaling); value = bytes all ranks processed / the
max-over-ranks time.  `--workload long` divides config 5's fixed 1024 sequences
over the ranks ("strong" scaling, still no collective).

`--parallelism heads` (N a divisor of the 8 KV heads) is north_star's second
mode: every rank holds KV heads [r*8/N, (r+1)*8/N) of EVERY block (cache
[num_blocks, 16, 8/N, 128]), computes the matching 32/N query heads of ALL
sequences and then ONE all_gather_into_tensor (RCCL over xGMI) inside the timed
step assembles [num_seqs, T, 32*128] on every rank ("strong" scaling: the total
work is fixed).  The line then carries the kernel time and the collective time
per step separately (HIP events) and `rccl_ranks_seen`.

One JSON line is printed by rank 0.  `roofline` is the dominant kernel's
algorithmic bytes (or FLOPs) per launch / its average launch duration measured
with HIP events on the launch stream; `cpu_baseline` is the CPU oracle
(op-for-op restatement of the reference's Rust path, f32, OpenMP) timed on a
bounded sample on this host.  The default run also measures the prefill
flash-attention kernel (BASELINE config 2) and the KV write and reports them
under "also".
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E peak (MI355X_MICROARCH.md, spec)
MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense bf16 MFMA peak (spec; not the 2:1-sparse figure)


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="decode",
                    choices=["decode", "shared", "long", "verify", "prefill", "kvwrite", "producer"])
    ap.add_argument("--parallelism", default="sequences", choices=["sequences", "heads"],
                    help="N > 1: shard sequences (no collective) or KV-head groups (+ one all-gather per step)")
    ap.add_argument("--tokens", type=int, default=8, help="new tokens per sequence (--workload verify)")
    ap.add_argument("--num-seqs", type=int, default=0)
    ap.add_argument("--ctx", type=int, default=0)
    ap.add_argument("--layers", type=int, default=0, help="disjoint layer caches rotated over")
    ap.add_argument("--batch", type=int, default=1, help="prefill batch")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-also", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--launcher-check", action="store_true",
                    help="CPU-only rehearsal of the N-rank launch path (gloo, no GPU work): rank 0 "
                         "prints {launcher_check, world, ranks_seen}")
    ap.add_argument("--launcher-check-fail-rank", type=int, default=-1, help=argparse.SUPPRESS)
    return ap.parse_args(argv)


def free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launcher_command(args, argv, port):
    """What `python bench.py --gpus N` (N > 1, not yet under a launcher) starts: N fresh
    processes, one rank per GPU, through torch.distributed.run -- the same command line the
    driver uses."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
            "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def launch_children(args, argv):
    """Parent of an N > 1 run started as plain `python bench.py --gpus N`.  Touches no GPU API
    (a process that has initialised HIP must not be the one that spawns the ranks); the ranks are
    fresh child processes, rank 0's JSON line passes through on stdout, and the exit code is
    non-zero if any rank failed."""
    import subprocess
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # RCCL needs dmabuf IPC on this pool
    env.setdefault("OMP_NUM_THREADS", "4")
    proc = subprocess.run(launcher_command(args, argv, free_port()), env=env)
    return proc.returncode


def launcher_check(args):
    """One rank of --launcher-check: rendezvous over gloo on the CPU, confirm the world size,
    count the ranks with one all-reduce.  No GPU API is touched."""
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("WORLD_SIZE=%d but --gpus %d" % (world, args.gpus))
    if world > 1:
        dist.init_process_group("gloo")
    if rank == args.launcher_check_fail_rank:
        raise SystemExit(3)
    seen = torch.ones(1, dtype=torch.int64)
    extra = {}
    if args.parallelism == "heads":
        # the head-sharded step's collective through the product code (sharding.all_gather_heads),
        # on a tensor that says which rank and head every column came from
        import nano_vllm_rs_amd as nv
        Hq, Hkv, D, N, T = 32, 8, 128, 3, 2
        if Hkv % world != 0:
            raise SystemExit("--parallelism heads needs --gpus to divide the %d KV heads" % Hkv)
        b, e = nv.sharding.q_head_range(Hq, Hkv, world, rank)
        local = torch.arange(b, e, dtype=torch.float32).repeat_interleave(D).expand(N, T, (e - b) * D).contiguous()
        full = nv.sharding.all_gather_heads(local, D) if world > 1 else local
        want = torch.arange(Hq, dtype=torch.float32).repeat_interleave(D).expand(N, T, Hq * D)
        extra = dict(parallelism="heads", gathered_shape=list(full.shape), heads_in_order=bool(torch.equal(full, want)))
    if world > 1:
        dist.all_reduce(seen)
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(dict(launcher_check=True, world=world, ranks_seen=int(seen.item()),
                              workload=args.workload, **extra)))
    return 0


def dist_setup(n):
    """Rank set-up of a worker.  Returns (rank, local_rank, world, rccl_ranks_seen): the last is
    the sum of a one per rank over an RCCL all-reduce (None at N = 1: no collective exists)."""
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != n:
        raise SystemExit("WORLD_SIZE=%d but --gpus %d" % (world, n))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    seen = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        t = torch.ones(1, dtype=torch.int64, device=torch.device("cuda", local_rank))
        dist.all_reduce(t)
        seen = int(t.item())
    return rank, local_rank, world, seen


def barrier(world):
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
    torch.cuda.synchronize()


def max_over_ranks(x, world, dev):
    if world == 1:
        return x
    import torch.distributed as dist
    t = torch.tensor([x], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


SETTLE_MS = 30.0   # untimed launches before the W warm-ups, see timed_loop


def timed_loop(fn, steps, warmup, world, dev):
    """W warm-ups, then exactly K steps between barrier+synchronize pairs.
    Returns (wall seconds max over ranks, mean per-launch ms from HIP events).
    Before the W warm-ups the step runs untimed for about SETTLE_MS (at most 64 launches): every
    case is built on the host first, the GPU idles meanwhile, and the first launches after that run
    at ramping clocks (5-10 % slow for ~10-30 launches, profiles/r03 kernel trace) -- with a small W
    (the driver's --steps 20 runs) the timed steps would measure the ramp, not the kernel."""
    fn(0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn(1)
    torch.cuda.synchronize()
    one = max(max_over_ranks((time.perf_counter() - t0) * 1e3, world, dev), 1e-3)   # (the same count on every rank)
    for i in range(min(64, int(SETTLE_MS / one))):
        fn(2 + i)
    for i in range(warmup):
        fn(i)
    barrier(world)
    ev0 = torch.cuda.Event(enable_timing=True)
    ev1 = torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for i in range(steps):
        fn(i)
    ev1.record()
    torch.cuda.synchronize()
    barrier(world)
    wall = time.perf_counter() - t0
    ev_ms = ev0.elapsed_time(ev1) / steps
    return max_over_ranks(wall, world, dev), max_over_ranks(ev_ms, world, dev)


def per_launch_ms(fn, steps):
    """Per-launch durations from one HIP-event pair around every launch (a pass of its own, after
    the timed region: an event record between launches costs host time the timed loop must not
    pay).  Returns (median, min) in ms."""
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    for i, (a, b) in enumerate(evs):
        a.record()
        fn(i)
        b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    return ts[len(ts) // 2], ts[0]


# ------------------------------------------------------------------------------------------
def build_decode_case(nv, dev, seed, num_seqs, ctx, T, Hq, Hkv, D, bs, layers, shared_prefix_tokens=0,
                      pool_blocks=None):
    """Block tables through the product BlockManager/BlockTable (allocate_with_prefix for the
    shared prefix), physical ids a seeded permutation of the pool, caches filled with N(0,1)."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    blocks_per_seq = -(-ctx // bs)
    shared_blocks = shared_prefix_tokens // bs
    need = shared_blocks + num_seqs * (blocks_per_seq - shared_blocks)
    if pool_blocks is None:
        pool_blocks = max(need + need // 4, need + 16)
    perm = torch.randperm(pool_blocks, generator=g)
    # the allocator hands out ascending ids (FIFO free list); the permutation maps them to
    # scattered physical blocks so tables are non-contiguous as BASELINE.md asks
    mgr = nv.BlockManager(pool_blocks, bs)
    tables = np.zeros((num_seqs, blocks_per_seq), dtype=np.int32)
    prefix_hashes = []
    parent = None
    for b in range(shared_blocks):
        parent = nv.hash_token_block(np.arange(b * bs, (b + 1) * bs, dtype=np.uint32), parent)
        prefix_hashes.append(parent)
    for s in range(num_seqs):
        ids = []
        for b in range(shared_blocks):
            bid, _ = mgr.allocate_with_prefix(prefix_hashes[b])
            ids.append(bid)
        ids.extend(mgr.allocate_many(blocks_per_seq - shared_blocks))
        tables[s] = perm[torch.tensor(ids)].numpy()
    if shared_blocks:
        assert mgr.ref_count(0) == num_seqs
    unique_blocks = need
    block_tables = torch.from_numpy(tables).to(dev)
    context_lens = torch.full((num_seqs,), ctx, dtype=torch.int32, device=dev)
    caches = []
    for _ in range(layers):
        kc = torch.randn((pool_blocks, bs, Hkv, D), device=dev, dtype=torch.bfloat16)
        vc = torch.randn((pool_blocks, bs, Hkv, D), device=dev, dtype=torch.bfloat16)
        caches.append((kc, vc))
    q = torch.randn((num_seqs, T, Hq, D), device=dev, dtype=torch.bfloat16)
    return dict(q=q, caches=caches, block_tables=block_tables, context_lens=context_lens,
                tables_np=tables, unique_blocks=unique_blocks, pool_blocks=pool_blocks)


def cpu_baseline_decode(case, ctx, T, Hq, Hkv, D, bs, scale, budget_s, bytes_per_seq):
    """Time the CPU oracle (reference's per-sequence paged_attention, f32, OpenMP) on a bounded
    sample of the same sequences; report it in the metric's unit (bf16-equivalent GB/s)."""
    import oracle
    kc, vc = case["caches"][0]
    tables = case["tables_np"]
    q = case["q"].float().cpu().numpy()
    n_total = tables.shape[0]
    done, t_used, outs = 0, 0.0, []
    # one untimed call to warm caches / thread pool
    sub = np.unique(tables[0])
    remap = {int(b): i for i, b in enumerate(sub)}
    while done < n_total and (done < 2 or t_used < budget_s):  # bounded by the budget
        ids = tables[done]
        # gather this sequence's blocks to the host as f32 (untimed: the reference holds f32 caches)
        sub = np.unique(ids)
        remap = {int(b): i for i, b in enumerate(sub)}
        idx = torch.from_numpy(sub.astype(np.int64)).to(kc.device)
        kch = kc[idx].float().cpu().numpy()
        vch = vc[idx].float().cpu().numpy()
        local_ids = [remap[int(b)] for b in ids]
        t0 = time.perf_counter()
        o = oracle.paged_attention(q[done:done + 1], kch, vch, local_ids, bs, ctx, Hq, Hkv, D, scale)
        t_used += time.perf_counter() - t0
        outs.append(o)
        done += 1
    gbs = done * bytes_per_seq / t_used / 1e9
    return dict(value=gbs, unit="GB/s", cores=oracle.num_threads(), kind="port",
                sample="%d of %d sequences (ctx %d) through oracle.paged_attention, one call per "
                       "sequence as the reference API works; %.1f s of CPU; bytes counted as the "
                       "same bf16 KV bytes the GPU metric uses" % (done, n_total, ctx, t_used),
                seconds=t_used), np.concatenate(outs, axis=0)


def run_decode(args, nv, dev, rank, world, workload):
    Hq, Hkv, D, bs = 32, 8, 128, 16
    scale = 1.0 / math.sqrt(D)
    heads_mode = args.parallelism == "heads"
    if heads_mode and Hkv % world != 0:
        raise SystemExit("--parallelism heads needs --gpus to divide the %d KV heads" % Hkv)
    if workload == "decode":      # BASELINE config 3
        num_seqs, ctx, T, shared = args.num_seqs or 256, args.ctx or 2048, 1, 0
        layers = args.layers or 4
        pool = 40960 if (num_seqs == 256 and ctx == 2048) else None
        name = "config3: paged decode, %d seqs x %d ctx, 32Q/8KV, hd128, block 16, T=1" % (num_seqs, ctx)
    elif workload == "shared":    # BASELINE config 4
        num_seqs, ctx, T, shared = args.num_seqs or 256, args.ctx or 2048, 4, (args.ctx or 2048) // 2
        layers = args.layers or 4
        pool = None
        name = ("config4: shared-prefix decode + spec-verify, %d seqs, T=4, ctx %d = %d shared + %d private"
                % (num_seqs, ctx, shared, ctx - shared))
    elif workload == "verify":    # config 3's batch with T new tokens per sequence
        num_seqs, ctx, T, shared = args.num_seqs or 256, args.ctx or 2048, args.tokens, 0
        layers = args.layers or 2
        pool = None
        name = "verify: %d seqs x %d ctx, T=%d new tokens, 32Q/8KV, hd128, block 16" % (num_seqs, ctx, T)
    else:                          # BASELINE config 5 (per-GPU shard of 1024/n sequences)
        total = args.num_seqs or 1024
        per = total if heads_mode else max(total // max(args.gpus, 1), 1)
        num_seqs, ctx, T, shared = per, args.ctx or 32768, 1, 0
        layers = args.layers or 1
        pool = None
        name = ("config5: long-context paged decode, %d seqs/GPU x %d ctx (%d total, %s-sharded %d-way)"
                % (num_seqs, ctx, total, "kv-head" if heads_mode else "sequence", args.gpus))
    if heads_mode:
        # every rank: ALL sequences (same tables on every rank: the seed does not depend on the
        # rank), its Hkv/n KV heads of every block, its Hq/n query heads
        Hq_l, Hkv_l, seed = Hq // world, Hkv // world, 3
        name += "; kv-head-sharded x%d: cache [nb,16,%d,128] per rank + one all-gather" % (world, Hkv_l)
    else:
        Hq_l, Hkv_l, seed = Hq, Hkv, 3 + rank
    case = build_decode_case(nv, dev, seed, num_seqs, ctx, T, Hq_l, Hkv_l, D, bs, layers, shared, pool)
    Hq_full, Hkv_full = Hq, Hkv
    Hq, Hkv = Hq_l, Hkv_l          # what this rank's launch sees
    token_bytes = 2 * Hkv * D * 2
    logical_bytes = num_seqs * ctx * token_bytes
    unique_bytes = case["unique_blocks"] * bs * token_bytes
    aux_bytes = 2 * num_seqs * T * Hq * D * 2 + case["block_tables"].numel() * 4
    need = nv.capi.lib().nvllm_paged_attn_workspace_bytes(num_seqs, T, Hq, Hkv, D, bs,
                                                          case["block_tables"].shape[1])
    ws = torch.empty(int(need), dtype=torch.uint8, device=dev) if need else None
    outs = [None]

    def kernel_step(i):
        kc, vc = case["caches"][i % layers]
        outs[0] = nv.paged_attention_batched(case["q"], kc, vc, case["block_tables"],
                                             case["context_lens"], scale, workspace=ws)

    def step(i):
        kernel_step(i)
        if heads_mode and world > 1:
            outs[0] = nv.sharding.all_gather_heads(outs[0], D)      # [num_seqs, T, Hq_full * D]

    wall, ev_ms = timed_loop(step, args.steps, args.warmup, world, dev)
    split = None
    if heads_mode:
        # the same step, kernel and collective bracketed separately (a pass of its own: three event
        # records per step are host work the timed loop does not carry)
        n = min(args.steps, 50)
        evs = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(n)]
        barrier(world)
        for i, (e0, e1, e2) in enumerate(evs):
            e0.record()
            kernel_step(i)
            e1.record()
            if world > 1:
                outs[0] = nv.sharding.all_gather_heads(outs[0], D)
            e2.record()
        torch.cuda.synchronize()
        k_ms = sorted(e0.elapsed_time(e1) for e0, e1, _ in evs)[n // 2]
        c_ms = sorted(e1.elapsed_time(e2) for _, e1, e2 in evs)[n // 2]
        split = dict(kernel_ms_median=max_over_ranks(k_ms, world, dev),
                     collective_ms_median=max_over_ranks(c_ms, world, dev), steps=n,
                     collective="all_gather_into_tensor, %d bytes per rank" % (num_seqs * T * Hq * D * 2),
                     output_shape=list(outs[0].shape))
        assert outs[0].shape[-1] == Hq_full * D
    algo_bytes = unique_bytes  # roofline numerator: bytes that must come from HBM
    res = dict(
        metric="paged_decode_attn_GBps", unit="GB/s",
        value=world * algo_bytes * args.steps / wall / 1e9,
        ms_per_step=wall / args.steps * 1e3,
        config=dict(workload=name, num_seqs_per_gpu=num_seqs, context_len=ctx, new_tokens=T,
                    num_q_heads=Hq, num_kv_heads=Hkv, head_dim=D, block_size=bs,
                    layer_caches_rotated=layers, pool_blocks=case["pool_blocks"],
                    parallelism=("kv-head-sharded x%d (one RCCL all-gather per step)" % world if heads_mode
                                 else "sequence-sharded x%d (no collective)" % world),
                    algorithmic_bytes_per_step_per_gpu=algo_bytes,
                    logical_bytes_per_step_per_gpu=logical_bytes, qo_table_bytes=aux_bytes),
        roofline=dict(bound="hbm", achieved=algo_bytes / (ev_ms * 1e-3) / 1e9, peak=HBM_PEAK_GBS,
                      unit="GB/s", frac=algo_bytes / (ev_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                      traffic=None, kernel="paged_attn_kernel",
                      launch_ms=ev_ms, effective_logical_GBps=logical_bytes / (ev_ms * 1e-3) / 1e9),
    )
    # per-GPU work fixed as N grows -> weak; a fixed total divided over the ranks -> strong
    res["scaling"] = "strong" if (heads_mode or workload == "long") else "weak"
    if split is not None:
        res["head_sharded_step"] = split
        # in head mode the roofline figure is the kernel's own time, the step adds the collective
        res["roofline"]["launch_ms"] = split["kernel_ms_median"]
        res["roofline"]["achieved"] = algo_bytes / (split["kernel_ms_median"] * 1e-3) / 1e9
        res["roofline"]["frac"] = res["roofline"]["achieved"] / HBM_PEAK_GBS
        res["roofline"]["step_ms_with_collective"] = ev_ms
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not heads_mode:   # the CPU leg runs at N=1 only
        budget = args.cpu_seconds if workload != "long" else args.cpu_seconds
        cb, want = cpu_baseline_decode(case, ctx, T, Hq, Hkv, D, bs, scale, budget,
                                       ctx * token_bytes)
        # the timed oracle outputs double as a parity check of the benchmarked launch
        step(0)
        torch.cuda.synchronize()
        got = outs[0][: want.shape[0]].float().cpu().numpy()
        err = np.abs(got - want)
        cb["parity_max_abs"] = float(err.max())
        cb["parity_mean_abs"] = float(err.mean())
        res["cpu_baseline"] = cb
    return res


def run_prefill(args, nv, dev, rank, world, steps, warmup, cpu=True):
    """BASELINE config 2: B=1, S=4096, 16 Q / 8 KV heads, hd128, causal, bf16."""
    B, S, Hq, Hkv, D = args.batch, args.ctx or 4096, 16, 8, 128
    g = torch.Generator(device="cpu").manual_seed(2 + rank)
    q = torch.randn((B, S, Hq, D), generator=g).to(dev, torch.bfloat16)
    k = torch.randn((B, S, Hkv, D), generator=g).to(dev, torch.bfloat16)
    v = torch.randn((B, S, Hkv, D), generator=g).to(dev, torch.bfloat16)
    cfg = nv.FlashAttentionConfig.new(D, True)
    outs = [None]

    def step(i):
        outs[0] = nv.flash_attention(q, k, v, cfg)

    wall, ev_ms = timed_loop(step, steps, warmup, world, dev)
    med_ms, min_ms = per_launch_ms(step, min(steps, 200))
    flops = 4.0 * B * Hq * S * S * D * 0.5   # causal: 68.72 GFLOP at B=1,S=4096
    res = dict(
        metric="prefill_flash_attn_TFLOPs", unit="TFLOP/s",
        value=world * flops * steps / wall / 1e12, ms_per_step=wall / steps * 1e3,
        config=dict(workload="config2: flash-attn prefill B=%d S=%d 16Q/8KV hd128 causal bf16" % (B, S),
                    algorithmic_flops_per_step_per_gpu=flops, parallelism="replicas x%d" % world),
        roofline=dict(bound="mfma", achieved=flops / (ev_ms * 1e-3) / 1e12, peak=MFMA_BF16_PEAK_TFLOPS,
                      unit="TFLOP/s", frac=flops / (ev_ms * 1e-3) / 1e12 / MFMA_BF16_PEAK_TFLOPS,
                      traffic=None, kernel="flash_attn_mfma_kernel", launch_ms=ev_ms,
                      launch_ms_median=med_ms, launch_ms_min=min_ms,
                      achieved_at_median=flops / (med_ms * 1e-3) / 1e12),
    )
    if B == 1 and S == 4096:
        res["roofline"]["traffic"], res["roofline"]["traffic_source"] = pmc_traffic("config2_prefill")
    if rank == 0 and world == 1 and cpu and not args.no_cpu_baseline:
        import oracle
        # sample: every head of batch element 0 (the whole workload at B=1), repeated to ~budget
        hs = Hq
        qs = q[:1].float().cpu().numpy()
        ks = k[:1].float().cpu().numpy()
        vs = v[:1].float().cpu().numpy()
        want = oracle.flash_attention_cpu(qs, ks, vs, True)   # untimed warm-up (thread pool, pages)
        reps, dt = 0, 0.0
        while reps < 2 or (dt < args.cpu_seconds and reps < 50):
            t0 = time.perf_counter()
            want = oracle.flash_attention_cpu(qs, ks, vs, True)
            dt += time.perf_counter() - t0
            reps += 1
        dt /= reps
        got = outs[0][:1].float().cpu().numpy()
        err = np.abs(got - want)
        res["cpu_baseline"] = dict(
            value=4.0 * hs * S * S * D * 0.5 / dt / 1e12, unit="TFLOP/s", cores=oracle.num_threads(),
            kind="port", seconds=dt,
            sample="all %d query heads of batch element 0 through oracle.flash_attention_cpu (64x64 "
                   "tiles, f32), mean of %d repetitions; causal FLOPs counted like the GPU metric"
                   % (hs, reps),
            parity_max_abs=float(err.max()), parity_mean_abs=float(err.mean()))
    return res


def run_kvwrite(args, nv, dev, rank, world, steps, warmup, num_tokens, order="random"):
    """K4 over `num_tokens` new rows.  order: "random" = every row to a random slot of the pool (the
    worst case, and the headline); "ascending_blocks" = whole blocks in ascending id order with gaps,
    what a prefill gets from a fresh free list."""
    Hkv, D, bs, nb = 8, 128, 16, 40960
    g = torch.Generator(device="cpu").manual_seed(6 + rank)
    nbuf = 4                                   # rotate so the caches do not serve the re-reads
    bufs = [(torch.randn((num_tokens, Hkv, D), generator=g).to(dev, torch.bfloat16),
             torch.randn((num_tokens, Hkv, D), generator=g).to(dev, torch.bfloat16)) for _ in range(nbuf)]
    kc = torch.zeros((nb, bs, Hkv, D), dtype=torch.bfloat16, device=dev)
    vc = torch.zeros_like(kc)
    if order == "random":
        slots = torch.randperm(nb * bs, generator=g)[:num_tokens].to(dev)
    else:
        blk = torch.sort(torch.randperm(nb, generator=g)[:(num_tokens + bs - 1) // bs]).values
        slots = (blk[:, None] * bs + torch.arange(bs)[None, :]).reshape(-1)[:num_tokens].contiguous().to(dev)

    def step(i):
        key, val = bufs[i % nbuf]
        nv.reshape_and_cache(key, val, kc, vc, slots)

    wall, ev_ms = timed_loop(step, steps, warmup, world, dev)
    bytes_moved = num_tokens * (2 * Hkv * D * 2) * 2 + num_tokens * 8   # read + written + slots
    return dict(metric="kv_write_GBps", unit="GB/s", value=world * bytes_moved * steps / wall / 1e9,
                ms_per_step=wall / steps * 1e3,
                config=dict(workload="reshape_and_cache, %d tokens, 8 KV heads, hd128, bf16, %s slots"
                                     % (num_tokens, order),
                            algorithmic_bytes_per_step_per_gpu=bytes_moved),
                roofline=dict(bound="hbm", achieved=bytes_moved / (ev_ms * 1e-3) / 1e9, peak=HBM_PEAK_GBS,
                              unit="GB/s", frac=bytes_moved / (ev_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                              traffic=None, kernel="reshape_and_cache_kernel", launch_ms=ev_ms))


def run_producer(args, nv, dev, rank, world, steps, warmup, num_tokens):
    """K5: per-head RMSNorm + RoPE + KV write over one step's new tokens (Qwen3-4B head shape)."""
    Hq, Hkv, D, bs = 32, 8, 128, 16
    nb = max(2 * num_tokens // bs, 4096)
    g = torch.Generator(device="cpu").manual_seed(7 + rank)
    nbuf = 4                                   # rotate so the caches do not serve the re-reads
    bufs = []
    for _ in range(nbuf):
        bufs.append((torch.randn((num_tokens, Hq, D), generator=g).to(dev, torch.bfloat16),
                     torch.randn((num_tokens, Hkv, D), generator=g).to(dev, torch.bfloat16),
                     torch.randn((num_tokens, Hkv, D), generator=g).to(dev, torch.bfloat16),
                     torch.zeros((nb, bs, Hkv, D), dtype=torch.bfloat16, device=dev),
                     torch.zeros((nb, bs, Hkv, D), dtype=torch.bfloat16, device=dev)))
    w = torch.ones(D, dtype=torch.bfloat16, device=dev)
    rot = nv.RotaryEmbedding(D, 40960, 1e6, torch.bfloat16, dev)
    pos = torch.randint(0, 40960, (num_tokens,), generator=g).to(dev, torch.int32)
    slots = torch.randperm(nb * bs, generator=g)[:num_tokens].to(dev)

    def step(i):
        q, k, v, kc, vc = bufs[i % nbuf]
        nv.qk_norm_rope_cache(q, k, v, w, w, 1e-6, rot, pos, kc, vc, slots)

    wall, ev_ms = timed_loop(step, steps, warmup, world, dev)
    bytes_moved = num_tokens * ((2 * Hq + 4 * Hkv) * D * 2 + 2 * D * 2 + 12)
    return dict(metric="qk_norm_rope_cache_GBps", unit="GB/s", value=world * bytes_moved * steps / wall / 1e9,
                ms_per_step=wall / steps * 1e3,
                config=dict(workload="qk_norm_rope_cache, %d tokens, 32Q/8KV, hd128, bf16" % num_tokens,
                            algorithmic_bytes_per_step_per_gpu=bytes_moved),
                roofline=dict(bound="hbm", achieved=bytes_moved / (ev_ms * 1e-3) / 1e9, peak=HBM_PEAK_GBS,
                              unit="GB/s", frac=bytes_moved / (ev_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                              traffic=None, kernel="qk_norm_rope_cache_d128_kernel", launch_ms=ev_ms))


PMC_TRAFFIC_FILES = ("profiles/r03_pmc_traffic.json", "profiles/r02_pmc_traffic.json")


def pmc_traffic(key):
    """(HBM bytes per launch of the dominant kernel, where the figure comes from).  The bytes are
    NOT measured in this run -- counters cannot be read inside the process: they come from the
    committed rocprofv3 PMC passes over this same command (FETCH_SIZE and WRITE_SIZE in separate
    runs, FETCH doubled per the gfx950 correction), newest file first.  (None, None) if the workload
    was not profiled."""
    for rel in PMC_TRAFFIC_FILES:
        try:
            with open(os.path.join(ROOT, rel)) as f:
                return json.load(f)[key]["traffic_bytes"], rel + " (committed rocprofv3 --pmc passes, not this run)"
        except Exception:
            continue
    return None, None


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: become the launcher; no GPU API before this point
        return launch_children(args, argv)
    if args.launcher_check:
        return launcher_check(args)
    rank, local_rank, world, rccl_seen = dist_setup(args.gpus)
    dev = torch.device("cuda", local_rank)
    import nano_vllm_rs_amd as nv
    nv.capi.lib()
    if args.workload in ("decode", "shared", "long", "verify"):
        res = run_decode(args, nv, dev, rank, world, args.workload)
        if args.workload == "decode" and not args.num_seqs and not args.ctx:
            res["roofline"]["traffic"], res["roofline"]["traffic_source"] = pmc_traffic("config3_decode")
    elif args.workload == "prefill":
        res = run_prefill(args, nv, dev, rank, world, args.steps, args.warmup)
    elif args.workload == "producer":
        res = run_producer(args, nv, dev, rank, world, args.steps, args.warmup, args.num_seqs or 32768)
    else:
        res = run_kvwrite(args, nv, dev, rank, world, args.steps, args.warmup, args.num_seqs or 4096)
    also = {}
    if args.workload == "decode" and not args.no_also and world == 1:
        torch.cuda.empty_cache()
        p = run_prefill(args, nv, dev, rank, world, max(args.steps, 20), max(args.warmup // 2, 5))
        also["prefill"] = {k: p[k] for k in ("metric", "value", "unit", "ms_per_step", "roofline", "config")
                           if k in p}
        if "cpu_baseline" in p:
            also["prefill"]["cpu_baseline"] = p["cpu_baseline"]
        saved = (args.steps, args.warmup, args.no_cpu_baseline, args.tokens)
        # (the case is built on the host first and the GPU idles meanwhile: with fewer than ~10 warm-up
        # launches the timed ones still ran at ramping clocks and read 5-10 % slow)
        args.steps, args.warmup, args.no_cpu_baseline = max(args.steps // 2, 5), max(args.warmup, 2), True
        for t in (8, 32):   # multi-token verify / short chunk on config 3's batch: decode kernel, prefill kernel
            args.tokens = t
            torch.cuda.empty_cache()
            w = run_decode(args, nv, dev, rank, world, "verify")
            also["verify_T%d" % t] = dict(value=w["value"], unit="GB/s", ms_per_step=w["ms_per_step"],
                                          frac=w["roofline"]["frac"])
        args.steps, args.warmup, args.no_cpu_baseline, args.tokens = saved
        torch.cuda.empty_cache()
        for n in (256, 4096, 32768):
            w = run_kvwrite(args, nv, dev, rank, world, args.steps, args.warmup, n)
            also["kv_write_%d" % n] = dict(value=w["value"], unit="GB/s", ms_per_step=w["ms_per_step"],
                                           frac=w["roofline"]["frac"])
        w = run_kvwrite(args, nv, dev, rank, world, args.steps, args.warmup, 32768, order="ascending_blocks")
        also["kv_write_32768_ascending_blocks"] = dict(value=w["value"], unit="GB/s",
                                                       ms_per_step=w["ms_per_step"], frac=w["roofline"]["frac"])
        for n in (256, 32768):
            w = run_producer(args, nv, dev, rank, world, args.steps, args.warmup, n)
            also["qk_norm_rope_cache_%d" % n] = dict(value=w["value"], unit="GB/s",
                                                     ms_per_step=w["ms_per_step"],
                                                     frac=w["roofline"]["frac"])
    props = torch.cuda.get_device_properties(dev)
    line = dict(
        metric=res["metric"], value=res["value"], unit=res["unit"], n_gpus=world, steps=args.steps,
        warmup=args.warmup, ms_per_step=res["ms_per_step"], higher_is_better=True,
        scaling=res.get("scaling", "weak"),
        vs_baseline=None, dtype="bf16", data="synthetic", config=res["config"], roofline=res["roofline"],
        untimed_settle_ms=SETTLE_MS,
        cpu_baseline=res.get("cpu_baseline"),
        device=dict(name=props.name, cus=props.multi_processor_count,
                    hbm_gib=round(props.total_memory / 2**30, 1),
                    peaks=dict(hbm_GBps=HBM_PEAK_GBS, mfma_bf16_TFLOPs=MFMA_BF16_PEAK_TFLOPS)),
    )
    if also:
        line["also"] = also
    if "head_sharded_step" in res:
        line["head_sharded_step"] = res["head_sharded_step"]
    if world > 1:
        line["rccl_ranks_seen"] = rccl_seen
    if rank == 0:
        print(json.dumps(line), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
