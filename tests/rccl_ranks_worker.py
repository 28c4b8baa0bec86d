"""One rank of tests/test_gpu_rccl_two_ranks.py (started by torch.distributed.run, one process per
GPU; not collected by pytest).  Both sharding modes of the paged-attention path through the HIP
kernels and RCCL: the assembled multi-rank result must equal the unsharded launch on this rank's
own GPU bit for bit."""
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
    rank = int(os.environ["RANK"])
    local_rank = int(os.environ["LOCAL_RANK"])
    world = int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist.init_process_group("nccl", device_id=dev)
    import bench
    import nano_vllm_rs_amd as nv
    sh = nv.sharding
    L = nv.capi.lib()
    assert L.nvllm_set_device(local_rank) == 0
    Hq, Hkv, D, bs = 32, 8, 128, 16
    scale = 1.0 / math.sqrt(D)
    N, ctx = 24, 1040
    try:
        seen = torch.ones(1, dtype=torch.int64, device=dev)
        dist.all_reduce(seen)
        assert int(seen.item()) == world
        for T in (1, 4):
            torch.manual_seed(1000 + T)          # the same caches and queries on every rank
            case = bench.build_decode_case(nv, dev, 70 + T, N, ctx, T, Hq, Hkv, D, bs, 1, 0, None)
            case["context_lens"] = torch.tensor([ctx - 7 * (i % 5) for i in range(N)], dtype=torch.int32, device=dev)
            kc, vc = case["caches"][0]
            # every rank must hold the same data, or "equal to the unsharded launch" means nothing
            chk = torch.stack([kc.float().sum(), vc.float().sum(), case["q"].float().sum()])
            lo, hi = chk.clone(), chk.clone()
            dist.all_reduce(lo, op=dist.ReduceOp.MIN)
            dist.all_reduce(hi, op=dist.ReduceOp.MAX)
            assert torch.equal(lo, hi), "ranks built different inputs"
            for splits in (1, 3):            # the split count is chosen from num_seqs x Hkv: pinned
                assert L.nvllm_debug_set_policy(b"NVLLM_PAGED_SPLITS", splits) == 0
                full = nv.paged_attention_batched(case["q"], kc, vc, case["block_tables"], case["context_lens"], scale)
                # kv-head-sharded: this rank's heads of every block, all sequences, one all-gather
                out_h = nv.paged_attention_batched(sh.shard_query_heads(case["q"], Hkv, world, rank),
                                                   sh.shard_cache_heads(kc, world, rank),
                                                   sh.shard_cache_heads(vc, world, rank),
                                                   case["block_tables"], case["context_lens"], scale)
                got_h = sh.all_gather_heads(out_h, D)
                # sequence-sharded: this rank's sequences, whole blocks, no collective in the path;
                # gather_sequences only replicates the result for the check
                b, e = sh.sequence_range(N, world, rank)
                out_s = nv.paged_attention_batched(case["q"][b:e].contiguous(), kc, vc,
                                                   case["block_tables"][b:e].contiguous(),
                                                   case["context_lens"][b:e].contiguous(), scale)
                got_s = sh.gather_sequences(out_s, N)
                torch.cuda.synchronize()
                assert got_h.shape == full.shape and torch.equal(got_h, full), ("heads", T, splits)
                assert got_s.shape == full.shape and torch.equal(got_s, full), ("sequences", T, splits)
            L.nvllm_debug_reset_policy()
        print("RCCL_RANK_OK rank=%d world=%d" % (rank, world), flush=True)
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
