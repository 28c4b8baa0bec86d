"""Both sharding modes through the HIP kernels AND RCCL (north_star (4)): two fresh rank
processes, one per GPU, each checks the assembled result against the unsharded launch on its own
GPU, bit for bit.  Needs two GPUs; on a one-GPU box the two-rank case is skipped and the same
worker runs as a single rank (world 1 over RCCL), so the script itself is exercised wherever the
GPU suite runs.  The ranks are child processes of torch.distributed.run started from here -- never
a re-exec of this process (which has touched the GPU)."""
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tests", "rccl_ranks_worker.py")


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run_ranks(world):
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), WORKER]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, (r.stdout + r.stderr)[-4000:]
    oks = [l for l in r.stdout.splitlines() if l.startswith("RCCL_RANK_OK")]
    assert len(oks) == world, r.stdout[-2000:]


def test_two_ranks_both_modes_bit_identical_to_unsharded(dev):
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (this box has %d)" % torch.cuda.device_count())
    _run_ranks(2)


def test_single_rank_rehearsal_of_the_worker(dev):
    _run_ranks(1)
