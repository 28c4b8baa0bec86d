"""CPU oracle for the nano-vllm-rs attention hot path.

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  The product package never imports this.
"""
from .oracle import *  # noqa: F401,F403
from .block_manager_oracle import BlockManagerOracle, OutOfBlocks  # noqa: F401
from .scheduler_oracle import SchedulerOracle, SequenceOracle, InvalidStateTransition  # noqa: F401
