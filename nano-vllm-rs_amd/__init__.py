"""MI355X-native attention hot path of nano-vllm-rs.

Python here is only the harness-side mirror of the C ABI in
include/nvllm_attn.h: ctypes bindings (`capi`), the host structures
(`core`: BlockTable, BlockManager, hash_token_block, ...) and the
reference-shaped operators (`attention`: flash_attention, prefill_attention,
paged_attention, write_kv_to_cache) over torch device tensors.  All compute
runs in libnvllm_attn.so (hand-written gfx950 HIP + C++ host code); nothing in
this package falls back to PyTorch or to the CPU oracle.
"""
from . import capi  # noqa: F401
from . import sharding  # noqa: F401
from . import scheduler  # noqa: F401
from .scheduler import Scheduler, SchedulerConfig, SchedulerOutputs  # noqa: F401
from .core import (  # noqa: F401
    DEFAULT_BLOCK_SIZE, BlockTable, BlockManager, BatchMetadata, KVCacheConfig, LayerKVCache, KVCache,
    compute_num_blocks, hash_token_block, NvllmError, ConfigError, OutOfBlocksError,
    BlockIndexOutOfBoundsError, TensorError,
)
from .attention import (  # noqa: F401
    FlashAttentionConfig, flash_attention, prefill_attention, paged_attention,
    paged_attention_batched, paged_attention_path, validate_tables, write_kv_to_cache, reshape_and_cache, RotaryEmbedding,
    qk_norm_rope_cache,
)
