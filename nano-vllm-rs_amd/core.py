"""Host structures of the hot path, mirroring the reference's `core` module.

Thin Python handles over the C++ implementations in csrc/nvllm_host.cpp
(through the C ABI).  Names, argument meaning and error behaviour follow
  src/core/block.rs, src/core/block_manager.rs, src/core/kv_cache.rs.
"""
import ctypes

import numpy as np

from . import capi

DEFAULT_BLOCK_SIZE = 16  # block.rs:14


class NvllmError(Exception):
    """crate::error::Error (src/error.rs:10-60)."""

    def __init__(self, code, message=None):
        self.code = code
        super().__init__(message if message is not None else capi.last_error())


class ConfigError(NvllmError):
    """Error::Config"""


class OutOfBlocksError(NvllmError):
    """Error::OutOfBlocks"""


class BlockIndexOutOfBoundsError(NvllmError):
    """Error::BlockIndexOutOfBounds"""


class TensorError(NvllmError):
    """Error::Tensor"""


_ERRORS = {
    capi.NVLLM_ERR_CONFIG: ConfigError,
    capi.NVLLM_ERR_OUT_OF_BLOCKS: OutOfBlocksError,
    capi.NVLLM_ERR_BLOCK_INDEX_OOB: BlockIndexOutOfBoundsError,
    capi.NVLLM_ERR_TENSOR: TensorError,
}


def check(rc):
    if rc != capi.NVLLM_OK:
        raise _ERRORS.get(rc, NvllmError)(rc)


def compute_num_blocks(seq_len, block_size):
    """block.rs:311"""
    return int(capi.lib().nvllm_compute_num_blocks(seq_len, block_size))


def hash_token_block(token_ids, parent_hash=None):
    """block.rs:43-57"""
    t = np.ascontiguousarray(token_ids, dtype=np.uint32)
    return int(capi.lib().nvllm_hash_token_block(
        t.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)), t.size,
        0 if parent_hash is None else 1, 0 if parent_hash is None else parent_hash))


class BlockTable:
    """block.rs:183-289"""

    def __init__(self, block_size=DEFAULT_BLOCK_SIZE):
        self._h = capi.lib().nvllm_block_table_new(block_size)
        if not self._h:
            raise ConfigError(capi.NVLLM_ERR_CONFIG)

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            capi.lib().nvllm_block_table_free(h)

    @property
    def handle(self):
        return self._h

    def block_size(self):
        return int(capi.lib().nvllm_block_table_block_size(self._h))

    def get_block_id(self, logical_block_idx):
        out = ctypes.c_int64()
        check(capi.lib().nvllm_block_table_get_block_id(self._h, logical_block_idx, ctypes.byref(out)))
        return out.value

    def append_block(self, block_id):
        capi.lib().nvllm_block_table_append_block(self._h, block_id)

    def num_blocks(self):
        return int(capi.lib().nvllm_block_table_num_blocks(self._h))

    def is_empty(self):
        return bool(capi.lib().nvllm_block_table_is_empty(self._h))

    def get_physical_block_ids(self):
        n = self.num_blocks()
        out = np.empty(max(n, 1), dtype=np.int64)
        capi.lib().nvllm_block_table_get_physical_block_ids(
            self._h, out.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)), n)
        return out[:n].tolist()

    def get_slot_mapping(self, seq_len):
        out = np.empty(max(seq_len, 1), dtype=np.int64)
        n = capi.lib().nvllm_block_table_get_slot_mapping(
            self._h, seq_len, out.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)), out.size)
        return out[:n].tolist()

    def clear(self):
        capi.lib().nvllm_block_table_clear(self._h)


class BlockManager:
    """block_manager.rs:40-383"""

    def __init__(self, num_blocks, block_size=DEFAULT_BLOCK_SIZE):
        self._h = capi.lib().nvllm_block_manager_new(num_blocks, block_size)
        if not self._h:
            raise ConfigError(capi.NVLLM_ERR_CONFIG)

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            capi.lib().nvllm_block_manager_free_handle(h)

    def block_size(self):
        return int(capi.lib().nvllm_block_manager_block_size(self._h))

    def num_blocks(self):
        return int(capi.lib().nvllm_block_manager_num_blocks(self._h))

    def num_free_blocks(self):
        return int(capi.lib().nvllm_block_manager_num_free_blocks(self._h))

    def num_used_blocks(self):
        return int(capi.lib().nvllm_block_manager_num_used_blocks(self._h))

    def has_free_blocks(self):
        return bool(capi.lib().nvllm_block_manager_has_free_blocks(self._h))

    def can_allocate(self, n):
        return bool(capi.lib().nvllm_block_manager_can_allocate(self._h, n))

    def allocate(self):
        out = ctypes.c_int64()
        check(capi.lib().nvllm_block_manager_allocate(self._h, ctypes.byref(out)))
        return out.value

    def allocate_many(self, n):
        out = np.empty(max(n, 1), dtype=np.int64)
        check(capi.lib().nvllm_block_manager_allocate_many(
            self._h, n, out.ctypes.data_as(ctypes.POINTER(ctypes.c_int64))))
        return out[:n].tolist()

    def free(self, block_id):
        return bool(capi.lib().nvllm_block_manager_free(self._h, block_id))

    def free_many(self, block_ids):
        ids = np.ascontiguousarray(block_ids, dtype=np.int64)
        return int(capi.lib().nvllm_block_manager_free_many(
            self._h, ids.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)), ids.size))

    def ref_count(self, block_id):
        """get_block(id).map(|b| b.ref_count()) -> None if the block is not in use."""
        r = int(capi.lib().nvllm_block_manager_ref_count(self._h, block_id))
        return None if r < 0 else r

    def prefix_hash_of(self, block_id):
        out = ctypes.c_uint64()
        if capi.lib().nvllm_block_manager_prefix_hash(self._h, block_id, ctypes.byref(out)):
            return out.value
        return None

    def increment_ref(self, block_id):
        r = int(capi.lib().nvllm_block_manager_increment_ref(self._h, block_id))
        return None if r < 0 else r

    def decrement_ref(self, block_id):
        r = int(capi.lib().nvllm_block_manager_decrement_ref(self._h, block_id))
        return None if r < 0 else r

    def get_cached_block(self, prefix_hash):
        r = int(capi.lib().nvllm_block_manager_get_cached_block(self._h, prefix_hash))
        return None if r < 0 else r

    def cache_block(self, block_id, prefix_hash):
        capi.lib().nvllm_block_manager_cache_block(self._h, block_id, prefix_hash)

    def allocate_with_prefix(self, prefix_hash):
        bid, cached = ctypes.c_int64(), ctypes.c_int()
        check(capi.lib().nvllm_block_manager_allocate_with_prefix(
            self._h, prefix_hash, ctypes.byref(bid), ctypes.byref(cached)))
        return bid.value, bool(cached.value)

    def is_prefix_cached(self, prefix_hash):
        return bool(capi.lib().nvllm_block_manager_is_prefix_cached(self._h, prefix_hash))

    def num_cached_prefixes(self):
        return int(capi.lib().nvllm_block_manager_num_cached_prefixes(self._h))

    def clear_prefix_cache(self):
        capi.lib().nvllm_block_manager_clear_prefix_cache(self._h)

    def reset(self):
        capi.lib().nvllm_block_manager_reset(self._h)


class KVCacheConfig:
    """kv_cache.rs:21-71 (dtype: a torch dtype; the reference defaults to F32, :51)."""

    def __init__(self, num_blocks, block_size, num_kv_heads, head_dim, num_layers, dtype=None):
        import torch
        self.num_blocks = num_blocks
        self.block_size = block_size
        self.num_kv_heads = num_kv_heads
        self.head_dim = head_dim
        self.num_layers = num_layers
        self.dtype = torch.float32 if dtype is None else dtype

    def with_dtype(self, dtype):
        self.dtype = dtype
        return self

    def _dtype_code(self):
        import torch
        codes = {torch.bfloat16: 0, torch.float32: 1}
        if self.dtype not in codes:
            raise ConfigError(capi.NVLLM_ERR_UNSUPPORTED, "dtype must be bfloat16 or float32")
        return codes[self.dtype]

    def layer_cache_size_bytes(self):
        """kv_cache.rs:62-65 (nvllm_kv_cache_layer_bytes)."""
        return int(capi.lib().nvllm_kv_cache_layer_bytes(self.num_blocks, self.block_size,
                                                         self.num_kv_heads, self.head_dim,
                                                         self._dtype_code()))

    def total_cache_size_bytes(self):
        """kv_cache.rs:68-70 (nvllm_kv_cache_total_bytes)."""
        return int(capi.lib().nvllm_kv_cache_total_bytes(self.num_blocks, self.block_size,
                                                         self.num_kv_heads, self.head_dim,
                                                         self.num_layers, self._dtype_code()))


class LayerKVCache:
    """kv_cache.rs:75-206: zero-initialised K and V [num_blocks, block_size, Hkv, D],
    resident in HBM (torch only provides the allocation)."""

    def __init__(self, config, device):
        import torch
        shape = (config.num_blocks, config.block_size, config.num_kv_heads, config.head_dim)
        self.config = config
        self._key_cache = torch.zeros(shape, dtype=config.dtype, device=device)
        self._value_cache = torch.zeros(shape, dtype=config.dtype, device=device)

    def key_cache(self):
        return self._key_cache

    def value_cache(self):
        return self._value_cache

    def _gather(self, cache, block_ids):
        import torch
        if len(block_ids) == 0:  # kv_cache.rs:124-126
            raise ConfigError(capi.NVLLM_ERR_CONFIG, "block_ids cannot be empty")
        c = self.config
        ids = torch.tensor([int(b) for b in block_ids], dtype=torch.int32, device=cache.device)
        out = torch.empty((len(block_ids), c.block_size, c.num_kv_heads, c.head_dim), dtype=c.dtype,
                          device=cache.device)
        st = torch.cuda.current_stream(cache.device).cuda_stream
        check(capi.lib().nvllm_gather_cache_blocks(cache.data_ptr(), ids.data_ptr(), len(block_ids),
                                                   out.data_ptr(), c.num_blocks, c.block_size,
                                                   c.num_kv_heads, c.head_dim, c._dtype_code(), st))
        return out

    def gather_keys(self, block_ids):
        """kv_cache.rs:123-132 -> [len(block_ids), block_size, Hkv, D]."""
        return self._gather(self._key_cache, block_ids)

    def gather_values(self, block_ids):
        """kv_cache.rs:135-144."""
        return self._gather(self._value_cache, block_ids)


class KVCache:
    """kv_cache.rs:210-279"""

    def __init__(self, config, device):
        self._config = config
        self._device = device
        self.layers = [LayerKVCache(config, device) for _ in range(config.num_layers)]

    def layer(self, idx):
        return self.layers[idx] if 0 <= idx < len(self.layers) else None

    def num_layers(self):
        return len(self.layers)

    def config(self):
        return self._config

    def device(self):
        return self._device

    def gather_keys_all_layers(self, block_ids):
        """kv_cache.rs:251-256"""
        return [l.gather_keys(block_ids) for l in self.layers]

    def gather_values_all_layers(self, block_ids):
        """kv_cache.rs:259-264"""
        return [l.gather_values(block_ids) for l in self.layers]


class BatchMetadata:
    """Flat per-step arrays for the batched kernels (C++ nvllm::core::BatchMetadata):
    what the reference's scheduler hands over one sequence at a time
    (src/scheduler/batch.rs:227-375 -> paged.rs:117-192) flattened into block_tables,
    context_lens, slot_mapping (block.rs:269-283 arithmetic) and query_start."""

    def __init__(self, max_blocks_per_seq):
        self._h = capi.lib().nvllm_batch_new(max_blocks_per_seq)
        if not self._h:
            raise ConfigError(capi.NVLLM_ERR_CONFIG)

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            capi.lib().nvllm_batch_free(h)

    def clear(self):
        capi.lib().nvllm_batch_clear(self._h)

    def add_sequence(self, table, context_len, num_new_tokens):
        check(capi.lib().nvllm_batch_add_sequence(self._h, table.handle, context_len, num_new_tokens))

    def num_seqs(self):
        return int(capi.lib().nvllm_batch_num_seqs(self._h))

    def num_tokens(self):
        return int(capi.lib().nvllm_batch_num_tokens(self._h))

    def max_blocks_per_seq(self):
        return int(capi.lib().nvllm_batch_max_blocks_per_seq(self._h))

    def _view(self, fn, ctype, n, dtype):
        if n == 0:
            return np.zeros(0, dtype=dtype)
        ptr = ctypes.cast(fn(self._h), ctypes.POINTER(ctype))
        return np.ctypeslib.as_array(ptr, shape=(n,)).copy()

    def block_tables(self):
        n = self.num_seqs()
        return self._view(capi.lib().nvllm_batch_block_tables, ctypes.c_int32,
                          n * self.max_blocks_per_seq(), np.int32).reshape(n, self.max_blocks_per_seq())

    def context_lens(self):
        return self._view(capi.lib().nvllm_batch_context_lens, ctypes.c_int32, self.num_seqs(), np.int32)

    def slot_mapping(self):
        return self._view(capi.lib().nvllm_batch_slot_mapping, ctypes.c_int64, self.num_tokens(), np.int64)

    def query_start(self):
        return self._view(capi.lib().nvllm_batch_query_start, ctypes.c_int32, self.num_seqs() + 1, np.int32)

    def positions(self):
        return self._view(capi.lib().nvllm_batch_positions, ctypes.c_int32, self.num_tokens(), np.int32)

    def upload(self, device, stream=None):
        """Device copies (block_tables, context_lens, slot_mapping, query_start, positions) as
        torch tensors."""
        import torch
        n, t, mb = self.num_seqs(), self.num_tokens(), self.max_blocks_per_seq()
        bt = torch.empty((n, mb), dtype=torch.int32, device=device)
        cl = torch.empty((n,), dtype=torch.int32, device=device)
        sm = torch.empty((t,), dtype=torch.int64, device=device)
        qs = torch.empty((n + 1,), dtype=torch.int32, device=device)
        ps = torch.empty((t,), dtype=torch.int32, device=device)
        st = stream if stream is not None else torch.cuda.current_stream(device).cuda_stream
        check(capi.lib().nvllm_batch_upload(self._h, bt.data_ptr(), cl.data_ptr(), sm.data_ptr(),
                                            qs.data_ptr(), ps.data_ptr(), st))
        return bt, cl, sm, qs, ps

