"""Reference-shaped attention operators over torch device tensors.

Mirrors nano_vllm::attention (src/attention/mod.rs:16-17):
  FlashAttentionConfig, flash_attention          src/attention/flash.rs:26-106
  prefill_attention, paged_attention,
  write_kv_to_cache                              src/attention/paged.rs:55-265
plus the batched forms the scheduler-side metadata builder feeds
(paged_attention_batched, reshape_and_cache).

torch is plumbing only (device memory + current stream); all compute is in
libnvllm_attn.so.  Tensors must live on a HIP device and be contiguous.
Deviation from the reference, on purpose: write_kv_to_cache updates the cache
IN PLACE and returns the same tensors (the reference returns new ones).
"""
import ctypes
import math

import numpy as np
import torch

from . import capi
from .core import BlockTable, ConfigError, LayerKVCache, check

_DT = {torch.bfloat16: 0, torch.float32: 1}


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr())


def _require_device(*tensors):
    for t in tensors:
        if not t.is_cuda:
            raise RuntimeError("nano_vllm_rs_amd operators need HIP device tensors "
                               "(no CPU fallback exists)")
        if not t.is_contiguous():
            raise ConfigError(capi.NVLLM_ERR_CONFIG, "tensor must be contiguous")


def _dtype_code(t):
    if t.dtype not in _DT:
        raise ConfigError(capi.NVLLM_ERR_UNSUPPORTED, "dtype must be bfloat16 or float32")
    return _DT[t.dtype]


class FlashAttentionConfig:
    """flash.rs:26-71.  block_size_q/kv are accepted for API parity; the GPU
    kernel's tiling is fixed and its result does not depend on them (the
    reference asserts block-size invariance, tests/flash_attention_test.rs:134-174)."""

    def __init__(self, head_dim=64, causal=True):
        self.block_size_q = 64
        self.block_size_kv = 64
        self.causal = causal
        self.softmax_scale = float(np.float32(1.0) / np.sqrt(np.float32(head_dim)))

    @classmethod
    def new(cls, head_dim, causal):
        return cls(head_dim, causal)

    def with_block_sizes(self, block_size_q, block_size_kv):
        self.block_size_q, self.block_size_kv = block_size_q, block_size_kv
        return self

    def with_scale(self, scale):
        self.softmax_scale = float(scale)
        return self


def flash_attention(query, key, value, config, return_lse=False):
    """flash.rs:85-90: q [B,Sq,Hq,D], k/v [B,Skv,Hkv,D] -> [B,Sq,Hq,D]."""
    _require_device(query, key, value)
    B, Sq, Hq, D = query.shape
    _, Skv, Hkv, _ = key.shape
    out = torch.empty_like(query)
    lse = torch.empty((B, Hq, Sq), dtype=torch.float32, device=query.device) if return_lse else None
    fn = (capi.lib().nvllm_flash_attn_fwd_bf16 if _dtype_code(query) == 0
          else capi.lib().nvllm_flash_attn_fwd_f32)
    check(fn(_ptr(query), _ptr(key), _ptr(value), _ptr(out), _ptr(lse) if return_lse else None,
             B, Sq, Skv, Hq, Hkv, D, config.softmax_scale, int(bool(config.causal)), _stream()))
    return (out, lse) if return_lse else out


def prefill_attention(query, key, value, num_heads, num_kv_heads, scale, causal):
    """paged.rs:55-63 -> [B, S, Hq*D]."""
    _require_device(query, key, value)
    B, S, Hq, D = query.shape
    if Hq != num_heads or key.shape[2] != num_kv_heads:
        raise ConfigError(capi.NVLLM_ERR_CONFIG, "head count mismatch")
    out = torch.empty((B, S, Hq * D), dtype=query.dtype, device=query.device)
    check(capi.lib().nvllm_prefill_attention(_ptr(query), _ptr(key), _ptr(value), _ptr(out), B, S,
                                             Hq, num_kv_heads, D, float(scale), int(bool(causal)),
                                             _dtype_code(query), _stream()))
    return out


def paged_attention(query, kv_cache, block_table, context_len, num_heads, num_kv_heads, head_dim,
                    scale):
    """paged.rs:117-126: ONE sequence, host BlockTable, the layer's cache -> [B,T,Hq*D]."""
    assert isinstance(kv_cache, LayerKVCache) and isinstance(block_table, BlockTable)
    kc, vc = kv_cache.key_cache(), kv_cache.value_cache()
    _require_device(query, kc, vc)
    B, T, Hq, D = query.shape
    if Hq != num_heads or D != head_dim or kc.shape[2] != num_kv_heads:
        raise ConfigError(capi.NVLLM_ERR_CONFIG, "shape mismatch")
    out = torch.empty((B, T, Hq * D), dtype=query.dtype, device=query.device)
    need = capi.lib().nvllm_paged_attention_seq_scratch_bytes(
        block_table.num_blocks(), T, Hq, num_kv_heads, D, block_table.block_size())
    scratch = torch.empty(max(int(need), 256), dtype=torch.uint8, device=query.device)
    check(capi.lib().nvllm_paged_attention_seq(
        _ptr(query), B, T, _ptr(kc), _ptr(vc), kc.shape[0], block_table.handle, context_len, Hq,
        num_kv_heads, D, float(scale), _dtype_code(query), _ptr(out), _ptr(scratch),
        scratch.numel(), _stream()))
    return out


PAGED_PATHS = ("decode", "k1_paged", "generic", "paged_mid")


def paged_attention_path(num_seqs, num_new_tokens, num_heads, num_kv_heads, head_dim, block_size,
                         max_blocks_per_seq, num_cache_blocks, dtype=torch.bfloat16):
    """nvllm_paged_attn_path: which kernel a batched launch of this shape runs (pure query)."""
    code = capi.lib().nvllm_paged_attn_path(num_seqs, num_new_tokens, num_heads, num_kv_heads, head_dim,
                                            block_size, max_blocks_per_seq, num_cache_blocks,
                                            0 if dtype == torch.bfloat16 else 1)
    check(min(code, 0))
    return PAGED_PATHS[code]


def validate_tables(block_tables, context_lens, block_size, num_cache_blocks):
    """nvllm_debug_validate_tables: (ids outside the pool among the blocks the context lengths use,
    context lengths outside their table row) -- what a launch over these tables would silently clamp."""
    _require_device(block_tables, context_lens)
    assert block_tables.dtype == torch.int32 and context_lens.dtype == torch.int32
    assert block_tables.is_contiguous() and context_lens.is_contiguous()
    counts = torch.empty(2, dtype=torch.int32, device=block_tables.device)
    check(capi.lib().nvllm_debug_validate_tables(block_tables.data_ptr(), context_lens.data_ptr(),
                                                 block_tables.shape[0], block_tables.shape[1], block_size,
                                                 num_cache_blocks, counts.data_ptr(), _stream()))
    bad_ids, bad_lens = counts.tolist()
    return bad_ids, bad_lens


def paged_attention_batched(query, key_cache, value_cache, block_tables, context_lens, scale,
                            workspace=None):
    """Batched form (nvllm_paged_attn_*): query [num_seqs,T,Hq,D], block_tables int32
    [num_seqs,max_blocks], context_lens int32 [num_seqs] -> [num_seqs,T,Hq*D]."""
    _require_device(query, key_cache, value_cache, block_tables, context_lens)
    N, T, Hq, D = query.shape
    nb, bs, Hkv, _ = key_cache.shape
    assert block_tables.dtype == torch.int32 and context_lens.dtype == torch.int32
    max_blocks = block_tables.shape[1]
    out = torch.empty((N, T, Hq * D), dtype=query.dtype, device=query.device)
    if workspace is None:
        need = capi.lib().nvllm_paged_attn_workspace_bytes(N, T, Hq, Hkv, D, bs, max_blocks)
        workspace = torch.empty(int(need), dtype=torch.uint8, device=query.device) if need else None
    fn = (capi.lib().nvllm_paged_attn_bf16 if _dtype_code(query) == 0
          else capi.lib().nvllm_paged_attn_f32)
    check(fn(_ptr(query), _ptr(key_cache), _ptr(value_cache), _ptr(block_tables),
             _ptr(context_lens), _ptr(out), _ptr(workspace) if workspace is not None else None,
             workspace.numel() if workspace is not None else 0, N, T, Hq, Hkv, D, bs, max_blocks,
             nb, float(scale), _stream()))
    return out


def reshape_and_cache(key, value, key_cache, value_cache, slot_mapping):
    """Batched KV write (nvllm_reshape_and_cache_*): key/value [num_tokens,Hkv,D],
    slot_mapping int64 [num_tokens] ON DEVICE; in place."""
    _require_device(key, value, key_cache, value_cache, slot_mapping)
    assert slot_mapping.dtype == torch.int64
    n, Hkv, D = key.shape
    nb, bs = key_cache.shape[0], key_cache.shape[1]
    fn = (capi.lib().nvllm_reshape_and_cache_bf16 if _dtype_code(key) == 0
          else capi.lib().nvllm_reshape_and_cache_f32)
    check(fn(_ptr(key), _ptr(value), _ptr(key_cache), _ptr(value_cache), _ptr(slot_mapping), n,
             Hkv, D, bs, nb, _stream()))


class RotaryEmbedding:
    """rope.rs:18-152 mirror: cos/sin tables [max_seq_len, dim] built on the host by
    nvllm_rope_cache_f32 (the reference's arithmetic) and held on `device` in `dtype`."""

    def __init__(self, dim, max_seq_len, theta, dtype, device):
        cos = np.empty((max_seq_len, dim), dtype=np.float32)
        sin = np.empty((max_seq_len, dim), dtype=np.float32)
        check(capi.lib().nvllm_rope_cache_f32(dim, max_seq_len, float(theta), cos.ctypes.data,
                                              sin.ctypes.data))
        self._dim = dim
        self._cos = torch.from_numpy(cos).to(device=device, dtype=dtype)
        self._sin = torch.from_numpy(sin).to(device=device, dtype=dtype)

    def dim(self):
        return self._dim

    def cos_cache(self):
        return self._cos

    def sin_cache(self):
        return self._sin


def qk_norm_rope_cache(query, key, value, q_weight, k_weight, eps, rotary, positions, key_cache,
                       value_cache, slot_mapping, k_out=None, q_out=None):
    """Steps 3-5 of Qwen3Attention::forward (attention.rs:229-254) fused: per-head RMSNorm of Q
    and K, RoPE at `positions`, K and V into their cache slots.  query [num_tokens,Hq,D],
    key/value [num_tokens,Hkv,D], positions int32 / slot_mapping int64 [num_tokens] on device.
    Returns q_out (query itself unless q_out is given)."""
    _require_device(query, key, value, q_weight, k_weight, positions, key_cache, value_cache,
                    slot_mapping)
    assert positions.dtype == torch.int32 and slot_mapping.dtype == torch.int64
    n, Hq, D = query.shape
    Hkv = key.shape[1]
    nb, bs = key_cache.shape[0], key_cache.shape[1]
    if q_out is None:
        q_out = query
    cos, sin = rotary.cos_cache(), rotary.sin_cache()
    assert cos.dtype == query.dtype and cos.shape[1] == D
    fn = (capi.lib().nvllm_qk_norm_rope_cache_bf16 if _dtype_code(query) == 0
          else capi.lib().nvllm_qk_norm_rope_cache_f32)
    check(fn(_ptr(query), _ptr(key), _ptr(value), _ptr(q_out),
             _ptr(k_out) if k_out is not None else None, _ptr(q_weight), _ptr(k_weight), float(eps),
             _ptr(cos), _ptr(sin), cos.shape[0], _ptr(positions), _ptr(key_cache),
             _ptr(value_cache), _ptr(slot_mapping), n, Hq, Hkv, D, bs, nb, _stream()))
    return q_out


def write_kv_to_cache(key, value, key_cache, value_cache, slot_mapping, block_size):
    """paged.rs:208-215: key/value [1,S,Hkv,D], HOST slot mapping (list of ints).
    Updates the caches in place and returns them."""
    _require_device(key, value, key_cache, value_cache)
    _, S, Hkv, D = key.shape
    slots = np.ascontiguousarray(slot_mapping, dtype=np.int64)
    scratch = torch.empty(max(S, 1) * 8, dtype=torch.uint8, device=key.device)
    check(capi.lib().nvllm_write_kv_to_cache(
        _ptr(key), _ptr(value), S, _ptr(key_cache), _ptr(value_cache), key_cache.shape[0],
        slots.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)), slots.size, block_size, Hkv, D,
        _dtype_code(key), _ptr(scratch), scratch.numel(), _stream()))
    return key_cache, value_cache
