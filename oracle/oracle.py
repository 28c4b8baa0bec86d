"""ctypes/numpy front-end of oracle/nvllm_oracle.c.

TEST INFRASTRUCTURE ONLY (see the header of nvllm_oracle.c): loaded by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg, never by the product.

Every function takes/returns float32 numpy arrays in the reference's layouts:
  q [B,Sq,Hq,D], k/v [B,Skv,Hkv,D], caches [num_blocks, block_size, Hkv, D].
"""
import ctypes
import os
import subprocess

import numpy as np

__all__ = [
    "build", "lib", "OracleConfigError", "OracleTensorError",
    "compute_num_blocks", "get_slot_mapping", "hash_token_block", "siphash",
    "flash_attention_cpu", "prefill_attention", "paged_attention",
    "write_kv_to_cache", "num_threads", "naive_sdpa", "bf16_round",
    "rms_norm", "rope_cache", "rope_apply", "qk_norm_rope",
]

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libnvllm_oracle.so")
_lib = None


class OracleConfigError(Exception):
    """Error::Config (src/error.rs:50-51)."""


class OracleTensorError(Exception):
    """Error::Tensor (src/error.rs:46-47): a candle shape/narrow/index failure."""


def build(force=False):
    """Compile the C restatement (gcc only; no reference sources involved)."""
    src = os.path.join(_HERE, "nvllm_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libnvllm_oracle.so"],
                              stdout=subprocess.DEVNULL)
    return _SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = ctypes.CDLL(_SO)
        i64, f32p, i64p = ctypes.c_int64, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_int64)
        L.oracle_compute_num_blocks.restype = i64
        L.oracle_compute_num_blocks.argtypes = [i64, i64]
        L.oracle_get_slot_mapping.restype = i64
        L.oracle_get_slot_mapping.argtypes = [i64p, i64, i64, i64, i64p]
        L.oracle_siphash.restype = ctypes.c_uint64
        L.oracle_siphash.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_uint64, ctypes.c_uint64,
                                     ctypes.c_char_p, i64]
        L.oracle_hash_token_block.restype = ctypes.c_uint64
        L.oracle_hash_token_block.argtypes = [ctypes.POINTER(ctypes.c_uint32), i64, ctypes.c_int,
                                              ctypes.c_uint64]
        L.oracle_flash_attention_cpu.restype = ctypes.c_int
        L.oracle_flash_attention_cpu.argtypes = [f32p, f32p, f32p, f32p] + [i64] * 8 + [
            ctypes.c_int, ctypes.c_float]
        L.oracle_prefill_attention.restype = ctypes.c_int
        L.oracle_prefill_attention.argtypes = [f32p, f32p, f32p, f32p] + [i64] * 5 + [
            ctypes.c_double, ctypes.c_int]
        L.oracle_paged_attention.restype = ctypes.c_int
        L.oracle_paged_attention.argtypes = [f32p, f32p, f32p, i64p] + [i64] * 9 + [
            ctypes.c_double, f32p]
        L.oracle_write_kv_to_cache.restype = ctypes.c_int
        L.oracle_write_kv_to_cache.argtypes = [f32p, f32p, f32p, f32p, i64p] + [i64] * 6
        L.oracle_rms_norm.restype = ctypes.c_int
        L.oracle_rms_norm.argtypes = [f32p, f32p, ctypes.c_double, i64, i64, f32p]
        L.oracle_rope_cache.restype = ctypes.c_int
        L.oracle_rope_cache.argtypes = [i64, i64, ctypes.c_double, f32p, f32p]
        L.oracle_rope_apply.restype = ctypes.c_int
        L.oracle_rope_apply.argtypes = [f32p, f32p, f32p] + [i64] * 6 + [f32p]
        L.oracle_num_threads.restype = ctypes.c_int
        _lib = L
    return _lib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def _pi(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_int64))


def _check(rc):
    if rc == -1:
        raise OracleConfigError()
    if rc != 0:
        raise OracleTensorError(rc)


def num_threads():
    return int(lib().oracle_num_threads())


def bf16_round(a):
    """Round float32 -> bfloat16 (round-to-nearest-even) -> float32, in numpy."""
    a = np.ascontiguousarray(a, dtype=np.float32)
    u = a.view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return (r & 0xFFFFFFFF).astype(np.uint32).view(np.float32).reshape(a.shape)


def compute_num_blocks(seq_len, block_size):
    return int(lib().oracle_compute_num_blocks(seq_len, block_size))


def get_slot_mapping(block_ids, block_size, seq_len):
    ids = np.ascontiguousarray(block_ids, dtype=np.int64)
    out = np.empty(max(int(seq_len), 1), dtype=np.int64)
    n = lib().oracle_get_slot_mapping(_pi(ids), ids.size, block_size, seq_len, _pi(out))
    return out[:n].copy()


def siphash(c_rounds, d_rounds, k0, k1, data):
    data = bytes(data)
    return int(lib().oracle_siphash(c_rounds, d_rounds, k0, k1, data, len(data)))


def hash_token_block(token_ids, parent_hash=None):
    t = np.ascontiguousarray(token_ids, dtype=np.uint32)
    return int(lib().oracle_hash_token_block(
        t.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)), t.size,
        0 if parent_hash is None else 1, 0 if parent_hash is None else parent_hash))


def flash_attention_cpu(q, k, v, causal=True, softmax_scale=None, block_size_q=64, block_size_kv=64):
    """flash.rs:137-275; FlashAttentionConfig::new defaults (64, 64, 1/sqrt(D)) (:39-46)."""
    q, k, v = _f32(q), _f32(k), _f32(v)
    B, Sq, Hq, D = q.shape
    _, Skv, Hkv, _ = k.shape
    if softmax_scale is None:
        softmax_scale = np.float32(1.0) / np.sqrt(np.float32(D))
    out = np.empty_like(q)
    _check(lib().oracle_flash_attention_cpu(_p(q), _p(k), _p(v), _p(out), B, Sq, Skv, Hq, Hkv, D,
                                            block_size_q, block_size_kv, int(bool(causal)),
                                            float(softmax_scale)))
    return out


def prefill_attention(q, k, v, num_heads, num_kv_heads, scale, causal):
    """paged.rs:55-95 -> [B, S, Hq*D]."""
    q, k, v = _f32(q), _f32(k), _f32(v)
    B, S, Hq, D = q.shape
    assert Hq == num_heads and k.shape[2] == num_kv_heads
    out = np.empty((B, S, Hq * D), dtype=np.float32)
    _check(lib().oracle_prefill_attention(_p(q), _p(k), _p(v), _p(out), B, S, Hq, num_kv_heads, D,
                                          float(scale), int(bool(causal))))
    return out


def paged_attention(query, key_cache, value_cache, block_ids, block_size, context_len, num_heads,
                    num_kv_heads, head_dim, scale):
    """paged.rs:117-192 -> [B, T, Hq*D] (B must be 1, as in the reference)."""
    query, key_cache, value_cache = _f32(query), _f32(key_cache), _f32(value_cache)
    B, T, Hq, D = query.shape
    assert Hq == num_heads and D == head_dim
    nb, bs, Hkv, _ = key_cache.shape
    assert bs == block_size and Hkv == num_kv_heads
    ids = np.ascontiguousarray(block_ids, dtype=np.int64)
    out = np.empty((B, T, Hq * D), dtype=np.float32)
    _check(lib().oracle_paged_attention(_p(query), _p(key_cache), _p(value_cache), _pi(ids),
                                        ids.size, nb, block_size, context_len, B, T, Hq, Hkv, D,
                                        float(scale), _p(out)))
    return out


def write_kv_to_cache(key, value, key_cache, value_cache, slot_mapping, block_size):
    """paged.rs:208-265; functional like the reference: returns NEW cache arrays."""
    key, value = _f32(key), _f32(value)
    kc, vc = _f32(key_cache).copy(), _f32(value_cache).copy()
    _, S, Hkv, D = key.shape
    nb, bs, _, _ = kc.shape
    assert bs == block_size
    slots = np.ascontiguousarray(slot_mapping, dtype=np.int64)
    _check(lib().oracle_write_kv_to_cache(_p(key), _p(value), _p(kc), _p(vc), _pi(slots),
                                          slots.size, S, nb, bs, Hkv, D))
    return kc, vc


def rms_norm(x, weight, eps):
    """norm.rs:83-100 over the last dimension."""
    x, weight = _f32(x), _f32(weight)
    D = x.shape[-1]
    out = np.empty_like(x)
    _check(lib().oracle_rms_norm(_p(x), _p(weight), float(eps), x.size // D, D, _p(out)))
    return out


def rope_cache(dim, max_seq_len, theta):
    """rope.rs:42-80 -> (cos, sin), each [max_seq_len, dim] f32."""
    cos = np.empty((max_seq_len, dim), dtype=np.float32)
    sin = np.empty((max_seq_len, dim), dtype=np.float32)
    _check(lib().oracle_rope_cache(dim, max_seq_len, float(theta), _p(cos), _p(sin)))
    return cos, sin


def rope_apply(x, cos, sin, start_pos):
    """rope.rs:107-152 on one tensor x [B,S,H,D] (apply() runs it on q and on k)."""
    x, cos, sin = _f32(x), _f32(cos), _f32(sin)
    B, S, H, D = x.shape
    out = np.empty_like(x)
    _check(lib().oracle_rope_apply(_p(x), _p(cos), _p(sin), cos.shape[0], start_pos, B, S, H, D,
                                   _p(out)))
    return out


def qk_norm_rope(q, k, q_weight, k_weight, eps, cos, sin, start_pos):
    """Steps 3-4 of Qwen3Attention::forward (src/model/attention.rs:229-234): per-head RMSNorm
    on Q and K, then RoPE from start_pos.  q [B,S,Hq,D], k [B,S,Hkv,D]."""
    qn, kn = rms_norm(q, q_weight, eps), rms_norm(k, k_weight, eps)
    return rope_apply(qn, cos, sin, start_pos), rope_apply(kn, cos, sin, start_pos)


def naive_sdpa(q, k, v, scale, causal):
    """The reference TEST's own comparison function (tests/flash_attention_test.rs:216-258),
    restated in float64 numpy; requires Hq == Hkv like the original."""
    q, k, v = (np.asarray(x, dtype=np.float64) for x in (q, k, v))
    qt, kt, vt = (x.transpose(0, 2, 1, 3) for x in (q, k, v))
    s = np.einsum("bhid,bhjd->bhij", qt, kt) * float(scale)
    if causal:
        Sq, Skv = s.shape[-2:]
        mask = np.triu(np.ones((Sq, Skv), dtype=bool), k=1)
        s = np.where(mask, -np.inf, s)
    s = s - s.max(axis=-1, keepdims=True)
    p = np.exp(s)
    p /= p.sum(axis=-1, keepdims=True)
    return np.einsum("bhij,bhjd->bhid", p, vt).transpose(0, 2, 1, 3)
