"""Oracle pins and host logic for the K5 row (per-head RMSNorm + RoPE).  CPU only.

Known answers the reference's own tests hold: RMSNorm of [1,2,3,4] (tests/model_norm_test.rs:47-72,
:75-95, both < 1e-5) and the RoPE table shapes (tests/model_rope_test.rs:11-17,125-131).  The
reference's RoPE tests are otherwise qualitative (rotation changes values, positions differ), so
RoPE NUMERICS ARE UNPINNED by the reference; the oracle restates rope.rs:42-152 and is
cross-checked here against an independent float64 numpy formula.
"""
import numpy as np
import pytest

import nano_vllm_rs_amd as nv
import oracle


def test_rmsnorm_known_answers():
    x = np.array([[[1.0, 2.0, 3.0, 4.0]]], np.float32)
    rms = np.sqrt((1 + 4 + 9 + 16) / 4.0 + 1e-6)
    out = oracle.rms_norm(x, np.ones(4, np.float32), 1e-6)           # model_norm_test.rs:47-72
    assert np.abs(out.ravel() - np.array([1, 2, 3, 4]) / rms).max() < 1e-5
    out = oracle.rms_norm(x, np.full(4, 2.0, np.float32), 1e-6)      # :75-95
    assert np.abs(out.ravel() - 2 * np.array([1, 2, 3, 4]) / rms).max() < 1e-5


def test_rope_cache_shapes_and_first_row():
    cos, sin = oracle.rope_cache(64, 1024, 10000.0)                  # model_rope_test.rs:11-17
    assert cos.shape == (1024, 64) and sin.shape == (1024, 64)
    assert (cos[0] == 1).all() and (sin[0] == 0).all()
    assert (cos[:, :32] == cos[:, 32:]).all() and (sin[:, :32] == sin[:, 32:]).all()
    with pytest.raises(oracle.OracleConfigError):                    # rope.rs:49 (panics there)
        oracle.rope_cache(5, 8, 10000.0)


def test_rope_matches_float64_formula():
    rng = np.random.default_rng(0)
    D, S, H, theta, start = 32, 7, 3, 1e6, 11
    cos, sin = oracle.rope_cache(D, 64, theta)
    x = rng.standard_normal((2, S, H, D)).astype(np.float32)
    got = oracle.rope_apply(x, cos, sin, start)
    half = D // 2
    inv = 1.0 / theta ** (2.0 * np.arange(half) / D)
    ang = np.arange(start, start + S)[:, None] * inv[None, :]
    c, s = np.cos(ang)[None, :, None, :], np.sin(ang)[None, :, None, :]
    x1, x2 = x[..., :half].astype(np.float64), x[..., half:].astype(np.float64)
    want = np.concatenate([x1 * c - x2 * s, x2 * c + x1 * s], -1)
    assert np.abs(got - want).max() < 2e-5      # f32 angle at positions < 64
    # the reference's qualitative checks (model_rope_test.rs:72-121)
    one = np.array([1.0, 2.0, 3.0, 4.0], np.float32).reshape(1, 1, 1, 4)
    c4, s4 = oracle.rope_cache(4, 1024, 10000.0)
    assert not np.array_equal(oracle.rope_apply(one, c4, s4, 5), one)
    ones = np.ones((1, 1, 1, 16), np.float32)
    c16, s16 = oracle.rope_cache(16, 100, 10000.0)
    assert np.abs(oracle.rope_apply(ones, c16, s16, 0) - oracle.rope_apply(ones, c16, s16, 1)).sum() > 1e-5
    with pytest.raises(oracle.OracleTensorError):                    # narrow past the table
        oracle.rope_apply(ones, c16, s16, 100)


@pytest.mark.parametrize("dim,n,theta", [(4, 1024, 10000.0), (64, 300, 10000.0), (128, 4096, 1e6)])
def test_host_rope_cache_is_bit_identical_to_oracle(dim, n, theta):
    import ctypes
    cos = np.empty((n, dim), np.float32)
    sin = np.empty((n, dim), np.float32)
    rc = nv.capi.lib().nvllm_rope_cache_f32(dim, n, theta, cos.ctypes.data, sin.ctypes.data)
    assert rc == 0
    ocos, osin = oracle.rope_cache(dim, n, theta)
    assert np.array_equal(cos, ocos) and np.array_equal(sin, osin)
    assert nv.capi.lib().nvllm_rope_cache_f32(5, 8, theta, cos.ctypes.data, sin.ctypes.data) == nv.capi.NVLLM_ERR_CONFIG


def test_inv_freq_division_is_f32_at_long_positions():
    """rope.rs:55 `1.0 / theta.powf(..) as f32`: the cast binds tighter than the division, so
    inv_freq = 1.0f32 / (pow as f32).  A f64 division rounded afterwards differs by one ulp in 25
    of the 64 frequencies at theta = 1e6, dim = 128 -- invisible at small positions, up to 4e-4 in
    cos/sin at position 32767.  Pin the oracle (and, through the bit-equality test above, the host
    table) on an independent numpy evaluation of the f32-division form at that position."""
    dim, theta, pos = 128, 1e6, 32767
    i = np.arange(dim // 2)
    powf = np.float32(theta ** (2.0 * i / dim))                 # f64 power, narrowed
    inv32 = np.float32(1.0) / powf                              # the reference's arithmetic
    inv64 = (1.0 / theta ** (2.0 * i / dim)).astype(np.float32)  # the misreading
    assert (inv32 != inv64).sum() >= 8                          # the two really differ
    cos, sin = oracle.rope_cache(dim, pos + 1, theta)
    ang = (np.float32(pos) * inv32).astype(np.float64)
    assert np.abs(cos[pos, :dim // 2] - np.cos(ang)).max() < 1e-6
    assert np.abs(sin[pos, :dim // 2] - np.sin(ang)).max() < 1e-6
    ang_wrong = (np.float32(pos) * inv64).astype(np.float64)
    assert np.abs(np.cos(ang_wrong) - np.cos(ang)).max() > 1e-4  # what the f64 division would cost
