"""Shared builders for the parity tests (seeded synthetic inputs)."""
import numpy as np

import oracle

MAX_ABS_TOL = 2e-2   # north_star: attention outputs within 2e-2 max-abs ...
MEAN_ABS_TOL = 1e-3  # ... and 1e-3 mean-abs of the CPU reference path (bf16)


def bf16_randn(rng, shape, sigma=1.0):
    """N(0, sigma) values already rounded to bf16, as float32."""
    return oracle.bf16_round(rng.standard_normal(shape, dtype=np.float32) * np.float32(sigma))


def assert_attention_close(got, want, max_tol=MAX_ABS_TOL, mean_tol=MEAN_ABS_TOL):
    got = np.asarray(got, dtype=np.float32)
    want = np.asarray(want, dtype=np.float32)
    assert got.shape == want.shape, (got.shape, want.shape)
    assert np.isfinite(got).all(), "non-finite values in kernel output"
    diff = np.abs(got - want)
    assert diff.max() <= max_tol, "max-abs %.3e > %.1e" % (diff.max(), max_tol)
    assert diff.mean() <= mean_tol, "mean-abs %.3e > %.1e" % (diff.mean(), mean_tol)


def make_paged_case(rng, ctx_lens, num_blocks, block_size, Hkv, D, shared_prefix_blocks=0,
                    poison=np.nan, slack_blocks=0):
    """Random non-contiguous block tables over a pool of `num_blocks` blocks.

    Returns (key_cache, value_cache, tables[list of id lists]).  Every slot that no
    sequence owns, and every slot past a sequence's context_len inside its last block,
    holds `poison` (NaN by default): a kernel that lets such a slot reach its output
    fails the finiteness check.  The first `shared_prefix_blocks` blocks are the same
    physical blocks in every table (prefix sharing).
    """
    perm = rng.permutation(num_blocks).tolist()
    kc = np.full((num_blocks, block_size, Hkv, D), poison, dtype=np.float32)
    vc = np.full((num_blocks, block_size, Hkv, D), poison, dtype=np.float32)
    shared = [perm.pop() for _ in range(shared_prefix_blocks)]
    for b in shared:
        kc[b] = bf16_randn(rng, (block_size, Hkv, D))
        vc[b] = bf16_randn(rng, (block_size, Hkv, D))
    tables = []
    for ctx in ctx_lens:
        nblk = -(-ctx // block_size)
        ids = list(shared[:min(nblk, len(shared))])
        while len(ids) < nblk:
            ids.append(perm.pop())
        for lb, b in enumerate(ids):
            if lb < len(shared):
                continue
            n_live = min(block_size, ctx - lb * block_size)
            kc[b, :n_live] = bf16_randn(rng, (n_live, Hkv, D))
            vc[b, :n_live] = bf16_randn(rng, (n_live, Hkv, D))
        for _ in range(slack_blocks):   # allocated but unused blocks after the context
            ids.append(perm.pop())
        tables.append(ids)
    return kc, vc, tables


def pad_tables(tables, max_blocks=None, fill=0):
    m = max(len(t) for t in tables) if max_blocks is None else max_blocks
    out = np.full((len(tables), m), fill, dtype=np.int32)
    for i, t in enumerate(tables):
        out[i, :len(t)] = t
    return out


def oracle_paged_batch(q, kc, vc, tables, ctx_lens, block_size, scale):
    """Loop the reference-shaped oracle over the batch (one call per sequence,
    exactly how the reference API is used).  NaN poison in dead slots of the
    caches is replaced by 0 for the oracle only where the reference would never
    look (it narrows to context_len) -- so no replacement is needed at all."""
    N, T, Hq, D = q.shape
    Hkv = kc.shape[2]
    outs = []
    for i in range(N):
        if ctx_lens[i] <= 0:
            outs.append(np.zeros((1, T, Hq * D), dtype=np.float32))
            continue
        nblk = -(-ctx_lens[i] // block_size)
        outs.append(oracle.paged_attention(q[i:i + 1], kc, vc, tables[i][:max(nblk, 1)], block_size,
                                           ctx_lens[i], Hq, Hkv, D, scale))
    return np.concatenate(outs, axis=0)


def hashed_bf16(shape, stream):
    """Deterministic ~N(0,1) values rounded to bf16 (as float32), made by a counter-based
    integer hash (splitmix64 of stream * 2^40 + index; the sum of the four 16-bit fields is
    Irwin-Hall, scaled to unit variance).  Pure uint64 arithmetic, so a fixture can commit the
    stream id instead of megabytes of inputs and any numpy reproduces the same bits."""
    n = int(np.prod(shape))
    with np.errstate(over="ignore"):
        z = np.arange(n, dtype=np.uint64) + (np.uint64(stream) << np.uint64(40))
        z = (z + np.uint64(0x9E3779B97F4A7C15))
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    s = np.zeros(n, dtype=np.float64)
    for k in range(4):
        s += ((z >> np.uint64(16 * k)) & np.uint64(0xFFFF)).astype(np.float64)
    x = (s / 65536.0 - 2.0) * np.sqrt(3.0)
    return oracle.bf16_round(x.astype(np.float32)).reshape(shape)
