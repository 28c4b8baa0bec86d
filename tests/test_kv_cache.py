"""Port of the reference's tests/kv_cache_test.rs against the product mirrors of
src/core/kv_cache.rs (KVCacheConfig sizes through the C ABI; LayerKVCache / KVCache on the GPU
with gather_keys / gather_values as a HIP kernel behind nvllm_gather_cache_blocks)."""
import numpy as np
import pytest
import torch

import nano_vllm_rs_amd as nv


def _config(dtype=None):
    return nv.KVCacheConfig(16, 4, 8, 64, 2, dtype)      # kv_cache_test.rs:6-14


def test_kv_cache_config():                               # kv_cache_test.rs:16-31
    c = _config()
    assert (c.num_blocks, c.block_size, c.num_kv_heads, c.head_dim, c.num_layers) == (16, 4, 8, 64, 2)
    assert c.dtype == torch.float32                       # kv_cache.rs:51 default F32
    assert c.layer_cache_size_bytes() == 131072           # 16 * 4 * 8 * 64 * 4
    assert c.total_cache_size_bytes() == 524288           # 2 layers * (K + V)
    assert _config().with_dtype(torch.bfloat16).layer_cache_size_bytes() == 65536
    L = nv.capi.lib()
    # BASELINE sizes: config 3 pool and config 5 pool, bf16, K or V of one layer
    assert L.nvllm_kv_cache_layer_bytes(40960, 16, 8, 128, 0) == 40960 * 32768
    assert L.nvllm_kv_cache_total_bytes(2 * 1024 * 1024, 16, 8, 128, 1, 0) == 128 * 2 ** 30
    assert L.nvllm_kv_cache_layer_bytes(-1, 16, 8, 128, 0) == -1
    assert L.nvllm_kv_cache_layer_bytes(1, 16, 8, 128, 7) == -1


def test_gather_rejects_an_empty_id_list_without_a_gpu():  # kv_cache.rs:124-126
    rc = nv.capi.lib().nvllm_gather_cache_blocks(None, None, 0, None, 16, 4, 8, 64, 1, None)
    assert rc == nv.capi.NVLLM_ERR_CONFIG
    assert "block_ids cannot be empty" in nv.capi.last_error()


@pytest.mark.gpu
def test_layer_kv_cache_creation(dev):                    # kv_cache_test.rs:33-44
    layer = nv.LayerKVCache(_config(), dev)
    assert tuple(layer.key_cache().shape) == (16, 4, 8, 64)
    assert tuple(layer.value_cache().shape) == (16, 4, 8, 64)
    assert float(layer.key_cache().abs().max()) == 0.0    # Tensor::zeros (kv_cache.rs:94-95)


@pytest.mark.gpu
def test_kv_cache_creation(dev):                          # kv_cache_test.rs:46-56
    cache = nv.KVCache(_config(), dev)
    assert cache.num_layers() == 2
    assert cache.layer(0) is not None and cache.layer(1) is not None and cache.layer(2) is None


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_gather_keys_values(dev, dtype):                  # kv_cache_test.rs:58-71, plus the contents
    layer = nv.LayerKVCache(_config(dtype), dev)
    g = torch.Generator(device="cpu").manual_seed(7)
    layer.key_cache().copy_(torch.randn((16, 4, 8, 64), generator=g).to(dtype))
    layer.value_cache().copy_(torch.randn((16, 4, 8, 64), generator=g).to(dtype))
    ids = [0, 2, 5]
    keys, values = layer.gather_keys(ids), layer.gather_values(ids)
    assert tuple(keys.shape) == (3, 4, 8, 64) and tuple(values.shape) == (3, 4, 8, 64)
    assert torch.equal(keys, layer.key_cache()[ids]) and torch.equal(values, layer.value_cache()[ids])
    with pytest.raises(nv.ConfigError):
        layer.gather_keys([])
    # an id outside the pool (the reference's index_select fails; no sync here): a zero block
    out = layer.gather_keys([1, 99, 3])
    assert torch.equal(out[0], layer.key_cache()[1]) and float(out[1].abs().max()) == 0.0


@pytest.mark.gpu
def test_gather_all_layers(dev):                          # kv_cache_test.rs:73-89
    cache = nv.KVCache(_config(), dev)
    all_keys = cache.gather_keys_all_layers([0, 1])
    all_values = cache.gather_values_all_layers([0, 1])
    assert len(all_keys) == 2 and len(all_values) == 2
    for keys in all_keys:
        assert tuple(keys.shape) == (2, 4, 8, 64)


@pytest.mark.gpu
def test_gather_at_baseline_block_shape(dev):
    """Qwen3 block shape (16 x 8 x 128 bf16 = 32 KiB), 2048 scattered ids = one config-5 sequence."""
    nb = 4096
    g = torch.Generator(device="cpu").manual_seed(8)
    layer = nv.LayerKVCache(nv.KVCacheConfig(nb, 16, 8, 128, 1, torch.bfloat16), dev)
    layer.key_cache().copy_(torch.randn((nb, 16, 8, 128), generator=g).to(torch.bfloat16))
    ids = torch.randperm(nb, generator=g)[:2048].tolist()
    assert torch.equal(layer.gather_keys(ids), layer.key_cache()[ids])
