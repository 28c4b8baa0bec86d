"""The C-ABI library loads and exports every symbol include/nvllm_attn.h declares, and the
binding table in capi.py covers exactly that set.  No GPU, no compute calls."""
import ctypes
import os
import re

import nano_vllm_rs_amd as nv

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "nvllm_attn.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return set(re.findall(r"\b(nvllm_[a-z0-9_]+)\s*\(", text))


def test_every_declared_symbol_is_exported():
    names = _declared()
    assert len(names) >= 50
    L = ctypes.CDLL(nv.capi.LIB_PATH)
    for n in sorted(names):
        assert hasattr(L, n), "libnvllm_attn.so does not export %s" % n


def test_binding_table_matches_header():
    assert set(nv.capi.SIGNATURES) == _declared()


def test_library_is_the_in_tree_hip_build():
    assert nv.capi.LIB_PATH.startswith(os.path.join(ROOT, "nano-vllm-rs_amd"))
    nv.capi.lib()
    assert isinstance(nv.capi.last_error(), str)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "nano-vllm-rs_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hpp", ".hip", ".h")):
                src = open(os.path.join(dirpath, f), errors="replace").read()
                assert "import oracle" not in src and "nvllm_oracle" not in src, f


def test_host_argument_validation_without_gpu():
    """Entry points reject bad arguments before touching the device (safe on a CPU-only box)."""
    L = nv.capi.lib()
    assert L.nvllm_paged_attn_bf16(None, None, None, None, None, None, None, 0, 1, 1, 32, 8, 128,
                                   16, 4, 8, 0.1, None) == nv.capi.NVLLM_ERR_NULL
    assert L.nvllm_reshape_and_cache_bf16(None, None, None, None, None, 0, 8, 128, 16, 4, None) == 0
    assert L.nvllm_reshape_and_cache_bf16(None, None, None, None, None, 4, 8, 128, 16, 4, None) \
        == nv.capi.NVLLM_ERR_NULL
    # split-K workspace: needed for a small batch, not for config 3 (2048 waves fill the chip)
    assert L.nvllm_paged_attn_workspace_bytes(16, 1, 32, 8, 128, 16, 128) > 0
    assert L.nvllm_paged_attn_workspace_bytes(256, 1, 32, 8, 128, 16, 128) == 0
    assert L.nvllm_paged_attn_workspace_bytes(4096, 1, 32, 8, 128, 16, 128) == 0   # splits = 1
    assert L.nvllm_paged_attn_workspace_bytes(4, 1, 4, 2, 8, 4, 3) == 0            # generic path
    t = nv.BlockTable(4)
    import numpy as np
    slots = np.zeros(2, dtype=np.int64)
    rc = L.nvllm_write_kv_to_cache(None, None, 3, None, None, 2,
                                   slots.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)), 2, 4, 2, 4,
                                   1, None, 0, None)
    assert rc == nv.capi.NVLLM_ERR_CONFIG
    assert "slot_mapping length 2 doesn't match seq_len 3" in nv.capi.last_error()   # paged.rs:219
    rc = L.nvllm_paged_attention_seq(None, 1, 1, None, None, 4, t.handle, 4, 4, 2, 8, 0.3, 1, None,
                                     None, 0, None)
    assert rc == nv.capi.NVLLM_ERR_CONFIG
    assert "BlockTable is empty" in nv.capi.last_error()                             # paged.rs:135


def test_workspace_and_launch_agree_or_the_call_fails(policy):
    """nvllm_paged_attn_workspace_bytes and the launch take the split decision from the same policy
    and arithmetic; a workspace smaller than that (or none) is NVLLM_ERR_CONFIG, never a silently
    different launch.  Validation only: the fake pointers are never dereferenced."""
    L = nv.capi.lib()
    fake = ctypes.c_void_p(0x1000)
    shape = (16, 1, 32, 8, 128, 16, 128)      # 16 sequences: split-K, workspace needed
    need = L.nvllm_paged_attn_workspace_bytes(*shape)
    assert need > 0
    args = lambda ws, nbytes: (fake, fake, fake, fake, fake, fake, ws, nbytes, 16, 1, 32, 8, 128, 16,
                               128, 4096, 0.1, None)
    assert L.nvllm_paged_attn_bf16(*args(None, 0)) == nv.capi.NVLLM_ERR_CONFIG
    assert "needs %d" % need in nv.capi.last_error()
    assert L.nvllm_paged_attn_bf16(*args(fake, need - 1)) == nv.capi.NVLLM_ERR_CONFIG
    assert L.nvllm_paged_attn_bf16(*args(ctypes.c_void_p(0x1008), need)) == nv.capi.NVLLM_ERR_CONFIG
    assert "16-byte aligned" in nv.capi.last_error()
    # the policy hook moves both sides together
    policy.setenv("NVLLM_PAGED_SPLITS", 1)
    assert L.nvllm_paged_attn_workspace_bytes(*shape) == 0
    policy.setenv("NVLLM_PAGED_SPLITS", 4)
    assert L.nvllm_paged_attn_workspace_bytes(*shape) == 16 * 32 * 4 * (128 + 2) * 4
    assert L.nvllm_debug_set_policy(b"NVLLM_NO_SUCH_KNOB", 1) == nv.capi.NVLLM_ERR_CONFIG
    L.nvllm_debug_reset_policy()
    assert L.nvllm_paged_attn_workspace_bytes(*shape) == need


def test_no_launch_path_reads_the_environment():
    """The NVLLM_* overrides are read once (launch_policy.cpp); no kernel source calls getenv."""
    csrc = os.path.join(ROOT, "nano-vllm-rs_amd", "csrc")
    for f in os.listdir(csrc):
        if f.endswith((".hip", ".cpp", ".hpp")) and f != "launch_policy.cpp":
            assert "getenv" not in open(os.path.join(csrc, f)).read(), f
