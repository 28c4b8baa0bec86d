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


def test_paged_attn_path_query_without_gpu():
    """nvllm_paged_attn_path is a pure function of the launch shape: decode kernel up to 48 query
    rows per kv head, the prefill kernel over the cache beyond -- on ANY cache size, the 2 Mi-block
    (64 GiB) pool of BASELINE config 5 included -- and the generic kernel for other head shapes."""
    pool5 = 2 * 1024 * 1024
    assert nv.paged_attention_path(256, 1, 32, 8, 128, 16, 128, 40960) == "decode"
    assert nv.paged_attention_path(1024, 1, 32, 8, 128, 16, 2048, pool5) == "decode"
    assert nv.paged_attention_path(256, 4, 32, 8, 128, 16, 128, 40960) == "decode"
    for T in (13, 16, 64, 512):
        assert nv.paged_attention_path(32, T, 32, 8, 128, 16, 256, pool5) == "k1_paged"
        assert nv.paged_attention_path(32, T, 32, 8, 128, 16, 256, 8 * pool5) == "k1_paged"
    assert nv.paged_attention_path(4, 1, 4, 2, 8, 4, 3, 8) == "generic"            # reference test shape
    assert nv.paged_attention_path(4, 64, 32, 8, 128, 16, 64, 4096, dtype=None) == "generic"  # f32
    L = nv.capi.lib()
    assert L.nvllm_paged_attn_path(0, 1, 32, 8, 128, 16, 4, 8, 0) == nv.capi.NVLLM_ERR_CONFIG
    assert L.nvllm_paged_attn_path(1, 1, 32, 5, 128, 16, 4, 8, 0) == nv.capi.NVLLM_ERR_CONFIG
    # NVLLM_PAGED_FLASH=0 pins long launches to the generic kernel; the workspace query follows
    try:
        assert L.nvllm_paged_attn_workspace_bytes(1, 16, 32, 8, 128, 16, 256) > 0
        assert L.nvllm_debug_set_policy(b"NVLLM_PAGED_FLASH", 0) == 0
        assert nv.paged_attention_path(32, 64, 32, 8, 128, 16, 256, pool5) == "generic"
        assert L.nvllm_paged_attn_workspace_bytes(1, 16, 32, 8, 128, 16, 256) == 0
        # a wave target below 1 would divide by zero in the split arithmetic: clamped
        assert L.nvllm_debug_set_policy(b"NVLLM_PAGED_WAVE_TARGET", -5) == 0
        assert L.nvllm_paged_attn_workspace_bytes(16, 1, 32, 8, 128, 16, 128) == 0
    finally:
        L.nvllm_debug_reset_policy()


def test_product_build_defines_no_lab_switch():
    """The K1 source carries lab switches (cycle stamps, ablations that give wrong results by design).
    They compile only under -DNVLLM_K1_LAB (an #error otherwise), and neither the product build
    (build.py) nor the Rust build script (rust/build.rs) may define any of them."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("nvllm_build", os.path.join(ROOT, "nano-vllm-rs_amd", "build.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    banned = re.compile(r"K1_ABL_|NVLLM_K1_STAMPS|NVLLM_K1_LAB|K1_FINE|K1_RT|K1_DBG_|K1_DEV_|K1_STAMP_SETUP")
    assert not any(banned.search(f) for f in b.FLAGS), b.FLAGS
    assert not banned.search(open(os.path.join(ROOT, "rust", "build.rs")).read())
    assert not banned.search(open(os.path.join(ROOT, "nano-vllm-rs_amd", "build.py")).read())
    src = open(os.path.join(ROOT, "nano-vllm-rs_amd", "csrc", "flash_attn_mfma.hip")).read()
    guard = src[src.index("#if !defined(NVLLM_K1_LAB)"):src.index("#endif", src.index("#if !defined(NVLLM_K1_LAB)"))]
    for name in set(re.findall(r"#if(?:def| defined)\s*\(?\s*((?:K1_|NVLLM_K1_)[A-Z_0-9]+)", src)) - {"NVLLM_K1_LAB"}:
        assert name in guard, "lab switch %s is not covered by the #error guard" % name
