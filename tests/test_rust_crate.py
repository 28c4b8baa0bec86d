"""rust/ is shipped as source (no Rust toolchain in the build image).  What can be checked without
a compiler: ffi.rs is exactly what scripts/gen_rust_ffi.py makes from include/nvllm_attn.h (every
symbol, argument counts), the safe wrappers only call symbols that exist with the right arity, the
wrappers carry the reference's public names, and build.rs compiles the same sources as build.py."""
import importlib.util
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RUST = os.path.join(ROOT, "rust")


def _header_decls():
    spec = importlib.util.spec_from_file_location("gen_rust_ffi", os.path.join(ROOT, "scripts", "gen_rust_ffi.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return {name: len(args) for name, args, _ in mod.declarations()}


def test_ffi_rs_is_the_generated_file():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "gen_rust_ffi.py"), "--check"])
    assert r.returncode == 0, "rust/src/attention/ffi.rs is stale: run scripts/gen_rust_ffi.py"


def test_ffi_rs_declares_every_header_symbol():
    decls = _header_decls()
    text = open(os.path.join(RUST, "src", "attention", "ffi.rs")).read()
    rust = {m.group(1): m.group(2) for m in re.finditer(r"pub fn (nvllm_\w+)\((.*?)\)", text)}
    assert set(rust) == set(decls) and len(decls) >= 100
    for name, params in rust.items():
        n = 0 if not params.strip() else len(params.split(","))
        assert n == decls[name], name
    import nano_vllm_rs_amd as nv
    assert set(rust) == set(nv.capi.SIGNATURES)          # the ctypes table binds the same set


def test_wrappers_call_existing_symbols_with_the_right_arity():
    decls = _header_decls()
    text = open(os.path.join(RUST, "src", "attention", "rocm.rs")).read()
    text = re.sub(r"//.*", "", text)
    calls = re.findall(r"ffi::(nvllm_\w+)", text)
    assert len(set(calls)) >= 12
    for name in set(calls):
        assert name in decls, name
    # argument counts of direct calls `ffi::name(...)`
    for m in re.finditer(r"ffi::(nvllm_\w+)\(", text):
        depth, i = 1, m.end()
        args, cur = 0, ""
        while depth:
            ch = text[i]
            if ch in "([":
                depth += 1
            elif ch in ")]":
                depth -= 1
            if depth == 1 and ch == ",":
                args += 1
                cur = ""
            elif depth >= 1:
                cur += ch
            i += 1
        args += 1 if cur.strip() else 0
        assert args == decls[m.group(1)], (m.group(1), args, decls[m.group(1)])


def test_wrappers_keep_the_reference_names():
    # src/attention/mod.rs:16-17 of the reference
    text = open(os.path.join(RUST, "src", "attention", "rocm.rs")).read()
    for name in ("pub struct FlashAttentionConfig", "pub fn flash_attention(", "pub fn prefill_attention(",
                 "pub fn paged_attention(", "pub fn write_kv_to_cache("):
        assert name in text, name


def test_build_rs_compiles_what_build_py_compiles():
    spec = importlib.util.spec_from_file_location("nvllm_build", os.path.join(ROOT, "nano-vllm-rs_amd", "build.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    text = open(os.path.join(RUST, "build.rs")).read()
    block = text[text.index("const SOURCES"):text.index("];")]
    assert sorted(re.findall(r'"([\w.]+)"', block)) == sorted(mod.SOURCES)
    assert "--offload-arch=gfx950" in text and "CARGO_FEATURE_ROCM" in text
