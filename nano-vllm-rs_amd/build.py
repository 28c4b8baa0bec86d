"""Build libnvllm_attn.so (hand-written gfx950 HIP kernels + C++ host + C ABI).

In-tree, explicit hipcc: the .so travels to the GPU box with the repo snapshot.
hipcc cross-compiles for gfx950 without a GPU.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIB = os.path.join(HERE, "libnvllm_attn.so")
SOURCES = [
    "nvllm_host.cpp",
    "nvllm_sched.cpp",
    "capi.cpp",
    "launch_policy.cpp",
    "reshape_and_cache.hip",
    "qk_norm_rope_cache.hip",
    "paged_attn.hip",
    "paged_attn_mid.hip",
    "flash_attn.hip",
    "flash_attn_mfma.hip",
]
HEADERS = ["nvllm_host.hpp", "nvllm_sched.hpp", "nvllm_kernels.hpp", os.path.join("..", "..", "include", "nvllm_attn.h")]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
         "-fno-gpu-rdc"]


def _newer(a, b):
    return not os.path.exists(b) or os.path.getmtime(a) > os.path.getmtime(b)


def build(force=False, verbose=False):
    os.makedirs(OBJ, exist_ok=True)
    hdr_paths = [os.path.join(CSRC, h) for h in HEADERS]
    procs = []
    objs = []
    for src in SOURCES:
        sp = os.path.join(CSRC, src)
        op = os.path.join(OBJ, src + ".o")
        objs.append(op)
        if force or _newer(sp, op) or any(_newer(h, op) for h in hdr_paths):
            cmd = [HIPCC] + FLAGS + ["-x", "hip", "-c", sp, "-o", op]
            if verbose:
                print(" ".join(cmd))
            procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    failed = False
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0 or (verbose and out):
            sys.stderr.write(out.decode(errors="replace"))
        if p.returncode != 0:
            failed = True
            sys.stderr.write("hipcc failed on %s\n" % src)
    if failed:
        raise RuntimeError("build of libnvllm_attn.so failed")
    if force or procs or not os.path.exists(LIB):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
    print(LIB)
