"""SURVEY section 5 (race detection / sanitizers): the host half of the library -- nvllm_host.cpp,
nvllm_sched.cpp, capi.cpp, launch_policy.cpp -- built by g++ with AddressSanitizer + UBSan (CPU
build only; GPU sanitizers are not available on this pool) and driven through the C ABI by
tests/host_sanitizer/driver.cpp: long random allocator and scheduler traces, every error path, the
batch roll-back.  The kernels' launchers are replaced by "no device" stand-ins
(tests/host_sanitizer/kernel_stubs.cpp), so no device code is involved."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "nano-vllm-rs_amd", "csrc")
HERE = os.path.join(ROOT, "tests", "host_sanitizer")
ROCM = os.environ.get("ROCM_PATH", "/opt/rocm")


@pytest.mark.skipif(shutil.which("g++") is None or not os.path.exists(os.path.join(ROCM, "include", "hip")),
                    reason="g++ or the HIP headers are missing")
def test_host_code_under_asan_and_ubsan(tmp_path):
    exe = str(tmp_path / "host_sanitizer_driver")
    srcs = [os.path.join(CSRC, f) for f in ("nvllm_host.cpp", "nvllm_sched.cpp", "capi.cpp", "launch_policy.cpp")]
    srcs += [os.path.join(HERE, "kernel_stubs.cpp"), os.path.join(HERE, "driver.cpp")]
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fno-omit-frame-pointer", "-fsanitize=address,undefined",
           "-fno-sanitize-recover=all", "-D__HIP_PLATFORM_AMD__", "-I" + os.path.join(ROCM, "include"),
           "-I" + CSRC, "-I" + os.path.join(ROOT, "include"), "-Wall", "-Wno-unused-function",
           "-Wno-unused-result"] + srcs + ["-L" + os.path.join(ROCM, "lib"), "-lamdhip64",
                                           "-Wl,-rpath," + os.path.join(ROCM, "lib"), "-o", exe]
    b = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert b.returncode == 0, b.stderr[-4000:]
    env = dict(os.environ)
    # leaks inside the HIP / HSA runtime's own start-up are not ours to fix
    supp = tmp_path / "lsan.supp"
    supp.write_text("leak:libamdhip64\nleak:libhsa-runtime64\nleak:libamd_comgr\n")
    env["LSAN_OPTIONS"] = "suppressions=%s:print_suppressions=0" % supp
    env["ASAN_OPTIONS"] = "abort_on_error=0:detect_stack_use_after_return=1"
    env["UBSAN_OPTIONS"] = "print_stacktrace=1"
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, (r.stdout + r.stderr)[-6000:]
    assert "host sanitizer driver: ok" in r.stdout
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr
