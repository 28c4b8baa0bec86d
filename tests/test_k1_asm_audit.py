"""Build-time audit of the K1 flash-attention kernel's ISA (no GPU needed).

K1 keeps its accumulators and Q fragments in literally named AGPRs and issues its MFMAs as
inline asm, which hipcc neither schedules around nor pads hazards for.  The audit
(scripts/k1_lab/check_asm.py) compiles nothing itself; this test cross-compiles the kernel to
gfx950 assembly and requires: no compiler-generated use of the owned AGPR range outside the asm
statements, no scratch (spill) traffic inside the step loops, no VALU write of an MFMA operand
within the two instructions in front of an MFMA statement, no VALU-written SGPR read by an LDS-DMA
load within 5 wait states.  Vector-register spills: none in the one-item-per-workgroup variants;
the chained ones (CHAIN) keep a few lane constants in scratch BETWEEN items."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_k1_isa_audit(tmp_path):
    src = os.path.join(ROOT, "nano-vllm-rs_amd", "csrc", "flash_attn_mfma.hip")
    out = str(tmp_path / "k1.s")
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17",
                           "-I" + os.path.join(ROOT, "nano-vllm-rs_amd", "csrc"),
                           "-I" + os.path.join(ROOT, "include"), "-x", "hip",
                           "--cuda-device-only", "-S", src, "-o", out])
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "k1_lab", "check_asm.py"), out],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    text = r.stdout
    assert "inside step loops: 0" in text
    spills = {l.split()[2]: int(l.split()[1]) for l in text.splitlines() if l.startswith("spills ")}
    assert len(spills) == 9, text                   # merge kernel + eight instantiations of the K1 kernel
    for name, n in spills.items():
        chained = "flash_attn_mfma_kernel" in name and name.split("flash_attn_mfma_kernelI")[1].startswith(
            ("Lb1ELb0ELb0ELb1", "Lb1ELb1ELb0ELb1"))
        # the chained variants keep a few lane constants in scratch BETWEEN items (never inside the
        # step loops: asserted above); everything else does not spill at all
        assert n <= (24 if chained else 0), (name, n)


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_mid_kernel_isa_audit(tmp_path):
    """The mid kernel (paged_attn_mid.hip) lives on two workgroups per CU: at most 256 registers per
    wave and exactly 80 KiB of LDS, no spill; its LDS-DMA statements are the only users of M0 and
    each is `s_mov_b32 m0 / s_nop 0 / 4 x buffer_load ... offset:1024 j lds` behind the s_nop that lets a
    VALU-written descriptor settle; every loop iteration has exactly one barrier (P.V runs one tile behind)."""
    import re
    src = os.path.join(ROOT, "nano-vllm-rs_amd", "csrc", "paged_attn_mid.hip")
    out = str(tmp_path / "mid.s")
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17",
                        "-I" + os.path.join(ROOT, "nano-vllm-rs_amd", "csrc"),
                        "-I" + os.path.join(ROOT, "include"), "-x", "hip", "--cuda-device-only", "-S", src,
                        "-o", out, "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    usage = r.stderr
    names = re.findall(r"Function Name: (\S+)", usage)
    assert len(names) == 2 and all("paged_attn_mid_kernel" in n for n in names), names   # NT and plain
    assert [int(x) for x in re.findall(r"VGPRs: (\d+)", usage)] and max(int(x) for x in re.findall(r"VGPRs: (\d+)", usage)) <= 256
    assert set(re.findall(r"VGPRs Spill: (\d+)", usage)) == {"0"}
    assert set(re.findall(r"SGPRs Spill: (\d+)", usage)) == {"0"}
    assert set(re.findall(r"LDS Size \[bytes/block\]: (\d+)", usage)) == {"81920"}
    assert set(re.findall(r"ScratchSize \[bytes/lane\]: (\d+)", usage)) == {"0"}
    lines = [l.strip() for l in open(out) if l.startswith("\t") and not l.strip().startswith((".", ";"))]
    dma = [i for i, l in enumerate(lines) if l.startswith("buffer_load_dwordx4") and l.endswith("lds")]
    # per kernel: K(0), K(1) in the prologue + V and K in the loop, each for up to four blocks (a wave
    # without rows fetches for the computing ones) = 16 requests x 4 pieces
    assert len(dma) == 2 * 64, len(dma)
    # a request = one write of M0 and four pieces told apart by the immediate offset (which moves the LDS
    # address and the memory address alike)
    m0 = [i for i, l in enumerate(lines) if re.search(r"\bm0\b", l)]
    assert len(m0) == len(dma) // 4 and all(lines[i].startswith("s_mov_b32 m0") for i in m0)
    for i in m0:
        assert lines[i - 1].startswith("s_nop 4") and lines[i + 1].startswith("s_nop 0"), lines[i - 1:i + 2]
        assert [k for k in range(i + 2, i + 6)] == [k for k in dma if i < k < i + 6], lines[i:i + 6]
        offs = [re.search(r"offset:(\d+)", lines[k]) for k in range(i + 2, i + 6)]
        assert [int(o.group(1)) if o else 0 for o in offs] == [0, 1024, 2048, 3072]
    nt = sum(" nt " in lines[i] or lines[i].endswith(" nt lds") for i in dma)
    assert nt == 64                                  # the NT instantiation carries the hint on every request
