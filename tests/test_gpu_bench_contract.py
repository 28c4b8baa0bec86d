"""bench.py prints ONE JSON line with the keys the driver reads (reduced sizes here: the contract,
not the numbers)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*flags):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *flags], cwd=ROOT,
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    return json.loads(lines[0])


def test_default_workload_line(dev):
    d = _run("--steps", "3", "--warmup", "1", "--num-seqs", "16", "--ctx", "512", "--layers", "1",
             "--no-also", "--cpu-seconds", "0.2")
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["metric"] == "paged_decode_attn_GBps" and d["unit"] == "GB/s"
    assert (d["n_gpus"], d["steps"], d["warmup"]) == (1, 3, 1)
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "bf16" and d["data"] == "synthetic" and "workload" in d["config"]
    assert d["value"] > 0 and d["ms_per_step"] > 0
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and "traffic" in r
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and c["sample"]
    assert c["parity_max_abs"] < 2e-2          # the timed oracle output doubles as a parity check


def test_prefill_workload_line(dev):
    d = _run("--workload", "prefill", "--steps", "3", "--warmup", "1", "--ctx", "512", "--no-cpu-baseline")
    assert d["metric"] == "prefill_flash_attn_TFLOPs" and d["unit"] == "TFLOP/s"
    assert d["roofline"]["bound"] == "mfma" and d["roofline"]["peak"] == 2500.0
    assert d["cpu_baseline"] is None and d["value"] > 0
