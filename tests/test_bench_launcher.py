"""The N > 1 entry of bench.py on the CPU: `python bench.py --gpus N` must start N fresh ranks by
itself (the driver may call it either way), pass rank 0's JSON line through and fail if a rank
fails.  --launcher-check rehearses exactly that path over gloo without touching a GPU."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(*argv, env=None):
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, BENCH] + list(argv), capture_output=True, text=True, env=e,
                          timeout=300)


def _json_line(stdout):
    lines = [l for l in stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, stdout
    return json.loads(lines[0])


def test_plain_invocation_self_launches_two_ranks():
    r = _run("--gpus", "2", "--launcher-check", "--workload", "long", "--steps", "3")
    assert r.returncode == 0, r.stdout + r.stderr
    line = _json_line(r.stdout)
    assert line == dict(launcher_check=True, world=2, ranks_seen=2, workload="long")


def test_head_sharded_mode_self_launches_and_gathers_in_head_order():
    """--parallelism heads: the launcher path is the same; the rehearsal runs the step's one
    collective (sharding.all_gather_heads) over gloo and checks that rank r's heads land in
    columns [r * Hq/n * D, (r + 1) * Hq/n * D)."""
    r = _run("--gpus", "2", "--launcher-check", "--parallelism", "heads")
    assert r.returncode == 0, r.stdout + r.stderr
    line = _json_line(r.stdout)
    assert line["world"] == 2 and line["ranks_seen"] == 2 and line["parallelism"] == "heads"
    assert line["gathered_shape"] == [3, 2, 32 * 128] and line["heads_in_order"] is True
    r = _run("--gpus", "4", "--launcher-check", "--parallelism", "heads")
    assert r.returncode == 0 and _json_line(r.stdout)["heads_in_order"] is True


def test_head_sharded_mode_needs_a_divisor_of_the_kv_heads():
    r = _run("--gpus", "3", "--launcher-check", "--parallelism", "heads")
    assert r.returncode != 0 and "divide the 8 KV heads" in r.stderr


def test_scaling_label_follows_the_workload():
    """fixed total work divided over the ranks is strong scaling, fixed per-GPU work is weak"""
    src = open(BENCH).read()
    assert 'res["scaling"] = "strong" if (heads_mode or workload == "long") else "weak"' in src
    assert 'scaling=res.get("scaling", "weak")' in src


def test_a_failing_rank_fails_the_run():
    r = _run("--gpus", "2", "--launcher-check", "--launcher-check-fail-rank", "1")
    assert r.returncode != 0


def test_single_rank_needs_no_launcher():
    r = _run("--launcher-check")
    assert r.returncode == 0 and _json_line(r.stdout)["world"] == 1


def test_world_size_mismatch_is_an_error():
    r = _run("--gpus", "4", "--launcher-check", env=dict(WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"))
    assert r.returncode != 0 and "WORLD_SIZE=2 but --gpus 4" in r.stderr


def test_launcher_command_is_the_drivers():
    sys.path.insert(0, ROOT)
    import bench
    args = bench.parse(["--gpus", "8", "--steps", "5", "--warmup", "2"])
    cmd = bench.launcher_command(args, ["--gpus", "8", "--steps", "5", "--warmup", "2"], 29511)
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "8"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-6:] == ["--gpus", "8", "--steps", "5", "--warmup", "2"] and cmd[-7] == BENCH


def test_parent_path_never_touches_the_gpu_api():
    """The launcher branch sits in front of every torch.cuda call of main()."""
    src = open(BENCH).read()
    body = src[src.index("def main("):]
    assert body.index("launch_children(args, argv)") < body.index("dist_setup(")
    assert "torch.cuda" not in src[src.index("def launch_children"):src.index("def launcher_check")]
