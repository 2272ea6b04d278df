"""CPU suite: bench.py's N > 1 path without a GPU.  `--dry-run` keeps the launcher (the script starts its own ranks
when no launcher did), the sharding (jpeg_decoder_amd/sharding.py), the process group (gloo) and the reporting, and
skips the kernel."""
import json
import os
import subprocess
import sys

import helpers

BENCH = os.path.join(helpers.ROOT, "bench.py")


def _run(*argv, env=None):
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    e.update(env or {})
    return subprocess.run([sys.executable, BENCH, *argv], capture_output=True, text=True, timeout=300, cwd=helpers.ROOT, env=e)


def _line(r):
    assert r.returncode == 0, r.stderr[-1500:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


def test_bench_starts_its_own_ranks_strong_images():
    """--gpus 2 with no launcher: two ranks are started, images are dealt round-robin, ONE line with n_gpus == 2."""
    res = _line(_run("--gpus", "2", "--dry-run", "--steps", "2", "--warmup", "0", "--workload", "c5", "--scaling", "strong",
                     "--images", "100", "--frames", "16"))
    assert res["n_gpus"] == 2 and res["scaling"] == "strong" and res["dry_run"] is True
    assert res["config"]["launches_per_step_rank0"] == 4          # 50 images on rank 0 in launches of 16: 16+16+16+2
    assert len(res["roofline"]["per_rank_avg_launch_us"]) == 2
    assert "rank i mod 2" in res["config"]["sharding"]


def test_bench_stripes_mode_and_weak_default():
    res = _line(_run("--gpus", "2", "--dry-run", "--steps", "1", "--warmup", "0", "--workload", "c2", "--scaling", "strong"))
    assert res["n_gpus"] == 2 and res["scaling"] == "strong" and "MCU-row stripes" in res["config"]["sharding"]
    assert res["config"]["blocks_per_frame_rank0"] == 786432 // 2       # half of the frame's MCU rows
    res = _line(_run("--gpus", "2", "--dry-run", "--steps", "1", "--warmup", "0"))
    assert res["n_gpus"] == 2 and res["scaling"] == "weak" and res["config"]["blocks_per_frame_rank0"] == 786432


def test_bench_never_reports_a_run_that_did_not_happen():
    # fewer GPUs visible than asked for (none here): refuses before starting anything
    r = _run("--gpus", "2", "--steps", "1", "--warmup", "0")
    import torch
    if torch.cuda.device_count() < 2:
        assert r.returncode != 0 and "visible" in (r.stderr + r.stdout) and "{" not in r.stdout
    # under a launcher whose world size differs from --gpus
    r = _run("--gpus", "3", "--dry-run", env={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE" in (r.stderr + r.stdout) and "{" not in r.stdout
