"""GPU suite: bench.py's N > 1 path with REAL kernels on the one GPU a test box has (`--share-device`: every rank on device
0, gloo for the barrier / max / gathers).  Checks the launcher, both partitions, the reporting, and -- through `--verify` --
that the raster a rank other than 0 produced for its shard equals the oracle's."""
import hashlib
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import helpers
from oracle import pyoracle

pytestmark = pytest.mark.gpu
BENCH = os.path.join(helpers.ROOT, "bench.py")


def _run(*argv):
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, BENCH, *argv, "--share-device", "--verify", "--no-cpu-baseline", "--no-end-to-end", "--prewarm-ms", "50"],
                       capture_output=True, text=True, timeout=600, cwd=helpers.ROOT, env=e)
    assert r.returncode == 0, (r.stdout[-500:], r.stderr[-1500:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


def _oracle_sha(w, h, samp, nbmcuV, first_frame_index):
    """SHA-256 of the raster the oracle gives for the frame bench.py generates at block offset first_frame_index * nb."""
    from jpeg_decoder_amd import synth
    d = helpers.oracle_desc(w, h, samp)
    if nbmcuV is not None:
        d.nbmcuV = nbmcuV
    nb = [helpers.plane_shape(d, c)[0] for c in range(d.nbcomp)]
    planes = [pyoracle.synth_blocks(nb[c], 0x5EED0000 + c, first_frame_index * nb[c], 0) for c in range(d.nbcomp)]
    lq, cq = synth.shaun_qtables()[:2]
    return hashlib.sha256(pyoracle.oracle.backend_run(d, planes, [lq, cq, None, None]).tobytes()).hexdigest()


def test_two_ranks_weak_c2():
    res = _run("--gpus", "2", "--steps", "3", "--warmup", "1", "--workload", "c2", "--frames", "2")
    assert res["n_gpus"] == 2 and res["scaling"] == "weak" and "share_device" in res
    us = res["roofline"]["per_rank_avg_launch_us"]
    assert len(us) == 2 and all(u > 0 for u in us)
    assert res["value"] > 0 and res["config"]["blocks_per_frame_rank0"] == 786432
    # rank 1's first frame is frame index 1 * frames_resident = 2 of the stream
    assert res["first_raster_sha256"][1] == _oracle_sha(4096, 4096, ((1, 1),) * 3, None, 2)
    assert res["first_raster_sha256"][0] == _oracle_sha(4096, 4096, ((1, 1),) * 3, None, 0)


def test_two_ranks_strong_c5_images():
    res = _run("--gpus", "2", "--steps", "2", "--warmup", "1", "--workload", "c5", "--scaling", "strong", "--images", "64", "--frames", "16")
    assert res["n_gpus"] == 2 and res["scaling"] == "strong" and res["config"]["launches_per_step_rank0"] == 2
    assert len(res["roofline"]["per_rank_avg_launch_us"]) == 2 and all(u > 0 for u in res["roofline"]["per_rank_avg_launch_us"])
    s420 = ((2, 2), (1, 1), (1, 1))
    assert res["first_raster_sha256"][1] == _oracle_sha(1920, 1080, s420, None, 32)      # rank 1 holds 32 images: its first = index 32


def test_two_ranks_strong_c3_stripes():
    res = _run("--gpus", "2", "--steps", "2", "--warmup", "1", "--workload", "c3", "--scaling", "strong", "--frames", "1")
    assert res["n_gpus"] == 2 and "MCU-row stripes" in res["config"]["sharding"]
    assert res["config"]["blocks_per_frame_rank0"] == 1572864 // 2
    s420 = ((2, 2), (1, 1), (1, 1))
    assert res["first_raster_sha256"][1] == _oracle_sha(8192, 8192, s420, 256, 1)         # half the MCU rows, stream index 1
