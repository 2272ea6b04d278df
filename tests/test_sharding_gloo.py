"""CPU suite: the N>1 path.  Two processes on gloo each decode their shard (stripes of one frame,
and images of a batch) and the gathered result must equal the single-process result.  The compute
stand-in here is the oracle (test code may use it); on GPUs bench.py runs the same partitioning with
jpeggpu_backend_run_batch_device and no data-path collective."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import helpers
from oracle import pyoracle
from jpeg_decoder_amd import sharding


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    orc = pyoracle.oracle
    QT = helpers.qtables_ramp()
    # (1) MCU-row stripes of one frame
    d = helpers.oracle_desc(333, 210, ((2, 2), (1, 1), (1, 1)))
    planes = helpers.synth_planes(d, 77)
    m0, m1 = sharding.shard_mcu_rows(d.nbmcuV, rank, world)
    sd, sub = sharding.stripe_planes(d, planes, m0, m1)
    mine = torch.from_numpy(orc.backend_run(sd, sub, QT).copy())
    sizes = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(sizes, torch.tensor([mine.shape[0]]))
    rows = max(int(s) for s in sizes)
    pad = torch.zeros((rows, mine.shape[1]), dtype=torch.uint8)
    pad[:mine.shape[0]] = mine
    got = [torch.zeros_like(pad) for _ in range(world)]
    dist.all_gather(got, pad)                      # test-side gather only; the data path has none
    whole = np.concatenate([g[:int(s)].numpy() for g, s in zip(got, sizes)])
    ok1 = np.array_equal(whole, orc.backend_run(d, planes, QT))
    # (2) independent images round-robin
    nimg = 5
    dd = helpers.oracle_desc(64, 48, ((1, 1),) * 3)
    imgs = [helpers.synth_planes(dd, 100 + i) for i in range(nimg)]
    mine_idx = sharding.shard_images(nimg, rank, world)
    outs = {i: orc.backend_run(dd, imgs[i], QT) for i in mine_idx}
    allidx = [None] * world
    dist.all_gather_object(allidx, {i: o.tobytes() for i, o in outs.items()})
    merged = {}
    for part in allidx:
        merged.update(part)
    ok2 = sorted(merged) == list(range(nimg)) and all(merged[i] == orc.backend_run(dd, imgs[i], QT).tobytes() for i in range(nimg))
    if rank == 0:
        q.put((ok1, ok2))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding_matches_single_process():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    ok = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert ok == (True, True)


def test_partitions_cover_everything_once():
    for n in (1, 2, 7, 68, 512):
        for w in (1, 2, 4, 8):
            rows = [sharding.shard_mcu_rows(n, r, w) for r in range(w)]
            assert rows[0][0] == 0 and rows[-1][1] == n
            assert all(rows[i][1] == rows[i + 1][0] for i in range(w - 1))
            imgs = sorted(sum((sharding.shard_images(n, r, w) for r in range(w)), []))
            assert imgs == list(range(n))
