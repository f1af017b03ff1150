"""Multi-process (gloo, world_size 2, CPU) test of the data-parallel path of SURVEY.md 8(e): the
per-image hot path shards by image with no data-path collective; only the control plane (barrier,
MAX of the wall time) talks.  Per-image work is stood in for by the CPU oracle (test infrastructure)
so the test checks exactly the host logic bench.py uses on N GPUs: shard ownership, that the union
of the shards reproduces the single-process result image by image, and the timing reduction."""
import os
import socket
import sys
import time

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from detectron2_amd.sharding import Stopwatch, global_image_ids, job_throughput, shard_range  # noqa: E402


def test_shard_range_partitions():
    for n in (0, 1, 2, 7, 16, 17):
        for world in (1, 2, 3, 8):
            got = [i for r in range(world) for i in shard_range(n, r, world)]
            assert got == list(range(n))
            sizes = [len(shard_range(n, r, world)) for r in range(world)]
            assert max(sizes) - min(sizes) <= 1
    assert global_image_ids(2, 3, 8) == [6, 7]
    assert job_throughput([2, 2, 2, 2], 0.5) == 16.0


def per_image_work(image_id):
    """one image of hot-path work on the CPU oracle, deterministic in the GLOBAL image id"""
    import oracle

    rng = np.random.default_rng(1000 + image_id)
    b = rng.uniform(0, 200, (300, 4)).astype(np.float32)
    b[:, 2:] += b[:, :2]
    s = ((rng.permutation(300) + 1) / 301).astype(np.float32)
    idx = rng.integers(0, 4, 300)
    keep = oracle.batched_nms(b, s, idx, 0.5)
    iou = oracle.pairwise_iou(b[:8], b)
    return keep.tolist(), float(iou.sum())


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ids = global_image_ids(2, rank, world)
        sw = Stopwatch(dist, torch.device("cpu"))
        sw.start()
        mine = {i: per_image_work(i) for i in ids}
        time.sleep(0.05 * (rank + 1))  # rank 1 is slower: the MAX must be reported on both ranks
        elapsed = sw.stop()
        gathered = [None] * world
        dist.all_gather_object(gathered, (ids, mine, elapsed))
        if rank == 0:
            q.put(gathered)
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_two_rank_gloo_sharding_matches_single_process():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    gathered = q.get(timeout=100)
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    ids0, res0, t0 = gathered[0]
    ids1, res1, t1 = gathered[1]
    assert ids0 == [0, 1] and ids1 == [2, 3]              # disjoint, contiguous, complete
    assert t0 == t1 and t0 >= 0.1                          # MAX over ranks, identical on every rank
    merged = {**res0, **res1}
    for i in range(4):                                     # union of shards == single-process result
        assert merged[i] == per_image_work(i)
    assert job_throughput([2, 2], t0) == 4 / t0
