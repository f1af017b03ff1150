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

from detectron2_amd.sharding import (MASK_RCNN_R50_FPN_GRADIENTS, GradientBuckets, Stopwatch, global_image_ids,  # noqa: E402
                                     job_throughput, pack_buckets, shard_range)


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


def test_gradient_bucket_layout():
    """44.1 M trainable parameters of Mask R-CNN R50-FPN (177 MB fp32 / 88 MB bf16); groups are never
    split; ready_after() says which leading buckets can be reduced once a group's gradients exist."""
    total = sum(n for _, n in MASK_RCNN_R50_FPN_GRADIENTS)
    assert total == 44120816
    lay = pack_buckets(MASK_RCNN_R50_FPN_GRADIENTS, 36 << 20, 2)
    assert [n for b in lay for n, _ in b] == [n for n, _ in MASK_RCNN_R50_FPN_GRADIENTS]
    assert all(sum(k for _, k in b) * 2 <= 36 << 20 or len(b) == 1 for b in lay)
    g = GradientBuckets(lay, torch.device("cpu"), None, torch.float32, torch.bfloat16)
    assert g.numel() == total and g.wire_bytes() == 2 * total and g.num_buckets == len(lay)
    assert g.ready_after("roi_heads.box_head") == 0 and g.ready_after("backbone.res3") == g.num_buckets
    assert g.ready_after(lay[0][-1][0]) == 1
    g.reduce(0)       # one process: no-ops, like DDP at world size 1
    g.finish()
    assert float(g.grads[0].abs().sum()) == 0.0


def _grad_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        out = {}
        for wire in (None, torch.bfloat16):
            g = GradientBuckets([[("a", 1000), ("b", 24)], [("c", 4096)]], torch.device("cpu"), dist, torch.float32, wire)
            for i, t in enumerate(g.grads):
                t.copy_(torch.arange(t.numel(), dtype=torch.float32) % 7 + 10 * rank + i)
            n_early = g.ready_after("b")
            for i in range(n_early):
                g.reduce(i)
            for i in range(n_early, g.num_buckets):
                g.reduce(i)
            g.finish()
            out[str(wire)] = [t.clone() for t in g.grads]
        if rank == 0:
            q.put(out)
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_two_rank_gradient_allreduce_averages():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_grad_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = q.get(timeout=100)
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    for key, tol in (("None", 0.0), ("torch.bfloat16", 0.1)):
        for i, t in enumerate(out[key]):
            want = torch.arange(t.numel(), dtype=torch.float32) % 7 + i + 5.0   # mean of rank 0 and rank 1
            assert float((t - want).abs().max()) <= tol, (key, i)


@pytest.mark.timeout(300)
def test_bench_launcher_spawns_n_ranks_and_reduces():
    """`python bench.py --gpus 2` started as ONE process becomes 2 ranks (torch.distributed.run), reports n_gpus 2,
    shards the images by rank, and the gradient all-reduce inside the step averages over both ranks.  --plumbing-only
    with gloo exercises exactly the launcher / process-group / all-reduce / timing code the GPU run uses, without a
    hot-path op (those have no CPU path)."""
    import json
    import subprocess

    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--plumbing-only", "--backend", "gloo",
                        "--steps", "2", "--grad-allreduce", "fp32"], env=env, capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout                      # rank 0 prints ONE JSON line
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["plumbing_only"] is True and d["buckets"] == 3
    assert d["allreduce_mean"] == [1.5, 1.5, 1.5] == [d["expected_mean"]] * 3
    # a torchrun world that disagrees with --gpus is refused
    env2 = dict(env, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    r2 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--plumbing-only", "--backend", "gloo"],
                        env=env2, capture_output=True, text=True, timeout=120)
    assert r2.returncode != 0 and "launcher started 1 ranks" in (r2.stderr + r2.stdout)
