"""Data-parallel sharding of the per-image hot path.

Every op of the hot path is independent per image (ROIs carry a batch index into their own image's
features; NMS, IoU and paste are per image; DCN is per sample), so the path shards by image with NO
data-path collective: rank r of W owns a contiguous, balanced slice of the global batch -- the
partition `torch.utils.data.DistributedSampler`-free equivalent of the reference's
`images-per-batch / world_size` split (detectron2/data/build.py:build_batch_data_loader).  The
only collectives are control-plane: a barrier around the timed region and a MAX of the wall time.
Gradient all-reduce of backbone / head parameters belongs to the caller's DistributedDataParallel
(detectron2/engine/defaults.py:60-79), not to these ops (they own no parameters except DCN
weights, which DDP reduces like any other nn.Parameter)."""
import time
from typing import List, Sequence

import torch


def shard_range(n_items: int, rank: int, world: int) -> range:
    """Contiguous balanced slice of range(n_items) owned by `rank` (first n % world ranks get one more)."""
    assert 0 <= rank < world and n_items >= 0
    base, rem = divmod(n_items, world)
    start = rank * base + min(rank, rem)
    return range(start, start + base + (1 if rank < rem else 0))


def global_image_ids(images_per_rank: int, rank: int, world: int) -> List[int]:
    """Weak scaling: the global batch has images_per_rank * world images; this rank's ids."""
    return list(shard_range(images_per_rank * world, rank, world))


class Stopwatch:
    """Timed region bracketed by barrier + device sync on both sides; elapsed = MAX over ranks."""

    def __init__(self, dist=None, device=None):
        self.dist, self.device = dist, device
        self._t0 = None

    def _sync(self):
        if self.device is not None and self.device.type == "cuda":
            torch.cuda.synchronize(self.device)
        if self.dist is not None:
            self.dist.barrier()
        if self.device is not None and self.device.type == "cuda":
            torch.cuda.synchronize(self.device)

    def start(self):
        self._sync()
        self._t0 = time.perf_counter()

    def stop(self) -> float:
        self._sync()
        elapsed = time.perf_counter() - self._t0
        if self.dist is not None:
            dev = self.device if self.device is not None else torch.device("cpu")
            t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
            elapsed = float(t.item())
        return elapsed


def job_throughput(units_per_rank: Sequence[int], elapsed_s: float) -> float:
    """Whole-job units / second: all ranks' units over the max-over-ranks time."""
    return sum(units_per_rank) / elapsed_s
