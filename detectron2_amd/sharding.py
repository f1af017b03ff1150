"""Data-parallel sharding of the per-image hot path.

Every op of the hot path is independent per image (ROIs carry a batch index into their own image's
features; NMS, IoU and paste are per image; DCN is per sample), so the path shards by image with NO
data-path collective: rank r of W owns a contiguous, balanced slice of the global batch -- the
partition `torch.utils.data.DistributedSampler`-free equivalent of the reference's
`images-per-batch / world_size` split (detectron2/data/build.py:build_batch_data_loader).  The
only collectives of the hot path itself are control-plane: a barrier around the timed region and a MAX of the
wall time.  The one data collective of a training step is the GRADIENT ALL-REDUCE of the model's parameters, which the
reference gets from DistributedDataParallel (detectron2/engine/defaults.py:60-79 `create_ddp_model`, optional
`fp16_compress_hook` :75-78; launched per engine/launch.py:27-84): `GradientBuckets` below is that step for a
caller that does not wrap its model in DDP -- flat buckets, asynchronous RCCL all-reduce (backend "nccl" on
ROCm) issued as soon as a bucket's gradients exist so that it overlaps the rest of the backward pass, averaged
over the ranks, optionally carried in bf16.  The hot-path ops own no parameters except the DCN weights, which
reduce like any other parameter."""
import time
from typing import List, Optional, Sequence, Tuple

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


# Trainable parameter groups of Mask R-CNN R50-FPN (configs/COCO-InstanceSegmentation/mask_rcnn_R_50_FPN_*.yaml:
# FrozenBN, stem + res2 frozen by MODEL.BACKBONE.FREEZE_AT = 2), in the order their gradients become available in
# the backward pass (heads first, res3 last) -- the order DDP fills its buckets in.  Element counts follow from the
# layer shapes (modeling/backbone/resnet.py, fpn.py, proposal_generator/rpn.py, roi_heads/{box_head,fast_rcnn,
# mask_head}.py): 44,120,816 parameters = 176.5 MB of fp32 gradients per step.
MASK_RCNN_R50_FPN_GRADIENTS: Tuple[Tuple[str, int], ...] = (
    ("roi_heads.mask_head", 4 * (256 * 256 * 9 + 256) + (256 * 256 * 4 + 256) + (256 * 80 + 80)),
    ("roi_heads.box_predictor", (1024 * 81 + 81) + (1024 * 320 + 320)),
    ("roi_heads.box_head", (256 * 7 * 7 * 1024 + 1024) + (1024 * 1024 + 1024)),
    ("proposal_generator.rpn_head", (256 * 256 * 9 + 256) + (256 * 3 + 3) + (256 * 12 + 12)),
    ("backbone.fpn", (256 + 512 + 1024 + 2048) * 256 + 4 * 256 + 4 * (256 * 256 * 9 + 256)),
    ("backbone.res5", (1024 * 512 + 512 * 512 * 9 + 512 * 2048 + 1024 * 2048) + 2 * (2048 * 512 + 512 * 512 * 9 + 512 * 2048)),
    ("backbone.res4", (512 * 256 + 256 * 256 * 9 + 256 * 1024 + 512 * 1024) + 5 * (1024 * 256 + 256 * 256 * 9 + 256 * 1024)),
    ("backbone.res3", (256 * 128 + 128 * 128 * 9 + 128 * 512 + 256 * 512) + 3 * (512 * 128 + 128 * 128 * 9 + 128 * 512)),
)


def pack_buckets(groups: Sequence[Tuple[str, int]], bucket_bytes: int, element_size: int):
    """DDP-style packing: consecutive groups share a bucket while it stays within `bucket_bytes`; a group is never
    split (one larger than the cap gets a bucket of its own).  -> [[(name, numel), ...], ...]"""
    cap = max(1, bucket_bytes // element_size)
    out, cur, cur_n = [], [], 0
    for name, numel in groups:
        if cur and cur_n + numel > cap:
            out.append(cur)
            cur, cur_n = [], 0
        cur.append((name, int(numel)))
        cur_n += int(numel)
    if cur:
        out.append(cur)
    return out


class GradientBuckets:
    """Bucketed, asynchronous gradient all-reduce (SUM, then / world: DDP's average).

    `buckets` = [[(name, numel), ...], ...]: the parameter groups of every flat bucket, in the order their gradients
    become ready (pack_buckets() builds it from a size cap).  xGMI is point-to-point and a ring all-reduce is bound
    per link, so few, large messages are right for MI355X -- tens of MB per bucket, not DDP's 25 MiB default tuned
    for NVSwitch.  `wire_dtype` = torch.bfloat16 mirrors the reference's fp16_compress_hook (cast, reduce, cast
    back); None keeps the gradient dtype.

    reduce(i) enqueues bucket i's all-reduce on the process group's own stream (async_op=True) after whatever the
    current stream has produced so far; finish() makes the current stream wait for all of them and applies the
    average.  With `dist` = None (one process) both are no-ops, like DDP at world size 1."""

    def __init__(self, buckets: Sequence[Sequence[Tuple[str, int]]], device, dist=None, grad_dtype=torch.float32,
                 wire_dtype: Optional[torch.dtype] = None, reduce_single_rank: bool = False):
        self.dist, self.device = dist, device
        self.reduce_single_rank = reduce_single_rank  # issue the collectives at world size 1 too (plumbing checks)
        self.world = dist.get_world_size() if dist is not None else 1
        self.grad_dtype, self.wire_dtype = grad_dtype, wire_dtype or grad_dtype
        self.layout: List[List[Tuple[str, int]]] = [list(b) for b in buckets if len(b)]
        self.grads = [torch.zeros(sum(n for _, n in b), dtype=grad_dtype, device=device) for b in self.layout]
        self.wire = [g if self.wire_dtype == grad_dtype else torch.empty_like(g, dtype=self.wire_dtype)
                     for g in self.grads]
        self._pending = []

    @property
    def num_buckets(self) -> int:
        return len(self.grads)

    def numel(self) -> int:
        return sum(g.numel() for g in self.grads)

    def wire_bytes(self) -> int:
        return sum(w.numel() * w.element_size() for w in self.wire)

    def ready_after(self, group_name: str) -> int:
        """Number of leading buckets that are complete once `group_name` and every group before it have their
        gradients: those can be reduced while the rest of the backward pass still runs."""
        names = [n for b in self.layout for n, _ in b]
        assert group_name in names, group_name
        last = names.index(group_name)
        done, seen = 0, 0
        for b in self.layout:
            seen += len(b)
            if seen - 1 <= last:
                done += 1
        return done

    def reduce(self, i: int):
        if self.dist is None or (self.world == 1 and not self.reduce_single_rank):
            return
        if self.wire[i] is not self.grads[i]:
            self.wire[i].copy_(self.grads[i])  # compress (fp16_compress_hook: cast before the all-reduce)
        self._pending.append((i, self.dist.all_reduce(self.wire[i], op=self.dist.ReduceOp.SUM, async_op=True)))

    def finish(self):
        for i, work in self._pending:
            work.wait()  # the current stream waits for the collective's stream
            if self.wire[i] is not self.grads[i]:
                self.grads[i].copy_(self.wire[i])
            self.grads[i].div_(self.world)
        self._pending = []
