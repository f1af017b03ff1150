#!/usr/bin/env python3
"""bench.py -- throughput of the Detectron2 detection hot path on MI355X.

One "step" = one pass of the TRAINING hot path of Mask R-CNN R50-FPN over one synthetic batch
(BASELINE.json configs[1]: 2 images, 1333x800 -> padded 800x1344, bf16 features, FPN p2..p5,
256 ch; SURVEY.md 8(d) inputs, seed 1234):
    per image : pairwise_iou(16 GT x 268,569 anchors)              [RPN matching, rpn.py:339]
                batched_nms(8,819 proposals, 5 levels, thr 0.7)     [proposal_utils.py:121]
                pairwise_iou(16 GT x 1,016 proposals)               [roi_heads.py:266]
    per batch : box  ROIPooler 7x7,   1024 ROIs over p2..p5  forward + backward   [roi_heads.py:_forward_box]
                mask ROIPooler 14x14, 256 ROIs over p2..p5  forward + backward   [roi_heads.py:_forward_mask]
                (ROIPooler = level assignment + ROIAlignV2 on the assigned level, poolers.py:206-263)
Everything else of the model (backbone convs, heads) is out of the hot path's scope and is NOT
in the step.  Inputs are resident in HBM before the timed region.  `value` = images / second
through the hot path, whole job (all ranks).  Multi-GPU: images shard across ranks, no data-path
collective (the ops own no parameters); weak scaling.

Extra JSON fields: `roofline` (dominant op, algorithmic bytes of SURVEY 8(d) / measured time),
`cpu_baseline` (the oracle timed on this host, bounded sample), `ops` (per-op breakdown).
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
STRIDES = (4, 8, 16, 32)
FEAT_HW = ((200, 336), (100, 168), (50, 84), (25, 42))  # 800x1344 padded input
IMG_H, IMG_W = 800, 1344
C = 256
IMAGES_PER_GPU = 2


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--layout", choices=["nchw", "nhwc"], default="nhwc",
                    help="feature memory format: nchw (reference default) or nhwc (torch.channels_last)")
    ap.add_argument("--dtype", choices=["bf16", "fp32", "fp16"], default="bf16")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    return ap.parse_args()


# ------------------------------------------------------------------------------------ inputs
def make_anchors():
    """Standard RPN anchors (sizes 32..512, ratios .5/1/2, strides 4..64) -> 268,569 x 4."""
    out = []
    for size, stride, (h, w) in zip((32, 64, 128, 256, 512), (4, 8, 16, 32, 64),
                                    FEAT_HW + ((13, 21),)):
        ys, xs = torch.meshgrid(torch.arange(h) * stride, torch.arange(w) * stride, indexing="ij")
        ctr = torch.stack([xs, ys, xs, ys], -1).reshape(-1, 1, 4).float()
        cells = []
        for r in (0.5, 1.0, 2.0):
            ww = math.sqrt(size * size / r)
            hh = ww * r
            cells.append([-ww / 2, -hh / 2, ww / 2, hh / 2])
        out.append((ctr + torch.tensor(cells)[None]).reshape(-1, 4))
    return torch.cat(out)


def make_boxes(gen, n, smin, smax):
    s = torch.exp(torch.empty(n).uniform_(math.log(smin), math.log(smax), generator=gen))
    ar = torch.exp(torch.empty(n).uniform_(math.log(0.5), math.log(2.0), generator=gen))
    w, h = s * ar.sqrt(), s / ar.sqrt()
    cx = torch.empty(n).uniform_(0, IMG_W, generator=gen)
    cy = torch.empty(n).uniform_(0, IMG_H, generator=gen)
    b = torch.stack([cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2], 1)
    b[:, 0::2] = b[:, 0::2].clamp(0, IMG_W)
    b[:, 1::2] = b[:, 1::2].clamp(0, IMG_H)
    return b


def assign_levels(boxes):
    """ROIPooler level assignment, detectron2/modeling/poolers.py:23-59 (canonical 224 @ level 4)."""
    sizes = ((boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1])).sqrt()
    lv = torch.floor(4 + torch.log2(sizes / 224 + 1e-8))
    return torch.clamp(lv, 2, 5).long() - 2


def image_generator(seed, image_id):
    """One RNG stream per GLOBAL image id: the synthetic batch does not depend on how it is sharded."""
    return torch.Generator().manual_seed(seed * 100003 + image_id)


class Workload:
    def __init__(self, dev, dtype, layout, seed=1234, image_ids=(0, 1)):
        n_img = len(image_ids)
        gens = [image_generator(seed, i) for i in image_ids]
        gen = gens[0]
        self.dev, self.n_img, self.image_ids = dev, n_img, list(image_ids)
        self.feats = []
        for (h, w) in FEAT_HW:
            f = torch.stack([torch.rand(C, h, w, generator=g) * 2 - 1 for g in gens]).to(dtype).to(dev)
            if layout == "nhwc":
                f = f.contiguous(memory_format=torch.channels_last)
            self.feats.append(f.requires_grad_(True))
        self.anchors = make_anchors().to(dev)
        assert self.anchors.shape[0] == 268569
        self.gt = [make_boxes(g, 16, 16, 512).to(dev) for g in gens]
        # RPN proposals entering NMS: 2000 per level p2-p5 + 819 for p6, distinct scores
        self.nms_in = []
        for gen in gens:
            per = (2000, 2000, 2000, 2000, 819)
            lv = torch.cat([torch.full((k,), i, dtype=torch.int64) for i, k in enumerate(per)])
            sz = torch.cat([torch.tensor([32.0, 64, 128, 256, 512])[i].repeat(k) for i, k in enumerate(per)])
            n = lv.numel()
            s = sz * torch.exp(torch.empty(n).uniform_(-0.5, 0.5, generator=gen))
            ar = torch.exp(torch.empty(n).uniform_(math.log(0.5), math.log(2.0), generator=gen))
            w, h = s * ar.sqrt(), s / ar.sqrt()
            cx = torch.empty(n).uniform_(0, IMG_W, generator=gen)
            cy = torch.empty(n).uniform_(0, IMG_H, generator=gen)
            b = torch.stack([cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2], 1)
            sc = torch.rand(n, generator=gen) + torch.arange(n) * 1e-9
            self.nms_in.append((b.to(dev), sc.to(dev), lv.to(dev)))
        self.props = [make_boxes(g, 1016, 16, 600).to(dev) for g in gens]
        # sampled ROIs: 512 / image (box head), 128 fg / image (mask head), as list[Boxes] per image
        from detectron2_amd.modeling import ROIPooler
        from detectron2_amd.structures import Boxes

        box_b = [make_boxes(g, 512, 16, 600) for g in gens]
        mask_b = [make_boxes(g, 128, 16, 600) for g in gens]
        self.box_lists = [Boxes(b.to(dev)) for b in box_b]
        self.mask_lists = [Boxes(b.to(dev)) for b in mask_b]
        self.box_level_counts = [torch.bincount(assign_levels(torch.cat(bb)), minlength=4).tolist()
                                 for bb in (box_b, mask_b)]
        scales = [1.0 / s for s in STRIDES]
        self.box_pooler = ROIPooler(7, scales, 0, "ROIAlignV2")
        self.mask_pooler = ROIPooler(14, scales, 0, "ROIAlignV2")
        mf = torch.channels_last if layout == "nhwc" else torch.contiguous_format
        self.gbox = torch.cat([torch.randn(512, C, 7, 7, generator=g) for g in gens]).to(dtype).to(dev) \
            .contiguous(memory_format=mf)
        self.gmask = torch.cat([torch.randn(128, C, 14, 14, generator=g) for g in gens]).to(dtype).to(dev) \
            .contiguous(memory_format=mf)
        self.esize = torch.empty((), dtype=dtype).element_size()

    # algorithmic bytes of ONE tile-gather launch (fine = FPN levels with > 512 tiles: p2, p3 here): the levels'
    # share of SURVEY 8(d)'s backward formula  s*K_l*C*R^2 (dY rows of the ROIs on those levels) + 2*s*N*C*H_l*W_l
    def alg_bytes_bwd_kernel(self, which, fine):
        s = self.esize
        counts, R = (self.box_level_counts[0], 7) if which == "box" else (self.box_level_counts[1], 14)
        tot = 0
        for l, (h, w) in enumerate(FEAT_HW):
            tiles = ((h + 7) // 8) * ((w + 7) // 8) * self.n_img
            if (tiles > 512) != fine:
                continue
            tot += s * counts[l] * C * R * R + 2 * s * self.n_img * C * h * w
        return tot

    # algorithmic (compulsory) bytes per op, SURVEY.md 8(d)
    def alg_bytes(self):
        s = self.esize
        feat = [self.n_img * C * h * w * s for (h, w) in FEAT_HW]
        d = {}
        for name, counts, R in (("roi_align_box", self.box_level_counts[0], 7),
                                ("roi_align_mask", self.box_level_counts[1], 14)):
            fwd = bwd = 0
            for l in range(4):
                k = counts[l]
                if k == 0:
                    continue
                fwd += feat[l] + 20 * k + s * k * C * R * R
                bwd += s * k * C * R * R + 2 * feat[l]
            d[name + "_fwd"], d[name + "_bwd"] = fwd, bwd
        d["roi_align_bwd"] = d["roi_align_box_bwd"] + d["roi_align_mask_bwd"]  # the step runs them in one backward pass
        n, m = 16, 268569
        d["pairwise_iou_rpn"] = self.n_img * (16 * (n + m) + 4 * n * m)
        d["pairwise_iou_roi"] = self.n_img * (16 * (16 + 1016) + 4 * 16 * 1016)
        nk = 8819
        d["batched_nms_rpn"] = self.n_img * (16 * nk + 8 * nk)  # boxes + keep list; bitmask is internal
        return d


class Timer:
    """Per-op HIP-event timing on torch's current stream (the stream every kernel is launched on).
    `only`: time just this op and call the others bare (two events cost ~10 us of host time; in the timed
    region only the roofline op carries them, the full breakdown comes from a separate, untimed pass)."""

    def __init__(self, only=None):
        self.pairs = {}
        self.only = only

    def run(self, name, fn):
        if self.only is not None and name != self.only:
            return fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        r = fn()
        b.record()
        self.pairs.setdefault(name, []).append((a, b))
        return r

    def totals_ms(self):
        return {k: sum(a.elapsed_time(b) for a, b in v) for k, v in self.pairs.items()}

    def counts(self):
        return {k: len(v) for k, v in self.pairs.items()}


def step(w, t=None):
    from detectron2_amd.layers import batched_nms_images
    from detectron2_amd.structures import pairwise_iou

    run = (lambda name, fn: t.run(name, fn)) if t is not None else (lambda name, fn: fn())
    # a training iteration produces NEW feature maps: drop the NHWC staging copies an NCHW run cached for the
    # previous step's tensors (the two poolers of one step still share one copy)
    from detectron2_amd.modeling import poolers as _poolers
    _poolers._NHWC_CACHE.clear()
    # RPN NMS of all images of the batch: one call, one launch per pipeline stage for the whole batch; the kept
    # counts come back with ONE host sync (the reference loops over images, one sync each).  The sync is deferred
    # past the anchor-labelling IoU, which does not depend on the proposals (RPN.forward computes the two in either
    # order: rpn.py label_and_sample_anchors / predict_proposals): the host enqueues it while the NMS pipeline runs.
    nms_done = run("batched_nms_rpn", lambda: batched_nms_images(w.nms_in, 0.7, defer=True))
    for i in range(w.n_img):
        run("pairwise_iou_rpn", lambda: pairwise_iou(w.gt[i], w.anchors))
    run("batched_nms_rpn_sync", nms_done)
    for i in range(w.n_img):
        run("pairwise_iou_roi", lambda: pairwise_iou(w.gt[i], w.props[i]))
    # both poolers forward, then ONE backward pass through both (a training iteration sums the box- and mask-head
    # losses and calls backward once: the autograd engine is entered once, both tile gathers run inside it)
    yb = run("roi_align_box_fwd", lambda: w.box_pooler(w.feats, w.box_lists))
    ym = run("roi_align_mask_fwd", lambda: w.mask_pooler(w.feats, w.mask_lists))
    run("roi_align_bwd", lambda: torch.autograd.backward([yb, ym], [w.gbox, w.gmask]))
    outs = [yb, ym]
    for f in w.feats:
        f.grad = None
    return outs


# ------------------------------------------------------------------------------------ CPU baseline
def cpu_baseline(w):
    """The oracle (plain-C port, 1 thread) on a bounded sample of the same workload: image 0's
    IoU + NMS inputs in full, and 16 ROIs x 32 channels of every level for ROIAlign fwd+bwd
    (scaled to the full ROI / channel count; ROIAlign cost is linear in both)."""
    import oracle

    t_img = 0.0
    gt, an = w.gt[0].cpu().numpy(), w.anchors.cpu().numpy()
    t0 = time.perf_counter(); oracle.pairwise_iou(gt, an); t_img += time.perf_counter() - t0
    b, s, lv = [x.cpu().numpy() for x in w.nms_in[0]]
    t0 = time.perf_counter(); oracle.batched_nms(b, s, lv, 0.7); t_img += time.perf_counter() - t0
    t0 = time.perf_counter(); oracle.pairwise_iou(gt, w.props[0].cpu().numpy()); t_img += time.perf_counter() - t0
    t_batch = 0.0
    cs = 32
    for lists, R in ((w.box_lists, 7), (w.mask_lists, 14)):
        allb = torch.cat([b.tensor.cpu() for b in lists])
        bidx = torch.cat([torch.full((len(b),), float(i)) for i, b in enumerate(lists)])
        rois_all = torch.cat([bidx[:, None], allb], 1)
        lv = assign_levels(allb)
        for l in range(4):
            rl = rois_all[lv == l]
            k = rl.shape[0]
            if k == 0:
                continue
            ks = min(16, k)
            x = w.feats[l].detach()[:, :cs].float().cpu().contiguous().numpy()
            r = rl[:ks].contiguous().numpy()
            t0 = time.perf_counter()
            y = oracle.roi_align_forward(x, r, (R, R), 1.0 / STRIDES[l], 0, True)
            oracle.roi_align_backward(y, r, x.shape, 1.0 / STRIDES[l], 0, True)
            dt = time.perf_counter() - t0
            t_batch += dt * (k / ks) * (C / cs)
    sec_per_batch = t_img * w.n_img + t_batch
    return {"value": round(w.n_img / sec_per_batch, 4), "unit": "img/s", "cores": 1, "kind": "port",
            "sample": "oracle/d2_oracle.c single thread: image 0 IoU+NMS in full; ROIAlign fwd+bwd on 16 ROIs x 32 "
                      "of 256 channels per level, scaled linearly to all ROIs/channels",
            "host_cores_available": os.cpu_count()}


def pmc_traffic(op, layout):
    """HBM bytes per launch of the op's kernels from the committed rocprofv3 PMC passes
    (profiles/<round>/pmc_traffic_<layout>.json, written by scripts/pmc_summary.py from separate
    FETCH_SIZE / WRITE_SIZE runs, gfx950 FETCH_SIZE x2 correction applied there); None if absent."""
    import glob

    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", f"pmc_traffic_{layout}.json"))):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        if op in d.get("ops", {}):
            best = d["ops"][op].get("hbm_bytes_per_launch")
    return best


ROOFLINE_OP = "roi_align_bwd"      # the op holding the dominant kernel (the box head's tile gather)
ROOFLINE_KERNEL_OP = "roi_align_box_bwd"  # algorithmic bytes / PMC traffic of that kernel's pooler


# ------------------------------------------------------------------------------------ main
def main():
    args = parse()
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    assert torch.cuda.is_available(), "bench.py needs a GPU (there is no CPU fallback)"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    dtype = {"bf16": torch.bfloat16, "fp32": torch.float32, "fp16": torch.float16}[args.dtype]
    from detectron2_amd.sharding import Stopwatch, global_image_ids

    # weak scaling: 2 images per GPU; rank r owns images [2r, 2r+1] of the global synthetic batch
    w = Workload(dev, dtype, args.layout, image_ids=global_image_ids(IMAGES_PER_GPU, rank, world))

    for _ in range(args.warmup):
        step(w)
    sw = Stopwatch(dist, dev)
    dom_timer = Timer(only=ROOFLINE_OP)  # HIP events around the roofline op only, inside the timed region
    # ... and HIP events on the launch stream right around its dominant KERNEL (the fine-level tile gather),
    # recorded by the library itself (d2amd_timing_*): this is the duration rocprofv3 reports for the kernel
    import ctypes
    from detectron2_amd import _C as _dc
    _dc.lib().d2amd_timing_enable(1)  # only the roofline kernel (fine levels, 7x7) inside the timed region
    sw.start()
    for _ in range(args.steps):
        step(w, dom_timer)
    elapsed = sw.stop()
    ktimes = {}
    def read_ktimes():
        for kn in ("pool_bwd_staged_r7", "pool_bwd_staged_r14", "pool_bwd_fine_r7", "pool_bwd_coarse_r7",
                   "pool_bwd_fine_r14", "pool_bwd_coarse_r14"):
            tot, cnt = ctypes.c_double(0.0), ctypes.c_int(0)
            _dc.check(_dc.lib().d2amd_timing_read(kn.encode(), ctypes.byref(tot), ctypes.byref(cnt)))
            if cnt.value and kn not in ktimes:
                ktimes[kn] = (tot.value / cnt.value, cnt.value)

    read_ktimes()
    _dc.lib().d2amd_timing_enable(15)  # the untimed breakdown pass times all four tile-gather launches
    # per-op breakdown: a separate, UNTIMED pass with events around every op (their host cost would otherwise
    # sit in the timed region: ~0.15 ms of a 0.9 ms step)
    timer = Timer()
    bsteps = min(args.steps, 20)
    for _ in range(bsteps):
        step(w, timer)
    torch.cuda.synchronize()
    read_ktimes()
    _dc.lib().d2amd_timing_enable(0)

    if rank == 0:
        ms_step = elapsed / args.steps * 1e3
        totals = {k: v * args.steps / bsteps for k, v in timer.totals_ms().items()}
        dom_ms_timed = dom_timer.totals_ms()[ROOFLINE_OP] / args.steps
        alg = w.alg_bytes()
        ops = {}
        for k, tot in totals.items():
            per_step_ms = tot / args.steps
            e = {"ms_per_step": round(per_step_ms, 4)}
            if k in alg:
                e["alg_MB"] = round(alg[k] / 1e6, 2)
                e["GBps"] = round(alg[k] / 1e9 / (per_step_ms / 1e3), 1)
                e["frac_hbm_peak"] = round(e["GBps"] / HBM_PEAK_GBS, 4)
            ops[k] = e
        counts = {k: v // bsteps for k, v in timer.counts().items()}
        for k in ops:
            ops[k]["launches_per_step"] = counts[k]
            ops[k]["ms_per_launch"] = round(ops[k]["ms_per_step"] / counts[k], 4)
        # roofline: the dominant kernel = the tile-gather launch of the box-head backward (pool_bwd_mfma_kernel /
        # pool_bwd_staged_kernel in the rocprofv3 stats; with D2AMD_POOL_NOSTAGED the fine-level launch of the two-launch register-gather
        # kernels), timed by HIP events on its launch stream inside the timed region.  Its algorithmic bytes:
        # SURVEY 8(d)'s backward formula for the levels the launch writes (dY read once + 2 x grad_input).
        dom = ROOFLINE_OP
        ops[dom]["ms_per_launch_timed_region"] = round(dom_ms_timed / counts[dom], 4)
        if args.layout == "nhwc" and "pool_bwd_staged_r7" in ktimes:
            # one launch for all FPN levels (LDS-staged tile gather): SURVEY 8(d)'s backward bytes of the whole op
            k_ms, k_n = ktimes["pool_bwd_staged_r7"]
            kb = alg[ROOFLINE_KERNEL_OP]
            roof = {"bound": "hbm", "kernel": "pool_bwd_mfma_kernel<T, 8> (16-bit I/O; fp32: pool_bwd_staged_kernel<float, 4, 8>), the 7x7 (box head) pooler's tile gather over all FPN levels, inside roi_align_bwd",
                    "achieved": round(kb / 1e9 / (k_ms / 1e3), 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(kb / 1e9 / (k_ms / 1e3) / HBM_PEAK_GBS, 4),
                    "traffic": pmc_traffic(ROOFLINE_KERNEL_OP, args.layout),
                    "traffic_note": "PMC bytes of the box-head pooler backward (records + tile lists/zero fill + tile gather)",
                    "alg_bytes_per_launch": int(kb), "ms_per_launch": round(k_ms, 4), "launches_timed": k_n,
                    "timing": "HIP events recorded by the library on the kernel's launch stream right around the launch, "
                              "mean over the timed steps",
                    "op": {"name": dom, "alg_bytes": int(alg[dom] / counts[dom]),
                           "ms_per_launch_events_around_op": round(dom_ms_timed / counts[dom], 4),
                           "frac": round(alg[dom] / counts[dom] / 1e9 / (dom_ms_timed / counts[dom] / 1e3) / HBM_PEAK_GBS, 4)},
                    "kernels_ms": {k: round(v[0], 4) for k, v in ktimes.items()}}
        elif args.layout == "nhwc" and "pool_bwd_fine_r7" in ktimes:
            k_ms, k_n = ktimes["pool_bwd_fine_r7"]
            kb = w.alg_bytes_bwd_kernel("box", fine=True)
            roof = {"bound": "hbm", "kernel": "pool_bwd_nhwc_kernel (fine FPN levels) of roi_align_box_bwd",
                    "achieved": round(kb / 1e9 / (k_ms / 1e3), 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(kb / 1e9 / (k_ms / 1e3) / HBM_PEAK_GBS, 4), "traffic": pmc_traffic(dom, args.layout),
                    "traffic_note": "PMC bytes are for the whole op (both tile-gather launches + records)",
                    "alg_bytes_per_launch": int(kb), "ms_per_launch": round(k_ms, 4), "launches_timed": k_n,
                    "timing": "HIP events recorded by the library on the kernel's launch stream right around the launch, "
                              "mean over the timed steps",
                    "op": {"name": dom, "alg_bytes": int(alg[dom] / counts[dom]),
                           "ms_per_launch_events_around_op": round(dom_ms_timed / counts[dom], 4),
                           "frac": round(alg[dom] / counts[dom] / 1e9 / (dom_ms_timed / counts[dom] / 1e3) / HBM_PEAK_GBS, 4)},
                    "kernels_ms": {k: round(v[0], 4) for k, v in ktimes.items()}}
        else:
            per_launch_bytes = alg[dom] / counts[dom]
            achieved = per_launch_bytes / 1e9 / (dom_ms_timed / counts[dom] / 1e3)
            roof = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": pmc_traffic(dom, args.layout),
                    "alg_bytes_per_launch": int(per_launch_bytes), "ms_per_launch": round(dom_ms_timed / counts[dom], 4),
                    "timing": "HIP events on the launch stream around the op (kernels + fork/join), mean over the timed steps"}
        gpu_ms = sum(v["ms_per_step"] for v in ops.values())
        out = {
            "metric": "img/s through the Mask R-CNN R50-FPN detection hot path (training ops), 1333x800 bs=2/GPU",
            "value": round(world * w.n_img * args.steps / elapsed, 2), "unit": "img/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_step, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype,
            "data": "synthetic",
            "config": {"workload": "maskrcnn_r50fpn_train_hotpath_bs2_800x1344 (BASELINE configs[1])",
                       "layout": args.layout, "global_batch": world * w.n_img,
                       "ops_per_step": counts,
                       "parallelism": f"dp{world} (images sharded, no data-path collective)"},
            "roofline": roof, "gpu_ms_per_step_sum_of_ops": round(gpu_ms, 4), "ops": ops,
            "ops_note": f"per-op times: separate untimed pass of {bsteps} steps with HIP events around every op",
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(w)
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
