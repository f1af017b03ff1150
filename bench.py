#!/usr/bin/env python3
"""bench.py -- throughput of the Detectron2 detection hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--rois uniform|clustered]
                    [--workload maskrcnn_train|retinanet_100k|dcn_r50|maskrcnn_infer|rrpn_micro]

`--gpus N` with N > 1 and no torchrun environment: bench.py launches itself as N ranks (one per GPU, RCCL) through
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...`; under torchrun it reads
RANK / LOCAL_RANK / WORLD_SIZE.  Rank 0 prints ONE JSON line.

Workloads (all inputs synthetic, resident in HBM before the timed region; SURVEY.md 8(d) shapes):

maskrcnn_train (default; BASELINE.json configs[1] / [2], the configuration the metric is quoted on)
    One step = one pass of the TRAINING hot path of Mask R-CNN R50-FPN over 2 images per GPU (1333x800 -> padded
    800x1344, bf16 FPN features p2..p5, 256 channels), CONNECTED as GeneralizedRCNN.forward connects it, captured once and
    replayed as ONE HIP graph per step with no host read:
      RPN       Matcher.match_boxes(16 GT x 268,569 anchors) per image + anchor sampling [rpn.py:307-364: pairwise_iou + Matcher]
                find_top_rpn_proposals_fused: per-level top-2000 of the objectness logits, decode, clip, per-level
                NMS 0.7 of 2 x 8,819 boxes, top 1000 -- counts left on the device   [rpn.py:468-533, proposal_utils.py:22-135]
      ROI heads label_and_sample_proposals on the NMS's device-side counts (match + sample 512 rows / image at a fixed
                shape, written in pooler format)                                   [roi_heads.py:257-295, sampling.py]
                box  ROIPooler 7x7 on the 512 sampled rows / image, mask ROIPooler 14x14 on their first 128 (one paired
                launch per direction)                                              [roi_heads.py:_forward_box / _forward_mask]
                mask targets: gt_masks[matched].crop_and_resize(fg boxes, 28) on (16, 800, 1344) bitmasks / image
                mask_rcnn_loss forward on (256, 80, 28, 28) logits, masked by the device-side foreground count [mask_head.py:33-112]
      backward  ONE autograd pass: mask_rcnn_loss backward + both poolers' backward into the FPN features
      N > 1     gradient all-reduce of the model's 44.1 M trainable parameters (RCCL over xGMI; the ROI heads'
                bucket is issued before the pooler backward and overlaps it; see detectron2_amd/sharding.py).  The line's
                `allreduce` block then carries `value_data_path` (the same step WITHOUT the collective: the hot path's own
                weak scaling -- compute data-path efficiency from this), `exposed_ms` (what the collective adds to a step; a
                full model hides it behind its backbone backward, which is outside this step) and `bus_GBps_alone`.
    Not in the step because out of the hot path's scope: backbone / head convolutions and FCs (their
    outputs -- logits, deltas, mask logits, dY of the pooled features -- are inputs here), the optimizer.
    `value` = images / second through the hot path, whole job.  `--rois clustered`: proposals drawn around 12 objects per
    image the way a trained RPN's are (also an `extra_workloads` entry of the default line, with the per-tile list
    histogram); `extra_workloads.nchw_drop_in`: the same step entered with NCHW features.
    `roofline` = the paired pooler backward (box + mask pooler in one launch): `frac` on the launch's COMPULSORY bytes (dY
    of both poolers once + every gradient tile written once), `frac_survey_units` on SURVEY 8(d)'s per-unit formula for the
    two units it processes, `frac_traffic` on its PMC bytes.

retinanet_100k (configs[3]): RetinaNet R50-FPN inference with TOPK_CANDIDATES_TEST 20000 x 5 levels = 100k candidates
    per image, SCORE_THRESH_TEST 0, 80 classes: dense_detector_inference_fused (threshold + top-k over 2 x 16.1 M
    class logits, decode, per-class NMS 0.5 of 2 x 100,000 boxes, top 100).  value = images / second.

dcn_r50 (configs[4]): the 13 ModulatedDeformConv (DCNv2) blocks of R50 res3-res5 (4 x 128ch @100x168, 6 x 256ch
    @50x84, 3 x 512ch @25x42), forward + backward (dX, d offset, d mask, dW), bf16, 2 images.  value = images / second;
    roofline bound = MFMA.

maskrcnn_infer, rrpn_micro: bench_extra.py (the inference chain as one HIP graph; the rotated operators).

Extra JSON fields: `roofline` (dominant kernel, algorithmic bytes / flops over its HIP-event time), `cpu_baseline` (the
oracle timed on this host), `ops` (per-op breakdown from a separate untimed pass), `extra_workloads` (the other workloads,
a few steps each, in the default line).
"""
import argparse
import ctypes
import json
import math
import os
import socket
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
MFMA_BF16_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16 MFMA peak
L1_PEAK_GBS = 37700.0      # aggregate vector-L1 rate: 64 B / clk / CU x 256 CUs x 2.3 GHz
STRIDES = (4, 8, 16, 32)
FEAT_HW = ((200, 336), (100, 168), (50, 84), (25, 42))  # 800x1344 padded input
ANCHOR_HW = FEAT_HW + ((13, 21),)
IMG_H, IMG_W = 800, 1344
C = 256
IMAGES_PER_GPU = 2
N_GT = 16
WORKLOADS = ("maskrcnn_train", "retinanet_100k", "dcn_r50", "maskrcnn_infer", "rrpn_micro")


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--workload", choices=WORKLOADS, default="maskrcnn_train")
    ap.add_argument("--layout", choices=["nchw", "nhwc"], default="nhwc",
                    help="feature memory format: nchw (reference default) or nhwc (torch.channels_last)")
    ap.add_argument("--dtype", choices=["bf16", "fp32", "fp16"], default="bf16")
    ap.add_argument("--rois", choices=["uniform", "clustered"], default="uniform",
                    help="maskrcnn_train: what the synthetic RPN head predicts.  uniform (default): random logits / deltas -> "
                         "proposals spread over the image.  clustered: a TRAINED RPN's picture -- anchors score by their IoU "
                         "with the 16 GT boxes and regress onto them with N(0, 0.1) jitter, plus ~30 %% isolated background "
                         "boxes: the 1,000 proposals and the 25 %% positives pile up on 16 objects (VERDICT r04, next 5)")
    ap.add_argument("--grad-allreduce", choices=["bf16", "fp32", "off"], default="bf16",
                    help="N > 1, maskrcnn_train: wire dtype of the gradient all-reduce (bf16 = the reference's "
                         "fp16_compress_hook idea, fp32 = plain DDP) or off")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--disconnected", action="store_true",
                    help="maskrcnn_train: the round-2 step -- the ROI heads pool FIXED sampled lists and the RPN's "
                         "proposals feed nothing; two HIP graphs with one host sync between them.  Default: the "
                         "connected step (RPN -> sampler -> poolers, counts on the device) replayed as ONE graph")
    ap.add_argument("--no-extra-workloads", action="store_true",
                    help="default run (maskrcnn_train, 1 GPU): do not append the short retinanet_100k / dcn_r50 runs")
    ap.add_argument("--no-graph", action="store_true",
                    help="maskrcnn_train: launch every op eagerly in the timed region (default: the two sync-free halves "
                         "of the step are captured once in HIP graphs and replayed)")
    ap.add_argument("--no-overlap", action="store_true",
                    help="maskrcnn_train: issue the independent branches of the step (anchor labelling | proposal path; "
                         "both poolers | proposal labelling + mask targets + loss) on ONE stream instead of forking "
                         "them onto side streams (detectron2_amd/streams.py)")
    ap.add_argument("--plumbing-only", action="store_true",
                    help="test hook: launcher + process group + gradient all-reduce + timing reduction only, no "
                         "hot-path op (runs without a GPU with --backend gloo); the JSON line says so")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl")
    ap.add_argument("--force-dist", action="store_true",
                    help="plumbing check on a 1-GPU box: create the process group and issue the gradient all-reduce even "
                         "at world size 1 (RCCL init, async collectives between the HIP graphs, barrier-bracketed timing)")
    a = ap.parse_args(argv)
    # (the eager workloads need ~20 warm-up steps in a fresh process -- allocator pools, lazily loaded code objects, side
    # streams: with 3-5 the timed region of a stand-alone run came out 15-45 % above the same workload's figure inside the
    # default line, which runs it in a warm process: gpurun_out/r4e1)
    dflt = {"maskrcnn_train": (200, 20), "retinanet_100k": (50, 20), "dcn_r50": (20, 10), "maskrcnn_infer": (30, 20),
            "rrpn_micro": (20, 20)}[a.workload]
    a.steps = dflt[0] if a.steps is None else a.steps
    a.warmup = dflt[1] if a.warmup is None else a.warmup
    return a


# ------------------------------------------------------------------------------------ launcher
def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def relaunch_under_torchrun(args):
    """`bench.py --gpus N` started as a plain process: become N ranks (engine/launch.py:27-84 does this with
    mp.start_processes; torchrun gives the same one-process-per-GPU layout and is what the driver uses)."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: required by RCCL on this host driver
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.call(cmd, env=env)


# ------------------------------------------------------------------------------------ inputs
def level_anchors():
    """Standard RPN anchors (sizes 32..512, ratios .5/1/2, strides 4..64), per level -> 268,569 x 4 in total."""
    out = []
    for size, stride, (h, w) in zip((32, 64, 128, 256, 512), (4, 8, 16, 32, 64), ANCHOR_HW):
        ys, xs = torch.meshgrid(torch.arange(h) * stride, torch.arange(w) * stride, indexing="ij")
        ctr = torch.stack([xs, ys, xs, ys], -1).reshape(-1, 1, 4).float()
        cells = []
        for r in (0.5, 1.0, 2.0):
            ww = math.sqrt(size * size / r)
            hh = ww * r
            cells.append([-ww / 2, -hh / 2, ww / 2, hh / 2])
        out.append((ctr + torch.tensor(cells)[None]).reshape(-1, 4))
    return out


def make_anchors():
    return torch.cat(level_anchors())


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


def blob_bitmasks(gen, boxes):
    """(G, H, W) bool: an ellipse inside every GT box ('Bernoulli-blob bitmasks')."""
    g = boxes.shape[0]
    yy = torch.arange(IMG_H).view(1, -1, 1).float() + 0.5
    xx = torch.arange(IMG_W).view(1, 1, -1).float() + 0.5
    cx, cy = ((boxes[:, 0] + boxes[:, 2]) / 2).view(g, 1, 1), ((boxes[:, 1] + boxes[:, 3]) / 2).view(g, 1, 1)
    rx = ((boxes[:, 2] - boxes[:, 0]) / 2).clamp(min=1).view(g, 1, 1)
    ry = ((boxes[:, 3] - boxes[:, 1]) / 2).clamp(min=1).view(g, 1, 1)
    return ((xx - cx) / rx) ** 2 + ((yy - cy) / ry) ** 2 <= 1.0


class Workload:
    """maskrcnn_train inputs of one rank (also used by scripts/microbench.py)."""

    def __init__(self, dev, dtype, layout, seed=1234, image_ids=(0, 1), full=True, rois="uniform"):
        from detectron2_amd.modeling import Matcher, ROIPooler
        from detectron2_amd.structures import BitMasks, Boxes

        n_img = len(image_ids)
        gens = [image_generator(seed, i) for i in image_ids]
        self.dev, self.n_img, self.image_ids, self.dtype, self.layout = dev, n_img, list(image_ids), dtype, layout
        self.overlap = True  # independent branches of the step on separate HIP streams (--no-overlap: one stream)
        self.feats = []
        for (h, w) in FEAT_HW:
            f = torch.stack([torch.rand(C, h, w, generator=g) * 2 - 1 for g in gens]).to(dtype).to(dev)
            if layout == "nhwc":
                f = f.contiguous(memory_format=torch.channels_last)
            self.feats.append(f.requires_grad_(True))
        self.anchor_levels = [a.to(dev) for a in level_anchors()]
        self.anchors = torch.cat(self.anchor_levels)
        assert self.anchors.shape[0] == 268569
        self.gt = [make_boxes(g, N_GT, 16, 512).to(dev) for g in gens]
        # RPN proposals entering NMS (stand-alone NMS input of SURVEY 8(d) micro (i); scripts/microbench.py)
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
        if not full:
            return
        # RPN head outputs (inputs of the proposal path): objectness logits / anchor deltas per level
        self.rpn_logits = [torch.stack([torch.randn(a.shape[0], generator=g) for g in gens]).to(dev)
                           for a in self.anchor_levels]
        self.rpn_deltas = [torch.stack([torch.randn(a.shape[0], 4, generator=g) * 0.2 for g in gens]).to(dev)
                           for a in self.anchor_levels]
        self.rois_mode = rois
        if rois == "clustered":
            self._clustered_rpn_head(gens)
        self.image_sizes = [(IMG_H, IMG_W)] * n_img
        self.anchor_matcher = Matcher([0.3, 0.7], [0, -1, 1], allow_low_quality_matches=True)   # rpn.py / defaults
        self.proposal_matcher = Matcher([0.5], [0, 1], allow_low_quality_matches=False)          # roi_heads.py
        self.props_with_gt = [torch.cat([p, g]) for p, g in zip(self.props, self.gt)]           # add_ground_truth_to_proposals
        # mask head: GT bitmasks, the matched GT of every fg ROI, GT classes, the mask head's logits
        self.gt_masks = [BitMasks(blob_bitmasks(g, b.cpu()).to(dev)) for g, b in zip(gens, self.gt)]
        self.fg_gt_index = [torch.randint(0, N_GT, (128,), generator=g).to(dev) for g in gens]
        self.fg_classes = torch.cat([torch.randint(0, 80, (128,), generator=g) for g in gens]).to(dev)
        self.mask_logits = torch.cat([torch.randn(128, 80, 28, 28, generator=g) for g in gens]).to(dtype).to(dev) \
            .requires_grad_(True)
        self.crop_status = torch.zeros(1, dtype=torch.int32, device=dev)
        # connected step: the ROI heads' inputs come from the RPN of the same step (label_and_sample_proposals_fixed):
        # 512 rows per image (box head), of which the first 128 feed the mask head (positives come first)
        self.gt_classes = [torch.randint(0, 80, (N_GT,), generator=g).to(dev) for g in gens]
        self.connected = False
        from detectron2_amd.modeling import DeviceKeyGenerator

        self.keygen = DeviceKeyGenerator(dev, seed=seed)
        self.loss_grad = torch.ones((), device=dev)  # d(total loss) / d(mask loss): passed in, not filled per step

    def _clustered_rpn_head(self, gens):
        """RPN head outputs of a TRAINED model (--rois clustered): objectness grows with the anchor's best IoU over the
        image's GT boxes, the deltas regress the anchor onto that GT box (Box2BoxTransform.get_deltas, weights 1) with
        N(0, 0.1) jitter -- proposals = GT jittered by ~0.1 x size -- and 0.2 % of the anchors are isolated false
        positives (~30 % of the 1,000 proposals that survive the NMS)."""
        dev = self.dev
        for i, g in enumerate(gens):
            gt = self.gt[i].cpu()
            for l, a in enumerate(self.anchor_levels):
                a = a.cpu()
                lt = torch.max(a[:, None, :2], gt[None, :, :2])
                rb = torch.min(a[:, None, 2:], gt[None, :, 2:])
                inter = (rb - lt).clamp(min=0).prod(dim=2)
                area_a = (a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1])
                area_g = (gt[:, 2] - gt[:, 0]) * (gt[:, 3] - gt[:, 1])
                iou = inter / (area_a[:, None] + area_g[None] - inter)
                best, arg = iou.max(dim=1)
                t = gt[arg]
                aw, ah = a[:, 2] - a[:, 0], a[:, 3] - a[:, 1]
                acx, acy = a[:, 0] + 0.5 * aw, a[:, 1] + 0.5 * ah
                tw, th = (t[:, 2] - t[:, 0]).clamp(min=1), (t[:, 3] - t[:, 1]).clamp(min=1)
                tcx, tcy = t[:, 0] + 0.5 * tw, t[:, 1] + 0.5 * th
                d = torch.stack([(tcx - acx) / aw, (tcy - acy) / ah, torch.log(tw / aw), torch.log(th / ah)], 1)
                near = best > 0.2
                d = torch.where(near[:, None], d, torch.zeros_like(d)) + torch.randn(a.shape[0], 4, generator=g) * 0.1
                logit = 8.0 * best + 0.5 * torch.randn(a.shape[0], generator=g) - 4.0
                fp = torch.rand(a.shape[0], generator=g) < 0.002
                logit = torch.where(fp, 2.0 + torch.rand(a.shape[0], generator=g) * 3.0, logit)
                self.rpn_logits[l][i] = logit.to(dev)
                self.rpn_deltas[l][i] = d.to(dev)

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
        d["roi_align_pair_fwd"] = d["roi_align_box_fwd"] + d["roi_align_mask_fwd"]  # pool_pair: both units, one launch
        # compulsory bytes of the FUSED backward launch: both poolers' dY read once, every level's
        # dX written once -- no zero fill, no second write, no read-back (the 8(d) formula charges those per pooler)
        lv = [l for l in range(4) if self.box_level_counts[0][l] or self.box_level_counts[1][l]]
        d["roi_align_pair_bwd_compulsory"] = (sum(s * k * C * 49 for k in self.box_level_counts[0]) +
                                              sum(s * k * C * 196 for k in self.box_level_counts[1]) + sum(feat[l] for l in lv))
        d["roi_align_box_bwd_compulsory"] = (sum(s * k * C * 49 for k in self.box_level_counts[0]) +
                                             sum(feat[l] for l in range(4) if self.box_level_counts[0][l]))
        d["backward"] = d["roi_align_box_bwd"] + d["roi_align_mask_bwd"]  # + the mask loss backward (2 x logits)
        d["backward"] += 2 * 256 * 80 * 784 * s
        n, m = N_GT, 268569
        d["match_anchors"] = self.n_img * (16 * (n + m) + 9 * m)       # boxes in, (int64 match, int8 label) out
        d["match_proposals"] = self.n_img * (16 * (n + 1032) + 9 * 1032)
        d["rpn_proposals"] = self.n_img * m * (4 + 4)                    # logits read by the select passes (>= 2x)
        d["mask_loss_fwd"] = 256 * 784 * (s + 1)
        d["label_and_sample_anchors"] = d["match_anchors"] + self.n_img * m * (1 + 4 + 1)  # + labels, keys in, labels out
        d["label_and_sample_proposals"] = self.n_img * (16 * (n + 1000) + 4 * (n + 1000) + 512 * 44)
        return d


class Timer:
    """Per-op HIP-event timing on torch's current stream (the stream every kernel is launched on).
    `only`: time just these ops and call the others bare (two events cost ~10 us of host time)."""

    def __init__(self, only=None):
        self.pairs = {}
        self.only = only

    def run(self, name, fn):
        if self.only is not None and name not in self.only:
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


def read_kernel_times(names):
    """{kernel: (mean ms, launches)} from the library's launch-stream events (d2amd_timing_*)."""
    from detectron2_amd import _C as _dc

    out = {}
    for kn in names:
        tot, cnt = ctypes.c_double(0.0), ctypes.c_int(0)
        _dc.check(_dc.lib().d2amd_timing_read(kn.encode(), ctypes.byref(tot), ctypes.byref(cnt)))
        if cnt.value:
            out[kn] = (tot.value / cnt.value, cnt.value)
    return out


# ------------------------------------------------------------------------------------ maskrcnn_train
def roi_branches(w):
    """The forward of the ROI heads as two branches (fork_join): both poolers | proposal labelling + mask targets +
    mask loss.  (Measured layouts, captured graph B incl. the backward, scripts/graph_split2.py: one stream 304-314 us,
    four branches 322-333 -- the poolers and the target rasteriser each fill the chip and only slow each other down --
    this one 278-280; with the poolers' branch captured FIRST -- fork_join(current_first=True): a graph's nodes reach the
    device in capture order -- 288 instead of 296 on a slower box.)  -> ((box features, mask features), (loss, stats))"""
    from detectron2_amd.modeling import mask_rcnn_loss_from_targets
    from detectron2_amd.structures import crop_and_resize_batch

    def poolers():
        # both poolers in one launch per direction (pool_pair: the forward values are the separate calls' bit for bit);
        # D2AMD_BENCH_POOL_SEPARATE=1 (the A/B): one call per pooler, the reference's structure
        if os.environ.get("D2AMD_BENCH_POOL_SEPARATE") == "1":
            return w.box_pooler(w.feats, w.box_lists), w.mask_pooler(w.feats, w.mask_lists)
        from detectron2_amd.modeling import pool_pair

        return pool_pair(w.box_pooler, w.mask_pooler, w.feats, w.box_lists, w.mask_lists)

    def labels_and_loss():
        _lab = [w.proposal_matcher.match_boxes(w.gt[i], w.props_with_gt[i]) for i in range(w.n_img)]
        tg = crop_and_resize_batch(w.gt_masks, [b.tensor for b in w.mask_lists], 28, w.fg_gt_index, w.crop_status)
        return mask_rcnn_loss_from_targets(w.mask_logits, w.fg_classes, tg)

    return poolers, labels_and_loss


def rpn_branches(w):
    """Anchor labelling | proposal path.  The labelling goes FIRST (= on the current stream): 180 us for the captured
    half against 203 us the other way round and 207-210 us on one stream (scripts/graph_split2.py; the proposal path
    alone is 156 us of 10-workgroup kernels, the labelling 60 us).  With the ranked selection that is 159.5 us; forking
    after the selection instead (find_top_rpn_proposals_fused: beside_nms, what the step does) 156.5 us."""
    from detectron2_amd.modeling import find_top_rpn_proposals_fused

    return (lambda: [w.anchor_matcher.match_boxes(w.gt[i], w.anchors) for i in range(w.n_img)],
            lambda: find_top_rpn_proposals_fused(w.anchor_levels, w.rpn_logits, w.rpn_deltas, w.image_sizes, 0.7, 2000,
                                                 1000, 0.0, True, defer=True))


def disconnected_step(w, t=None, grads=None):
    """The round-2 step (--disconnected): the ROI half pools FIXED lists, the RPN's proposals feed nothing."""
    from detectron2_amd.modeling import find_top_rpn_proposals_fused, mask_rcnn_loss_from_targets
    from detectron2_amd.modeling import poolers as _poolers
    from detectron2_amd.streams import fork_join
    from detectron2_amd.structures import crop_and_resize_batch

    run = (lambda name, fn: t.run(name, fn)) if t is not None else (lambda name, fn: fn())
    # a training iteration produces NEW feature maps: drop the NHWC staging copies an NCHW run cached for the
    # previous step's tensors (the two poolers of one step still share one copy)
    _poolers._NHWC_CACHE.clear()
    # RPN: the proposal path is enqueued first (selection + decode + NMS of both images), its one host sync is
    # deferred past the anchor labelling, which does not depend on the proposals (RPN.forward computes the two in either
    # order: rpn.py label_and_sample_anchors / predict_proposals): the device works through both while the host waits
    if t is None and w.overlap:  # independent branches on separate streams (detectron2_amd/streams.py)
        rpn_done = find_top_rpn_proposals_fused(w.anchor_levels, w.rpn_logits, w.rpn_deltas, w.image_sizes, 0.7, 2000,
                                                1000, 0.0, True, defer=True, beside_nms=rpn_branches(w)[0])
    else:
        rpn_done = run("rpn_proposals", lambda: find_top_rpn_proposals_fused(
            w.anchor_levels, w.rpn_logits, w.rpn_deltas, w.image_sizes, 0.7, 2000, 1000, 0.0, True, defer=True))
        for i in range(w.n_img):
            run("match_anchors", lambda: w.anchor_matcher.match_boxes(w.gt[i], w.anchors))
    props = run("rpn_proposals_sync", rpn_done)
    # ROI heads: proposal labelling (the sampled lists themselves are fixed inputs: subsample_labels is out of scope)
    if t is None and w.overlap:
        (yb, ym), (loss, _stats) = fork_join(*roi_branches(w), current_first=True)
    else:
        for i in range(w.n_img):
            run("match_proposals", lambda: w.proposal_matcher.match_boxes(w.gt[i], w.props_with_gt[i]))
        if os.environ.get("D2AMD_BENCH_POOL_SEPARATE") == "1":
            yb = run("roi_align_box_fwd", lambda: w.box_pooler(w.feats, w.box_lists))
            ym = run("roi_align_mask_fwd", lambda: w.mask_pooler(w.feats, w.mask_lists))
        else:
            from detectron2_amd.modeling import pool_pair

            yb, ym = run("roi_align_pair_fwd", lambda: pool_pair(w.box_pooler, w.mask_pooler, w.feats, w.box_lists,
                                                                 w.mask_lists))
        tg = run("mask_targets", lambda: crop_and_resize_batch(
            w.gt_masks, [b.tensor for b in w.mask_lists], 28, w.fg_gt_index, w.crop_status))
        loss, _stats = run("mask_loss_fwd", lambda: mask_rcnn_loss_from_targets(w.mask_logits, w.fg_classes, tg))
    # N > 1: the ROI heads' weight gradients exist before the poolers' backward runs -> their bucket's all-reduce
    # overlaps it; the remaining buckets (RPN head, FPN, backbone) follow the feature gradients
    n_early = grads.ready_after("roi_heads.box_head") if grads is not None else 0
    if grads is not None:
        run("allreduce_issue", lambda: [grads.reduce(i) for i in range(n_early)])
    # ONE backward pass (a training iteration sums the losses and calls backward once)
    run("backward", lambda: torch.autograd.backward([yb, ym, loss], [w.gbox, w.gmask, None]))
    if grads is not None:
        run("allreduce_issue", lambda: [grads.reduce(i) for i in range(n_early, grads.num_buckets)])
        run("allreduce_wait", grads.finish)
    for f in w.feats:
        f.grad = None
    w.mask_logits.grad = None
    return props


ROI_BATCH, ROI_POS_FRACTION, RPN_BATCH, RPN_POS_FRACTION, MASK_ROWS = 512, 0.25, 256, 0.5, 128


def anchor_labels(w, keys=None):
    """RPN.label_and_sample_anchors (rpn.py:307-364) for the batch: fused IoU + Matcher of every image's ground truth
    against the 268,569 anchors in ONE launch per pass, then _subsample_labels (256 per image, in place) -- all on the
    device.  -> (labels [N, A] int8 in {-1, 0, 1}, matched ground-truth index [N, A], counts [N, 2])"""
    from detectron2_amd.modeling import subsample_anchor_labels_

    ab = os.environ.get("D2AMD_BENCH_ABLATE", "")  # (sensitivity experiments only: the step then computes LESS)
    if "anchors" in ab:
        return None, None, None
    matches, labels = w.anchor_matcher.match_boxes_batch(w.gt, w.anchors)
    if "subsample" in ab:
        return labels, matches, None
    labels, counts = subsample_anchor_labels_(labels, RPN_BATCH, RPN_POS_FRACTION, keys=keys)
    return labels, matches, counts


def step_keys(w):
    """Every random key of a step in ONE launch: [N, 268,569] for the anchor sampler, [N, 1,000 + G] for the proposal
    sampler (torch's generator: inside a captured graph the Philox offset advances per replay)."""
    na = w.anchors.shape[0]
    k = w.keygen.uniform(w.n_img * (na + 1000 + N_GT))  # (device-resident generator: nothing in front of a graph replay)
    roi = k[w.n_img * na:].view(w.n_img, 1000 + N_GT)
    return k[:w.n_img * na].view(w.n_img, na), [roi[i] for i in range(w.n_img)]


def connected_forward(w, run=None, rpn_keys=None, roi_keys=None, sync=False):
    """The training hot path with the ROI heads fed by the RPN of the same step (GeneralizedRCNN.forward: rpn.py:431-480
    -> roi_heads.py:220-295 -> poolers.py:206): RPN selection + NMS (anchor labelling + sampling beside the NMS) ->
    label_and_sample_proposals on the NMS's device-side counts -> box pooler on the 512 sampled rows per image, mask
    pooler + mask targets + masked mask loss on their first 128 rows.  Nothing waits for the host.
    sync=True (tests / A-B): the reference's data flow instead -- ONE host sync after the NMS, exact-size proposal
    lists into the sampler -- which must give the same bits for the same keys."""
    from detectron2_amd.modeling import (find_top_rpn_proposals_fused, label_and_sample_proposals_fixed,
                                         mask_rcnn_loss_from_targets)
    from detectron2_amd.streams import fork_join
    from detectron2_amd.structures import Boxes, crop_and_resize_batch

    bare = run is None  # no per-op events: the independent branches may go to side streams
    run = run or (lambda name, fn: fn())
    n = w.n_img
    keys_ready = None
    if w.overlap and bare:
        # side branch beside the NMS: the step's keys first (an event tells the main path when), then anchor labelling +
        # sampling; joined at the END of the forward -- it outlasts the NMS by ~25 us and nothing needs it before.
        # (The proposal sampler can draw its keys inside its kernel -- label_and_sample_proposals_fixed(keygen=...) --
        # and the main path then waits for nothing from this branch.  Measured: 0.54-0.56 ms against 0.40-0.41.  Without
        # that edge the replayed graph runs this branch BEHIND the targets / loss branch on the same queue, at the end
        # of the forward, where the join waits for all ~105 us of it: the wait for the keys is what pins it here.)
        def side():
            nonlocal rpn_keys, roi_keys, keys_ready
            if roi_keys is None or rpn_keys is None:
                rk, ok = step_keys(w)
                rpn_keys = rk if rpn_keys is None else rpn_keys
                roi_keys = ok if roi_keys is None else roi_keys
                keys_ready = torch.cuda.Event()
                keys_ready.record()
            return anchor_labels(w, rpn_keys)

        # (forking this branch at the very START of the step instead was measured -- D2AMD_BENCH_FORK experiments,
        # gpurun_out/r3z*: 0.465-0.48 against 0.458 ms; the branch the graph does not launch on starts ~10 us late, and
        # the matcher slows the selection's latency-bound chain; r06 at 0.325 ms, same box: 0.3277-0.3281 against 0.3229-0.3266)
        done = find_top_rpn_proposals_fused(w.anchor_levels, w.rpn_logits, w.rpn_deltas, w.image_sizes, 0.7, 2000, 1000,
                                            0.0, True, defer=True, beside_nms=side, join_beside=False,
                                            host_result=sync)
        anchors_out = done.beside
        if keys_ready is not None:
            torch.cuda.current_stream().wait_event(keys_ready)
    else:
        if roi_keys is None or rpn_keys is None:
            rk, ok = step_keys(w)
            rpn_keys, roi_keys = (rk if rpn_keys is None else rpn_keys), (ok if roi_keys is None else roi_keys)
        done = run("rpn_proposals", lambda: find_top_rpn_proposals_fused(
            w.anchor_levels, w.rpn_logits, w.rpn_deltas, w.image_sizes, 0.7, 2000, 1000, 0.0, True, defer=True))
        anchors_out = run("label_and_sample_anchors", lambda: anchor_labels(w, rpn_keys))
    if sync:
        props = done()  # the host sync of the reference's data flow
        pb = [p.proposal_boxes.tensor for p in props]
        keys = [torch.cat([roi_keys[i][:pb[i].shape[0]], roi_keys[i][1000:]]) for i in range(n)]
        samp = label_and_sample_proposals_fixed(pb, w.gt, w.gt_classes, keys=keys, batch_size_per_image=ROI_BATCH,
                                                positive_fraction=ROI_POS_FRACTION, num_classes=80, head_rows=MASK_ROWS)
    else:
        dp = done.device
        samp = run("label_and_sample_proposals", lambda: label_and_sample_proposals_fixed(
            dp.boxes, w.gt, w.gt_classes, limits=dp.limits, limit_stride=2, keys=roi_keys,
            batch_size_per_image=ROI_BATCH, positive_fraction=ROI_POS_FRACTION, num_classes=80, head_rows=MASK_ROWS))
    mask_boxes = [samp["boxes"][i, :MASK_ROWS] for i in range(n)]

    def poolers():  # (the sampler wrote its rows in pooler format: no conversion launch in front of either pooler)
        # both poolers in one launch per direction (pool_pair_rois: the forward values are the separate calls' bit for
        # bit); D2AMD_BENCH_POOL_SEPARATE=1 (the A/B): one call per pooler, the reference's structure
        if os.environ.get("D2AMD_BENCH_POOL_SEPARATE") == "1":
            return (run("roi_align_box_fwd", lambda: w.box_pooler.pool_rois(w.feats, samp["rois"])),
                    run("roi_align_mask_fwd", lambda: w.mask_pooler.pool_rois(w.feats, samp["head_rois"])))
        from detectron2_amd.modeling import pool_pair_rois

        return run("roi_align_pair_fwd", lambda: pool_pair_rois(w.box_pooler, w.mask_pooler, w.feats, samp["rois"],
                                                                samp["head_rois"], plan=plan))

    # The binning of the poolers' backward depends on the sampled ROIs alone and CAN be issued here, on the targets / loss
    # branch beside the poolers' forward (PairBackwardPlan): D2AMD_BENCH_PREBIN=1.  Measured at 0.358 ms (same box): 0.3621 /
    # 0.362 with it against 0.3583 / 0.3574 -- both branches already end together, and the backward waits for the longer.
    # r06 (step 0.324 ms, same box, scripts/r06_prebin_ab.sh): on the targets branch (=1) 0.3339-0.3375; on a THIRD branch
    # forked beside the two (=2) 0.3278-0.3286; off 0.3226-0.3248 -- the third branch's two graph edges and the phase
    # split's zero-fill launch cost more than the 25 us of binning they take out of the backward.
    plan = None
    prebin = os.environ.get("D2AMD_BENCH_PREBIN", "0")
    if w.overlap and bare and prebin in ("1", "2") and os.environ.get("D2AMD_BENCH_POOL_SEPARATE") != "1":
        from detectron2_amd.modeling import PairBackwardPlan

        plan = PairBackwardPlan()

    def targets_and_loss():
        if plan is not None and prebin == "1":
            plan.prepare(w.box_pooler, w.mask_pooler, w.feats, samp["rois"], samp["head_rois"])
        idx = [samp["gt_index"][i, :MASK_ROWS].contiguous() for i in range(n)]
        cls = samp["head_classes"].reshape(-1)  # (contiguous: written by the sampler, no copy launch on this branch)
        if "crop" in os.environ.get("D2AMD_BENCH_ABLATE", ""):  # (sensitivity experiment: constant targets)
            tg = w._abl_tg if hasattr(w, "_abl_tg") else crop_and_resize_batch(w.gt_masks, mask_boxes, 28, idx, w.crop_status)
            w._abl_tg = tg
        else:
            tg = run("mask_targets", lambda: crop_and_resize_batch(w.gt_masks, mask_boxes, 28, idx, w.crop_status))
        # background / padding rows among the 128 do not count (class 80 / -1): masked loss, row count on the device
        return run("mask_loss_fwd", lambda: mask_rcnn_loss_from_targets(w.mask_logits, cls, tg,
                                                                        ignore_invalid_rows=True))

    # (A/B at 0.355 ms, same box: the ROI half WITHOUT this fork 0.3604 / 0.3607 against 0.3573 / 0.3598 -- its two edges
    # cost ~14 us and the branches slow each other down, for 38 us of overlapped work; targets + loss behind the anchor
    # labelling on that branch's stream instead 0.3553 / 0.3574 against 0.3536 / 0.3534: profiles/r04/LOG.md)
    if w.overlap and bare and plan is not None and prebin == "2":  # the binning on a THIRD branch
        (yb, ym), (loss, stats), _ = fork_join(
            poolers, targets_and_loss,
            lambda: plan.prepare(w.box_pooler, w.mask_pooler, w.feats, samp["rois"], samp["head_rois"]), current_first=True)
        done.join_beside()
    elif w.overlap and bare:
        (yb, ym), (loss, stats) = fork_join(poolers, targets_and_loss, current_first=True)
        done.join_beside()  # the anchor labels are part of the forward's result
    else:
        (yb, ym), (loss, stats) = poolers(), targets_and_loss()
    return {"anchors": anchors_out, "sample": samp, "box_features": yb, "mask_features": ym, "loss": loss,
            "stats": stats, "done": done}


def roi_tile_histogram(w, out):
    """Per-tile list lengths of the pooler backward for the ROIs this step sampled (8 x 8-pixel tiles of each FPN level;
    an ROI is on the list of every tile its box, grown by the bilinear footprint of one pixel, overlaps): what the tile
    gather's split planner and heavy-first queues see.  Host-side arithmetic on the sampled rows, outside any timing."""
    samp = out["sample"]
    rois = samp["rois"].reshape(-1, 5).float().cpu()          # box head: 512 rows per image
    head = samp["head_rois"].reshape(-1, 5).float().cpu()     # mask head: the first 128 rows per image
    edges = [0, 1, 3, 9, 17, 41, 10 ** 9]
    names = ["0", "1-2", "3-8", "9-16", "17-40", "> 40"]
    hist = dict.fromkeys(names, 0)
    longest, tiles_nonempty, entries = 0, 0, 0
    for lv, (h, wd) in enumerate(FEAT_HW):
        ty, tx = (h + 7) // 8, (wd + 7) // 8
        cnt = torch.zeros(w.n_img, ty + 1, tx + 1, dtype=torch.int64)
        for r in (rois, head):
            b = r[:, 1:]
            keep = (assign_levels(b) == lv) & ((b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1]) > 0)
            if not bool(keep.any()):
                continue
            img = r[keep, 0].long()
            bb = b[keep] / STRIDES[lv] - 0.5
            x0 = ((bb[:, 0] - 1).floor().clamp(0, wd - 1) / 8).long()
            x1 = ((bb[:, 2] + 1).ceil().clamp(0, wd - 1) / 8).long()
            y0 = ((bb[:, 1] - 1).floor().clamp(0, h - 1) / 8).long()
            y1 = ((bb[:, 3] + 1).ceil().clamp(0, h - 1) / 8).long()
            one = torch.ones_like(img)
            for (yy, xx, sgn) in ((y0, x0, 1), (y0, x1 + 1, -1), (y1 + 1, x0, -1), (y1 + 1, x1 + 1, 1)):
                cnt.index_put_((img, yy, xx), sgn * one, accumulate=True)
        c = cnt.cumsum(1).cumsum(2)[:, :ty, :tx]
        for k in range(6):
            hist[names[k]] += int(((c >= edges[k]) & (c < edges[k + 1])).sum())
        longest = max(longest, int(c.max()))
        tiles_nonempty += int((c > 0).sum())
        entries += int(c.sum())
    return {"tiles_by_list_length": hist, "longest_list": longest, "tiles_with_rois": tiles_nonempty,
            "list_entries": entries, "rois": int(rois.shape[0] + head.shape[0])}


def connected_step(w, t=None, grads=None):
    """One eager pass of the connected step (per-op events when `t`), forward + ONE backward."""
    from detectron2_amd.modeling import poolers as _poolers

    _poolers._NHWC_CACHE.clear()
    run = (lambda name, fn: fn()) if t is None else (lambda name, fn: t.run(name, fn))
    out = connected_forward(w, None if t is None else run)
    n_early = grads.ready_after("roi_heads.box_head") if grads is not None else 0
    if grads is not None:
        run("allreduce_issue", lambda: [grads.reduce(i) for i in range(n_early)])
    run("backward", lambda: torch.autograd.backward([out["box_features"], out["mask_features"], out["loss"]],
                                                    [w.gbox, w.gmask, w.loss_grad]))
    if grads is not None:
        run("allreduce_issue", lambda: [grads.reduce(i) for i in range(n_early, grads.num_buckets)])
        run("allreduce_wait", grads.finish)
    for f in w.feats:
        f.grad = None
    w.mask_logits.grad = None
    return out


class GraphedConnectedStep:
    """The connected step as ONE HIP graph: RPN half, samplers, ROI-head half and the backward in one hipGraphLaunch,
    no host read anywhere in it.  N > 1: the graph is cut in front of the backward so that the ROI heads' gradient
    bucket can be handed to RCCL there (two launches, still no host sync)."""

    def __init__(self, w, grads):
        self.w, self.grads = w, grads
        self.split = grads is not None

        def fwd():
            self.out = connected_forward(w)
            return self.out["loss"].detach()

        def bwd():
            for f in w.feats:
                f.grad = None
            w.mask_logits.grad = None
            torch.autograd.backward([self.out["box_features"], self.out["mask_features"], self.out["loss"]],
                                    [w.gbox, w.gmask, w.loss_grad])

        def whole():
            loss = fwd()
            bwd()
            self.out = None  # no reference into the captured autograd graph survives the capture
            return loss

        if self.split:
            self.gf, self.loss = GraphedStep._capture(fwd, backward=bwd)
            self.gb = self.gf.second
        else:
            self.g, self.loss = GraphedStep._capture(whole)

    def __call__(self):
        g = self.grads
        if not self.split:
            self.g.replay()
            return self.loss
        self.gf.replay()
        n_early = g.ready_after("roi_heads.box_head")
        for i in range(n_early):
            g.reduce(i)
        self.gb.replay()
        for i in range(n_early, g.num_buckets):
            g.reduce(i)
        g.finish()
        return self.loss


class GraphedStep:
    """The same step with its two sync-free halves captured in HIP graphs (torch.cuda.CUDAGraph drives
    hipStreamBeginCapture / hipGraphLaunch; every kernel of libd2amd.so is launched on torch's current stream, so the
    capture sees them all):
      graph A  RPN selection + decode + NMS of the batch, then the anchor labelling        -> ONE host sync (counts)
      graph B  proposal labelling, both poolers forward, mask targets, mask loss, ONE backward pass
    The step is launch-bound when issued eagerly (~40 kernels of 5-80 us through Python): replaying removes the host
    from the critical path.  Inputs live at fixed addresses (a training loop would copy its batch into them); the
    proposal lists are rebuilt from graph A's buffers after the sync, like the eager path does."""

    def __init__(self, w, grads):
        from detectron2_amd.modeling import find_top_rpn_proposals_fused, mask_rcnn_loss_from_targets
        from detectron2_amd.structures import crop_and_resize_batch

        self.w, self.grads = w, grads

        from detectron2_amd.streams import fork_join

        def part_a():
            lab, rpn = rpn_branches(w)
            if w.overlap:  # labelling beside the NMS only: 156.5 us, forked at the start 159.5 us
                done = find_top_rpn_proposals_fused(w.anchor_levels, w.rpn_logits, w.rpn_deltas, w.image_sizes, 0.7,
                                                    2000, 1000, 0.0, True, defer=True, beside_nms=lab)
                labels = done.beside
            else:
                done, labels = rpn(), lab()
            return done, labels

        def part_b():
            lab = None
            if w.overlap:
                (yb, ym), (loss, _) = fork_join(*roi_branches(w), current_first=True)
            else:
                lab = [w.proposal_matcher.match_boxes(w.gt[i], w.props_with_gt[i]) for i in range(w.n_img)]
                yb = w.box_pooler(w.feats, w.box_lists)
                ym = w.mask_pooler(w.feats, w.mask_lists)
                tg = crop_and_resize_batch(w.gt_masks, [b.tensor for b in w.mask_lists], 28, w.fg_gt_index,
                                           w.crop_status)
                loss, _ = mask_rcnn_loss_from_targets(w.mask_logits, w.fg_classes, tg)
            for f in w.feats:
                f.grad = None
            w.mask_logits.grad = None
            torch.autograd.backward([yb, ym, loss], [w.gbox, w.gmask, None])
            return lab, loss.detach()  # no reference into the captured autograd graph survives the capture

        self.ga, self.out_a = self._capture(part_a)
        self.gb, self.out_b = self._capture(part_b)

    @staticmethod
    def _capture(fn, backward=None):
        """backward: a second callable captured as its own graph right behind `fn`'s, in the same memory pool (it
        consumes the autograd graph `fn` built); returned as `.second` of the first graph."""
        import gc

        gc.collect()  # no autograd graph of an eager step may survive into the capture (its AccumulateGrad nodes are
        # bound to the default stream and would pull the capture onto it)
        cur = torch.cuda.current_stream()
        side = torch.cuda.Stream()
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            for _ in range(3):
                fn()
                if backward is not None:
                    backward()
        cur.wait_stream(side)
        torch.cuda.synchronize()
        # thread_local: with a process group alive, RCCL's watchdog thread polls its events (hipEventQuery) whenever
        # it likes; under the default global capture mode that call fails with "operation not permitted when stream is
        # capturing" and takes the process down (seen once in ~10 runs at world size 1)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            out = fn()
        if backward is not None:
            g.second = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g.second, pool=g.pool(), capture_error_mode="thread_local"):
                backward()
        return g, out

    def __call__(self):
        self.ga.replay()
        props = self.out_a[0]()  # the one host sync of the step
        g = self.grads
        n_early = g.ready_after("roi_heads.box_head") if g is not None else 0
        for i in range(n_early):
            g.reduce(i)  # RCCL, on its own stream: overlaps graph B's pooler backward
        self.gb.replay()
        if g is not None:
            for i in range(n_early, g.num_buckets):
                g.reduce(i)
            g.finish()
        return props


def _threads():
    return max(1, min(32, os.cpu_count() or 1))


def _median_time(fn, runs=5):
    fn()  # warm-up
    ts = []
    for _ in range(runs):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return float(np.median(ts))


def cpu_baseline_maskrcnn(w):
    """The oracle (oracle/d2_oracle.c + numpy restatements, kind "port") on the SAME step, unsampled: image 0's
    per-image ops in full (x n_img), both poolers forward + backward over all ROIs and all 256 channels (channel
    slabs on a thread pool: ctypes releases the GIL; torchvision's CPU roi_align is parallel over ROIs, the
    reference's own CPU kernels are single-threaded), mask targets and mask loss of the batch.  Median of 5 runs
    after one warm-up."""
    from concurrent.futures import ThreadPoolExecutor

    import oracle
    from oracle import mask_head as omh
    from oracle import rpn as orpn

    T = _threads()
    gt, an = w.gt[0].cpu().numpy(), w.anchors.cpu().numpy()
    anchors_l = [a.cpu().numpy() for a in w.anchor_levels]
    lg = [x[:1].cpu().numpy() for x in w.rpn_logits]
    dl = [x[:1].cpu().numpy() for x in w.rpn_deltas]
    pg = w.props_with_gt[0].cpu().numpy()

    from oracle import sampling as osp

    krng = np.random.default_rng(0)
    akeys, pkeys = krng.random(an.shape[0], dtype=np.float32), krng.random(pg.shape[0], dtype=np.float32)
    gcls = w.gt_classes[0].cpu().numpy()

    def per_image():
        q = oracle.pairwise_iou(gt, an)
        _idx, lab = oracle.matcher(q, [0.3, 0.7], [0, -1, 1], True)
        if w.connected:
            osp.subsample_anchor_labels(lab, akeys, RPN_BATCH, RPN_POS_FRACTION)
        orpn.find_top_rpn_proposals(anchors_l, lg, dl, [(IMG_H, IMG_W)], 0.7, 2000, 1000, 0.0)
        if w.connected:  # pairwise_iou + Matcher + subsample_labels of the proposals (+ appended ground truth)
            osp.label_and_sample_fixed(pg[:-N_GT], len(pg) - N_GT, gt, gcls, pkeys, [0.5], [0, 1], ROI_BATCH,
                                       ROI_POS_FRACTION, 80)
        else:
            oracle.matcher(oracle.pairwise_iou(gt, pg), [0.5], [0, 1], False)

    t_img = _median_time(per_image)
    jobs = []
    for lists, R, gout in ((w.box_lists, 7, w.gbox), (w.mask_lists, 14, w.gmask)):
        allb = torch.cat([b.tensor.cpu() for b in lists])
        bidx = torch.cat([torch.full((len(b),), float(i)) for i, b in enumerate(lists)])
        rois_all = torch.cat([bidx[:, None], allb], 1)
        lv = assign_levels(allb)
        g_all = gout.detach().float().cpu()
        for l in range(4):
            sel = lv == l
            if int(sel.sum()) == 0:
                continue
            r = rois_all[sel].contiguous().numpy()
            x = w.feats[l].detach().float().cpu().contiguous()
            g = g_all[sel].contiguous()
            cs = C // T if C % T == 0 else C
            for c0 in range(0, C, cs):
                jobs.append((x[:, c0:c0 + cs].contiguous().numpy(), r, g[:, c0:c0 + cs].contiguous().numpy(), R,
                             1.0 / STRIDES[l]))

    def one(job):
        x, r, g, R, sc = job
        oracle.roi_align_forward(x, r, (R, R), sc, 0, True)
        oracle.roi_align_backward(g, r, x.shape, sc, 0, True)

    with ThreadPoolExecutor(T) as ex:
        t_pool = _median_time(lambda: list(ex.map(one, jobs)))
    masks = [m.tensor.cpu().numpy().astype(np.float32)[:, None] for m in w.gt_masks]
    idx = [i.cpu().numpy() for i in w.fg_gt_index]
    mb = [b.tensor.cpu().numpy() for b in w.mask_lists]
    logits = w.mask_logits.detach().float().cpu().numpy()
    cls = w.fg_classes.cpu().numpy()

    def mask_part():
        tg = []
        for m, i, b in zip(masks, idx, mb):
            rois = np.concatenate([i.astype(np.float32)[:, None], b], 1)
            tg.append(oracle.roi_align_forward(m, rois, (28, 28), 1.0, 0, True)[:, 0] >= 0.5)
        tg = np.concatenate(tg)
        omh.mask_rcnn_loss(logits, cls, tg)
        omh.mask_rcnn_loss_grad(logits, cls, tg)

    t_mask = _median_time(mask_part, runs=3)
    sec = t_img * w.n_img + t_pool + t_mask
    try:
        refpy = cpu_reference_python(w)
    except Exception as e:  # (a reported baseline must never take the line down)
        refpy = {"error": repr(e)}
    return {"value": round(w.n_img / sec, 4), "unit": "img/s", "cores": T, "kind": "port", "reference_python": refpy,
            "sample": f"the full step, unsampled, median of 5 runs: per-image ops (IoU+Matcher x2, RPN top-k/decode/NMS) "
                      f"of image 0 on 1 thread x {w.n_img} images = {t_img * w.n_img:.3f} s; both poolers fwd+bwd, all "
                      f"ROIs x 256 channels in channel slabs on {T} threads = {t_pool:.3f} s; mask targets + loss "
                      f"fwd/bwd (1 thread, 3 runs) = {t_mask:.3f} s",
            "host_cores_available": os.cpu_count()}


def cpu_reference_python(w):
    """The REFERENCE's own Python on the host's cores, where it runs without torchvision: image 0's `pairwise_iou` +
    `Matcher` (structures/boxes.py:312-358, modeling/matcher.py) for anchors and proposals, `subsample_labels`
    (modeling/sampling.py) and `find_top_rpn_proposals` (proposal_utils.py:22-135; its batched_nms = torchvision, absent
    here: the C port is bound in, and the decode feeding it is Box2BoxTransform.apply_deltas restated in torch).  Loaded
    by oracle/ref.py from the tree or the staged bytecode; torch threads = all cores; medians of 3.  The ROIAlign-backed
    reference functions (ROIPooler, BitMasks.crop_and_resize, mask_rcnn_loss) need torchvision and cannot run."""
    import oracle
    from oracle import ref

    if not ref.have_py():
        return None
    nthr = min(32, os.cpu_count() or 1)  # (256 torch threads on these small ops only fight each other: 4.9 s vs 0.02 s)
    torch.set_num_threads(nthr)
    boxes_mod, mt, sp = ref.py_boxes(), ref.py_matcher(), ref.py_sampling()
    Boxes = boxes_mod.Boxes

    def nms(b, s, i, t):
        return torch.from_numpy(oracle.batched_nms(b.numpy(), s.numpy(), i.numpy(), t))

    pu = ref.py_proposal_utils(nms)
    gt, an, pg = w.gt[0].cpu(), w.anchors.cpu(), w.props_with_gt[0].cpu()
    am, pm = mt.Matcher([0.3, 0.7], [0, -1, 1], True), mt.Matcher([0.5], [0, 1], False)
    labels = {}

    def match():
        _i, labels["a"] = am(boxes_mod.pairwise_iou(Boxes(gt), Boxes(an)))
        pm(boxes_mod.pairwise_iou(Boxes(gt), Boxes(pg)))

    t_match = _median_time(match, runs=3)
    t_samp = _median_time(lambda: sp.subsample_labels(labels["a"].clone(), RPN_BATCH, RPN_POS_FRACTION, 0), runs=3)
    props = []
    for a, d in zip(w.anchor_levels, w.rpn_deltas):
        a, d = a.cpu(), d[:1].cpu()
        wd, ht = a[:, 2] - a[:, 0], a[:, 3] - a[:, 1]
        cx, cy = a[:, 0] + 0.5 * wd, a[:, 1] + 0.5 * ht
        dw, dh = d[..., 2].clamp(max=math.log(1000.0 / 16)), d[..., 3].clamp(max=math.log(1000.0 / 16))
        pcx, pcy = d[..., 0] * wd + cx, d[..., 1] * ht + cy
        pw, ph = torch.exp(dw) * wd, torch.exp(dh) * ht
        props.append(torch.stack([pcx - 0.5 * pw, pcy - 0.5 * ph, pcx + 0.5 * pw, pcy + 0.5 * ph], -1))
    lg = [x[:1].cpu() for x in w.rpn_logits]
    t_rpn = _median_time(lambda: pu.find_top_rpn_proposals(props, lg, [(IMG_H, IMG_W)], 0.7, 2000, 1000, 0.0, True), runs=3)
    return {"kind": "reference", "cores": nthr, "unit": "s per image",
            "pairwise_iou+Matcher (anchors 16 x 268,569 and proposals 16 x 1,032)": round(t_match, 4),
            "subsample_labels (268,569 anchor labels)": round(t_samp, 4),
            "find_top_rpn_proposals (2000 / 1000, NMS = the C port)": round(t_rpn, 4),
            "note": "the reference's own Python (oracle/ref.py), image 0, medians of 3 after a warm-up; beside the port's "
                    "figure in `sample`, not part of `value`"}


def pmc_source(op=None, layout="nhwc"):
    """Where `roofline.traffic` comes from: the committed PMC summary that holds the op (newest round wins) and the
    commit that last touched it -- the driver's run does NOT re-measure traffic (counters need their own rocprofv3
    passes); the figure is replayed from that file."""
    import glob
    import subprocess

    src = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", f"pmc_traffic_{layout}.json"))):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        if op is None or op in d.get("ops", {}):
            src = f
    if src is None:
        return None
    rel = os.path.relpath(src, ROOT)
    try:
        commit = subprocess.run(["git", "-C", ROOT, "log", "-1", "--format=%h", "--", rel], capture_output=True,
                                text=True, timeout=10).stdout.strip() or None
    except Exception:
        commit = None
    return {"file": rel, "commit": commit, "measured_in_this_run": False}


def pmc_traffic(op, layout, key="hbm_bytes_per_launch"):
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
            best = d["ops"][op].get(key) or (d["ops"][op].get("hbm_bytes_per_launch") if key == "hbm_bytes_per_launch" else None)
    return best


def bench_maskrcnn(args, ctx):
    from detectron2_amd import _C as _dc
    from detectron2_amd.sharding import (MASK_RCNN_R50_FPN_GRADIENTS, GradientBuckets, Stopwatch, global_image_ids)

    dev, rank, world, dist = ctx["dev"], ctx["rank"], ctx["world"], ctx["dist"]
    dtype = {"bf16": torch.bfloat16, "fp32": torch.float32, "fp16": torch.float16}[args.dtype]
    # weak scaling: 2 images per GPU; rank r owns images [2r, 2r+1] of the global synthetic batch
    w = Workload(dev, dtype, args.layout, image_ids=global_image_ids(IMAGES_PER_GPU, rank, world), rois=getattr(args, "rois", "uniform"))
    w.overlap = not args.no_overlap
    w.connected = not args.disconnected
    grads = make_gradient_buckets(args, dev, dist, world)
    step = connected_step if w.connected else disconnected_step
    for _ in range(args.warmup):
        step(w, None, grads)
    sw = Stopwatch(dist, dev)
    KERN = "pool_bwd_staged_r7"
    KERN_SEL = "pool_bwd_pair,pool_bwd_staged_r7"  # the paired tile gather (both poolers, one launch) when the chain takes it
    use_graph = not args.no_graph
    if use_graph:
        gstep = GraphedConnectedStep(w, grads) if w.connected else GraphedStep(w, grads)
        for _ in range(3):
            gstep()
        sw.start()
        for _ in range(args.steps):
            gstep()
        elapsed = sw.stop()
        # The contract's timed region is the window above (`value` / `ms_per_step`).  Boxes differ by +-7 % and a window
        # is ~0.1 s: four more windows of the same K steps give the spread printed beside it (`ms_per_step_windows`).
        windows = [elapsed / args.steps * 1e3]
        if world == 1:
            for _ in range(4):
                sw.start()
                for _ in range(args.steps):
                    gstep()
                windows.append(sw.stop() / args.steps * 1e3)
        allreduce_info = measure_allreduce(args, w, grads, sw, elapsed, dist, world) if grads is not None else None
        # events inside a replayed graph cannot be read: the roofline kernel is timed by the library's launch-stream
        # events in an eager pass of the same steps right after the timed region
        _dc.lib().d2amd_timing_select(KERN_SEL.encode())
        dom_timer = Timer(only=("backward",))
        for _ in range(args.steps):
            step(w, dom_timer, grads)
        torch.cuda.synchronize()
    else:
        _dc.lib().d2amd_timing_select(KERN_SEL.encode())  # HIP events around the roofline kernel only, in the timed region
        dom_timer = Timer(only=("backward",))
        sw.start()
        for _ in range(args.steps):
            step(w, dom_timer, grads)
        elapsed = sw.stop()
        allreduce_info = None
        windows = [elapsed / args.steps * 1e3]
    ktimes = read_kernel_times(KERN_SEL.split(","))
    paired = "pool_bwd_pair" in ktimes
    if paired:
        KERN = "pool_bwd_pair"
    knames = ["pool_bwd_pair", "pool_bwd_staged_r7", "pool_bwd_staged_r14", "pool_fwd_pair", "pool_fwd_r7", "pool_fwd_r14",
              "nms_mask", "nms_reduce"]
    _dc.lib().d2amd_timing_select(",".join(knames).encode())
    timer = Timer()  # per-op breakdown: a separate, UNTIMED pass with events around every op
    bsteps = min(args.steps, 20)
    for _ in range(bsteps):
        step(w, timer, grads)
    torch.cuda.synchronize()
    for k, v in read_kernel_times(knames).items():
        ktimes.setdefault(k, v)
    _dc.lib().d2amd_timing_select(None)
    if rank != 0:
        return None
    ms_step = elapsed / args.steps * 1e3
    alg = w.alg_bytes()
    counts = {k: v // bsteps for k, v in timer.counts().items()}
    ops = {}
    for k, tot in timer.totals_ms().items():
        e = {"ms_per_step": round(tot / bsteps, 4), "launches_per_step": counts[k]}
        if k in alg:
            e["alg_MB"] = round(alg[k] / 1e6, 2)
            e["GBps"] = round(alg[k] / 1e6 / max(e["ms_per_step"], 1e-9), 1)
            e["frac_hbm_peak"] = round(e["GBps"] / HBM_PEAK_GBS, 4)
        ops[k] = e
    # (the eager pass behind the graphs, NOT the timed region; its first step pays one-off costs and is left out)
    dom_pairs = dom_timer.pairs["backward"]
    dom_pairs = dom_pairs[1:] if len(dom_pairs) > 1 else dom_pairs
    dom_ms_timed = sum(a.elapsed_time(b) for a, b in dom_pairs) / len(dom_pairs)
    ops["backward"]["ms_per_step_eager_after_graphs" if use_graph else "ms_per_step_timed_region"] = round(dom_ms_timed, 4)
    if args.layout == "nhwc" and KERN in ktimes:
        k_ms, k_n = ktimes[KERN]
        # the paired launch processes TWO of SURVEY 8(d)'s units (the box head's and the mask head's ROIAlign backward: the
        # reference zero-fills and writes dX once per pooler and sums the two); the one-unit figure is kept beside it
        kb_survey = alg["roi_align_box_bwd"] + (alg["roi_align_mask_bwd"] if paired else 0)
        kb = alg["roi_align_pair_bwd_compulsory"] if paired else alg["roi_align_box_bwd_compulsory"]
        pmc_key = "roi_align_pair_bwd" if paired else "roi_align_box_bwd"
        roof = {"bound": "hbm",
                "kernel": ("pool_bwd_kcat_kernel<T, 4, 48> (d2amd_roi_pooler_backward_pair): ONE K-concatenated tile gather "
                           "over all FPN levels for the 7x7 (box head) AND the 14x14 (mask head) pooler, inside `backward`" if paired else
                           "pool_bwd_kcat_kernel<T, 4, 48> (16-bit I/O; fp32: pool_bwd_staged_kernel<float, 4, 8>): the 7x7 "
                           "(box head) pooler's tile gather over all FPN levels, inside `backward`"),
                "achieved": round(kb / 1e6 / k_ms, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(kb / 1e6 / k_ms / HBM_PEAK_GBS, 4),
                "traffic": pmc_traffic(pmc_key, args.layout),
                "traffic_source": pmc_source(pmc_key, args.layout),
                "traffic_note": "PMC bytes of the pooler backward (records + tile lists + tile gather); frac_traffic = the "
                                "gather kernel's own PMC bytes / its time / peak",
                "alg_bytes_per_launch": int(kb), "ms_per_launch": round(k_ms, 4), "launches_timed": k_n,
                "units_per_launch": 2 if paired else 1,
                "alg_bytes_note": "`frac` charges the launch with its COMPULSORY bytes: both poolers' dY read once + every "
                                  "level's dX written once (no zero fill, no second write: the fused launch does neither).  "
                                  "`frac_survey_units` is the SURVEY 8(d) formula -- per pooler s*K*C*R^2 (dY) + 2*s*sum_l "
                                  "N*C*H_l*W_l (zero fill + write of dX), which the reference's two launches would move and "
                                  "this one does not" + ("; box unit %d B + mask unit %d B" % (alg["roi_align_box_bwd"], alg["roi_align_mask_bwd"]) if paired else ""),
                "frac_survey_units": round(kb_survey / 1e6 / k_ms / HBM_PEAK_GBS, 4),
                "survey_bytes_per_launch": int(kb_survey),
                "timing": "HIP events recorded by the library on the kernel's launch stream right around the launch, mean over "
                          + ("an eager pass of the same number of steps right after the timed region (the timed region "
                             "replays HIP graphs, inside which events cannot be read)" if use_graph else "the timed steps"),
                "kernels_ms": {k: round(v[0], 4) for k, v in ktimes.items()}}
        ktraffic = pmc_traffic(pmc_key, args.layout, "kernel_hbm_bytes_per_launch") or roof["traffic"]
        if ktraffic:
            roof["traffic_kernel"] = ktraffic  # the gather kernel alone (the op's figure includes records + binning)
            roof["frac_traffic"] = round(ktraffic / 1e6 / k_ms / HBM_PEAK_GBS, 4)
    else:
        per = alg["backward"]
        roof = {"bound": "hbm", "kernel": "backward (both poolers' tile gather + mask loss backward)",
                "achieved": round(per / 1e6 / dom_ms_timed, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(per / 1e6 / dom_ms_timed / HBM_PEAK_GBS, 4), "traffic": None,
                "alg_bytes_per_launch": int(per), "ms_per_launch": round(dom_ms_timed, 4),
                "timing": "HIP events on the launch stream around the op, mean over the timed steps",
                "kernels_ms": {k: round(v[0], 4) for k, v in ktimes.items()}}
    out = {
        "metric": "img/s through the Mask R-CNN R50-FPN detection hot path (training ops), 1333x800 bs=2/GPU",
        "value": round(world * w.n_img * args.steps / elapsed, 2), "unit": "img/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_step, 4),
        "ms_per_step_windows": {"n": len(windows), "min": round(min(windows), 4),
                                "median": round(sorted(windows)[len(windows) // 2], 4), "max": round(max(windows), 4),
                                "note": "windows of `steps` replays each; the first one is the contract's timed region "
                                        "(`value`, `ms_per_step`)"},
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": "maskrcnn_r50fpn_train_hotpath_bs2_800x1344 (BASELINE configs[1]; configs[2] at n_gpus 8)",
                   "layout": args.layout, "global_batch": world * w.n_img, "ops_per_step": counts,
                   "rois": w.rois_mode + (" (a trained RPN's picture: proposals and positives clustered on the 16 GT boxes)"
                                          if w.rois_mode == "clustered" else " (random RPN head: proposals spread over the image)"),
                   "step": ("connected: RPN selection + NMS -> label_and_sample_proposals on the NMS's device-side counts -> "
                            "box pooler on the 512 sampled rows / image, mask pooler + targets + masked loss on their "
                            "first 128 rows (both poolers as ONE launch per direction: pool_pair_rois / "
                            "d2amd_roi_pooler_backward_pair); anchor labelling + sampling beside the NMS; no host read inside "
                            "the step"
                            if w.connected else "disconnected (round 2): fixed ROI lists, the RPN's proposals feed nothing"),
                   "launch": (("ONE HIP graph per step (one hipGraphLaunch, no host sync)" if grads is None else
                               "2 HIP graphs per step (forward | backward, cut for the gradient all-reduce), no host sync")
                              if use_graph and w.connected else
                              "2 HIP graphs per step (RPN half | ROI-head half + backward), one host sync between them"
                              if use_graph else "eager: every op launched from Python"),
                   "streams": ("independent branches forked onto side streams (anchor labelling + sampling beside the NMS; "
                               "both poolers | mask targets + loss), joined before the backward"
                               if w.overlap else "one stream"),
                   "parallelism": f"dp{world}: images sharded, no data-path collective; "
                                  + (grads_description(grads) if grads is not None else
                                     "no gradient all-reduce (world size 1, like DDP)")},
        "roofline": roof, "gpu_ms_per_step_sum_of_ops": round(sum(v["ms_per_step"] for v in ops.values()), 4),
        "ops": ops,
        "ops_note": f"per-op times: separate untimed pass of {bsteps} steps with HIP events around every op",
    }
    if w.connected:
        try:  # (diagnostic: the distribution the tile gather worked on; one eager step after everything timed)
            out["roi_tiles"] = roi_tile_histogram(w, connected_step(w))
        except Exception as e:
            out["roi_tiles"] = {"error": f"{type(e).__name__}: {e}"}
    if allreduce_info is not None:
        out["allreduce"] = allreduce_info
    if not args.no_cpu_baseline and world == 1:
        out["cpu_baseline"] = cpu_baseline_maskrcnn(w)
    return out


def measure_allreduce(args, w, grads, sw, elapsed, dist, world):
    """N > 1 (or --force-dist): what the gradient all-reduce costs the step, separated from the data path.
    `value` of the line includes the collective (it is part of a training step); this block adds the same step WITHOUT
    it (`value_data_path`: the hot path's own weak scaling), the collective's exposed time inside the step, its time
    and bandwidth alone, and a value check (every rank contributes rank + 1: the average must be (world + 1) / 2).
    The backbone backward that would hide buckets 2 and 3 in a full model is outside the hot path: they are exposed here
    by construction, so `value` at N > 1 is the worst case for this collective."""
    dev = w.dev
    rank = dist.get_rank()
    for g in grads.grads:
        g.fill_(float(rank + 1))
    for i in range(grads.num_buckets):
        grads.reduce(i)
    grads.finish()
    torch.cuda.synchronize()
    ok = all(abs(float(g[0]) - (world + 1) / 2.0) < 1e-3 and abs(float(g[-1]) - (world + 1) / 2.0) < 1e-3 for g in grads.grads)
    sw.start()
    for _ in range(args.steps):
        for i in range(grads.num_buckets):
            grads.reduce(i)
        grads.finish()
    t_alone = sw.stop() / args.steps
    plain = GraphedConnectedStep(w, None) if w.connected else GraphedStep(w, None)
    for _ in range(3):
        plain()
    sw.start()
    for _ in range(args.steps):
        plain()
    t_plain = sw.stop() / args.steps
    t_step = elapsed / args.steps
    wire = grads.wire_bytes()
    XGMI_LINK_GBS, XGMI_LINKS = 153.0, 7  # MI355X_MICROARCH.md: 7 xGMI links x ~153 GB/s per GPU, point to point
    bus = wire * 2.0 * (world - 1) / world if world > 1 else wire
    return {"backend": dist.get_backend(), "world": world, "values_ok": bool(ok),
            "wire_MB": round(wire / 1e6, 1), "buckets": grads.num_buckets,
            "alone_ms": round(t_alone * 1e3, 4), "bus_GBps_alone": round(bus / 1e9 / max(t_alone, 1e-9), 1),
            "xgmi_peak_GBps_ring_per_link": XGMI_LINK_GBS, "xgmi_peak_GBps_all_links": XGMI_LINK_GBS * XGMI_LINKS,
            "ms_per_step_data_path": round(t_plain * 1e3, 4),
            "value_data_path": round(world * w.n_img / t_plain, 2),
            "exposed_ms": round(max(t_step - t_plain, 0.0) * 1e3, 4),
            "note": "value = the step WITH the collective; value_data_path = the same step without it (the hot path's own "
                    "scaling); exposed_ms = what the collective adds to the step; bus_GBps_alone = 2 (p - 1) / p x wire "
                    "bytes / its time alone (p = 1: wire bytes / time: RCCL's local copy)"}


def make_gradient_buckets(args, dev, dist, world):
    from detectron2_amd.sharding import MASK_RCNN_R50_FPN_GRADIENTS, GradientBuckets

    if dist is None or (world == 1 and not args.force_dist) or args.grad_allreduce == "off":
        return None
    g = dict(MASK_RCNN_R50_FPN_GRADIENTS)
    names = [n for n, _ in MASK_RCNN_R50_FPN_GRADIENTS]
    # three buckets in gradient-ready order: ROI heads | RPN head + FPN + res5 | res4 + res3 (34 / 38 / 17 MB in bf16)
    layout = [[(n, g[n]) for n in names[0:3]], [(n, g[n]) for n in names[3:6]], [(n, g[n]) for n in names[6:8]]]
    wire = torch.bfloat16 if args.grad_allreduce == "bf16" else None
    return GradientBuckets(layout, dev, dist, torch.float32, wire, reduce_single_rank=args.force_dist)


def grads_description(grads):
    return (f"gradient all-reduce of {grads.numel():,} parameters ({grads.wire_bytes() / 1e6:.1f} MB on the wire, "
            f"{str(grads.wire_dtype).replace('torch.', '')}, {grads.num_buckets} buckets, RCCL, async; the ROI heads' "
            f"bucket overlaps the pooler backward) inside the timed step")


# ------------------------------------------------------------------------------------ retinanet_100k
RETINA_A = [9 * 16800, 9 * 4200, 9 * 1050, 9 * 273, 9 * 77]  # anchors per level at 800x1344 (p3..p7, 9 / location)
RETINA_K, RETINA_TOPK, RETINA_NMS, RETINA_MAXDET = 80, 20000, 0.5, 100


def retina_inputs(dev, image_ids, seed=1234):
    gens = [image_generator(seed + 7, i) for i in image_ids]
    g0 = torch.Generator().manual_seed(seed + 99)
    anchors = []
    for li, a in enumerate(RETINA_A):
        s = 32.0 * 2 ** li
        anchors.append(make_boxes(g0, a, s * 0.7, s * 1.5).to(dev))
    logits = [torch.stack([torch.randn(a, RETINA_K, generator=g) * 1.2 - 4.6 for g in gens]).to(dev) for a in RETINA_A]
    deltas = [torch.stack([torch.randn(a, 4, generator=g) * 0.2 for g in gens]).to(dev) for a in RETINA_A]
    return anchors, logits, deltas


def bench_retinanet(args, ctx):
    from detectron2_amd import _C as _dc
    from detectron2_amd.modeling import dense_detector_inference_fused
    from detectron2_amd.sharding import Stopwatch, global_image_ids

    dev, rank, world, dist = ctx["dev"], ctx["rank"], ctx["world"], ctx["dist"]
    ids = global_image_ids(IMAGES_PER_GPU, rank, world)
    anchors, logits, deltas = retina_inputs(dev, ids)
    sizes = [(IMG_H, IMG_W)] * len(ids)

    def one():
        return dense_detector_inference_fused(anchors, logits, deltas, sizes, 0.0, RETINA_TOPK, RETINA_NMS,
                                              RETINA_MAXDET)

    # The step is ~60 launches (selection + per-image NMS pipelines on side streams) and ONE host read: captured once in
    # a HIP graph and replayed, like the headline step; the read of the kept counts stays outside.  D2AMD_BENCH_RETINA_EAGER=1
    # (or a capture error): the eager calls.
    execution = "eager"
    if os.environ.get("D2AMD_BENCH_RETINA_EAGER") != "1":
        try:
            g, fin = GraphedStep._capture(lambda: dense_detector_inference_fused(
                anchors, logits, deltas, sizes, 0.0, RETINA_TOPK, RETINA_NMS, RETINA_MAXDET, defer=True))

            def one():  # noqa: F811
                g.replay()
                return fin()

            ref = dense_detector_inference_fused(anchors, logits, deltas, sizes, 0.0, RETINA_TOPK, RETINA_NMS, RETINA_MAXDET)
            got = one()
            assert all(torch.equal(a.pred_boxes.tensor, b.pred_boxes.tensor) and torch.equal(a.scores, b.scores)
                       for a, b in zip(ref, got)), "graph replay differs from the eager call"
            execution = "one HIP graph per step (selection + both images' NMS pipelines) + one host read"
        except Exception as e:  # keep the eager path
            print(f"[bench] retinanet_100k: graph capture failed ({type(e).__name__}: {e}); eager", file=sys.stderr)
            torch.cuda.synchronize()
    for _ in range(args.warmup):
        one()
    knames = ["nms_mask", "nms_reduce"]
    graphed = execution != "eager"
    if not graphed:
        _dc.lib().d2amd_timing_select(",".join(knames).encode())
    sw = Stopwatch(dist, dev)
    sw.start()
    for _ in range(args.steps):
        res = one()
    elapsed = sw.stop()
    if graphed:  # a replayed graph has no per-kernel events: an eager pass of the same steps right after the timed region
        _dc.lib().d2amd_timing_select(",".join(knames).encode())
        for _ in range(args.steps):
            dense_detector_inference_fused(anchors, logits, deltas, sizes, 0.0, RETINA_TOPK, RETINA_NMS, RETINA_MAXDET)
        torch.cuda.synchronize()
    ktimes = read_kernel_times(knames)
    _dc.lib().d2amd_timing_select(None)
    if rank != 0:
        return None
    n_img = len(ids)
    n_box = sum(min(a * RETINA_K, RETINA_TOPK) for a in RETINA_A)
    # SURVEY 8(d) NMS: pairs = upper triangle within class; bytes = 16N + 2 x 8 * sum_c n_c * ceil(n_c / 64)
    # (bitmask write + read) + 8 N_keep, per image; classes ~ uniform over 80
    per_cls = n_box / RETINA_K
    pairs = RETINA_K * per_cls * (per_cls - 1) / 2
    mask_bytes = 8 * RETINA_K * per_cls * math.ceil(per_cls / 64)
    alg_img = 16 * n_box + 2 * mask_bytes + 8 * RETINA_MAXDET
    dom = max(ktimes, key=lambda k: ktimes[k][0]) if ktimes else None
    roof = {"bound": "hbm", "kernel": None, "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None,
            "traffic": None}
    if dom:
        k_ms, k_n = ktimes[dom]
        # 100k candidates per image exceed the batched pipeline's 12,288 (RANK_MAX_N): batched_nms_images runs ONE
        # LAUNCH PER IMAGE on side streams (layers/ops.py: nms_images) -- bytes and pairs per launch are one image's
        per_launch_images = 1 if n_box > 12288 else n_img
        kb = per_launch_images * alg_img
        roof = {"bound": "hbm",
                "kernel": {"nms_mask": "nms_mask_kernel (wavefront suppression bitmask, per class)",
                           "nms_reduce": "nms_reduce_kernel (greedy reduction over the bitmask)"}[dom],
                "achieved": round(kb / 1e6 / k_ms, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(kb / 1e6 / k_ms / HBM_PEAK_GBS, 4), "traffic": pmc_traffic("retinanet_" + dom, "nhwc"),
                "traffic_source": pmc_source("retinanet_" + dom, "nhwc"),
                "alg_bytes_per_launch": int(kb), "ms_per_launch": round(k_ms, 4), "launches_timed": k_n,
                "alg_bytes_note": "SURVEY 8(d) NMS: 16N boxes + bitmask write+read 2*8*sum_c n_c*ceil(n_c/64) + 8*N_keep; "
                                  "the kernel is VALU/LDS bound (IoU tests), so this fraction is small by construction: "
                                  "see pairs_per_s",
                "images_per_launch": per_launch_images,
                "pairs_per_s": round(per_launch_images * pairs / (k_ms / 1e3), 1),
                "timing": "HIP events recorded by the library on the kernel's launch stream, mean over "
                          + ("an eager pass of the same number of steps right after the timed region (the timed region "
                             "replays a HIP graph, which has no per-kernel events)" if graphed else "the timed steps"),
                "kernels_ms": {k: round(v[0], 4) for k, v in ktimes.items()}}
    out = {
        "metric": "img/s through the RetinaNet R50-FPN inference hot path (select + decode + batched NMS), 100k candidates/img",
        "value": round(world * n_img * args.steps / elapsed, 2), "unit": "img/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "fp32", "data": "synthetic",
        "config": {"workload": "retinanet_r50fpn_inference_100k_candidates_800x1344 (BASELINE configs[3])",
                   "candidates_per_image": n_box, "class_logits_per_image": sum(RETINA_A) * RETINA_K,
                   "num_classes": RETINA_K, "topk_candidates": RETINA_TOPK, "score_thresh": 0.0, "nms_thresh": RETINA_NMS,
                   "detections_kept": [len(r) for r in res], "global_batch": world * n_img, "execution": execution,
                   "parallelism": f"dp{world}: images sharded, replicas only (inference, no collective)"},
        "roofline": roof,
    }
    if not args.no_cpu_baseline and world == 1:
        out["cpu_baseline"] = cpu_baseline_retinanet(anchors, logits, deltas)
    return out


def cpu_baseline_retinanet(anchors, logits, deltas):
    """oracle/dense_detector.py (numpy restatement + the C NMS port) on image 0, in full; median of 3 runs."""
    from oracle import dense_detector as odd

    an = [a.cpu().numpy() for a in anchors]
    lg = [x[0].cpu().numpy() for x in logits]
    dl = [x[0].cpu().numpy() for x in deltas]
    t = _median_time(lambda: odd.inference_single_image(an, lg, dl, 0.0, RETINA_TOPK, RETINA_NMS, RETINA_MAXDET), runs=3)
    return {"value": round(1.0 / t, 4), "unit": "img/s", "cores": 1, "kind": "port",
            "sample": f"image 0 in full (16.1 M class logits -> 100k candidates -> per-class NMS -> top 100), numpy "
                      f"selection + oracle/d2_oracle.c NMS, 1 thread, median of 3 runs after a warm-up = {t:.3f} s",
            "host_cores_available": os.cpu_count()}


# ------------------------------------------------------------------------------------ dcn_r50
DCN_STAGES = (("res3", 4, 128, 100, 168), ("res4", 6, 256, 50, 84), ("res5", 3, 512, 25, 42))


def bench_dcn(args, ctx):
    from detectron2_amd import _C as _dc
    from detectron2_amd.layers import ModulatedDeformConv
    from detectron2_amd.sharding import Stopwatch, global_image_ids

    dev, rank, world, dist = ctx["dev"], ctx["rank"], ctx["world"], ctx["dist"]
    dtype = {"bf16": torch.bfloat16, "fp32": torch.float32, "fp16": torch.float16}[args.dtype]
    ids = global_image_ids(IMAGES_PER_GPU, rank, world)
    n_img = len(ids)
    gen = image_generator(4321, ids[0])
    blocks, flops_fwd = [], 0.0
    for tag, nblk, ch, h, wd in DCN_STAGES:
        for b in range(nblk):
            mod = ModulatedDeformConv(ch, ch, 3, padding=1, bias=False).to(dev).to(dtype)
            # --layout nhwc (default): channels_last activations and gradients, the kernels' native layout (a model run
            # with memory_format=torch.channels_last; d2amd_dcn_params.layout = NHWC: no transposes in or out);
            # --layout nchw: the reference's DeformBottleneckBlock layout (resnet.py:303-327), transposed per call
            mf = torch.channels_last if (args.layout == "nhwc" and dtype != torch.float32) else torch.contiguous_format
            x = torch.randn(n_img, ch, h, wd, generator=gen).to(dev).to(dtype).contiguous(memory_format=mf)
            off = (torch.randn(n_img, 18, h, wd, generator=gen) * 2).to(dev).to(dtype)
            msk = torch.sigmoid(torch.randn(n_img, 9, h, wd, generator=gen)).to(dev).to(dtype)
            gy = torch.randn(n_img, ch, h, wd, generator=gen).to(dev).to(dtype).contiguous(memory_format=mf)
            blocks.append((tag, mod, x.requires_grad_(True), off.requires_grad_(True), msk.requires_grad_(True), gy))
            flops_fwd += 2.0 * ch * ch * 9 * n_img * h * wd

    def one(t=None):
        run = (lambda name, fn: t.run(name, fn)) if t is not None else (lambda name, fn: fn())
        for tag, mod, x, off, msk, gy in blocks:
            y = run("fwd_" + tag, lambda: mod(x, off, msk))
            run("bwd_" + tag, lambda: torch.autograd.backward([y], [gy]))
            x.grad = off.grad = msk.grad = mod.weight.grad = None

    # The step is ~400 launches through Python and autograd (13 x (forward + backward)): issued eagerly it is as much a
    # measurement of the host as of the device (3.3-4.3 ms on boxes whose kernels take the same time).  Like the other
    # workloads it is captured once in a HIP graph and replayed; D2AMD_BENCH_DCN_EAGER=1 (or a capture error): eager.
    execution, step = "eager", one
    if os.environ.get("D2AMD_BENCH_DCN_EAGER") != "1":
        try:
            g, _ = GraphedStep._capture(one)
            step, execution = g.replay, "one HIP graph per step (13 x (ModulatedDeformConv forward + autograd backward))"
        except Exception as e:  # keep the eager path
            print(f"[bench] dcn_r50: graph capture failed ({type(e).__name__}: {e}); eager", file=sys.stderr)
            torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    knames = ["dcn_fwd_col", "dcn_fwd_gemm", "dcn_bwd_dcol_gemm", "dcn_bwd_coord", "dcn_bwd_gather", "dcn_bwd_weight",
              "dcn_fwd", "dcn_bwd_data"]  # (the last two: the fused kernels of r01-r04, --layout nchw / D2AMD_DCN_FUSED=1)
    graphed = execution != "eager"
    if not graphed:
        _dc.lib().d2amd_timing_select(",".join(knames).encode())
    sw = Stopwatch(dist, dev)
    sw.start()
    for _ in range(args.steps):
        step()
    elapsed = sw.stop()
    if graphed:  # a replayed graph has no per-kernel events: an eager pass right after the timed region
        _dc.lib().d2amd_timing_select(",".join(knames).encode())
        for _ in range(min(args.steps, 10)):
            one()
        torch.cuda.synchronize()
    ktimes = read_kernel_times(knames)
    _dc.lib().d2amd_timing_select(None)
    timer = Timer()
    bsteps = min(args.steps, 5)
    for _ in range(bsteps):
        one(timer)
    torch.cuda.synchronize()
    if rank != 0:
        return None
    # every kernel is launched once per block; its algorithmic flops per launch = the GEMM it carries, averaged over
    # the 13 blocks (all three stages have the same 2*C*9C*H*W): fwd C x 9C x P, bwd-data 9C x C x P, bwd-weight C x 9C x P
    per_launch = flops_fwd / len(blocks)
    roof = {"bound": "mfma", "kernel": None, "achieved": None, "peak": MFMA_BF16_TFLOPS, "unit": "TFLOP/s",
            "frac": None, "traffic": None}
    if ktimes and args.dtype != "fp32":
        gemms = [k for k in ktimes if k in ("dcn_fwd_gemm", "dcn_bwd_dcol_gemm", "dcn_bwd_weight", "dcn_fwd", "dcn_bwd_data")]
        dom = max(gemms or ktimes, key=lambda k: ktimes[k][0] * ktimes[k][1])  # the matrix kernel that takes the most time
        k_ms, k_n = ktimes[dom]
        roof = {"bound": "mfma",
                "kernel": {"dcn_fwd_gemm": "gemm_nt_kernel (Y = col Wp^T + bias: dense NT GEMM, LDS-DMA staged, on the column dcn_col_kernel wrote)",
                           "dcn_bwd_dcol_gemm": "gemm_nt_kernel (dcol = dY Wt^T: dense NT GEMM; writes the 16-bit column the coordinate-gradient "
                                                "kernel and the dX gather read)",
                           "dcn_fwd": "dcn_fwd_wave_kernel / dcn_fwd_tc_kernel (gather + MFMA, no column buffer)",
                           "dcn_bwd_data": "dcn_bwd_data_ws_kernel (wave-specialised: dcol = W^T dY on MFMA by the matrix waves -> "
                                           "16-bit column rows + d offset / d mask by the consumer waves)",
                           "dcn_bwd_gather": "dcn_gather_dx_kernel (dX = per-pixel gather of the column rows; HBM/L2 bound, no flops counted)",
                           "dcn_bwd_weight": "dcn_bww_gemm_kernel (dW = dY^T col: dense split-K MFMA GEMM over the column the forward saved)"}[dom],
                "achieved": round(per_launch / 1e9 / k_ms, 1), "peak": MFMA_BF16_TFLOPS, "unit": "TFLOP/s",
                "frac": round(per_launch / 1e9 / k_ms / MFMA_BF16_TFLOPS, 4), "traffic": pmc_traffic(dom, args.layout),
                "traffic_source": pmc_source(dom, args.layout),
                "traffic_all_kernels": {k: pmc_traffic(k, args.layout) for k in ktimes},
                "frac_step": round(3 * flops_fwd / 1e12 / (elapsed / args.steps) / (MFMA_BF16_TFLOPS / 1e3) / 1e3, 4),
                "frac_step_note": "the whole step's 3 GEMMs per block (386 GFLOP) over its wall time, against the dense bf16 MFMA "
                                  "peak: includes the column / coordinate-gradient / gather kernels, which carry no flops",
                "alg_flops_per_launch": per_launch, "ms_per_launch": round(k_ms, 4), "launches_timed": k_n,
                "alg_flops_note": "SURVEY 8(d) DCN: 2*Co*Ci*kh*kw*N*Ho*Wo per block and GEMM (9.9 GFLOP for 2 images, "
                                  "identical for res3/res4/res5); mean over the 13 blocks' launches",
                "timing": "HIP events recorded by the library on the kernel's launch stream, mean over "
                          + ("an eager pass right after the timed region (the timed region replays a HIP graph, which has no "
                             "per-kernel events)" if graphed else "the timed steps"),
                "kernels_ms": {k: round(v[0], 4) for k, v in ktimes.items()},
                "kernels_frac_mfma": {k: round(per_launch / 1e9 / v[0] / MFMA_BF16_TFLOPS, 4) for k, v in ktimes.items()}}
        # the other bound of these kernels: the deformable GATHER.  Every (position, tap) reads 4 corners x C channels x
        # 2 B through the CUs' vector L1 (the sampled pixels are served by L2; nothing of it is HBM-compulsory): the
        # kernel cannot be faster than those bytes at the aggregate L1 rate (64 B/clk/CU: 37.7 TB/s,
        # MI355X_MICROARCH.md) -- 310 / 155 / 77 MB for res3 / res4 / res5, a 2-8 us floor that sits ABOVE the MFMA floor
        gather_bytes = sum(n_img * h * wd * 9 * 4 * ch * 2 * nblk for _t, nblk, ch, h, wd in DCN_STAGES) / len(blocks)
        roof["gather_floor"] = {"bytes_per_launch": int(gather_bytes), "peak_GBps": L1_PEAK_GBS,
                                "floor_ms": round(gather_bytes / 1e6 / L1_PEAK_GBS, 4),
                                "frac_of_floor": {k: round(gather_bytes / 1e6 / L1_PEAK_GBS / v[0], 4)
                                                  for k, v in ktimes.items() if k != "dcn_bwd_gather"}}
    ops = {k: {"ms_per_step": round(v / bsteps, 4), "launches_per_step": timer.counts()[k] // bsteps}
           for k, v in timer.totals_ms().items()}
    out = {
        "metric": "img/s through the 13 DCNv2 blocks of R50 res3-res5 (ModulatedDeformConv forward + backward), 1333x800 bs=2/GPU",
        "value": round(world * n_img * args.steps / elapsed, 2), "unit": "img/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": "dcnv2_r50_res3-5_13_blocks_fwd+bwd_bs2_800x1344 (BASELINE configs[4])",
                   "layout": args.layout if dtype != torch.float32 else "nchw", "global_batch": world * n_img,
                   "gflop_per_step": round(3 * flops_fwd / 1e9, 1), "execution": execution,
                   "parallelism": f"dp{world}: images sharded; DCN weight gradients reduce with the model's (not in this step)"},
        "roofline": roof, "ops": ops,
        "step_tflops": round(3 * flops_fwd / 1e12 / (elapsed / args.steps), 1),
    }
    if not args.no_cpu_baseline and world == 1:
        out["cpu_baseline"] = cpu_baseline_dcn(blocks)
    return out


def cpu_baseline_dcn(blocks):
    """oracle/d2_oracle.c DCNv2 forward + backward on ONE block (the first res5 block: smallest maps, same flops),
    its images on separate threads; scaled by the 13 blocks of the step (every block has the same flops)."""
    from concurrent.futures import ThreadPoolExecutor

    import oracle

    tag, mod, x, off, msk, gy = [b for b in blocks if b[0] == "res5"][0]
    n = x.shape[0]
    xs = x.detach().float().cpu().contiguous().numpy()
    of = off.detach().float().cpu().numpy()
    mk = msk.detach().float().cpu().numpy()
    g = gy.detach().float().cpu().contiguous().numpy()
    wt = mod.weight.detach().float().cpu().numpy()

    def one(i):
        oracle.deform_conv_forward(xs[i:i + 1], of[i:i + 1], wt, mask=mk[i:i + 1], padding=1)
        oracle.deform_conv_backward(xs[i:i + 1], of[i:i + 1], wt, g[i:i + 1], mask=mk[i:i + 1], padding=1)

    t0 = time.perf_counter()
    with ThreadPoolExecutor(n) as ex:
        list(ex.map(one, range(n)))
    t = time.perf_counter() - t0
    return {"value": round(n / (t * len(blocks)), 4), "unit": "img/s", "cores": n, "kind": "port",
            "sample": f"1 of the 13 blocks (res5, {n} images on {n} threads, forward + backward, one run = {t:.2f} s), "
                      f"scaled x13 (equal flops per block)",
            "host_cores_available": os.cpu_count()}


# ------------------------------------------------------------------------------------ plumbing-only (tests)
def bench_plumbing(args, ctx):
    """Launcher + process group + GradientBuckets + Stopwatch with NO hot-path op: what tests/test_sharding_gloo.py
    drives on CPU with --backend gloo.  Not a measurement."""
    from detectron2_amd.sharding import Stopwatch, global_image_ids

    dev, rank, world, dist = ctx["dev"], ctx["rank"], ctx["world"], ctx["dist"]
    grads = make_gradient_buckets(args, dev, dist, world)
    sw = Stopwatch(dist, dev)
    sw.start()
    check = None
    for _ in range(args.steps):
        if grads is not None:
            for i, g in enumerate(grads.grads):
                g.fill_(float(rank + 1))
            for i in range(grads.num_buckets):
                grads.reduce(i)
            grads.finish()
            check = [float(g[0]) for g in grads.grads]
    elapsed = sw.stop()
    if rank != 0:
        return None
    return {"metric": "plumbing only (no hot-path op): not a measurement", "value": None, "unit": "img/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "plumbing_only": True,
            "ms_per_step": round(elapsed / max(args.steps, 1) * 1e3, 4), "backend": args.backend,
            "image_ids_rank0": global_image_ids(IMAGES_PER_GPU, rank, world),
            "allreduce_mean": check, "expected_mean": (world + 1) / 2.0,
            "buckets": grads.num_buckets if grads is not None else 0,
            "parallelism": grads_description(grads) if grads is not None else "no all-reduce"}


# ------------------------------------------------------------------------------------ main
def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(relaunch_under_torchrun(args))
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    # The contract is ONE JSON line on stdout.  Libraries write there too (RCCL prints its version banner to the C
    # stdout when the first communicator is created, and the buffer is flushed at exit: behind the JSON line), so the
    # real stdout is kept aside for that line and file descriptor 1 is pointed at stderr for everything else.
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started {world} ranks")
    gpu = args.backend == "nccl"
    if gpu:
        assert torch.cuda.is_available(), "bench.py needs a GPU (there is no CPU fallback)"
        assert torch.cuda.device_count() > local, f"rank {rank}: no GPU {local} ({torch.cuda.device_count()} visible)"
        torch.cuda.set_device(local)
        dev = torch.device("cuda", local)
    else:
        assert args.plumbing_only, "--backend gloo only serves --plumbing-only"
        dev = torch.device("cpu")
    dist = None
    if world > 1 or args.force_dist:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(free_port()))
        if gpu:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
    ctx = {"dev": dev, "rank": rank, "world": world, "dist": dist}
    if args.plumbing_only:
        out = bench_plumbing(args, ctx)
    else:
        import bench_extra

        fns = {"maskrcnn_train": bench_maskrcnn, "retinanet_100k": bench_retinanet, "dcn_r50": bench_dcn,
               "maskrcnn_infer": bench_extra.bench_maskrcnn_infer, "rrpn_micro": bench_extra.bench_rrpn_micro}
        out = fns[args.workload](args, ctx)
        # The default run also carries BASELINE configs[3] and [4] and the two SURVEY 8(d) workloads of bench_extra.py (a
        # few steps each, no CPU leg) so that the driver's ONE line records them: `extra_workloads` = their own metric /
        # value / ms_per_step / roofline.
        if args.workload == "maskrcnn_train" and world == 1 and not args.no_extra_workloads and not args.force_dist:
            import copy

            extra = {}
            # (steps / warm-up: a single slow step of 10 moved dcn_r50 from 3.3 to 4.7 ms on one box: 20-50 steps, 10-20 warm)
            for name, fn, steps, warm in (("retinanet_100k", bench_retinanet, 50, 20), ("dcn_r50", bench_dcn, 20, 10),
                                          ("maskrcnn_infer", bench_extra.bench_maskrcnn_infer, 30, 10),
                                          ("rrpn_micro", bench_extra.bench_rrpn_micro, 20, 10)):
                a2 = copy.copy(args)
                a2.workload, a2.steps, a2.warmup, a2.no_cpu_baseline = name, steps, warm, True
                torch.cuda.synchronize()
                r = fn(a2, ctx)
                extra[name] = {k: r[k] for k in ("metric", "value", "unit", "ms_per_step", "steps", "warmup", "dtype",
                                                 "config", "roofline", "ops") if k in r}
            # the same connected step (a) fed NCHW features -- what an UNMODIFIED model hands the poolers: channels_last
            # staging copies and the transposes of results / gradients included -- and (b) with a
            # trained RPN's clustered proposals (weak 8): their own short lines
            for name, over in (("nchw_drop_in", {"layout": "nchw"}), ("clustered_rois", {"rois": "clustered"})):
                a2 = copy.copy(args)
                a2.steps, a2.warmup, a2.no_cpu_baseline = 100, 20, True
                for k, v in over.items():
                    setattr(a2, k, v)
                torch.cuda.synchronize()
                r = bench_maskrcnn(a2, ctx)
                extra[name] = {k: r[k] for k in ("metric", "value", "unit", "ms_per_step", "steps", "warmup", "dtype", "config",
                                                 "roofline", "roi_tiles") if k in r}
                extra[name]["vs_default_step"] = round(r["ms_per_step"] / out["ms_per_step"], 3)
            # configs[4] entered with NCHW activations, as the reference's DeformBottleneckBlock hands them over
            # (resnet.py:303-327): staged channels_last per call (layers/deform_conv.py: _stage_nhwc), results back in NCHW
            a2 = copy.copy(args)
            a2.workload, a2.steps, a2.warmup, a2.no_cpu_baseline, a2.layout = "dcn_r50", 20, 10, True, "nchw"
            torch.cuda.synchronize()
            r = bench_dcn(a2, ctx)
            extra["dcn_r50_nchw"] = {k: r[k] for k in ("metric", "value", "unit", "ms_per_step", "steps", "warmup", "dtype",
                                                       "config") if k in r}
            extra["dcn_r50_nchw"]["vs_channels_last"] = round(r["ms_per_step"] / extra["dcn_r50"]["ms_per_step"], 3)
            out["extra_workloads"] = extra
            # FLAT copies of the figures above (the driver's record keeps top-level scalars only: BENCH_rNN.parsed drops
            # nested objects)
            flat = {"dcn_r50_ms": extra["dcn_r50"].get("ms_per_step"),
                    "dcn_r50_frac_step": (extra["dcn_r50"].get("roofline") or {}).get("frac_step"),
                    "maskrcnn_infer_ms": extra["maskrcnn_infer"].get("ms_per_step"),
                    "retinanet_100k_ms": extra["retinanet_100k"].get("ms_per_step"),
                    "rrpn_micro_ms": extra["rrpn_micro"].get("ms_per_step"),
                    "nchw_drop_in_ms": extra["nchw_drop_in"].get("ms_per_step"),
                    "dcn_r50_nchw_ms": extra["dcn_r50_nchw"].get("ms_per_step"),
                    "clustered_rois_ms": extra["clustered_rois"].get("ms_per_step")}
            out.update({k: v for k, v in flat.items() if v is not None})
        if args.workload == "maskrcnn_train":
            km = (out.get("roofline") or {}).get("kernels_ms") or {}
            for key, name in (("pool_bwd_pair_us", "pool_bwd_pair"), ("pool_fwd_pair_us", "pool_fwd_pair")):
                if name in km:
                    out[key] = round(km[name] * 1e3, 2)
            chain = [km.get(k) for k in ("nms_mask", "nms_reduce")]  # (order / finalize: rocprof stats under profiles/)
            if all(v is not None for v in chain):
                out["nms_mask_reduce_us"] = round(sum(chain) * 1e3, 2)
            for k in ("frac", "frac_traffic", "frac_survey_units"):
                if k in (out.get("roofline") or {}):
                    out["roofline_" + k] = out["roofline"][k]
    if rank == 0:
        json_out.write(json.dumps(out) + "\n")
        json_out.flush()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
