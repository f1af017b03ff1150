"""Whole-model context for the hot path (NOT bench.py, not the product): the reference's own `GeneralizedRCNN` (Mask R-CNN
R50-FPN, configs/COCO-InstanceSegmentation/mask_rcnn_R_50_FPN_1x.yaml) and `RetinaNet`, imported UNCHANGED from the
byte-compiled package of the test infrastructure (tests/_reference_model.py: oracle/_ref/pkg + stand-ins for the absent
third-party packages), training iterations of 2 synthetic 800 x 1333 images on one MI355X with `detectron2.layers` bound to
this library -- forward + backward + SGD step, the convolutions on MIOpen as in any PyTorch-ROCm run.  What it answers:
BASELINE.json's target line ("Mask R-CNN R50-FPN training >= MODEL_ZOO.md's reported img/s on 1 MI355X": 61.3 img/s is the
8 x V100 aggregate, 7.66 img/s one V100's share) for the model the hot path sits in, and how much of an iteration the hot
path still is.

    python scripts/model_iteration_bench.py [--iters 30] [--amp] [--channels-last] [--model maskrcnn|retinanet]
-> one JSON line per configuration on stdout."""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--amp", action="store_true", help="torch.autocast(bfloat16), as the reference's AMPTrainer")
    ap.add_argument("--channels-last", action="store_true")
    ap.add_argument("--model", default="maskrcnn", choices=["maskrcnn", "retinanet"])
    ap.add_argument("--backend", default="product", choices=["product", "reference"])
    ap.add_argument("--size", type=int, nargs=2, default=[800, 1333])
    ap.add_argument("--fused", action="store_true",
                    help="detectron2_amd.integrate.patch: the FUSED callers bound into the model (Level 1 of INTEGRATION.md)")
    ap.add_argument("--count-syncs", action="store_true",
                    help="host synchronisations per iteration (torch.cuda.set_sync_debug_mode('warn'), counted over 3 iterations)")
    args = ap.parse_args()

    import _reference_model as rm

    rm.install()
    from detectron2.utils.events import EventStorage

    torch.backends.cudnn.benchmark = False  # (the ROI heads change shape every iteration: benchmark mode re-tunes MIOpen each time, 1.4 s per iteration)
    cfg = rm.mask_rcnn_cfg() if args.model == "maskrcnn" else rm.retinanet_cfg()
    model = rm.build_model(cfg, seed=0, device="cuda")
    with torch.no_grad():  # keep the random-init activations in a trained model's range (tests/test_gpu_reference_models.py)
        for m in model.modules():
            if hasattr(m, "conv3") and hasattr(m.conv3, "norm") and hasattr(m.conv3.norm, "weight"):
                m.conv3.norm.weight.fill_(0.2)
    if args.channels_last:
        model = model.to(memory_format=torch.channels_last)
    model.train()
    params = [p for p in model.parameters() if p.requires_grad]
    opt = torch.optim.SGD(params, lr=1e-5, momentum=0.9, weight_decay=1e-4)
    inputs = rm.make_inputs(2, tuple(args.size), 8, seed=3, device="cuda", masks=args.model == "maskrcnn")

    def iteration():
        opt.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=args.amp):
            losses = model(inputs)
            loss = sum(losses.values())
        loss.backward()
        opt.step()
        return loss

    import contextlib

    class _Late(contextlib.ExitStack):  # (entered INSIDE rm.backend: the test infrastructure binds pairwise_iou too)
        def __enter__(self):
            r = super().__enter__()
            if args.fused:
                import detectron2
                from detectron2_amd import integrate

                self.enter_context(integrate.patch(detectron2, models=[model], layers=False))
            return r

    stack = _Late()
    syncs = None
    with rm.backend(args.backend), stack, EventStorage(0):
        for _ in range(args.warmup):
            iteration()
        if args.count_syncs:
            import warnings

            torch.cuda.synchronize()
            torch.cuda.set_sync_debug_mode("warn")
            with warnings.catch_warnings(record=True) as rec:
                warnings.simplefilter("always")
                for _ in range(3):
                    iteration()
            torch.cuda.set_sync_debug_mode("default")
            syncs = round(sum("synchroniz" in str(w.message).lower() for w in rec) / 3.0, 1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.iters):
            loss = iteration()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / args.iters
    print(json.dumps({"model": args.model, "backend": args.backend, "fused_callers": args.fused, "host_syncs_per_iteration": syncs,
                      "amp_bf16": args.amp, "channels_last": args.channels_last,
                      "images_per_iteration": 2, "image_size": args.size, "iterations": args.iters,
                      "s_per_iteration": round(dt, 5), "img_per_s": round(2 / dt, 2), "final_loss": float(loss.detach()),
                      "trainable_parameters": int(sum(p.numel() for p in params)),
                      "note": "the reference's unmodified model code; detectron2.layers bound to libd2amd (product) or to "
                              "plain-torch / host restatements (reference backend: a checker, not a fast baseline); "
                              "convolutions = MIOpen; synthetic images, 8 ground-truth instances each; forward + backward + SGD"}))


if __name__ == "__main__":
    main()
