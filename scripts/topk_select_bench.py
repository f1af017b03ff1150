"""The dense-detector selection alone (one stream, back to back: kernels of consecutive calls cannot overlap, unlike the
bench step whose per-image NMS runs on side streams) at the bench's RetinaNet shapes: 2 x 16.1 M class logits,
score_thresh 0, 20,000 candidates per level.   python scripts/topk_select_bench.py [reps]"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.getcwd())
import bench  # noqa: E402
from detectron2_amd.modeling import dense_select_predictions  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
dev = torch.device("cuda")
anchors, logits, deltas = bench.retina_inputs(dev, [0, 1])
for thr, topk in ((0.0, bench.RETINA_TOPK), (0.05, 1000)):
    for _ in range(5):
        dense_select_predictions(anchors, logits, deltas, thr, topk)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        dense_select_predictions(anchors, logits, deltas, thr, topk)
    torch.cuda.synchronize()
    print(json.dumps({"score_thresh": thr, "topk": topk, "dense_select_ms": round((time.perf_counter() - t0) / reps * 1e3, 4)}))
