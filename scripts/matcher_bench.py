"""Device time of the fused matcher vs pairwise_iou + (matrix) Matcher at the RPN size (16 x 268,569)."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from detectron2_amd.modeling import Matcher
from detectron2_amd.structures import pairwise_iou
from scripts.dcn_bench import timeit
w = bench.Workload(torch.device("cuda", 0), torch.bfloat16, "nhwc")
gt, an = w.gt[0], w.anchors
mt = Matcher([0.3, 0.7], [0, -1, 1], allow_low_quality_matches=True)
out = {"fused_match_boxes_ms": round(timeit(lambda: mt.match_boxes(gt, an), rep=50), 4),
       "pairwise_iou_ms": round(timeit(lambda: pairwise_iou(gt, an), rep=50), 4),
       "iou_plus_matrix_matcher_ms": round(timeit(lambda: mt(pairwise_iou(gt, an)), rep=50), 4)}
n, m = 16, an.shape[0]
out["alg_MB_fused"] = round((2 * 16 * m + 9 * m + 16 * n) / 1e6, 2)
out["GBps_fused"] = round(out["alg_MB_fused"] / out["fused_match_boxes_ms"], 1)
print(json.dumps(out))
