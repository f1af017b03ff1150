"""How does match_boxes scale with M (ground truth) and N (predictions)?"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from detectron2_amd.modeling import Matcher
dev = torch.device("cuda", 0)
anchors = bench.make_anchors().to(dev)
g = torch.Generator().manual_seed(0)
def timeit(fn, rep=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(rep): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / rep * 1e3
for low in (True, False):
    mt = Matcher([0.3, 0.7], [0, -1, 1], allow_low_quality_matches=low)
    for M in (1, 4, 16, 64, 256):
        gt = bench.make_boxes(g, M, 16, 512).to(dev)
        for N in (1032, 67000, 268569):
            print(f"low={low} M={M:4d} N={N:7d}: {timeit(lambda: mt.match_boxes(gt, anchors[:N])):7.1f} us (host+device per call)")
