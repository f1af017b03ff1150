"""Device time of the box / mask pooler forward (back-to-back launches between two events).  python scripts/pool_fwd_ab.py [tag]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from scripts.microbench import timeit
tag = sys.argv[1] if len(sys.argv) > 1 else "default"
w = bench.Workload(torch.device("cuda", 0), torch.bfloat16, "nhwc")
feats = [f.detach() for f in w.feats]
out = []
for name, pooler, lists in (("box", w.box_pooler, w.box_lists), ("mask", w.mask_pooler, w.mask_lists)):
    out.append(f"{name}: {timeit(lambda: pooler(feats, lists), rep=50) * 1e3:.1f} us")
print(f"[{tag}] " + " | ".join(out), flush=True)
