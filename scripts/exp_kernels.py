"""Run the hot-path ops a few times each (meant to be run under `rocprofv3 --kernel-trace --stats`,
which gives the per-kernel device durations that host-bound event loops cannot)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from detectron2_amd.layers import batched_nms, nms
from scripts.microbench import nms_inputs
dev = torch.device("cuda", 0)
layout = sys.argv[1] if len(sys.argv) > 1 else "nhwc"
w = bench.Workload(dev, torch.bfloat16, layout)
gen = torch.Generator().manual_seed(7)
b2, s2, i2 = nms_inputs(gen, 20000, 80, dev)
b3, s3, i3 = nms_inputs(gen, 100000, 80, dev)
for it in range(10):
    bench.disconnected_step(w)
    batched_nms(b2, s2, i2, 0.5)
    batched_nms(b3, s3, i3, 0.5)
    nms(b2[:4096], s2[:4096], 0.5)
torch.cuda.synchronize()
