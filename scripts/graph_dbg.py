import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from detectron2_amd.modeling import rpn_select_proposals
dev = torch.device("cuda", 0)
w = bench.Workload(dev, torch.bfloat16, "nhwc")
def fn():
    return rpn_select_proposals(w.anchor_levels, w.rpn_logits, w.rpn_deltas, w.image_sizes, 2000, 0.0)
for _ in range(3): out = fn()
torch.cuda.synchronize()
print("eager flags", out[4].item(), torch.isfinite(out[0]).all().item(), out[2].sum().item())
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    fn()
torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
with torch.cuda.graph(g):
    out = fn()
for i in range(4):
    g.replay(); torch.cuda.synchronize()
    print("replay", i, "flags", out[4].item(), torch.isfinite(out[0]).all().item(), out[2].sum().item(), out[1][0, :3].tolist())
