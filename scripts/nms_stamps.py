import os, sys, torch
sys.path.insert(0, "/root/repo")
import bench
from detectron2_amd.layers import batched_nms
w = bench.Workload(torch.device("cuda", 0), torch.bfloat16, "nhwc")
b, s, l = w.nms_in[0]
for _ in range(3): batched_nms(b, s, l, 0.7)
os.environ["D2AMD_NMS_STAMPS"] = "1"
k = batched_nms(b, s, l, 0.7)
print("kept", k.numel())
