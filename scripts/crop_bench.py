"""Stand-alone time of the mask-target crop of the connected step (bench.Workload's boxes / masks).  D2AMD_LIB_PATH selects the build."""
import sys, torch
sys.path.insert(0, "/root/repo")
import bench
from detectron2_amd.structures import crop_and_resize_batch
w = bench.Workload(torch.device("cuda", 0), torch.bfloat16, "nhwc")
out = bench.connected_forward(w)
samp = out["sample"]
n = w.n_img
mask_boxes = [samp["boxes"][i, :bench.MASK_ROWS] for i in range(n)]
idx = [samp["gt_index"][i, :bench.MASK_ROWS].contiguous() for i in range(n)]
for _ in range(5): crop_and_resize_batch(w.gt_masks, mask_boxes, 28, idx, w.crop_status)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(50): crop_and_resize_batch(w.gt_masks, mask_boxes, 28, idx, w.crop_status)
b.record(); torch.cuda.synchronize()
print("crop_and_resize_batch: %.1f us per call (incl. launch)" % (a.elapsed_time(b) / 50 * 1e3))
