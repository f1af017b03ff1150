"""Where does bitmask_crop_kernel spend its time?  Variations of the bench's mask-target inputs."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from detectron2_amd.structures import BitMasks, crop_and_resize_batch

def timeit(fn, rep=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(rep): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / rep * 1e3

dev = torch.device("cuda", 0)
w = bench.Workload(dev, torch.bfloat16, "nhwc")
boxes = [b.tensor for b in w.mask_lists]
def run(masks, bxs, idx): return lambda: crop_and_resize_batch(masks, bxs, 28, idx, w.crop_status)
print("bench inputs            us", round(timeit(run(w.gt_masks, boxes, w.fg_gt_index)), 1))
zero = [BitMasks(torch.zeros_like(m.tensor)) for m in w.gt_masks]
print("all-zero masks          us", round(timeit(run(zero, boxes, w.fg_gt_index)), 1))
ones = [BitMasks(torch.ones_like(m.tensor)) for m in w.gt_masks]
print("all-one masks           us", round(timeit(run(ones, boxes, w.fg_gt_index)), 1))
small = [torch.cat([b[:, :2], b[:, :2] + 20], 1) for b in boxes]
print("boxes 20x20 (1 sample)  us", round(timeit(run(w.gt_masks, small, w.fg_gt_index)), 1))
mid = [torch.cat([b[:, :2] * 0.5, b[:, :2] * 0.5 + 100], 1) for b in boxes]
print("boxes 100x100 (16 samp) us", round(timeit(run(w.gt_masks, mid, w.fg_gt_index)), 1))
big = [torch.cat([b[:, :2] * 0.2, b[:, :2] * 0.2 + 600], 1) for b in boxes]
print("boxes 600x600 (484)     us", round(timeit(run(w.gt_masks, big, w.fg_gt_index)), 1))
sz = [(b[:, 2] - b[:, 0]).clamp(min=1) * (b[:, 3] - b[:, 1]).clamp(min=1) for b in boxes]
print("box side percentiles", torch.cat(sz).sqrt().quantile(torch.tensor([.1, .5, .9, .99], device=dev)).tolist())
one = [boxes[0][:1], boxes[1][:0]]
print("one box                 us", round(timeit(run(w.gt_masks, one, [w.fg_gt_index[0][:1], w.fg_gt_index[1][:0]])), 1))
