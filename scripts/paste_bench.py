"""Device time of paste_masks_in_image at the SURVEY 8(d) micro shape (100 x 28 x 28 -> 100 x 800 x 1333)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from detectron2_amd.layers import paste_masks_in_image
from scripts.microbench import timeit
import bench
dev = torch.device("cuda", 0)
gen = torch.Generator().manual_seed(1234)
n, H, W = 100, 800, 1333
masks = torch.rand(n, 28, 28, generator=gen).to(dev)
# boxes as in tests/layers/test_mask_ops.py:175-179 scaled to the image
b = torch.rand(n, 4, generator=gen)
x0, y0 = b[:, 0] * W * 0.6, b[:, 1] * H * 0.6
boxes = torch.stack([x0, y0, x0 + 20 + b[:, 2] * W * 0.4, y0 + 20 + b[:, 3] * H * 0.4], 1).to(dev)
for dt in (torch.float32, torch.bfloat16):
    m = masks.to(dt)
    ms = timeit(lambda: paste_masks_in_image(m, boxes, (H, W), 0.5), rep=20)
    by = n * H * W + m.numel() * m.element_size() + 16 * n
    print(f"paste_masks {dt}: {ms * 1e3:.1f} us  {by / 1e6:.1f} MB  {by / 1e9 / (ms / 1e3):.0f} GB/s  frac {by / 1e9 / (ms / 1e3) / 8000:.3f}")
# references for the write bound: the same output filled by torch (fill kernel) and by this kernel with boxes that
# touch no pixel row (every workgroup only writes zeros)
out = torch.empty(n, H, W, dtype=torch.uint8, device=dev)
ms = timeit(lambda: out.zero_(), rep=20)
print(f"torch zero_ of the output: {ms * 1e3:.1f} us  {out.numel() / 1e9 / (ms / 1e3):.0f} GB/s")
dead = boxes.clone(); dead[:, 1] = -500.0; dead[:, 3] = -400.0
ms = timeit(lambda: paste_masks_in_image(masks, dead, (H, W), 0.5), rep=20)
print(f"paste_masks, all boxes above the image (zero fill only): {ms * 1e3:.1f} us")
