"""How many ROIs cover each 8 x 8 tile of the pooler backward (CPU only: the bench's synthetic ROI lists).
The tile gather processes a tile's ROI list serially in one workgroup, so the kernel cannot end before its heaviest
tile does:  max ROIs per tile x time per item  vs  total items x time per item / 512 resident workgroups.
usage: python scripts/pool_bwd_tile_load.py"""
import math
import sys
from collections import Counter

import torch

sys.path.insert(0, ".")
import bench  # noqa: E402

g = [bench.image_generator(1234, i) for i in (0, 1)]
for gen in g:  # the draws Workload.__init__ makes before the sampled ROI lists (bench.py:196-217)
    bench.make_boxes(gen, bench.N_GT, 16, 512)
for gen in g:
    n = 8819
    torch.empty(n).uniform_(-0.5, 0.5, generator=gen)
    for _ in range(3):
        torch.empty(n).uniform_(0, 1, generator=gen)
    torch.rand(n, generator=gen)
for gen in g:
    bench.make_boxes(gen, 1016, 16, 600)
box_b = [bench.make_boxes(gen, 512, 16, 600) for gen in g]
mask_b = [bench.make_boxes(gen, 128, 16, 600) for gen in g]


def tiles(blist, tw=8, th=8):
    cnt = Counter()
    for img, boxes in enumerate(blist):
        lv = bench.assign_levels(boxes).numpy()
        for b, l in zip(boxes.numpy(), lv):
            x0, y0, x1, y1 = b / bench.STRIDES[l] - 0.5
            fx0, fx1 = math.floor(max(x0, 0)), math.floor(max(x1, 0)) + 1
            fy0, fy1 = math.floor(max(y0, 0)), math.floor(max(y1, 0)) + 1
            for ty in range(fy0 // th, fy1 // th + 1):
                for tx in range(fx0 // tw, fx1 // tw + 1):
                    cnt[(img, int(l), ty, tx)] += 1
    return cnt


for name, bl, us in (("box head (7x7)", box_b, 71.0), ("mask head (14x14)", mask_b, 47.5)):
    c = tiles(bl)
    total, heaviest = sum(c.values()), max(c.values())
    print(f"{name}: {total} (tile, ROI) items in {len(c)} non-empty tiles; heaviest tile {heaviest} ROIs")
    for l in range(4):
        v = sorted((n for (i, ll, ty, tx), n in c.items() if ll == l), reverse=True)
        print(f"   level p{l + 2}: {len(v):5d} tiles, {sum(v):5d} items, mean {sum(v) / max(len(v), 1):5.1f}, max {v[:4]}")
    per_item = us / heaviest
    print(f"   measured kernel {us} us = {heaviest} x {per_item:.2f} us per item;  balanced over 512 workgroups the same "
          f"items would take {total * per_item / 512:.0f} us")
