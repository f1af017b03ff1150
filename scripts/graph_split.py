"""Where the graphed step's time goes: graph A (RPN half + anchor labelling) alone, graph B (ROI half + backward)
alone, the host-side result read between them, and the full step -- with and without side streams."""
import sys
import time

import torch

sys.path.insert(0, ".")
import bench  # noqa: E402

dev = torch.device("cuda:0")


def t(fn, n=200):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


for overlap in (True, False):
    w = bench.Workload(dev, torch.bfloat16, "nhwc")
    w.overlap = overlap
    for _ in range(3):
        bench.disconnected_step(w)
    g = bench.GraphedStep(w, None)
    a_only = t(lambda: g.ga.replay())
    a_sync = t(lambda: (g.ga.replay(), g.out_a[0]()))
    b_only = t(lambda: g.gb.replay())
    full = t(g)
    print(f"overlap={overlap}: graph A {a_only:.1f} us (with the result read {a_sync:.1f}), graph B {b_only:.1f} us, "
          f"step {full:.1f} us (A + B = {a_only + b_only:.1f})")
