import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from detectron2_amd.layers import batched_nms, batched_nms_images
dev = torch.device("cuda", 0)
w = bench.Workload(dev, torch.bfloat16, "nhwc")
def t(fn, rep=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(rep): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / rep * 1e3
print("loop  ms", round(t(lambda: [batched_nms(b, s, l, 0.7) for b, s, l in w.nms_in]), 4))
print("multi ms", round(t(lambda: batched_nms_images(w.nms_in, 0.7)), 4))
print("one   ms", round(t(lambda: batched_nms(*w.nms_in[0], 0.7)), 4))
