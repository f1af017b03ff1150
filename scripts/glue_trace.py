"""Which Python lines launch the torch (non-libd2amd) kernels of the maskrcnn_train step?  Eager step under
torch.profiler with stacks; prints the aten ops that ran device kernels, grouped by the innermost repo frames."""
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, ".")
import bench  # noqa: E402

dev = torch.device("cuda:0")
w = bench.Workload(dev, torch.bfloat16, "nhwc")
for _ in range(3):
    bench.disconnected_step(w)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    for _ in range(3):
        bench.disconnected_step(w)
    torch.cuda.synchronize()
rows = []
for e in prof.key_averages(group_by_stack_n=12):
    dt = getattr(e, "device_time_total", 0) or getattr(e, "cuda_time_total", 0)
    if dt <= 0 or not e.key.startswith("aten::"):
        continue
    st = [f for f in e.stack if "/repo/" in f or "bench.py" in f][:3]
    rows.append((dt / 3, e.count / 3, e.key, " <- ".join(s.split("/repo/")[-1] for s in st)))
rows.sort(reverse=True)
for dt, c, k, st in rows[:40]:
    print(f"{dt:8.1f} us/step  {c:4.1f}x  {k:28s} {st}")
