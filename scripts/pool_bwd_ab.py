"""Device time of the box / mask pooler backward (back-to-back launches between two events) under the current
environment (D2AMD_POOL_NOQUEUE, D2AMD_POOL_QTHR_*, D2AMD_BWD_CFG ...).  python scripts/pool_bwd_ab.py [tag]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from detectron2_amd import _C

tag = sys.argv[1] if len(sys.argv) > 1 else "default"
w = bench.Workload(torch.device("cuda", 0), torch.bfloat16, "nhwc")
out = []
for name, pooler, lists, grad in (("box", w.box_pooler, w.box_lists, w.gbox), ("mask", w.mask_pooler, w.mask_lists, w.gmask)):
    y = pooler(w.feats, lists)
    for _ in range(5):
        torch.autograd.grad([y], w.feats, [grad], retain_graph=True)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    rep = 50
    a.record()
    for _ in range(rep):
        torch.autograd.grad([y], w.feats, [grad], retain_graph=True)
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / rep
    # kernel-level timing recorded by the library on the launch streams
    _C.lib().d2amd_timing_enable(15)
    for _ in range(20):
        torch.autograd.grad([y], w.feats, [grad], retain_graph=True)
    torch.cuda.synchronize()
    ks = {}
    import ctypes
    r = 7 if name == "box" else 14
    for k in (f"pool_bwd_staged_r{r}", f"pool_bwd_fine_r{r}", f"pool_bwd_coarse_r{r}"):
        tot, n = ctypes.c_double(0), ctypes.c_int(0)
        if _C.lib().d2amd_timing_read(k.encode(), ctypes.byref(tot), ctypes.byref(n)) == 0 and n.value:
            ks[k] = round(tot.value / n.value, 4)
    _C.lib().d2amd_timing_enable(0)
    out.append(f"{name}: op {ms:.4f} ms  kernels {ks}")
print(f"[{tag}] " + " | ".join(out), flush=True)
