"""Debug: the res3 NHWC DCN block (bf16 / f16) repeated N times under the current D2AMD_DCN_SIDE_MODE; every run's
outputs against the first run of the process (the data-gradient path is deterministic up to the fp32 atomics of
d(offset) / d(mask) with csplit > 1)."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tests")
import _dcn_cases as dc
from detectron2_amd import layers
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
for dt in (torch.bfloat16, torch.float16):
    case = dc.make_full("res3", rounding=dt)
    first = None
    worst = {}
    for it in range(n):
        out = dc.run_module(layers.modulated_deform_conv, layers.deform_conv, case, "cuda", dt, True)
        if first is None:
            first = out
            continue
        for k in out:
            d = np.abs(out[k].astype(np.float64) - first[k].astype(np.float64))
            rel = float((d / (np.abs(first[k].astype(np.float64)) + 1e-3)).max())
            if rel > worst.get(k, (0, 0))[0]:
                worst[k] = (rel, int((d > 0).sum()), it)
            if (d > 0).any() and len(sys.argv) > 2:
                idx = np.argwhere(d > 0)
                print("   it", it, k, "n", len(idx), [(tuple(int(q) for q in i), float(out[k][tuple(i)]), float(first[k][tuple(i)])) for i in idx[:10]])
    print(str(dt), {k: (round(v[0], 5), v[1], v[2]) for k, v in worst.items()})
