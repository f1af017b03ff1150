"""Device-side cost of the dense-detector selection at the BASELINE configs[3] shapes (RetinaNet R50-FPN, 800 x 1344,
2 images: 2 x 16.1 M class scores) -- kernels via rocprofv3, host-inclusive wall time here."""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from detectron2_amd.modeling import dense_detector_inference_fused, dense_select_predictions
dev = torch.device("cuda", 0)
torch.manual_seed(3)
sizes = [9 * 16800, 9 * 4200, 9 * 1050, 9 * 273, 9 * 77]
N, K = 2, 80
A = [torch.rand(a, 4, device=dev) * 100 for a in sizes]
A = [torch.cat([x[:, :2], x[:, :2] + x[:, 2:] + 8], 1) for x in A]
Lg = [torch.randn(N, a, K, device=dev) * 1.2 - 4.6 for a in sizes]  # prior-probability 0.01 bias
D = [torch.randn(N, a, 4, device=dev) * 0.2 for a in sizes]
def t(fn, rep=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(rep): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / rep * 1e3
def reference_like():  # what the reference launches per image and level, on the same device (torch ops)
    out = []
    for i in range(N):
        for l in range(len(sizes)):
            sc = Lg[l][i].sigmoid()
            keep = sc > 0.05
            s2 = sc[keep]; idx = torch.nonzero(keep)
            k = min(1000, idx.shape[0])
            s3, o = s2.topk(k)
            out.append((s3, idx[o]))
    return out
res = {"dense_select_ms": round(t(lambda: dense_select_predictions(A, Lg, D, 0.05, 1000)), 4),
       "dense_detector_inference_fused_ms": round(t(lambda: dense_detector_inference_fused(A, Lg, D, [(800, 1344)] * N, 0.05, 1000, 0.5, 100)), 4),
       "torch_sigmoid_threshold_nonzero_topk_per_level_ms": round(t(reference_like), 4),
       "scores_bytes_MB": round(sum(x.numel() for x in Lg) * 4 / 1e6, 1)}
print(json.dumps(res))
