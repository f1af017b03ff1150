"""Device-side cost of the fused RPN proposal path at the BASELINE configs[1] shapes (2 images)."""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from detectron2_amd.modeling import find_top_rpn_proposals_fused, rpn_select_proposals
dev = torch.device("cuda", 0)
torch.manual_seed(3)
sizes = [201600, 50400, 12600, 3150, 819]
H, W = 800, 1344
A, Lg, D = [], [], []
for l, a in enumerate(sizes):
    s = 32.0 * 2 ** l
    c = torch.rand(a, 2) * torch.tensor([W, H]); wh = s * torch.exp(torch.rand(a, 2) - 0.5)
    A.append(torch.cat([c - wh / 2, c + wh / 2], 1).to(dev))
    Lg.append((torch.randn(2, a) + torch.arange(a) * 1e-7).to(dev))
    D.append((torch.randn(2, a, 4) * torch.tensor([0.2, 0.2, 0.3, 0.3])).to(dev))
hw = [(H, W)] * 2
def t(fn, rep=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(rep): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / rep * 1e3
out = {"select_ms": round(t(lambda: rpn_select_proposals(A, Lg, D, hw, 2000, 0.0)), 4),
       "find_top_rpn_proposals_fused_ms": round(t(lambda: find_top_rpn_proposals_fused(A, Lg, D, hw, 0.7, 2000, 1000, 0.0, True)), 4)}
print(json.dumps(out))
