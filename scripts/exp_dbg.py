import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = torch.device("cuda", 0)
w = bench.Workload(dev, torch.bfloat16, "nhwc")
for name, pooler, lists, grad in (("box7", w.box_pooler, w.box_lists, w.gbox),):
    y = pooler(w.feats, lists)
    for blk in sys.argv[1:]:
        os.environ["D2AMD_DBG_BLOCK"] = blk
        for rep in range(2):
            torch.autograd.grad([y], w.feats, [grad], retain_graph=True)
            torch.cuda.synchronize()
