"""Device time of the mask-head glue kernels (256 x 80 x 28 x 28 bf16), pieces separately."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scripts.microbench import timeit
from detectron2_amd import _C
from detectron2_amd.modeling import mask_rcnn_loss_from_targets
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(0)
for dt in (torch.bfloat16, torch.float32):
    ml = torch.randn(256, 80, 28, 28, generator=g).to(dev).to(dt).requires_grad_(True)
    mc = torch.randint(0, 80, (256,), generator=g).to(dev)
    mg = (torch.rand(256, 28, 28, generator=g) < 0.4).to(dev)
    f = timeit(lambda: mask_rcnn_loss_from_targets(ml, mc, mg))
    loss, _ = mask_rcnn_loss_from_targets(ml, mc, mg)
    def fb():
        l, _ = mask_rcnn_loss_from_targets(ml, mc, mg)
        l.backward()
        ml.grad = None
    fbt = timeit(fb)
    L = _C.lib()
    grad = torch.empty_like(ml)
    one = torch.ones((), device=dev)
    t8 = mg.view(torch.uint8)
    x = ml.detach()
    bk = timeit(lambda: L.d2amd_mask_rcnn_loss_backward(_C.ptr(x), _C.ptr(mc), _C.ptr(t8), _C.ptr(one), 256, 80, 784, _C.dtype_code(x), _C.ptr(grad), _C.stream()))
    print(f"{dt}: forward {f*1e3:.1f} us, forward+backward (autograd) {fbt*1e3:.1f} us, backward kernel alone {bk*1e3:.1f} us ({grad.numel()*grad.element_size()/1e6:.0f} MB written)")
