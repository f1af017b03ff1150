"""detectron2_amd._C_shim on the GPU: the reference's OWN Python wrappers (detectron2/layers/deform_conv.py:16-309:
caller-allocated outputs, zero-filled gradients, scratch `columns` / `ones`) restated call for call against the shim,
compared with this package's DeformConv / ModulatedDeformConv modules."""
import pytest
import torch

import detectron2_amd._C_shim as _C
from detectron2_amd.layers import DeformConv, ModulatedDeformConv

pytestmark = pytest.mark.gpu
DEV = "cuda"


def test_v1_through_the_extension_shaped_calls():
    torch.manual_seed(0)
    n, c, co, h, w = 2, 8, 12, 9, 11
    x = torch.randn(n, c, h, w, device=DEV, requires_grad=True)
    off = torch.randn(n, 18, h, w, device=DEV, requires_grad=True)
    mod = DeformConv(c, co, 3, padding=1).to(DEV)
    y = mod(x, off)
    g = torch.randn_like(y)
    y.backward(g)
    # deform_conv.py:58-104: forward with caller-allocated output and two scratch buffers
    output = x.new_empty(y.shape)
    bufs = [x.new_empty(0), x.new_empty(0)]
    assert _C.deform_conv_forward(x.detach(), mod.weight.detach(), off.detach(), output, bufs[0], bufs[1], 3, 3, 1, 1, 1,
                                  1, 1, 1, 1, 1, 64) == 1
    assert torch.equal(output, y.detach())
    # deform_conv.py:106-141: zero-filled gradients, accumulate semantics
    gi, goff, gw = torch.zeros_like(x), torch.zeros_like(off), torch.zeros_like(mod.weight)
    assert _C.deform_conv_backward_input(x.detach(), off.detach(), g, gi, goff, mod.weight.detach(), bufs[0], 3, 3, 1, 1,
                                         1, 1, 1, 1, 1, 1, 64) == 1
    assert _C.deform_conv_backward_filter(x.detach(), off.detach(), g, gw, bufs[0], bufs[1], 3, 3, 1, 1, 1, 1, 1, 1, 1, 1,
                                          1, 64) == 1
    assert torch.allclose(gi, x.grad, rtol=1e-5, atol=1e-6) and torch.allclose(goff, off.grad, rtol=1e-5, atol=1e-6)
    assert torch.allclose(gw, mod.weight.grad, rtol=1e-5, atol=1e-5)
    before = gw.clone()  # a second call accumulates (scale 0.5), like the reference's addmm_ into gradWeight
    _C.deform_conv_backward_filter(x.detach(), off.detach(), g, gw, bufs[0], bufs[1], 3, 3, 1, 1, 1, 1, 1, 1, 1, 1, 0.5, 64)
    assert torch.allclose(gw, before * 1.5, rtol=1e-5, atol=1e-5)


def test_v2_through_the_extension_shaped_calls():
    torch.manual_seed(1)
    n, c, co, h, w = 2, 8, 8, 7, 10
    x = torch.randn(n, c, h, w, device=DEV, requires_grad=True)
    off = torch.randn(n, 18, h, w, device=DEV, requires_grad=True)
    msk = torch.rand(n, 9, h, w, device=DEV, requires_grad=True)
    mod = ModulatedDeformConv(c, co, 3, padding=1, bias=True).to(DEV)
    with torch.no_grad():
        mod.bias.uniform_(-1, 1)
    y = mod(x, off, msk)
    g = torch.randn_like(y)
    y.backward(g)
    output = x.new_empty(y.shape)
    bufs = [x.new_empty(0), x.new_empty(0)]
    assert _C.modulated_deform_conv_forward(x.detach(), mod.weight.detach(), mod.bias.detach(), bufs[0], off.detach(),
                                            msk.detach(), output, bufs[1], 3, 3, 1, 1, 1, 1, 1, 1, 1, 1, True) is None
    assert torch.equal(output, y.detach())
    gi, goff, gm = torch.zeros_like(x), torch.zeros_like(off), torch.zeros_like(msk)
    gw, gb = torch.zeros_like(mod.weight), torch.zeros_like(mod.bias)
    _C.modulated_deform_conv_backward(x.detach(), mod.weight.detach(), mod.bias.detach(), bufs[0], off.detach(),
                                      msk.detach(), bufs[1], gi, gw, gb, goff, gm, g, 3, 3, 1, 1, 1, 1, 1, 1, 1, 1, True)
    for got, want in ((gi, x.grad), (goff, off.grad), (gm, msk.grad), (gw, mod.weight.grad), (gb, mod.bias.grad)):
        assert torch.allclose(got, want, rtol=1e-5, atol=1e-5)
    assert _C.get_cuda_version().startswith("HIP ") and _C.has_cuda() is False
