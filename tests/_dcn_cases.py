"""Deformable-convolution cases shared by the golden generator (tests/golden/make_dcn_reference_gpu.py, which runs the
REFERENCE's own kernels on a GPU box) and the tests that consume its output (tests/test_gpu_dcn_reference.py,
tests/test_oracle_golden.py).  Inputs come from a seeded CPU generator, so every process builds the same tensors; the
golden file stores a checksum of each input so that a drift of torch's CPU RNG fails loudly instead of silently."""
import numpy as np
import torch

# name: (seed, B, C, Co, H, W, groups, deformable_groups, stride, pad, dilation, kernel, modulated, offset scale)
SMALL = {
    "v2_base":      (201, 2, 64, 64, 12, 14, 1, 1, 1, 1, 1, 3, True, 1.5),
    "v2_groups":    (202, 2, 128, 64, 9, 11, 2, 1, 1, 1, 1, 3, True, 1.5),
    "v2_dg2":       (203, 1, 128, 96, 10, 9, 1, 2, 1, 1, 1, 3, True, 1.5),
    "v2_g2_dg4":    (204, 1, 256, 64, 7, 9, 2, 4, 1, 1, 1, 3, True, 1.5),
    "v2_stride2":   (205, 2, 64, 64, 17, 19, 1, 1, 2, 1, 1, 3, True, 1.5),
    "v2_dil2":      (206, 1, 64, 32, 16, 15, 1, 1, 1, 2, 2, 3, True, 1.5),
    "v2_nopad":     (207, 3, 64, 64, 9, 9, 1, 1, 1, 0, 1, 3, True, 1.5),
    "v2_tiny":      (208, 1, 64, 64, 4, 5, 1, 1, 1, 1, 1, 3, True, 1.5),
    "v2_c24":       (209, 2, 24, 40, 10, 12, 1, 1, 1, 1, 1, 3, True, 1.5),   # generic (non-MFMA) kernels
    "v2_5x5":       (210, 1, 64, 64, 12, 13, 1, 1, 1, 2, 1, 5, True, 1.5),
    "v2_far":       (211, 2, 64, 64, 13, 15, 1, 1, 1, 1, 1, 3, True, 6.0),   # many samples leave the image
    "v1_base":      (212, 2, 64, 64, 12, 14, 1, 1, 1, 1, 1, 3, False, 1.5),
    "v1_groups":    (213, 2, 128, 64, 9, 11, 2, 1, 1, 1, 1, 3, False, 1.5),
    "v1_dg2":       (214, 1, 128, 96, 10, 9, 1, 2, 1, 1, 1, 3, False, 1.5),
    "v1_stride2":   (215, 2, 64, 64, 17, 19, 1, 1, 2, 1, 1, 3, False, 1.5),
}
# BASELINE configs[4]: the R50 DCNv2 block shapes, 2 images.  Inputs are rounded to `rounding`
# before the reference's fp32 kernels see them, so the 16-bit product paths are compared with exact-input maths.
FULL = {
    "res3": (301, 2, 128, 128, 100, 168),
    "res4": (302, 2, 256, 256, 50, 84),
    "res5": (303, 2, 512, 512, 25, 42),
}
FULL_SAMPLES = 16384   # elements of each full-size tensor kept in the golden file (indices are stored with them)
KEYS = ("out", "grad_input", "grad_offset", "grad_mask", "grad_weight", "grad_bias")


def make_small(name, rounding=None):
    seed, B, C, Co, H, W, groups, dg, stride, pad, dil, k, modulated, off_scale = SMALL[name]
    g = torch.Generator().manual_seed(seed)
    Ho = (H + 2 * pad - (dil * (k - 1) + 1)) // stride + 1
    Wo = (W + 2 * pad - (dil * (k - 1) + 1)) // stride + 1
    q = (lambda t: t.to(rounding).float()) if rounding is not None else (lambda t: t)
    x = q(torch.randn(B, C, H, W, generator=g))
    off = q(torch.randn(B, dg * 2 * k * k, Ho, Wo, generator=g) * off_scale)
    msk = q(torch.sigmoid(torch.randn(B, dg * k * k, Ho, Wo, generator=g))) if modulated else None
    w = q(torch.randn(Co, C // groups, k, k, generator=g) * 0.05)
    bias = q(torch.randn(Co, generator=g)) if modulated else None
    go = q(torch.randn(B, Co, Ho, Wo, generator=g))
    kw = dict(stride=stride, padding=pad, dilation=dil, groups=groups, deformable_groups=dg)
    return dict(x=x, offset=off, mask=msk, weight=w, bias=bias, grad_out=go, kw=kw)


def make_full(name, rounding=torch.bfloat16):
    """offsets ~ N(0, 2^2), mask = sigmoid(N(0, 1)), Kaiming-scaled weights: SURVEY 8(d)'s DCN micro inputs."""
    seed, B, C, Co, H, W = FULL[name]
    g = torch.Generator().manual_seed(seed)
    q = lambda t: t.to(rounding).float()
    x = q(torch.randn(B, C, H, W, generator=g))
    off = q(torch.randn(B, 18, H, W, generator=g) * 2.0)
    msk = q(torch.sigmoid(torch.randn(B, 9, H, W, generator=g)))
    w = q(torch.randn(Co, C, 3, 3, generator=g) * (2.0 / (9 * C)) ** 0.5)
    bias = q(torch.randn(Co, generator=g) * 0.1)
    go = q(torch.randn(B, Co, H, W, generator=g))
    kw = dict(stride=1, padding=1, dilation=1, groups=1, deformable_groups=1)
    return dict(x=x, offset=off, mask=msk, weight=w, bias=bias, grad_out=go, kw=kw)


def input_checksum(case):
    """fp64 sum of |.| over every input tensor: stored in the golden file, recomputed by its readers."""
    return float(sum(case[k].double().abs().sum().item() for k in ("x", "offset", "mask", "weight", "bias", "grad_out")
                     if case[k] is not None))


def sample_indices(name, key, numel):
    """The (deterministic) flat indices of tensor `key` of full-size case `name` that the golden file keeps."""
    if numel <= FULL_SAMPLES:
        return np.arange(numel, dtype=np.int64)
    rng = np.random.default_rng(FULL[name][0] * 16 + KEYS.index(key))
    return np.sort(rng.choice(numel, FULL_SAMPLES, replace=False)).astype(np.int64)


def run_module(fn_v2, fn_v1, case, device, dtype=torch.float32, channels_last=False):
    """Forward + backward through `modulated_deform_conv` / `deform_conv`-shaped callables (the product's, or the
    reference's own autograd Functions).  -> dict of float32 numpy arrays keyed like KEYS."""
    kw = case["kw"]
    a = (kw["stride"], kw["padding"], kw["dilation"], kw["groups"], kw["deformable_groups"])
    t = lambda v: None if v is None else v.to(device=device, dtype=dtype).requires_grad_(True)
    x, off, msk, w, b = (t(case[k]) for k in ("x", "offset", "mask", "weight", "bias"))
    go = case["grad_out"].to(device=device, dtype=dtype)
    if channels_last:
        x = x.detach().contiguous(memory_format=torch.channels_last).requires_grad_(True)
        go = go.contiguous(memory_format=torch.channels_last)
    if msk is not None:
        y = fn_v2(x, off, msk, w, b, *a)
    else:
        y = fn_v1(x, off, w, *a)
    y.backward(go)
    f = lambda v: v.detach().float().cpu().numpy()
    res = dict(out=f(y), grad_input=f(x.grad), grad_offset=f(off.grad), grad_weight=f(w.grad))
    if msk is not None:
        res.update(grad_mask=f(msk.grad), grad_bias=f(b.grad))
    return res
