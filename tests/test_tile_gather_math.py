"""CPU check of the ALGORITHM behind the tile-gather ROIAlign backward (csrc/roi_pool.hip).

The device kernel never scatters: every 8x8 tile of grad_input gathers, for each ROI whose
footprint touches it, G[y,x] += inv_count * sum_ph sum_pw Wy[y][ph] * Wx[x][pw] * dY[ph,pw], where
Wy[y][ph] is the total bilinear weight the g_h samples of bin `ph` put on pixel row y (the
axis-aligned sampling grid is separable).  This file restates that in numpy -- same per-axis tap
classification as the kernel (and as the reference's bilinear_interpolate_gradient,
ROIAlignRotated_cpu.cpp:64-107 per axis) -- and checks it against the oracle's sample-by-sample
scatter.  It also pins the ROIPooler level-assignment restatement (poolers.py:23-59).
"""
import math

import numpy as np
import pytest

import oracle

f32 = np.float32


def axis_tap(y, size):
    """one axis of the bilinear footprint -> (valid, lo, hi, wlo, whi), all fp32 arithmetic"""
    y = f32(y)
    valid = not (y < f32(-1.0) or y > f32(size))
    if y < 0:
        y = f32(0)
    lo = int(y)
    if lo >= size - 1:
        hi = lo = size - 1
        y = f32(lo)
    else:
        hi = lo + 1
    l = f32(y - f32(lo))
    return valid, lo, hi, f32(f32(1) - l), l


def axis_weights(start, bin_sz, grid, P, size):
    """dense [size, P] matrix of per-axis weights (what the kernel evaluates per tile row)"""
    Wm = np.zeros((size, P), f32)
    for p in range(P):
        for i in range(grid):
            pos = f32(start + f32(p) * bin_sz + f32(f32(i) + f32(.5)) * bin_sz / f32(grid))
            valid, lo, hi, wlo, whi = axis_tap(pos, size)
            if not valid:
                continue
            Wm[lo, p] += wlo
            Wm[hi, p] += whi
    return Wm


def tile_gather_backward(gout, rois, shape, scale, sr, aligned, tile=8):
    N, C, H, W = shape
    K, _, PH, PW = gout.shape
    gin = np.zeros(shape, f32)
    off = f32(0.5) if aligned else f32(0)
    recs = []
    for k in range(K):
        b = int(rois[k, 0])
        sw = f32(rois[k, 1] * f32(scale) - off)
        sh = f32(rois[k, 2] * f32(scale) - off)
        ew = f32(rois[k, 3] * f32(scale) - off)
        eh = f32(rois[k, 4] * f32(scale) - off)
        rw, rh = f32(ew - sw), f32(eh - sh)
        if not aligned:
            rw, rh = max(rw, f32(1)), max(rh, f32(1))
        bw, bh = f32(rw / f32(PW)), f32(rh / f32(PH))
        gh = sr if sr > 0 else int(math.ceil(rh / f32(PH)))
        gw = sr if sr > 0 else int(math.ceil(rw / f32(PW)))
        # conservative footprint rectangle (rows/cols that can receive gradient)
        fy0 = max(int(math.floor(max(sh, 0))), 0)
        fy1 = min(int(math.floor(sh + rh)) + 1, H - 1)
        fx0 = max(int(math.floor(max(sw, 0))), 0)
        fx1 = min(int(math.floor(sw + rw)) + 1, W - 1)
        recs.append((b, sh, sw, bh, bw, gh, gw, fy0, fy1, fx0, fx1))
    for n in range(N):
        for ty in range(0, H, tile):
            for tx in range(0, W, tile):
                for k, (b, sh, sw, bh, bw, gh, gw, fy0, fy1, fx0, fx1) in enumerate(recs):
                    if b != n or gh <= 0 or gw <= 0:
                        continue
                    if fy1 < ty or fy0 >= ty + tile or fx1 < tx or fx0 >= tx + tile:
                        continue
                    Wy = axis_weights(sh, bh, gh, PH, H)[ty:ty + tile]      # [rows, PH]
                    Wx = axis_weights(sw, bw, gw, PW, W)[tx:tx + tile]      # [cols, PW]
                    inv = f32(1) / f32(gh * gw)
                    g = np.einsum("yp,cpq,xq->cyx", Wy, gout[k], Wx).astype(f32) * inv
                    gin[n, :, ty:ty + tile, tx:tx + tile] += g
    return gin


@pytest.mark.parametrize("P,sr,aligned", [(7, 0, True), (14, 2, True), (7, 0, False), (5, 3, True)])
def test_tile_gather_equals_scatter(P, sr, aligned):
    rng = np.random.default_rng(P * 10 + sr)
    N, C, H, W = 2, 3, 21, 30
    rois = []
    for _ in range(12):
        b = rng.integers(0, N)
        x1, y1 = rng.uniform(-20, W * 4), rng.uniform(-20, H * 4)
        w, h = np.exp(rng.uniform(np.log(2), np.log(150), 2))
        rois.append([b, x1, y1, x1 + w, y1 + h])
    rois.append([0, 0, 0, 0, 0])                 # empty box
    rois.append([1, -50, -50, W * 4 + 50, H * 4 + 50])  # larger than the image
    rois.append([0, W * 4 + 10, 5, W * 4 + 30, 40])     # outside
    rois = np.asarray(rois, f32)
    g = rng.standard_normal((len(rois), C, P, P)).astype(f32)
    exp = oracle.roi_align_backward(g, rois, (N, C, H, W), 0.25, sr, aligned)
    got = tile_gather_backward(g, rois, (N, C, H, W), 0.25, sr, aligned)
    assert np.abs(got - exp).max() <= 1e-5 * max(1.0, np.abs(exp).max())


def assign_levels_restated(boxes, min_level, max_level, canonical_size, canonical_level):
    """fp32 restatement of detectron2/modeling/poolers.py:51-59 (what the device kernel evaluates)"""
    out = []
    for x1, y1, x2, y2 in boxes.astype(f32):
        area = f32(f32(x2 - x1) * f32(y2 - y1))
        size = f32(np.sqrt(area))
        r = f32(f32(size / f32(canonical_size)) + f32(1e-8))
        lv = np.floor(f32(f32(canonical_level) + f32(np.log2(r))))
        lv = min(max(lv, min_level), max_level)
        out.append(int(lv) - min_level)
    return np.asarray(out)


def test_level_assignment_restatement_matches_torch():
    import torch

    rng = np.random.default_rng(0)
    wh = np.exp(rng.uniform(np.log(1), np.log(1500), (4000, 2)))
    xy = rng.uniform(0, 500, (4000, 2))
    boxes = np.concatenate([xy, xy + wh], 1).astype(f32)
    boxes[:8, 2:] = boxes[:8, :2] + np.array([[224, 224], [112, 112], [448, 448], [0, 0], [1, 1],
                                              [111.99, 112.01], [896, 896], [223.9, 224.1]], f32)
    t = torch.from_numpy(boxes)
    sizes = torch.sqrt((t[:, 2] - t[:, 0]) * (t[:, 3] - t[:, 1]))
    ref = torch.clamp(torch.floor(4 + torch.log2(sizes / 224 + 1e-8)), min=2, max=5).to(torch.int64) - 2
    got = assign_levels_restated(boxes, 2, 5, 224, 4)
    assert np.array_equal(got, ref.numpy())


def axis_weights_closed_form(start, bin_sz, grid, P, size):
    """the v9 device formulation (csrc/roi_pool.hip axis_weight): a valid sample at y puts
    max(0, 1 - |clamp(y, 0, size - 1) - pix|) on pixel pix; sample spacing bin / grid divided once"""
    Wm = np.zeros((size, P), f32)
    step = f32(f32(bin_sz) / f32(grid))
    pix = np.arange(size, dtype=f32)
    for p in range(P):
        y0 = f32(f32(start) + f32(p) * f32(bin_sz) + f32(0.5) * step)
        for i in range(grid):
            y = f32(y0 + f32(i) * step)
            if y < f32(-1.0) or y > f32(size):
                continue
            yc = min(max(y, f32(0)), f32(size - 1))
            Wm[:, p] += np.maximum(f32(1) - np.abs(yc - pix), f32(0)).astype(f32)
    return Wm


def test_closed_form_axis_weights_equal_the_tap_classification():
    """Both formulations of the per-axis weights agree to fp32 rounding (the backward's bar is 1e-4), including samples
    below 0, beyond the last pixel, outside [-1, size] and degenerate (zero-size) bins."""
    rng = np.random.default_rng(3)
    worst = 0.0
    for _ in range(300):
        size = int(rng.integers(1, 40))
        P = int(rng.integers(1, 15))
        start = f32(rng.uniform(-6, size + 3))
        bin_sz = f32(0.0 if rng.random() < 0.05 else np.exp(rng.uniform(np.log(0.05), np.log(9.0))))
        grid = int(rng.integers(1, 9))
        a = axis_weights(start, bin_sz, grid, P, size)
        b = axis_weights_closed_form(start, bin_sz, grid, P, size)
        worst = max(worst, float(np.abs(a - b).max()))
    assert worst < 2e-5, worst


def _split_16(w, kind):
    """hi + lo split of fp32 weights into two 16-bit values (the MFMA kernel's weight image)"""
    import torch

    dt = torch.bfloat16 if kind == "bf16" else torch.float16
    t = torch.from_numpy(w)
    hi = t.to(dt)
    lo = (t - hi.float()).to(dt)
    return hi.float().numpy(), lo.float().numpy()


@pytest.mark.parametrize("kind,bits", [("bf16", 15), ("f16", 20)])
def test_hi_lo_weight_split_keeps_fp32_accuracy(kind, bits):
    """pool_bwd_mfma_kernel contracts (hi + lo) . dY on the matrix cores with fp32 accumulation: the split weights
    reproduce the fp32 weights to >= `bits` significant bits, so a 32-bin contraction matches the fp32-weight result
    far inside the 16-bit output rounding."""
    rng = np.random.default_rng(11)
    w = (rng.random(4096) * np.exp(rng.uniform(-6, 0, 4096))).astype(f32)  # weights in (0, 1], several decades
    hi, lo = _split_16(w, kind)
    err = np.abs((hi + lo).astype(np.float64) - w)
    floor = 2.0 ** -24 if kind == "f16" else 0.0  # f16: the low part of a small weight is subnormal (spacing 2^-24)
    assert (err <= np.maximum(2.0 ** -bits * w, floor)).all(), (err / w).max()
    dy = np.float32(rng.standard_normal((4096 // 32, 32)))
    ref = (w.reshape(-1, 32).astype(np.float64) * dy).sum(1)
    got = ((hi + lo).reshape(-1, 32).astype(np.float64) * dy).sum(1)
    assert np.abs(got - ref).max() < 2.0 ** -12 * np.abs(ref).max()  # 16-bit outputs round at 2^-9 / 2^-11
