"""Independent fp64 torch-autograd formulations used ONLY by tests to cross-check the oracle's
gradients where the reference itself has no CPU implementation (DCN backward) -- SURVEY 8(c).
Written from the operator definition (deformable sampling + dense contraction), not from the
oracle, so the two can disagree."""
import torch
import torch.nn.functional as F


def dcn_torch(x, offset, weight, mask=None, bias=None, stride=(1, 1), padding=(0, 0),
              dilation=(1, 1), groups=1, deformable_groups=1):
    """Deformable conv v1/v2.  offset channels: 2k = dy, 2k+1 = dx for tap k = i*kw + j."""
    B, C, H, W = x.shape
    Co, Cg, kh, kw = weight.shape
    Ho = (H + 2 * padding[0] - (dilation[0] * (kh - 1) + 1)) // stride[0] + 1
    Wo = (W + 2 * padding[1] - (dilation[1] * (kw - 1) + 1)) // stride[1] + 1
    dg = deformable_groups
    cpg = C // dg
    ys = torch.arange(Ho, dtype=x.dtype, device=x.device) * stride[0] - padding[0]
    xs = torch.arange(Wo, dtype=x.dtype, device=x.device) * stride[1] - padding[1]
    cols = []
    off = offset.view(B, dg, kh * kw, 2, Ho, Wo)
    msk = mask.view(B, dg, kh * kw, Ho, Wo) if mask is not None else None
    xg = x.view(B * dg, cpg, H, W)
    for i in range(kh):
        for j in range(kw):
            k = i * kw + j
            py = ys[None, None, :, None] + i * dilation[0] + off[:, :, k, 0]  # B,dg,Ho,Wo
            px = xs[None, None, None, :] + j * dilation[1] + off[:, :, k, 1]
            # bilinear with zero padding outside [0,H-1]x[0,W-1]: grid_sample, align_corners=True
            gy = 2 * py / max(H - 1, 1) - 1 if H > 1 else py * 0
            gx = 2 * px / max(W - 1, 1) - 1 if W > 1 else px * 0
            if H == 1 or W == 1:
                raise NotImplementedError
            grid = torch.stack([gx, gy], -1).view(B * dg, Ho, Wo, 2)
            v = F.grid_sample(xg, grid, mode="bilinear", padding_mode="zeros", align_corners=True)
            v = v.view(B, dg, cpg, Ho, Wo)
            if msk is not None:
                v = v * msk[:, :, k, None]
            cols.append(v.reshape(B, C, Ho, Wo))
    col = torch.stack(cols, 2)  # B, C, K2, Ho, Wo
    col = col.view(B, groups, (C // groups) * kh * kw, Ho * Wo)
    w = weight.view(groups, Co // groups, Cg * kh * kw)
    out = torch.einsum("gok,bgkl->bgol", w, col).reshape(B, Co, Ho, Wo)
    if bias is not None:
        out = out + bias.view(1, -1, 1, 1)
    return out
