"""`subsample_labels` -- detectron2/modeling/sampling.py:9-54, the last step of anchor / proposal labelling
(rpn.py:307-364, roi_heads.py:257-295).  Same contract: (pos_idx, neg_idx), at most
`int(num_samples * positive_fraction)` random positives, the rest filled with random negatives, fewer if there are not
enough; indices are int64 on the labels' device.

The result is RNG-defined (the reference draws two `torch.randperm`s on the device), so there is no bit-level parity to
hold -- what must hold is the distribution: every subset of the right size equally likely.  This version draws ONE
uniform key per element and takes the smallest keys of each class (a uniform random subset), with ONE host sync (the
two class counts) instead of the reference's two `nonzero` syncs and two device sorts inside `randperm`.  Plain
PyTorch on purpose: there is no arithmetic here to put on the matrix cores, and the random stream is torch's
(`generator` / the global seed), so runs reproduce under `torch.manual_seed` like the reference's."""
import torch

__all__ = ["subsample_labels", "label_and_sample_proposals_fixed"]


def subsample_labels(labels: torch.Tensor, num_samples: int, positive_fraction: float, bg_label: int,
                     generator: torch.Generator = None):
    """labels (N,): -1 ignore, bg_label negative, anything else positive.  Returns (pos_idx, neg_idx)."""
    assert labels.dim() == 1, labels.shape
    n = labels.numel()
    dev = labels.device
    if n == 0:
        e = torch.empty(0, dtype=torch.int64, device=dev)
        return e, e.clone()
    pos_mask = (labels != -1) & (labels != bg_label)
    neg_mask = labels == bg_label
    counts = torch.stack([pos_mask.sum(), neg_mask.sum()]).tolist()  # the one host sync
    num_pos = min(counts[0], int(num_samples * positive_fraction))
    num_neg = min(counts[1], num_samples - num_pos)
    key = torch.rand(n, device=dev, generator=generator)
    two = torch.full((), 2.0, device=dev)  # sorts after every real key
    pos_idx = torch.topk(torch.where(pos_mask, key, two), num_pos, largest=False, sorted=False).indices
    neg_idx = torch.topk(torch.where(neg_mask, key, two), num_neg, largest=False, sorted=False).indices
    return pos_idx, neg_idx


def label_and_sample_proposals_fixed(proposal_boxes, gt_boxes, gt_classes, limits=None, keys=None,
                                     thresholds=(0.5,), labels=(0, 1), batch_size_per_image: int = 512,
                                     positive_fraction: float = 0.25, num_classes: int = 80,
                                     proposal_append_gt: bool = True, generator: torch.Generator = None):
    """`ROIHeads.label_and_sample_proposals` (roi_heads/roi_heads.py:219-295) for a batch with a FIXED output shape and
    no host sync -- d2amd_label_and_sample_proposals (include/d2amd.h), one workgroup per image.

    proposal_boxes: per image a [max_p_i, 4] fp32 HIP tensor (e.g. `finish.gathered[i][0]` of
    `batched_nms_images(..., gather=...)`: rows in keep order, valid up to a count the DEVICE knows);
    limits: per image None or an int64 HIP tensor of up to 4 words -- the image uses min(max_p_i, words) proposals;
    gt_boxes / gt_classes: per image [G_i, 4] fp32 / [G_i] int64 HIP tensors (Matcher thresholds / labels as
    roi_heads.py:176-180 builds them: [0.5] / [0, 1], no low-quality matches);
    keys: per image [max_p_i + G_i] uniform fp32 (default: torch.rand with `generator`).
    Sampling rule: `subsample_labels` above (smallest keys per group), ties by candidate index.

    -> dict of HIP tensors: boxes [N, S, 4], classes [N, S] (class, num_classes = background, -1 = padding),
    gt_index [N, S] (matched ground truth), index [N, S] (candidate index into [proposals[:n]; gt], -1 = padding),
    counts [N, 2] int32 = (positives, rows).  Positives first, then negatives, then padding; S = batch_size_per_image."""
    import ctypes

    from .. import _C

    n_img = len(proposal_boxes)
    assert len(gt_boxes) == n_img and len(gt_classes) == n_img
    dev = proposal_boxes[0].device if n_img else torch.device("cuda")
    S = int(batch_size_per_image)
    out = {"boxes": torch.empty((n_img, S, 4), dtype=torch.float32, device=dev),
           "classes": torch.empty((n_img, S), dtype=torch.int64, device=dev),
           "gt_index": torch.empty((n_img, S), dtype=torch.int64, device=dev),
           "index": torch.empty((n_img, S), dtype=torch.int64, device=dev),
           "counts": torch.empty((n_img, 2), dtype=torch.int32, device=dev)}
    if n_img == 0:
        return out
    imgs = (_C.SampleImage * n_img)()
    hold = []
    for i in range(n_img):
        p = proposal_boxes[i].detach().float().contiguous().reshape(-1, 4)
        g = gt_boxes[i].detach().float().contiguous().reshape(-1, 4)
        c = gt_classes[i].detach().to(torch.int64).contiguous().reshape(-1)
        _C.require_gpu(p, g, c, op="label_and_sample_proposals")
        assert c.shape[0] == g.shape[0], (c.shape, g.shape)
        k = keys[i] if keys is not None else torch.rand(p.shape[0] + g.shape[0], device=dev, generator=generator)
        k = k.detach().float().contiguous().reshape(-1)
        assert k.shape[0] == p.shape[0] + g.shape[0], (k.shape, p.shape, g.shape)
        lim = None if limits is None or limits[i] is None else limits[i].detach().to(torch.int64).contiguous().reshape(-1)
        hold += [p, g, c, k, lim]
        imgs[i].proposals, imgs[i].gt_boxes, imgs[i].gt_classes, imgs[i].keys = (
            _C.ptr(p).value, _C.ptr(g).value, _C.ptr(c).value, _C.ptr(k).value)
        imgs[i].limits = None if lim is None else _C.ptr(lim).value
        imgs[i].max_proposals, imgs[i].num_gt = int(p.shape[0]), int(g.shape[0])
        imgs[i].n_limits = 0 if lim is None else int(lim.shape[0])
    T = len(thresholds)
    thr = (ctypes.c_float * max(T, 1))(*[float(t) for t in thresholds])
    lab = (ctypes.c_int8 * (T + 1))(*[int(v) for v in labels])
    with _C.on_device(dev):
        _C.check(_C.lib().d2amd_label_and_sample_proposals(
            imgs, n_img, thr, lab, T, S, int(S * positive_fraction), int(num_classes), int(bool(proposal_append_gt)),
            _C.ptr(out["boxes"]), _C.ptr(out["classes"]), _C.ptr(out["gt_index"]), _C.ptr(out["index"]),
            _C.ptr(out["counts"]), _C.stream()))
    out["_hold"] = hold  # inputs stay alive until the caller drops the result (the launch is asynchronous)
    return out
