"""`subsample_labels` -- detectron2/modeling/sampling.py:9-54, the last step of anchor / proposal labelling
(rpn.py:287-364, roi_heads.py:181-295) -- on the device: d2amd_subsample_labels
(csrc/subsample.hip; include/d2amd.h).

The reference's result is RNG-defined (two `torch.randperm` draws behind two `nonzero` host syncs), so there is no
bit-level parity with it to hold; what must hold is the contract -- at most `int(num_samples * positive_fraction)`
positives, the rest filled with negatives, fewer if there are not enough, int64 indices on the labels' device -- and
the distribution: every subset of the right size equally likely.  Rule here (the same in label_sample.hip and
oracle/sampling.py): ONE uniform key per element (torch.rand: `generator` / the global seed, so runs reproduce under
`torch.manual_seed` like the reference's), the smallest keys of each group, ties towards the lower index; GIVEN the
keys the result is exact and checked bit for bit against the oracle (tests/test_gpu_subsample.py).

Three entries:
  subsample_labels(labels, ...)          the reference's signature and variable-size result: ONE host read (the two
                                         sample sizes) instead of two nonzero syncs
  subsample_labels_batch(labels[N,n])    fixed shape, no host sync: index lists padded with -1 + counts on the device
  subsample_anchor_labels_(labels[N,n])  RPN._subsample_labels (rpn.py:287-305) for the batch: the int8 label vectors
                                         rewritten to -1 / 0 / 1 in place, no host sync (a captured step can hold it)
There is no CPU path: a CPU tensor raises NotImplementedError like every op of this package."""
import torch

from .. import _C

__all__ = ["subsample_labels", "subsample_labels_batch", "subsample_anchor_labels_", "label_and_sample_proposals_fixed",
           "DeviceKeyGenerator"]


class DeviceKeyGenerator:
    """Uniform sampling keys from a generator whose state lives ON THE DEVICE (d2amd_uniform_keys: Philox4x32-10).
    `torch.rand` inside a captured HIP graph costs two host-side fill launches in front of every replay (torch feeds
    its generator's seed and offset that way); this one advances its offset in the kernel, so a replayed step draws
    fresh keys with nothing in front of the graph.  Seeded from torch's generator (`generator` or the global one), so
    `torch.manual_seed` reproduces a run; the stream itself is this class's, not torch's."""

    def __init__(self, device, generator: torch.Generator = None, seed: int = None):
        if seed is None:
            seed = int(torch.randint(0, 2 ** 62, (1,), generator=generator).item())
        self.state = torch.tensor([seed, 0, 0], dtype=torch.int64, device=device)

    def uniform(self, *shape) -> torch.Tensor:
        """fp32 tensor of `shape`, values in [0, 1)"""
        out = torch.empty(*shape, dtype=torch.float32, device=self.state.device)
        with _C.on_device(out.device):
            _C.check(_C.lib().d2amd_uniform_keys(_C.ptr(self.state), _C.ptr(out), out.numel(), _C.stream()))
        return out


def _subsample_device(labels2d, keys, num_samples, max_pos, bg_label, want_idx, labels_out):
    N, n = labels2d.shape
    dev = labels2d.device
    pos = torch.empty((N, max_pos), dtype=torch.int64, device=dev) if want_idx else None
    neg = torch.empty((N, num_samples), dtype=torch.int64, device=dev) if want_idx else None
    counts = torch.empty((N, 2), dtype=torch.int32, device=dev)
    L = _C.lib()
    with _C.on_device(dev):
        ws_bytes = L.d2amd_subsample_labels_workspace_bytes(N, n, num_samples, max_pos)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        _C.check(L.d2amd_subsample_labels(_C.ptr(labels2d), labels2d.element_size(), N, n, _C.ptr(keys), num_samples,
                                          max_pos, int(bg_label), _C.ptr(pos), _C.ptr(neg), _C.ptr(counts),
                                          _C.ptr(labels_out), _C.ptr(ws), ws_bytes, _C.stream()))
    return pos, neg, counts


def _prep(labels, keys, generator, op):
    _C.require_gpu(labels, keys, op=op)
    if labels.dtype not in (torch.int8, torch.int64):
        labels = labels.to(torch.int64)
    labels = labels.detach().contiguous()
    if keys is None:
        keys = torch.rand(labels.shape, device=labels.device, generator=generator)
    keys = keys.detach().float().contiguous()
    assert keys.shape == labels.shape, (keys.shape, labels.shape)
    return labels, keys


def subsample_labels_batch(labels: torch.Tensor, num_samples: int, positive_fraction: float, bg_label: int,
                           keys: torch.Tensor = None, generator: torch.Generator = None):
    """labels [N, n] (int8 or int64): `subsample_labels` of every row, fixed shape, nothing waits for the host.
    -> (pos_idx [N, int(num_samples * positive_fraction)], neg_idx [N, num_samples]) int64 padded with -1 (valid
    entries first, ascending (key, index)), counts [N, 2] int32 = (sampled positives, sampled negatives)."""
    assert labels.dim() == 2, labels.shape
    labels, keys = _prep(labels, keys, generator, "subsample_labels_batch")
    return _subsample_device(labels, keys, int(num_samples), int(num_samples * positive_fraction), bg_label, True, None)


def subsample_anchor_labels_(labels: torch.Tensor, num_samples: int, positive_fraction: float,
                             keys: torch.Tensor = None, generator: torch.Generator = None):
    """RPN._subsample_labels (rpn.py:287-305) for a batch: labels [N, n] int8 in {-1, 0, 1} (Matcher labels) are
    rewritten IN PLACE to -1 except at the sampled positives (1) and negatives (0).  Returns (labels, counts [N, 2]
    int32 on the device); no host sync."""
    assert labels.dim() == 2 and labels.dtype == torch.int8 and labels.is_contiguous(), (labels.shape, labels.dtype)
    lab, keys = _prep(labels, keys, generator, "subsample_anchor_labels_")
    _, _, counts = _subsample_device(lab, keys, int(num_samples), int(num_samples * positive_fraction), 0, False, lab)
    return labels, counts


def subsample_labels(labels: torch.Tensor, num_samples: int, positive_fraction: float, bg_label: int,
                     generator: torch.Generator = None, keys: torch.Tensor = None):
    """labels (N,): -1 ignore, bg_label negative, anything else positive.  Returns (pos_idx, neg_idx) int64 -- the
    reference's signature (modeling/sampling.py:9-54); one host read for the two result sizes."""
    assert labels.dim() == 1, labels.shape
    _C.require_gpu(labels, op="subsample_labels")
    if labels.numel() == 0:
        e = torch.empty(0, dtype=torch.int64, device=labels.device)
        return e, e.clone()
    pos, neg, counts = subsample_labels_batch(labels[None], num_samples, positive_fraction, bg_label, keys=None if
                                              keys is None else keys[None], generator=generator)
    num_pos, num_neg = counts[0].tolist()  # the one host sync (the result's shape)
    return pos[0, :num_pos], neg[0, :num_neg]


def label_and_sample_proposals_fixed(proposal_boxes, gt_boxes, gt_classes, limits=None, keys=None, limit_stride=1,
                                     thresholds=(0.5,), labels=(0, 1), batch_size_per_image: int = 512,
                                     positive_fraction: float = 0.25, num_classes: int = 80,
                                     proposal_append_gt: bool = True, generator: torch.Generator = None,
                                     head_rows: int = 0, keygen: "DeviceKeyGenerator" = None):
    """`ROIHeads.label_and_sample_proposals` (roi_heads/roi_heads.py:219-295) for a batch with a FIXED output shape and
    no host sync -- d2amd_label_and_sample_proposals (include/d2amd.h), one workgroup per image.

    proposal_boxes: per image a [max_p_i, 4] fp32 HIP tensor (e.g. `finish.gathered[i][0]` of
    `batched_nms_images(..., gather=...)`: rows in keep order, valid up to a count the DEVICE knows);
    limits: per image None or an int64 HIP tensor -- the image uses min(max_p_i, the limit words) proposals; the words
    are limits[0], limits[limit_stride], ... (up to 4).  The NMS result row of find_top_rpn_proposals_fused is
    {kept, flags, finite, 0}: pass the row with limit_stride = 2 (kept and finite; `DeviceProposals.limits`);
    gt_boxes / gt_classes: per image [G_i, 4] fp32 / [G_i] int64 HIP tensors (Matcher thresholds / labels as
    roi_heads.py:176-180 builds them: [0.5] / [0, 1], no low-quality matches);
    keys: per image [max_p_i + G_i] uniform fp32 (default: torch.rand with `generator`);
    keygen (instead of keys): a `DeviceKeyGenerator` -- the keys are drawn INSIDE the sampler's kernel from its
    device-resident state (image i's keys are outputs sum_{j<i}(max_p_j + G_j) ... of one `keygen.uniform` draw, and the
    generator advances as by one draw): no key launch, and in a captured step no cross-stream wait for one.
    Sampling rule: `subsample_labels` above (smallest keys per group), ties by candidate index.

    -> dict of HIP tensors: boxes [N, S, 4], classes [N, S] (class, num_classes = background, -1 = padding),
    gt_index [N, S] (matched ground truth), index [N, S] (candidate index into [proposals[:n]; gt], -1 = padding),
    counts [N, 2] int32 = (positives, rows).  Positives first, then negatives, then padding; S = batch_size_per_image.
    Also "rois" [N * S, 5] (pooler format: image, x1, y1, x2, y2 -- `ROIPooler.pool_rois` takes it as it is) and, with
    head_rows > 0, "head_rois" [N * head_rows, 5]: the first head_rows rows of every image (the mask head's), and
    "head_classes" [N, head_rows]: their classes, contiguous."""
    import ctypes

    n_img = len(proposal_boxes)
    assert len(gt_boxes) == n_img and len(gt_classes) == n_img
    dev = proposal_boxes[0].device if n_img else torch.device("cuda")
    S = int(batch_size_per_image)
    out = {"boxes": torch.empty((n_img, S, 4), dtype=torch.float32, device=dev),
           "classes": torch.empty((n_img, S), dtype=torch.int64, device=dev),
           "gt_index": torch.empty((n_img, S), dtype=torch.int64, device=dev),
           "index": torch.empty((n_img, S), dtype=torch.int64, device=dev),
           "counts": torch.empty((n_img, 2), dtype=torch.int32, device=dev),
           "rois": torch.empty((n_img * S, 5), dtype=torch.float32, device=dev)}
    H = int(head_rows)
    assert 0 <= H <= S
    if H:
        out["head_rois"] = torch.empty((n_img * H, 5), dtype=torch.float32, device=dev)
        out["head_classes"] = torch.empty((n_img, H), dtype=torch.int64, device=dev)
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
        if keys is None and keygen is not None:
            k = None
        else:
            k = keys[i] if keys is not None else torch.rand(p.shape[0] + g.shape[0], device=dev, generator=generator)
            k = k.detach().float().contiguous().reshape(-1)
            assert k.shape[0] == p.shape[0] + g.shape[0], (k.shape, p.shape, g.shape)
        lim = None if limits is None or limits[i] is None else limits[i].detach().reshape(-1)
        assert lim is None or (lim.dtype == torch.int64 and lim.is_contiguous()), "limits: contiguous int64 words"
        hold += [p, g, c, k, lim]
        imgs[i].proposals, imgs[i].gt_boxes, imgs[i].gt_classes, imgs[i].keys = (
            _C.ptr(p).value, _C.ptr(g).value, _C.ptr(c).value, None if k is None else _C.ptr(k).value)
        imgs[i].limits = None if lim is None else _C.ptr(lim).value
        imgs[i].max_proposals, imgs[i].num_gt = int(p.shape[0]), int(g.shape[0])
        st = max(int(limit_stride), 1)
        imgs[i].n_limits = 0 if lim is None else min(4, (int(lim.shape[0]) + st - 1) // st)
        imgs[i].limit_stride = st
    T = len(thresholds)
    thr = (ctypes.c_float * max(T, 1))(*[float(t) for t in thresholds])
    lab = (ctypes.c_int8 * (T + 1))(*[int(v) for v in labels])
    with _C.on_device(dev):
        _C.check(_C.lib().d2amd_label_and_sample_proposals(
            imgs, n_img, thr, lab, T, S, int(S * positive_fraction), int(num_classes), int(bool(proposal_append_gt)),
            _C.ptr(out["boxes"]), _C.ptr(out["classes"]), _C.ptr(out["gt_index"]), _C.ptr(out["index"]),
            _C.ptr(out["counts"]), _C.ptr(out["rois"]), _C.ptr(out.get("head_rois")), _C.ptr(out.get("head_classes")), H,
            _C.ptr(keygen.state) if (keys is None and keygen is not None) else None, _C.stream()))
    out["_hold"] = hold  # inputs stay alive until the caller drops the result (the launch is asynchronous)
    return out
