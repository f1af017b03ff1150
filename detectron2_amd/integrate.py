"""Level-1 integration as ONE call: bind an imported `detectron2` package's hot-path entry points to
this library -- the layer modules AND the fused callers either side of them -- without touching a model's code:

    import detectron2, detectron2_amd.integrate as integrate
    patched = integrate.patch(detectron2)          # before build_model(cfg) (classes are swapped), or with models=[m]
    ...
    patched.undo()                                 # (also a context manager)

What is bound, and what it replaces (reference file:line under the Detectron2 tree):

  names re-exported by `detectron2.layers` (layers/__init__.py:2-16)      -> detectron2_amd.layers.*
      ROIAlign, roi_align, ROIAlignRotated, roi_align_rotated, DeformConv, ModulatedDeformConv, deform_conv,
      modulated_deform_conv, nms, batched_nms, nms_rotated, batched_nms_rotated, paste_masks_in_image,
      pairwise_iou_rotated
  structures/boxes.py:312-377 pairwise_iou / pairwise_ioa                  -> the fused IoU kernels; under a `Matcher`
      the matrix is never written: pairwise_iou returns a LAZY matrix, `Matcher.__call__` (modeling/matcher.py:61-127)
      recognises it and runs Matcher.match_boxes (rpn.py:331-353, roi_heads.py:257-295 read nothing else of it);
      any other use of the lazy object materialises the matrix
  modeling/poolers.py:112-263 ROIPooler                                    -> the fused multi-level pooler (class for models
      built afterwards; instances inside `models` are converted in place and restored by undo()); the box head's and the
      mask head's poolers of one iteration chain into the PAIRED backward by themselves
  proposal_generator/rpn.py:482-512 RPN.predict_proposals                  -> find_top_rpn_proposals_fused (decode + per-level
      top-k + clip + NMS + top-k for the batch, one host read)
  roi_heads/fast_rcnn.py:44-170 fast_rcnn_inference                        -> fast_rcnn_inference_fused
  roi_heads/mask_head.py:33-158 mask_rcnn_loss / mask_rcnn_inference       -> the device-side glue (bit masks cropped for the
      whole batch in one launch)

NOT bound: the samplers (`subsample_labels`, sampling.py:9-54, is defined by torch's RNG; this library's sampler is
defined by explicit keys -- the same distribution, other draws: a bound sampler changes WHICH rows are sampled, so it is
offered separately, `samplers=True`).

Product glue only: no CPU path, no fallback -- a tensor that is not on a HIP device raises as everywhere in this package.
"""
import sys
from typing import Iterable, List, Optional

import torch

__all__ = ["patch", "Patched", "LazyIoU"]


class LazyIoU:
    """`pairwise_iou(gt, boxes)` not yet evaluated.  `Matcher.__call__` consumes it without the matrix; anything else
    (`.max(...)`, indexing, torch functions through `.tensor()`) evaluates it once and forwards."""

    def __init__(self, boxes1, boxes2):
        self.boxes1, self.boxes2 = boxes1, boxes2
        self._m = None

    def tensor(self):
        if self._m is None:
            from .structures import Boxes, pairwise_iou

            self._m = pairwise_iou(Boxes(self.boxes1), Boxes(self.boxes2))
        return self._m

    def __getattr__(self, name):  # (only names the object does not have: everything of the matrix)
        return getattr(self.tensor(), name)

    def __getitem__(self, idx):
        return self.tensor()[idx]

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        un = lambda a: a.tensor() if isinstance(a, LazyIoU) else a
        return func(*[un(a) for a in args], **{k: un(v) for k, v in (kwargs or {}).items()})


class Patched:
    """Handle of one `patch()`: `.undo()` restores every binding (names, methods, converted pooler instances)."""

    def __init__(self):
        self._names, self._attrs, self._mods = [], [], []

    def _set(self, obj, name, new):
        self._attrs.append((obj, name, obj.__dict__.get(name, getattr(obj, name))))
        setattr(obj, name, new)

    def undo(self):
        for obj, name, old in reversed(self._attrs):
            setattr(obj, name, old)
        for parent, name, old in reversed(self._mods):
            setattr(parent, name, old)
        self._attrs, self._mods = [], []

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.undo()
        return False


def _rebind_everywhere(h: Patched, pkg: str, name: str, orig, new):
    """Every module of the package that bound `name` to `orig` at import time (from ... import name) gets `new`."""
    for mod in list(sys.modules.values()):
        if mod is None or not getattr(mod, "__name__", "").startswith(pkg):
            continue
        if mod.__dict__.get(name) is orig:
            h._set(mod, name, new)


def _convert_pooler(ref):
    """The reference's ROIPooler instance -> this library's, same configuration (poolers.py:112-204)."""
    from .modeling.poolers import ROIPooler

    lp = ref.level_poolers[0]
    kind = type(lp).__name__
    if kind == "ROIAlign":
        ptype = "ROIAlignV2" if lp.aligned else "ROIAlign"
    elif kind == "ROIAlignRotated":
        ptype = "ROIAlignRotated"
    else:
        return None  # (ROIPool: torchvision's max pooling, on no BASELINE config)
    return ROIPooler(tuple(ref.output_size), [float(p.spatial_scale) for p in ref.level_poolers], int(lp.sampling_ratio),
                     ptype, canonical_box_size=ref.canonical_box_size, canonical_level=ref.canonical_level)


def patch(detectron2=None, models: Iterable = (), layers: bool = True, fused: bool = True, samplers: bool = False,
          only: Optional[Iterable[str]] = None) -> Patched:
    """Bind `detectron2` (default: the imported package) to this library; see the module docstring.  `models`: already
    built models whose ROIPooler instances are converted in place.  `only`: a subset of the fused bindings
    {"matcher", "pooler", "rpn", "box_inference", "mask_head", "dense"} (default: all of them)."""
    want = lambda k: only is None or k in set(only)
    if detectron2 is None:
        import detectron2  # noqa: F811
    pkg = detectron2.__name__
    import importlib

    imp = lambda n: importlib.import_module(pkg + "." + n)
    from . import layers as L
    from . import modeling as M
    from . import structures as S

    h = Patched()
    d2_boxes = imp("structures.boxes")
    D2Boxes = d2_boxes.Boxes
    if layers:
        d2l = imp("layers")
        for name in ("ROIAlign", "roi_align", "ROIAlignRotated", "roi_align_rotated", "DeformConv", "ModulatedDeformConv",
                     "deform_conv", "modulated_deform_conv", "nms", "batched_nms", "nms_rotated", "batched_nms_rotated",
                     "pairwise_iou_rotated"):
            if hasattr(d2l, name) and hasattr(L, name):
                _rebind_everywhere(h, pkg, name, getattr(d2l, name), getattr(L, name))

        def paste_masks_in_image(masks, boxes, image_shape, threshold: float = 0.5):
            return L.paste_masks_in_image(masks, getattr(boxes, "tensor", boxes), image_shape, threshold)

        _rebind_everywhere(h, pkg, "paste_masks_in_image", imp("layers.mask_ops").paste_masks_in_image, paste_masks_in_image)

        def pairwise_ioa(b1, b2):
            return S.pairwise_ioa(S.Boxes(b1.tensor), S.Boxes(b2.tensor))

        _rebind_everywhere(h, pkg, "pairwise_ioa", d2_boxes.pairwise_ioa, pairwise_ioa)
        if not fused:
            def pairwise_iou(b1, b2):
                return S.pairwise_iou(S.Boxes(b1.tensor), S.Boxes(b2.tensor))

            _rebind_everywhere(h, pkg, "pairwise_iou", d2_boxes.pairwise_iou, pairwise_iou)
    if not fused:
        return h

    # ---- IoU + Matcher without the matrix
    if not want("matcher"):
        def pairwise_iou(b1, b2):
            return S.pairwise_iou(S.Boxes(b1.tensor), S.Boxes(b2.tensor))

        if layers:
            _rebind_everywhere(h, pkg, "pairwise_iou", d2_boxes.pairwise_iou, pairwise_iou)

    def pairwise_iou_lazy(b1, b2):
        return LazyIoU(b1.tensor, b2.tensor)

    if want("matcher"):
        _rebind_everywhere(h, pkg, "pairwise_iou", d2_boxes.pairwise_iou, pairwise_iou_lazy)
    d2_matcher = imp("modeling.matcher").Matcher
    ref_call = d2_matcher.__call__

    def matcher_call(self, match_quality_matrix):
        if isinstance(match_quality_matrix, LazyIoU):
            mine = getattr(self, "_d2amd", None)
            if mine is None:
                mine = M.Matcher(list(self.thresholds[1:-1]), list(self.labels), self.allow_low_quality_matches)
                self._d2amd = mine
            q = match_quality_matrix
            if q.boxes1.shape[0] == 0:  # (matcher.py:79-88: no ground truth -> everything unmatched; no kernel)
                return ref_call(self, q.tensor())
            return mine.match_boxes(q.boxes1, q.boxes2)
        return ref_call(self, match_quality_matrix)

    if want("matcher"):
        h._set(d2_matcher, "__call__", matcher_call)

    # ---- ROIPooler: the class (models built from now on) and the instances of `models`
    d2_poolers = imp("modeling.poolers")
    ref_pooler_cls = d2_poolers.ROIPooler
    if want("pooler"):
        _rebind_everywhere(h, pkg, "ROIPooler", ref_pooler_cls, M.ROIPooler)
    for model in (models if want("pooler") else ()):
        for parent in model.modules():
            for name, child in list(parent.named_children()):
                if isinstance(child, ref_pooler_cls):
                    mine = _convert_pooler(child)
                    if mine is not None:
                        mine.to(next(iter(model.parameters())).device)
                        h._mods.append((parent, name, child))
                        setattr(parent, name, mine)

    # ---- RPN: decode + selection + NMS for the batch
    Instances = imp("structures").Instances
    rpn = imp("modeling.proposal_generator.rpn")
    ref_predict = rpn.RPN.predict_proposals

    def predict_proposals(self, anchors, pred_objectness_logits, pred_anchor_deltas, image_sizes):
        tr = self.box2box_transform
        if (type(tr).__name__ != "Box2BoxTransform" or len(anchors) == 0 or anchors[0].tensor.shape[-1] != 4
                or not pred_objectness_logits[0].is_cuda):
            return ref_predict(self, anchors, pred_objectness_logits, pred_anchor_deltas, image_sizes)
        with torch.no_grad():
            props = M.find_top_rpn_proposals_fused(
                [a.tensor for a in anchors], [x.detach() for x in pred_objectness_logits],
                [x.detach() for x in pred_anchor_deltas], [tuple(s) for s in image_sizes], self.nms_thresh,
                self.pre_nms_topk[self.training], self.post_nms_topk[self.training], self.min_box_size, self.training,
                weights=tuple(tr.weights), scale_clamp=tr.scale_clamp)
        out = []
        for p in props:
            res = Instances(tuple(p.image_size))
            res.proposal_boxes = D2Boxes(p.proposal_boxes.tensor)
            res.objectness_logits = p.objectness_logits
            out.append(res)
        return out

    if want("rpn"):
        h._set(rpn.RPN, "predict_proposals", predict_proposals)

    # ---- box head inference
    fr = imp("modeling.roi_heads.fast_rcnn")

    def fast_rcnn_inference(boxes, scores, image_shapes, score_thresh, nms_thresh, topk_per_image):
        dets, rows = M.fast_rcnn_inference_fused(boxes, scores, image_shapes, score_thresh, nms_thresh, topk_per_image)
        out = []
        for d in dets:
            res = Instances(tuple(d.image_size))
            res.pred_boxes = D2Boxes(d.pred_boxes.tensor)
            res.scores = d.scores
            res.pred_classes = d.pred_classes
            out.append(res)
        return out, rows

    if want("box_inference"):
        _rebind_everywhere(h, pkg, "fast_rcnn_inference", fr.fast_rcnn_inference, fast_rcnn_inference)

    # ---- mask head glue
    mh = imp("modeling.roi_heads.mask_head")
    d2_masks = imp("structures.masks")

    class _Row:  # (what mask_rcnn_loss reads of an Instances; reference BitMasks -> this library's, same storage)
        def __init__(self, inst):
            self._n = len(inst)
            self.gt_classes, self.proposal_boxes = inst.gt_classes, inst.proposal_boxes
            gm = inst.gt_masks
            self.gt_masks = S.BitMasks(gm.tensor) if isinstance(gm, d2_masks.BitMasks) and gm.tensor.is_cuda else gm

        def __len__(self):
            return self._n

    def mask_rcnn_loss(pred_mask_logits, instances, vis_period: int = 0):
        if vis_period > 0:  # (the reference's image logging path: unchanged)
            return ref_loss(pred_mask_logits, instances, vis_period)
        return M.mask_rcnn_loss(pred_mask_logits, [_Row(i) for i in instances])

    ref_loss = mh.mask_rcnn_loss
    if want("mask_head"):
        _rebind_everywhere(h, pkg, "mask_rcnn_loss", ref_loss, mask_rcnn_loss)
        _rebind_everywhere(h, pkg, "mask_rcnn_inference", mh.mask_rcnn_inference, M.mask_rcnn_inference)

    # ---- RetinaNet inference: per-level threshold + top-k + decode and the per-class NMS of the batch in one device
    # pipeline (retinanet.py:257-309, dense_detector.py:186-260: a Python loop over images x levels with a nonzero / topk /
    # index chain and one batched_nms per image)
    try:
        rn = imp("modeling.meta_arch.retinanet")
    except ImportError:  # (a trimmed package without the dense detectors)
        rn = None
    if rn is not None and want("dense"):
        ref_forward_inference = rn.RetinaNet.forward_inference

        def forward_inference(self, images, features, predictions):
            tr = self.box2box_transform
            if type(tr).__name__ != "Box2BoxTransform" or len(features) == 0 or not features[0].is_cuda:
                return ref_forward_inference(self, images, features, predictions)
            pred_logits, pred_anchor_deltas = self._transpose_dense_predictions(predictions, [self.num_classes, 4])
            anchors = self.anchor_generator(features)
            dets = M.dense_detector_inference_fused(
                [a.tensor for a in anchors], pred_logits, pred_anchor_deltas, [tuple(sz) for sz in images.image_sizes],
                self.test_score_thresh, self.test_topk_candidates, self.test_nms_thresh, self.max_detections_per_image,
                weights=tuple(tr.weights), scale_clamp=tr.scale_clamp)
            out = []
            for d in dets:
                res = Instances(tuple(d.image_size))
                res.pred_boxes = D2Boxes(d.pred_boxes.tensor)
                res.scores = d.scores
                res.pred_classes = d.pred_classes
                out.append(res)
            return out

        h._set(rn.RetinaNet, "forward_inference", forward_inference)

    if samplers:
        smp = imp("modeling.sampling")

        def subsample_labels(labels, num_samples, positive_fraction, bg_label):
            return M.subsample_labels(labels, num_samples, positive_fraction, bg_label)

        _rebind_everywhere(h, pkg, "subsample_labels", smp.subsample_labels, subsample_labels)
    return h
