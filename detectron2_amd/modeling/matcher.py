"""Matcher -- mirrors detectron2/modeling/matcher.py:8-127 (same constructor, same `__call__` on an
M x N match-quality matrix, same return dtypes: int64 matches, int8 labels) and adds the fused entry
`match_boxes(gt_boxes, boxes)` = `self(pairwise_iou(gt_boxes, boxes))` that never writes the matrix
(d2amd_match_boxes): the RPN calls it with G x 268,569 anchors per image (proposal_generator/
rpn.py:307-364), the ROI heads with G x ~1,000 proposals (roi_heads/roi_heads.py:257-295);
`match_boxes_batch(gt_boxes_per_image, boxes)` does it for all images of a batch against the same boxes
(the anchors) in one launch per pass (d2amd_match_boxes_batch; the reference loops over the images).
The reference's `assert torch.all(match_quality_matrix >= 0)` is not evaluated (host sync)."""
import ctypes
from typing import List

import torch

from .. import _C


class Matcher:
    def __init__(self, thresholds: List[float], labels: List[int], allow_low_quality_matches: bool = False):
        thresholds = thresholds[:]
        assert thresholds[0] > 0
        thresholds.insert(0, -float("inf"))
        thresholds.append(float("inf"))
        assert all([low <= high for (low, high) in zip(thresholds[:-1], thresholds[1:])])
        assert all([l in [-1, 0, 1] for l in labels])
        assert len(labels) == len(thresholds) - 1
        self.thresholds = thresholds
        self.labels = labels
        self.allow_low_quality_matches = allow_low_quality_matches
        t = self.thresholds[1:-1]
        self._thr = (ctypes.c_float * max(len(t), 1))(*t)
        self._lab = (ctypes.c_int8 * len(labels))(*labels)
        self._T = len(t)

    def _out(self, n, device):
        return (torch.empty(n, dtype=torch.int64, device=device), torch.empty(n, dtype=torch.int8, device=device))

    def __call__(self, match_quality_matrix):
        assert match_quality_matrix.dim() == 2
        _C.require_gpu(match_quality_matrix, op="Matcher")
        q = match_quality_matrix.detach().float().contiguous()
        m, n = q.shape
        matches, labels = self._out(n, q.device)
        L = _C.lib()
        with _C.on_device(q.device):
            ws_bytes = L.d2amd_matcher_workspace_bytes(m)
            ws = torch.empty(ws_bytes, dtype=torch.uint8, device=q.device)
            _C.check(L.d2amd_match_quality_matrix(_C.ptr(q), m, n, self._thr, self._lab, self._T,
                                                  int(self.allow_low_quality_matches), _C.ptr(matches),
                                                  _C.ptr(labels), _C.ptr(ws), ws_bytes, _C.stream()))
        return matches, labels

    def match_boxes(self, gt_boxes, boxes):
        """== self(pairwise_iou(gt_boxes, boxes)) without the matrix.  Arguments: Boxes or Tensor[.,4]."""
        g = gt_boxes if isinstance(gt_boxes, torch.Tensor) else gt_boxes.tensor
        b = boxes if isinstance(boxes, torch.Tensor) else boxes.tensor
        _C.require_gpu(g, b, op="Matcher.match_boxes")
        g = g.detach().float().contiguous()
        b = b.detach().float().contiguous()
        assert g.dim() == 2 and g.shape[1] == 4 and b.dim() == 2 and b.shape[1] == 4
        m, n = g.shape[0], b.shape[0]
        matches, labels = self._out(n, b.device)
        L = _C.lib()
        with _C.on_device(b.device):
            ws_bytes = L.d2amd_matcher_workspace_bytes(m)
            ws = torch.empty(ws_bytes, dtype=torch.uint8, device=b.device)
            _C.check(L.d2amd_match_boxes(_C.ptr(g), m, _C.ptr(b), n, self._thr, self._lab, self._T,
                                         int(self.allow_low_quality_matches), _C.ptr(matches), _C.ptr(labels),
                                         _C.ptr(ws), ws_bytes, _C.stream()))
        return matches, labels

    def match_boxes_batch(self, gt_boxes_list, boxes):
        """[self(pairwise_iou(g, boxes)) for g in gt_boxes_list] for the whole batch: the images' ground truth against the
        SAME boxes (the RPN's anchors, rpn.py:331-353) in one launch per pass.
        -> (matches [len(gt_boxes_list), N] int64, labels [len(gt_boxes_list), N] int8); row i is image i's result."""
        b = boxes if isinstance(boxes, torch.Tensor) else boxes.tensor
        gs = [(g if isinstance(g, torch.Tensor) else g.tensor) for g in gt_boxes_list]
        _C.require_gpu(b, *gs, op="Matcher.match_boxes_batch")
        b = b.detach().float().contiguous()
        gs = [g.detach().float().contiguous().reshape(-1, 4) for g in gs]
        assert b.dim() == 2 and b.shape[1] == 4
        cnt, n = len(gs), b.shape[0]
        matches = torch.empty((cnt, n), dtype=torch.int64, device=b.device)
        labels = torch.empty((cnt, n), dtype=torch.int8, device=b.device)
        if cnt == 0 or n == 0:
            return matches, labels
        L = _C.lib()
        ms = (ctypes.c_int * cnt)(*[int(g.shape[0]) for g in gs])
        ptrs = (ctypes.c_void_p * cnt)(*[g.data_ptr() if g.shape[0] else None for g in gs])
        with _C.on_device(b.device):
            ws_bytes = L.d2amd_match_boxes_batch_workspace_bytes(ms, cnt)
            ws = torch.empty(ws_bytes, dtype=torch.uint8, device=b.device)
            _C.check(L.d2amd_match_boxes_batch(ptrs, ms, cnt, _C.ptr(b), n, self._thr, self._lab, self._T,
                                               int(self.allow_low_quality_matches), _C.ptr(matches), _C.ptr(labels),
                                               _C.ptr(ws), ws_bytes, _C.stream()))
        return matches, labels
