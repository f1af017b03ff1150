"""pairwise_iou / pairwise_ioa / pairwise_intersection -- mirrors
detectron2/structures/boxes.py:312-377, one fused HIP kernel each (no [N,M,2] temporary).
`Boxes` here is only the thin tensor holder those functions take; the full container class is
out of scope -- any object with a `.tensor` (N,4) attribute is accepted, so the
reference's own `Boxes` works unchanged."""
import torch

from .. import _C


class Boxes:
    def __init__(self, tensor: torch.Tensor):
        if not isinstance(tensor, torch.Tensor):
            tensor = torch.as_tensor(tensor, dtype=torch.float32)
        if tensor.numel() == 0:
            tensor = tensor.reshape((-1, 4)).to(dtype=torch.float32)
        assert tensor.dim() == 2 and tensor.size(-1) == 4, tensor.size()
        self.tensor = tensor

    def area(self):
        box = self.tensor
        return (box[:, 2] - box[:, 0]) * (box[:, 3] - box[:, 1])

    def __len__(self):
        return self.tensor.shape[0]

    @property
    def device(self):
        return self.tensor.device


def _t(b):
    return b if isinstance(b, torch.Tensor) else b.tensor


def _pairwise(boxes1, boxes2, mode):
    b1, b2 = _t(boxes1), _t(boxes2)
    _C.require_gpu(b1, b2, op="pairwise_iou")
    b1 = b1.detach().float().contiguous()
    b2 = b2.detach().float().contiguous()
    n, m = b1.shape[0], b2.shape[0]
    out = torch.empty((n, m), dtype=torch.float32, device=b1.device)
    if n and m:
        with _C.on_device(b1.device):
            _C.check(_C.lib().d2amd_pairwise_iou(_C.ptr(b1), n, _C.ptr(b2), m, mode, _C.ptr(out), _C.stream()))
    return out


def pairwise_intersection(boxes1, boxes2) -> torch.Tensor:
    """Intersection area of all N x M pairs (xmin, ymin, xmax, ymax boxes) -> [N,M]."""
    return _pairwise(boxes1, boxes2, 2)


def pairwise_iou(boxes1, boxes2) -> torch.Tensor:
    """IoU of all N x M pairs -> [N,M]; 0 where the intersection is empty."""
    return _pairwise(boxes1, boxes2, 0)


def pairwise_ioa(boxes1, boxes2) -> torch.Tensor:
    """Intersection over boxes2 area -> [N,M]."""
    return _pairwise(boxes1, boxes2, 1)
