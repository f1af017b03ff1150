"""The reference's OWN model code -- GeneralizedRCNN, RetinaNet, DeformBottleneckBlock and everything under them (rpn.py,
roi_heads.py, fast_rcnn.py, mask_head.py, poolers.py, postprocessing.py, resnet.py, fpn.py ...) -- imported UNCHANGED
from the byte-compiled package tree (oracle/_ref/pkg, oracle/build_ref.py: build_pkg) and run on two operator backends:

  "product"    torchvision.ops.{roi_align, nms, batched_nms}, detectron2._C (the five DCN entry points),
               layers.paste_masks_in_image, structures.pairwise_iou / pairwise_ioa  ->  detectron2_amd (HIP kernels)
  "reference"  the same names -> plain-torch restatements of torchvision's ops (below; greedy NMS on the host via the
               C oracle), the reference's own deform_conv kernels compiled as HIP (oracle/_ref/_d2ref_C.so), and the
               reference's own pure-torch mask paste / IoU

TEST INFRASTRUCTURE (SURVEY 8 row g).  Nothing under detectron2_amd/ imports this.

Third-party packages the image lacks (fvcore, iopath, yacs, omegaconf, termcolor, pycocotools, torchvision, cv2, ...)
are served by a meta-path finder that fabricates permissive stub modules; the few names the model code really RUNS are
given small real implementations here, each citing what it stands in for."""
import contextlib
import importlib
import importlib.abc
import importlib.machinery
import math
import os
import sys
import types

import numpy as np
import torch
import torch.nn.functional as F

_HERE = os.path.dirname(os.path.abspath(__file__))
_PKG = os.path.join(os.path.dirname(_HERE), "oracle", "_ref", "pkg")
STUB_TOPS = ("fvcore", "iopath", "yacs", "omegaconf", "termcolor", "pycocotools", "torchvision", "cv2", "hydra", "lvis",
             "shapely", "panopticapi", "cityscapesscripts", "tensorboard", "mmdet", "mmcv", "onnx", "caffe2", "timm",
             "fairscale", "psutil_stub")


# ---------------------------------------------------------------------------------------------- permissive stubs
class _DummyMeta(type):
    def __getattr__(cls, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return _dummy_class(name)

    def __iter__(cls):
        return iter(())


def _dummy_class(name="Dummy"):
    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        if len(a) == 1 and callable(a[0]) and not k:
            return a[0]  # used as a decorator
        return _dummy_class(name)()

    def __getattr__(self, n):
        if n.startswith("__"):
            raise AttributeError(n)
        return _dummy_class(n)()

    return _DummyMeta(name, (), {"__init__": __init__, "__call__": __call__, "__getattr__": __getattr__,
                                 "__iter__": lambda self: iter(()), "__len__": lambda self: 0,
                                 "__enter__": lambda self: self, "__exit__": lambda self, *e: False})


class _StubModule(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        v = _dummy_class(name)
        setattr(self, name, v)
        return v


class _StubFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path=None, target=None):
        if fullname.split(".")[0] in STUB_TOPS:
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        m = _StubModule(spec.name)
        m.__path__ = []
        return m

    def exec_module(self, module):
        pass


# ---------------------------------------------------------------------------------------------- small real pieces
class Registry:
    """fvcore.common.registry.Registry: name -> object, `register` as call or decorator, `get`."""

    def __init__(self, name):
        self._name, self._map = name, {}

    def register(self, obj=None):
        if obj is None:
            def deco(o):
                self._map[o.__name__] = o
                return o
            return deco
        self._map[obj.__name__] = obj

    def get(self, name):
        if name not in self._map:
            raise KeyError("No object named '{}' found in '{}' registry!".format(name, self._name))
        return self._map[name]

    def __contains__(self, name):
        return name in self._map


class CfgNode(dict):
    """yacs.config.CfgNode / fvcore.common.config.CfgNode, as far as config/defaults.py and the model builders use it:
    attribute access on a nested dict, clone, freeze / defrost, merge_from_list."""

    def __init__(self, init_dict=None, key_list=None, new_allowed=False):
        super().__init__()
        self.__dict__["_frozen"] = False
        for k, v in (init_dict or {}).items():
            self[k] = CfgNode(v) if isinstance(v, dict) and not isinstance(v, CfgNode) else v

    def __getattr__(self, name):
        if name in self:
            return self[name]
        raise AttributeError(name)

    def __setattr__(self, name, value):
        if self.__dict__.get("_frozen"):
            raise AttributeError("Attempted to set {} to {}, but CfgNode is immutable".format(name, value))
        self[name] = value

    def clone(self):
        import copy

        return copy.deepcopy(self)

    def __deepcopy__(self, memo):
        import copy

        c = type(self)()
        for k, v in self.items():
            dict.__setitem__(c, k, copy.deepcopy(v, memo))
        return c

    def freeze(self):
        self._set_frozen(True)

    def defrost(self):
        self._set_frozen(False)

    def is_frozen(self):
        return self.__dict__["_frozen"]

    def _set_frozen(self, f):
        self.__dict__["_frozen"] = f
        for v in self.values():
            if isinstance(v, CfgNode):
                v._set_frozen(f)

    def merge_from_list(self, lst):
        for k, v in zip(lst[0::2], lst[1::2]):
            d = self
            parts = k.split(".")
            for p in parts[:-1]:
                d = d[p]
            d[parts[-1]] = v

    def get(self, k, default=None):
        return self[k] if k in self else default


class HistoryBuffer:
    """fvcore.common.history_buffer.HistoryBuffer (utils/events.py keeps one per scalar)."""

    def __init__(self, max_length=1000000):
        self._data = []

    def update(self, value, iteration=None):
        self._data.append((value, iteration))

    def latest(self):
        return self._data[-1][0]

    def median(self, window_size):
        return float(np.median([x[0] for x in self._data[-window_size:]]))

    def avg(self, window_size):
        return float(np.mean([x[0] for x in self._data[-window_size:]]))

    def global_avg(self):
        return float(np.mean([x[0] for x in self._data]))

    def values(self):
        return self._data


def c2_xavier_fill(module):
    """fvcore.nn.weight_init.c2_xavier_fill"""
    torch.nn.init.kaiming_uniform_(module.weight, a=1)
    if module.bias is not None:
        torch.nn.init.constant_(module.bias, 0)


def c2_msra_fill(module):
    """fvcore.nn.weight_init.c2_msra_fill"""
    torch.nn.init.kaiming_normal_(module.weight, mode="fan_out", nonlinearity="relu")
    if module.bias is not None:
        torch.nn.init.constant_(module.bias, 0)


def smooth_l1_loss(input, target, beta, reduction="none"):
    """fvcore.nn.smooth_l1_loss"""
    if beta < 1e-5:
        loss = torch.abs(input - target)
    else:
        n = torch.abs(input - target)
        loss = torch.where(n < beta, 0.5 * n ** 2 / beta, n - 0.5 * beta)
    return loss.mean() if reduction == "mean" else loss.sum() if reduction == "sum" else loss


def sigmoid_focal_loss(inputs, targets, alpha=-1, gamma=2, reduction="none"):
    """fvcore.nn.sigmoid_focal_loss(_jit)"""
    p = torch.sigmoid(inputs)
    ce = F.binary_cross_entropy_with_logits(inputs, targets, reduction="none")
    p_t = p * targets + (1 - p) * (1 - targets)
    loss = ce * ((1 - p_t) ** gamma)
    if alpha >= 0:
        loss = (alpha * targets + (1 - alpha) * (1 - targets)) * loss
    return loss.mean() if reduction == "mean" else loss.sum() if reduction == "sum" else loss


def giou_loss(boxes1, boxes2, reduction="none", eps=1e-7):
    """fvcore.nn.giou_loss"""
    x1, y1, x2, y2 = boxes1.unbind(-1)
    x1g, y1g, x2g, y2g = boxes2.unbind(-1)
    xk1, yk1, xk2, yk2 = torch.max(x1, x1g), torch.max(y1, y1g), torch.min(x2, x2g), torch.min(y2, y2g)
    inter = torch.zeros_like(x1)
    m = (yk2 > yk1) & (xk2 > xk1)
    inter[m] = (xk2[m] - xk1[m]) * (yk2[m] - yk1[m])
    union = (x2 - x1) * (y2 - y1) + (x2g - x1g) * (y2g - y1g) - inter
    iou = inter / (union + eps)
    area_c = (torch.max(x2, x2g) - torch.min(x1, x1g)) * (torch.max(y2, y2g) - torch.min(y1, y1g))
    loss = 1 - (iou - (area_c - union) / (area_c + eps))
    return loss.mean() if reduction == "mean" else loss.sum() if reduction == "sum" else loss


# ---------------------------------------------------------------------------------------------- operator backends
def torch_roi_align(input, boxes, output_size, spatial_scale=1.0, sampling_ratio=-1, aligned=False):
    """torchvision.ops.roi_align restated in plain torch (differentiable): the pixel model documented at
    layers/roi_align.py:15-35 and the sampling of the in-tree twin ROIAlignRotated_cpu.cpp:27-129 at angle 0 --
    bin = mean of g_h x g_w bilinear samples, g = sampling_ratio or ceil(roi / pooled); samples outside [-1, size] are 0,
    coordinates clamped to [0, size - 1].  One ROI at a time (its grid size is its own); small inputs only."""
    ph, pw = (output_size, output_size) if isinstance(output_size, int) else output_size
    if boxes.shape[0] == 0:
        return input.new_zeros((0, input.shape[1], ph, pw))
    N, C, H, W = input.shape
    off = 0.5 if aligned else 0.0
    outs = []
    for r in boxes.detach().float().cpu().numpy():  # (one host copy; the geometry below is fp32 like the op's)
        b = int(r[0])
        x1, y1, x2, y2 = [float(np.float32(np.float32(v) * np.float32(spatial_scale)) - np.float32(off)) for v in r[1:]]
        rw, rh = x2 - x1, y2 - y1
        if not aligned:
            rw, rh = max(rw, 1.0), max(rh, 1.0)
        bw, bh = rw / pw, rh / ph
        gh = sampling_ratio if sampling_ratio > 0 else int(math.ceil(rh / ph))
        gw = sampling_ratio if sampling_ratio > 0 else int(math.ceil(rw / pw))
        gh, gw = max(gh, 1), max(gw, 1)
        ys = y1 + (torch.arange(ph * gh, device=input.device, dtype=torch.float32) + 0.5) * (bh / gh)
        xs = x1 + (torch.arange(pw * gw, device=input.device, dtype=torch.float32) + 0.5) * (bw / gw)

        def axis(c, size):
            valid = (c >= -1.0) & (c <= size)
            c = c.clamp(min=0.0)
            lo = c.floor().long()
            hi_edge = lo >= size - 1
            lo = torch.where(hi_edge, torch.full_like(lo, size - 1), lo)
            hi = torch.where(hi_edge, lo, lo + 1)
            c = torch.where(hi_edge, lo.to(c.dtype), c)
            frac = c - lo.to(c.dtype)
            return lo, hi, frac, valid

        ylo, yhi, ly, vy = axis(ys, H)
        xlo, xhi, lx, vx = axis(xs, W)
        img = input[b].float()                                                     # (C, H, W)
        top = img[:, ylo][:, :, xlo] * (1 - lx) + img[:, ylo][:, :, xhi] * lx
        bot = img[:, yhi][:, :, xlo] * (1 - lx) + img[:, yhi][:, :, xhi] * lx
        val = top * (1 - ly)[:, None] + bot * ly[:, None]
        val = val * (vy[:, None] & vx[None, :]).to(val.dtype)
        outs.append(val.reshape(C, ph, gh, pw, gw).mean(dim=(2, 4)))
    return torch.stack(outs).to(input.dtype)


def host_nms(boxes, scores, iou_threshold):
    """torchvision.ops.nms: greedy, score-descending, the C oracle on the host (oracle/d2_oracle.c: nms)."""
    import oracle

    if boxes.numel() == 0:
        return torch.empty((0,), dtype=torch.int64, device=boxes.device)
    keep = oracle.nms(boxes.detach().float().cpu().numpy(), scores.detach().float().cpu().numpy(), float(iou_threshold))
    return torch.from_numpy(np.asarray(keep, np.int64)).to(boxes.device)


def host_batched_nms(boxes, scores, idxs, iou_threshold):
    """torchvision.ops.boxes.batched_nms as per-category NMS (results agree with the coordinate-offset
    variant up to fp rounding of the offset coordinates)."""
    import oracle

    if boxes.numel() == 0:
        return torch.empty((0,), dtype=torch.int64, device=boxes.device)
    keep = oracle.batched_nms(boxes.detach().float().cpu().numpy(), scores.detach().float().cpu().numpy(),
                              idxs.detach().cpu().numpy(), float(iou_threshold))
    return torch.from_numpy(np.asarray(keep, np.int64)).to(boxes.device)


class Backend:
    """What the stubbed operator names dispatch to at CALL time (so one imported model runs on either)."""
    name = "reference"


BACKEND = Backend()


def _ops(kind):
    if kind == "product":
        import detectron2_amd.layers as L
        from detectron2_amd import _C_shim

        return dict(roi_align=L.roi_align, nms=L.nms, batched_nms=L.batched_nms, C=_C_shim)
    class _LazyC:
        def __getattr__(self, name):  # (the compiled reference DCN is a HIP module: loaded when a DCN entry is first called)
            from oracle import ref

            return getattr(ref.compiled_dcn(), name)

    return dict(roi_align=torch_roi_align, nms=host_nms, batched_nms=host_batched_nms, C=_LazyC())


class _DispatchC(types.ModuleType):
    """`detectron2._C`: the five DCN entry points (vision.cpp:85-102) of the active backend."""

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return getattr(BACKEND.ops["C"], name)


@contextlib.contextmanager
def backend(kind):
    """Run the reference's model code with its operators bound to `kind` ("product" | "reference")."""
    prev = getattr(BACKEND, "ops", None), BACKEND.name
    BACKEND.ops, BACKEND.name = _ops(kind), kind
    d2 = sys.modules["detectron2"]
    import detectron2.layers.mask_ops as mo
    import detectron2.structures.boxes as sb

    saved = (mo.paste_masks_in_image, sb.pairwise_iou, sb.pairwise_ioa)
    patched = []
    try:
        if kind == "product":
            import detectron2_amd.layers as L
            import detectron2_amd.structures as S

            def paste(masks, boxes, image_shape, threshold=0.5):
                return L.paste_masks_in_image(masks, getattr(boxes, "tensor", boxes), image_shape, threshold)

            def iou(b1, b2):
                return S.pairwise_iou(S.Boxes(b1.tensor), S.Boxes(b2.tensor))

            def ioa(b1, b2):
                return S.pairwise_ioa(S.Boxes(b1.tensor), S.Boxes(b2.tensor))

            # every module that bound these names at import time
            for mod in list(sys.modules.values()):
                if mod is None or not getattr(mod, "__name__", "").startswith("detectron2"):
                    continue
                for nm, fn, orig in (("paste_masks_in_image", paste, saved[0]), ("pairwise_iou", iou, saved[1]),
                                     ("pairwise_ioa", ioa, saved[2])):
                    if mod.__dict__.get(nm) is orig:
                        patched.append((mod, nm, orig))
                        setattr(mod, nm, fn)
        yield d2
    finally:
        for mod, nm, orig in patched:
            setattr(mod, nm, orig)
        BACKEND.ops, BACKEND.name = prev


_INSTALLED = {}


def install():
    """Make `import detectron2` resolve to the reference's package (bytecode tree) with the stubs above.  Idempotent;
    returns the package.  Raises if the bytecode tree is missing (oracle/build_ref.py builds it where /root/reference
    exists; it travels to the GPU box under oracle/_ref/)."""
    if _INSTALLED:
        return _INSTALLED["pkg"]
    from oracle import build_ref

    if not build_ref.build_pkg():
        raise RuntimeError("reference package bytecode missing: run oracle/build_ref.py where /root/reference exists")
    for n in list(sys.modules):
        if n == "detectron2" or n.startswith("detectron2."):
            raise RuntimeError("a `detectron2` package is already imported (%s): the model-level tests need a fresh process "
                               "or must run before tests/_reference_surface.py's Surface" % n)
    sys.meta_path.append(_StubFinder())
    sys.path.insert(0, _PKG)

    def mod(name, **attrs):
        m = _StubModule(name)
        m.__path__ = []
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules[name] = m
        parent, _, leaf = name.rpartition(".")
        if parent and parent in sys.modules:
            setattr(sys.modules[parent], leaf, m)
        return m

    mod("fvcore", __version__="0.1.5")
    mod("cv2", __version__="4.0.0")  # (utils/env.py:73-76 reads the version; visualizer.py imports it; never called here)
    mod("fvcore.common")
    mod("fvcore.common.registry", Registry=Registry)
    mod("fvcore.common.config", CfgNode=CfgNode)
    mod("fvcore.common.history_buffer", HistoryBuffer=HistoryBuffer)
    wi = mod("fvcore.nn.weight_init", c2_xavier_fill=c2_xavier_fill, c2_msra_fill=c2_msra_fill)
    mod("fvcore.nn", weight_init=wi, smooth_l1_loss=smooth_l1_loss, sigmoid_focal_loss=sigmoid_focal_loss,
        sigmoid_focal_loss_jit=sigmoid_focal_loss, giou_loss=giou_loss)
    sys.modules["fvcore.nn.weight_init"] = wi
    mod("termcolor", colored=lambda s, *a, **k: s)
    # torchvision: only the operator names the reference calls (layers/roi_align.py:58-65, layers/nms.py:6,22)
    tv = mod("torchvision", __version__="0.19.1")
    ops = mod("torchvision.ops")
    ops.roi_align = lambda *a, **k: BACKEND.ops["roi_align"](*a, **k)
    ops.nms = lambda *a, **k: BACKEND.ops["nms"](*a, **k)
    ops.batched_nms = lambda *a, **k: BACKEND.ops["batched_nms"](*a, **k)

    def no_tv_dcn(*a, **k):
        raise NotImplementedError("torchvision.ops.deform_conv2d (the reference's CPU DCN forward) is not installed")

    ops.deform_conv2d = no_tv_dcn
    bx = mod("torchvision.ops.boxes", nms=ops.nms, batched_nms=ops.batched_nms)
    ops.boxes = bx
    tv.ops = ops
    sys.modules["detectron2._C"] = _DispatchC("detectron2._C")
    BACKEND.ops = _ops("reference")
    # @torch.jit.script CLASSES (box_regression.py:20,119) are compiled at import and need their source, which the GPU
    # box does not have; in eager mode a scripted class IS the Python class, so the decorator is the identity here
    real_script = torch.jit.script
    torch.jit.script = lambda obj, *a, **k: obj if isinstance(obj, type) else real_script(obj, *a, **k)
    try:
        pkg = importlib.import_module("detectron2")
        pkg._C = sys.modules["detectron2._C"]
        for m in ("detectron2.config", "detectron2.structures", "detectron2.layers", "detectron2.modeling",
                  "detectron2.utils.events"):
            importlib.import_module(m)
    finally:
        torch.jit.script = real_script
    _INSTALLED["pkg"] = pkg
    return pkg


# ---------------------------------------------------------------------------------------------- model builders
def mask_rcnn_cfg():
    """configs/Base-RCNN-FPN.yaml + COCO-InstanceSegmentation/mask_rcnn_R_50_FPN_1x.yaml, set in code (no yaml / model
    zoo here); weights: seeded random."""
    install()
    from detectron2.config import get_cfg

    cfg = get_cfg()
    cfg.MODEL.META_ARCHITECTURE = "GeneralizedRCNN"
    cfg.MODEL.BACKBONE.NAME = "build_resnet_fpn_backbone"
    cfg.MODEL.RESNETS.OUT_FEATURES = ["res2", "res3", "res4", "res5"]
    cfg.MODEL.RESNETS.DEPTH = 50
    cfg.MODEL.FPN.IN_FEATURES = ["res2", "res3", "res4", "res5"]
    cfg.MODEL.ANCHOR_GENERATOR.SIZES = [[32], [64], [128], [256], [512]]
    cfg.MODEL.ANCHOR_GENERATOR.ASPECT_RATIOS = [[0.5, 1.0, 2.0]]
    cfg.MODEL.RPN.IN_FEATURES = ["p2", "p3", "p4", "p5", "p6"]
    cfg.MODEL.RPN.PRE_NMS_TOPK_TRAIN = 2000
    cfg.MODEL.RPN.PRE_NMS_TOPK_TEST = 1000
    cfg.MODEL.RPN.POST_NMS_TOPK_TRAIN = 1000
    cfg.MODEL.RPN.POST_NMS_TOPK_TEST = 1000
    cfg.MODEL.ROI_HEADS.NAME = "StandardROIHeads"
    cfg.MODEL.ROI_HEADS.IN_FEATURES = ["p2", "p3", "p4", "p5"]
    cfg.MODEL.ROI_BOX_HEAD.NAME = "FastRCNNConvFCHead"
    cfg.MODEL.ROI_BOX_HEAD.NUM_FC = 2
    cfg.MODEL.ROI_BOX_HEAD.POOLER_RESOLUTION = 7
    cfg.MODEL.ROI_MASK_HEAD.NAME = "MaskRCNNConvUpsampleHead"
    cfg.MODEL.ROI_MASK_HEAD.NUM_CONV = 4
    cfg.MODEL.ROI_MASK_HEAD.POOLER_RESOLUTION = 14
    cfg.MODEL.MASK_ON = True
    return cfg


def retinanet_cfg():
    """configs/Base-RetinaNet.yaml + COCO-Detection/retinanet_R_50_FPN_1x.yaml in code."""
    install()
    from detectron2.config import get_cfg

    cfg = get_cfg()
    cfg.MODEL.META_ARCHITECTURE = "RetinaNet"
    cfg.MODEL.BACKBONE.NAME = "build_retinanet_resnet_fpn_backbone"
    cfg.MODEL.RESNETS.OUT_FEATURES = ["res3", "res4", "res5"]
    cfg.MODEL.RESNETS.DEPTH = 50
    cfg.MODEL.ANCHOR_GENERATOR.SIZES = [[x, x * 2 ** (1.0 / 3), x * 2 ** (2.0 / 3)] for x in [32, 64, 128, 256, 512]]
    cfg.MODEL.FPN.IN_FEATURES = ["res3", "res4", "res5"]
    cfg.MODEL.RETINANET.IOU_THRESHOLDS = [0.4, 0.5]
    cfg.MODEL.RETINANET.IOU_LABELS = [0, -1, 1]
    cfg.MODEL.RETINANET.SMOOTH_L1_LOSS_BETA = 0.0
    return cfg


def build_model(cfg, seed=0, device="cuda"):
    from detectron2.modeling import build_model as build

    torch.manual_seed(seed)
    cfg = cfg.clone()
    cfg.MODEL.DEVICE = device
    return build(cfg)


def make_inputs(n_images=2, size=(800, 800), n_gt=8, seed=0, device="cuda", masks=True):
    """batched_inputs as DatasetMapper would hand them over: uint8-range images (CHW float), Instances with gt_boxes,
    gt_classes and (for Mask R-CNN) BitMasks of the boxes' ellipses."""
    from detectron2.structures import BitMasks, Boxes, Instances

    g = torch.Generator().manual_seed(seed)
    H, W = size
    out = []
    for _ in range(n_images):
        img = torch.rand(3, H, W, generator=g) * 255
        wh = torch.exp(torch.rand(n_gt, 2, generator=g) * (math.log(0.6 * min(H, W)) - math.log(24.0)) + math.log(24.0))
        xy = torch.rand(n_gt, 2, generator=g) * (torch.tensor([W, H], dtype=torch.float32) - wh)
        boxes = torch.cat([xy, xy + wh], dim=1)
        inst = Instances((H, W))
        inst.gt_boxes = Boxes(boxes)
        inst.gt_classes = torch.randint(0, 80, (n_gt,), generator=g)
        if masks:
            yy, xx = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
            cx, cy = (boxes[:, 0] + boxes[:, 2]) / 2, (boxes[:, 1] + boxes[:, 3]) / 2
            rx, ry = (boxes[:, 2] - boxes[:, 0]) / 2, (boxes[:, 3] - boxes[:, 1]) / 2
            m = ((xx[None] - cx[:, None, None]) / rx[:, None, None]) ** 2 + ((yy[None] - cy[:, None, None]) / ry[:, None, None]) ** 2 <= 1
            inst.gt_masks = BitMasks(m)
        out.append({"image": img, "instances": inst.to(device), "height": H, "width": W})
    return out
