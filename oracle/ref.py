"""The REAL reference, where it can be run -- TEST INFRASTRUCTURE ONLY.

* ``compiled()``: torch.ops.d2ref.* = the reference's own C++ CPU ops, built by
  oracle/build_ref.py into oracle/_ref/libd2ref.so (travels to the GPU box prebuilt).
* ``py_mask_ops()`` / ``py_boxes()``: the reference's own Python modules loaded by file path
  from /root/reference (only exists in the build container; used to generate tests/golden/).
"""
import importlib.util
import os
import sys
import types

_HERE = os.path.dirname(os.path.abspath(__file__))
_REF_LIB = os.path.join(_HERE, "_ref", "libd2ref.so")
REF_ROOT = os.environ.get("D2_REFERENCE_ROOT", "/root/reference")
_loaded = False


def have_compiled():
    return os.path.exists(_REF_LIB)


def have_tree():
    return os.path.isdir(os.path.join(REF_ROOT, "detectron2", "layers"))


def compiled():
    """Returns torch.ops.d2ref (nms_rotated, box_iou_rotated, roi_align_rotated_forward/backward)."""
    global _loaded
    import torch

    if not _loaded:
        if not have_compiled():
            from . import build_ref

            if not build_ref.build(verbose=False):
                raise RuntimeError("compiled reference unavailable (no oracle/_ref, no reference tree)")
        torch.ops.load_library(_REF_LIB)
        _loaded = True
    return torch.ops.d2ref


def _load_by_path(name, relpath):
    path = os.path.join(REF_ROOT, relpath)
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def py_mask_ops():
    """detectron2/layers/mask_ops.py (deps: torch, numpy, PIL only)."""
    return _load_by_path("_d2ref_mask_ops", "detectron2/layers/mask_ops.py")


def py_boxes():
    """detectron2/structures/boxes.py (deps: torch, numpy only)."""
    return _load_by_path("_d2ref_boxes", "detectron2/structures/boxes.py")


def _load_with_layers_stub(name, relpath):
    """Load a reference module whose only non-torch import is `nonzero_tuple` from detectron2.layers
    (layers/wrappers.py:150-162: `x.nonzero().unbind(1)` outside scripting); the package itself is not importable
    here, so a stub module providing exactly that helper is registered for the duration of the load."""
    saved = {k: sys.modules.get(k) for k in ("detectron2", "detectron2.layers")}
    pkg, layers = types.ModuleType("detectron2"), types.ModuleType("detectron2.layers")

    def nonzero_tuple(x):
        if x.dim() == 0:
            return x.unsqueeze(0).nonzero().unbind(1)
        return x.nonzero().unbind(1)

    layers.nonzero_tuple = nonzero_tuple
    pkg.layers = layers
    sys.modules["detectron2"], sys.modules["detectron2.layers"] = pkg, layers
    try:
        return _load_by_path(name, relpath)
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v


def py_matcher():
    """detectron2/modeling/matcher.py."""
    return _load_with_layers_stub("_d2ref_matcher", "detectron2/modeling/matcher.py")


def py_sampling():
    """detectron2/modeling/sampling.py (subsample_labels)."""
    return _load_with_layers_stub("_d2ref_sampling", "detectron2/modeling/sampling.py")


def _with_stubs(stubs, fn):
    saved = {k: sys.modules.get(k) for k in stubs}
    sys.modules.update(stubs)
    try:
        return fn()
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v


def py_box_regression():
    """detectron2/modeling/box_regression.py (Box2BoxTransform).  Its module-level imports of the loss helpers
    (fvcore.nn, detectron2.layers.{ciou,diou}_loss), unused by apply_deltas / get_deltas, are stubbed."""
    import torch

    fv, fvnn = types.ModuleType("fvcore"), types.ModuleType("fvcore.nn")
    fvnn.giou_loss = fvnn.smooth_l1_loss = lambda *a, **k: (_ for _ in ()).throw(NotImplementedError())
    fv.nn = fvnn
    pkg, layers, structs = (types.ModuleType(n) for n in ("detectron2", "detectron2.layers", "detectron2.structures"))
    layers.cat = lambda ts, dim=0: torch.cat(ts, dim)
    layers.ciou_loss = layers.diou_loss = fvnn.giou_loss
    structs.Boxes = py_boxes().Boxes
    pkg.layers, pkg.structures = layers, structs
    stubs = {"fvcore": fv, "fvcore.nn": fvnn, "detectron2": pkg, "detectron2.layers": layers,
             "detectron2.structures": structs}
    return _with_stubs(stubs, lambda: _load_by_path("_d2ref_box_regression", "detectron2/modeling/box_regression.py"))


def py_proposal_utils(batched_nms):
    """detectron2/modeling/proposal_generator/proposal_utils.py (find_top_rpn_proposals) with
    detectron2.layers.batched_nms = the given function (torchvision is not installed: the oracle's restatement
    is passed in), cat / move_device_like as in layers/wrappers.py, Boxes / Instances from the reference files."""
    import torch

    pkg, layers, structs = (types.ModuleType(n) for n in ("detectron2", "detectron2.layers", "detectron2.structures"))
    layers.batched_nms = batched_nms
    layers.cat = lambda ts, dim=0: torch.cat(ts, dim)
    layers.move_device_like = lambda src, dst: src.to(dst.device)
    structs.Boxes = py_boxes().Boxes
    structs.Instances = _load_by_path("_d2ref_instances", "detectron2/structures/instances.py").Instances
    pkg.layers, pkg.structures = layers, structs
    stubs = {"detectron2": pkg, "detectron2.layers": layers, "detectron2.structures": structs}
    return _with_stubs(stubs, lambda: _load_by_path("_d2ref_proposal_utils",
                                                    "detectron2/modeling/proposal_generator/proposal_utils.py"))


class _EventRecorder:
    """Stand-in for detectron2.utils.events.EventStorage: keeps what mask_rcnn_loss logs."""

    def __init__(self):
        self.scalars = {}
        self.iter = 0

    def put_scalar(self, name, value, **kw):
        self.scalars[name] = float(value)

    def put_image(self, *a, **k):
        pass


def py_mask_head():
    """detectron2/modeling/roi_heads/mask_head.py (mask_rcnn_loss, mask_rcnn_inference).  Module-level imports that
    the two functions do not use (fvcore weight_init, configurable, the conv layer classes, the registry) are
    stubbed; `cat` / `move_device_like` as in layers/wrappers.py; Instances from the reference file;
    get_event_storage returns an _EventRecorder exposed as `module._d2_events`."""
    import torch

    fv, fvnn, fvwi = (types.ModuleType(n) for n in ("fvcore", "fvcore.nn", "fvcore.nn.weight_init"))
    fv.nn, fvnn.weight_init = fvnn, fvwi
    names = ("detectron2", "detectron2.config", "detectron2.layers", "detectron2.layers.wrappers",
             "detectron2.structures", "detectron2.utils", "detectron2.utils.events", "detectron2.utils.registry")
    pkg, cfg, layers, wrappers, structs, utils, events, registry = (types.ModuleType(n) for n in names)
    cfg.configurable = lambda f=None, **k: f
    layers.Conv2d, layers.ConvTranspose2d = torch.nn.Conv2d, torch.nn.ConvTranspose2d
    layers.ShapeSpec = object
    layers.get_norm = lambda *a, **k: None
    layers.cat = lambda ts, dim=0: torch.cat(ts, dim)
    wrappers.move_device_like = lambda src, dst: src.to(dst.device)
    structs.Instances = _load_by_path("_d2ref_instances", "detectron2/structures/instances.py").Instances
    rec = _EventRecorder()
    events.get_event_storage = lambda: rec

    class Registry:
        def __init__(self, name):
            self.name = name

        def register(self, obj=None):
            return obj if obj is not None else (lambda o: o)

    registry.Registry = Registry
    pkg.config, pkg.layers, pkg.structures, pkg.utils = cfg, layers, structs, utils
    layers.wrappers, utils.events, utils.registry = wrappers, events, registry
    stubs = dict(zip(names, (pkg, cfg, layers, wrappers, structs, utils, events, registry)))
    stubs.update({"fvcore": fv, "fvcore.nn": fvnn, "fvcore.nn.weight_init": fvwi})
    mod = _with_stubs(stubs, lambda: _load_by_path("_d2ref_mask_head", "detectron2/modeling/roi_heads/mask_head.py"))
    mod._d2_events = rec
    mod._d2_Instances = structs.Instances
    return mod


def py_dense_detector():
    """detectron2/modeling/meta_arch/dense_detector.py (DenseDetector._decode_per_level_predictions /
    _decode_multi_level_predictions).  Loaded under its real dotted name so that its relative import of
    `..postprocessing` resolves to a stub; the visualisation / backbone imports it does not use for decoding are
    stubbed too.  Returns (module, Box2BoxTransform class, Boxes class, Instances class)."""
    import torch

    names = ("detectron2", "detectron2.data", "detectron2.data.detection_utils", "detectron2.layers",
             "detectron2.modeling", "detectron2.modeling.meta_arch", "detectron2.modeling.postprocessing",
             "detectron2.structures", "detectron2.utils", "detectron2.utils.events")
    pkg, data, du, layers, modeling, meta, post, structs, utils, events = (types.ModuleType(n) for n in names)
    for m in (pkg, data, modeling, meta, utils):
        m.__path__ = []  # mark as packages
    du.convert_image_to_rgb = lambda *a, **k: None
    layers.move_device_like = lambda src, dst: src.to(dst.device)
    layers.cat = lambda ts, dim=0: torch.cat(ts, dim)
    modeling.Backbone = torch.nn.Module
    post.detector_postprocess = lambda *a, **k: None
    structs.Boxes = py_boxes().Boxes
    structs.Instances = _load_by_path("_d2ref_instances", "detectron2/structures/instances.py").Instances
    structs.ImageList = object
    events.get_event_storage = lambda: _EventRecorder()
    br = py_box_regression()
    stubs = dict(zip(names, (pkg, data, du, layers, modeling, meta, post, structs, utils, events)))
    mod = _with_stubs(stubs, lambda: _load_by_path("detectron2.modeling.meta_arch.dense_detector",
                                                    "detectron2/modeling/meta_arch/dense_detector.py"))
    sys.modules.pop("detectron2.modeling.meta_arch.dense_detector", None)
    return mod, br.Box2BoxTransform, structs.Boxes, structs.Instances
