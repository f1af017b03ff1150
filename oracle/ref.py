"""The REAL reference, where it can be run -- TEST INFRASTRUCTURE ONLY.

* ``compiled()``: torch.ops.d2ref.* = the reference's own C++ CPU ops, built by
  oracle/build_ref.py into oracle/_ref/libd2ref.so (travels to the GPU box prebuilt).
* ``py_mask_ops()`` / ``py_boxes()``: the reference's own Python modules loaded by file path
  from /root/reference (only exists in the build container; used to generate tests/golden/).
"""
import importlib.machinery
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


def have_py():
    """The reference's Python modules are loadable: from the tree, or from the bytecode oracle/build_ref.py staged in
    oracle/_ref/py/ (what the GPU box has)."""
    from . import build_ref

    return have_tree() or all(os.path.exists(build_ref.pyc_path(m)) for m in build_ref.PY_MODULES)


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
    if os.path.exists(path):
        spec = importlib.util.spec_from_file_location(name, path)
    else:  # no tree here (the GPU box): the bytecode build_ref.build_py() compiled from the same file
        from . import build_ref

        pyc = build_ref.pyc_path(relpath)
        if not os.path.exists(pyc):
            raise FileNotFoundError(f"reference module {relpath}: neither {path} nor {pyc}")
        spec = importlib.util.spec_from_loader(name, importlib.machinery.SourcelessFileLoader(name, pyc))
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


def py_fast_rcnn(batched_nms):
    """detectron2/modeling/roi_heads/fast_rcnn.py (fast_rcnn_inference / fast_rcnn_inference_single_image) with
    detectron2.layers.batched_nms = the given function (torchvision is not installed).  The module-level imports the
    two functions do not use (configurable, the federated-loss helper, the loss functions, the event storage) are stubbed;
    Boxes / Instances are the reference's own (pure torch)."""
    import torch

    names = ("detectron2", "detectron2.config", "detectron2.data", "detectron2.data.detection_utils", "detectron2.layers",
             "detectron2.modeling", "detectron2.modeling.box_regression", "detectron2.structures", "detectron2.utils",
             "detectron2.utils.events")
    pkg, cfg, data, du, layers, modeling, boxreg, structs, utils, events = (types.ModuleType(n) for n in names)
    for m in (pkg, data, modeling, utils):
        m.__path__ = []
    cfg.configurable = lambda f=None, **k: f if f is not None else (lambda g: g)
    du.get_fed_loss_cls_weights = lambda *a, **k: None
    layers.ShapeSpec = object
    layers.batched_nms = batched_nms
    layers.cat = lambda ts, dim=0: torch.cat(ts, dim)
    layers.cross_entropy = torch.nn.functional.cross_entropy
    layers.nonzero_tuple = lambda x: x.nonzero().unbind(1)
    # (Box2BoxTransform is a torch.jit.script class: it needs its source, which the GPU box does not have -- and the
    # two inference functions never touch it)
    boxreg.Box2BoxTransform = object
    boxreg._dense_box_regression_loss = None
    structs.Boxes = py_boxes().Boxes
    structs.Instances = _load_by_path("_d2ref_instances", "detectron2/structures/instances.py").Instances
    events.get_event_storage = lambda: _EventRecorder()
    stubs = dict(zip(names, (pkg, cfg, data, du, layers, modeling, boxreg, structs, utils, events)))
    return _with_stubs(stubs, lambda: _load_by_path("_d2ref_fast_rcnn", "detectron2/modeling/roi_heads/fast_rcnn.py"))


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


def py_callers(layers_impl):
    """the reference's OWN callers of the hot path -- modeling/poolers.py (ROIPooler),
    proposal_generator/proposal_utils.py (find_top_rpn_proposals), roi_heads/mask_head.py (mask_rcnn_loss /
    mask_rcnn_inference), structures/masks.py (BitMasks) + layers/mask_ops.py -- loaded unchanged, with the names they
    import from `detectron2.layers` resolved by `layers_impl` (the replacement: detectron2_amd.layers).  The pure-torch
    helpers of layers/wrappers.py (cat, nonzero_tuple, shapes_to_tensor, move_device_like) are restated here; Boxes /
    Instances are the reference's own classes (pure torch).  -> a namespace of the loaded classes / functions."""
    import torch

    boxes_mod = py_boxes()
    inst_mod = _load_by_path("_d2ref_instances", "detectron2/structures/instances.py")
    names = ("detectron2", "detectron2.config", "detectron2.layers", "detectron2.layers.wrappers",
             "detectron2.layers.roi_align", "detectron2.structures", "detectron2.structures.boxes", "detectron2.utils",
             "detectron2.utils.events", "detectron2.utils.registry", "detectron2.utils.tracing", "detectron2.utils.memory",
             "torchvision", "torchvision.ops", "pycocotools", "pycocotools.mask", "fvcore", "fvcore.nn",
             "fvcore.nn.weight_init")
    m = {n: types.ModuleType(n) for n in names}
    for n in ("detectron2", "detectron2.layers", "detectron2.structures", "detectron2.utils", "torchvision", "pycocotools",
              "fvcore", "fvcore.nn"):
        m[n].__path__ = []
    L = m["detectron2.layers"]
    for k in ("ROIAlign", "ROIAlignRotated", "batched_nms", "nms", "paste_masks_in_image"):
        setattr(L, k, getattr(layers_impl, k))
    L.cat = lambda ts, dim=0: ts[0] if len(ts) == 1 else torch.cat(ts, dim)          # wrappers.py:65-72
    L.nonzero_tuple = lambda x: (x.unsqueeze(0) if x.dim() == 0 else x).nonzero().unbind(1)  # :158-169
    L.shapes_to_tensor = lambda x, device=None: torch.as_tensor(x, device=device)   # :20-41 (eager branch)
    L.move_device_like = lambda src, dst: src.to(dst.device)                          # :172-177
    L.Conv2d, L.ConvTranspose2d, L.ShapeSpec = torch.nn.Conv2d, torch.nn.ConvTranspose2d, object
    L.get_norm = lambda *a, **k: None
    m["detectron2.layers.wrappers"].move_device_like = L.move_device_like
    m["detectron2.layers.roi_align"].ROIAlign = layers_impl.ROIAlign
    m["detectron2.config"].configurable = lambda f=None, **k: f
    S = m["detectron2.structures"]
    S.Boxes, S.Instances = boxes_mod.Boxes, inst_mod.Instances
    m["detectron2.structures.boxes"].Boxes = boxes_mod.Boxes
    rec = _EventRecorder()
    m["detectron2.utils.events"].get_event_storage = lambda: rec

    class Registry:
        def __init__(self, name):
            self.name = name

        def register(self, obj=None):
            return obj if obj is not None else (lambda o: o)

    m["detectron2.utils.registry"].Registry = Registry
    m["detectron2.utils.tracing"].assert_fx_safe = lambda cond, msg: cond
    m["detectron2.utils.tracing"].is_fx_tracing = lambda: False
    m["detectron2.utils.memory"].retry_if_cuda_oom = lambda f: f
    m["torchvision.ops"].RoIPool = type("RoIPool", (torch.nn.Module,), {})
    for a, b in (("detectron2", "layers"), ("detectron2", "structures"), ("detectron2", "utils"), ("detectron2", "config"),
                 ("torchvision", "ops"), ("pycocotools", "mask"), ("fvcore", "nn")):
        setattr(m[a], b, m[a + "." + b])
    m["fvcore.nn"].weight_init = m["fvcore.nn.weight_init"]

    def load():
        ns = types.SimpleNamespace()
        ns.poolers = _load_by_path("_d2ref_poolers", "detectron2/modeling/poolers.py")
        ns.proposal_utils = _load_by_path("_d2ref_proposal_utils_g1", "detectron2/modeling/proposal_generator/proposal_utils.py")
        ns.mask_head = _load_by_path("_d2ref_mask_head_g1", "detectron2/modeling/roi_heads/mask_head.py")
        ns.mask_ops = _load_by_path("_d2ref_mask_ops_g1", "detectron2/layers/mask_ops.py")
        ns.masks = _load_by_path("detectron2.structures.masks", "detectron2/structures/masks.py")  # (relative import)
        sys.modules.pop("detectron2.structures.masks", None)
        ns.Boxes, ns.Instances, ns.events = boxes_mod.Boxes, inst_mod.Instances, rec
        return ns

    return _with_stubs(m, load)


_REF_DCN = os.path.join(_HERE, "_ref", "_d2ref_C.so")


def have_dcn():
    return os.path.exists(_REF_DCN)


def compiled_dcn():
    """The reference's own csrc/deformable/*.cu compiled as HIP for gfx950 (oracle/build_ref.py: build_dcn) -- a
    pybind11 module with detectron2._C's five DCN entry points (vision.cpp:86-102).  GPU CHECKER only."""
    if "_d2ref_C" in sys.modules:
        return sys.modules["_d2ref_C"]
    if not have_dcn():
        from . import build_ref

        if not build_ref.build_dcn(verbose=False):
            raise RuntimeError("compiled reference DCN unavailable (no oracle/_ref/_d2ref_C.so, no reference tree)")
    import torch  # noqa: F401  (libtorch must be loaded first)

    spec = importlib.util.spec_from_file_location("_d2ref_C", _REF_DCN)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    sys.modules["_d2ref_C"] = mod
    return mod


def py_deform_conv():
    """detectron2/layers/deform_conv.py -- the reference's own autograd Functions and modules (_DeformConv,
    _ModulatedDeformConv, DeformConv, ModulatedDeformConv) -- loaded unchanged with `detectron2._C` = compiled_dcn(),
    `.wrappers` = the reference's own layers/wrappers.py, torchvision's deform_conv2d (its CPU forward, not installed)
    stubbed to raise.  What this gives: the reference's DCN forward AND backward exactly as upstream runs them on a GPU
    (deform_conv.py:62-133, 221-281 -> deform_conv_cuda.cu:272-1221)."""
    names = ("detectron2", "detectron2.layers", "detectron2.utils", "detectron2.utils.develop", "detectron2.utils.env",
             "torchvision", "torchvision.ops")
    m = {n: types.ModuleType(n) for n in names}
    for n in ("detectron2", "detectron2.layers", "detectron2.utils", "torchvision"):
        m[n].__path__ = []
    import torch

    m["detectron2.utils.env"].TORCH_VERSION = tuple(int(x) for x in torch.__version__.split(".")[:2])

    def _no_tv(*a, **k):
        raise NotImplementedError("torchvision is not installed: the reference's CPU DCN forward cannot run here")

    m["torchvision.ops"].deform_conv2d = _no_tv

    def _dummy(name, *a):
        raise ImportError(name)

    m["detectron2.utils.develop"].create_dummy_class = lambda name, *a: (lambda *x, **k: _dummy(name))
    m["detectron2.utils.develop"].create_dummy_func = lambda name, *a: (lambda *x, **k: _dummy(name))
    m["detectron2"]._C = compiled_dcn()
    m["detectron2._C"] = m["detectron2"]._C
    m["detectron2"].layers, m["detectron2"].utils = m["detectron2.layers"], m["detectron2.utils"]
    m["detectron2.utils"].develop, m["detectron2.utils"].env = m["detectron2.utils.develop"], m["detectron2.utils.env"]
    m["torchvision"].ops = m["torchvision.ops"]

    def load():
        w = _load_by_path("detectron2.layers.wrappers", "detectron2/layers/wrappers.py")
        m["detectron2.layers"].wrappers = w
        mod = _load_by_path("detectron2.layers.deform_conv", "detectron2/layers/deform_conv.py")
        sys.modules.pop("detectron2.layers.deform_conv", None)
        sys.modules.pop("detectron2.layers.wrappers", None)
        return mod

    return _with_stubs(m, load)
