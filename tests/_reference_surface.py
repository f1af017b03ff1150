"""A `detectron2` package tree, built in sys.modules for the duration of a test, whose hot-path names resolve to
`detectron2_amd` -- what the reference's OWN unit tests import when tests/test_gpu_reference_tests.py runs them.

TEST INFRASTRUCTURE.  What is bound to what:

  detectron2.layers{,.roi_align,.roi_align_rotated,.rotated_boxes}   detectron2_amd.layers  (the product)
  detectron2.modeling.poolers.ROIPooler / .matcher.Matcher            detectron2_amd.modeling (the product)
  detectron2.structures.pairwise_iou / pairwise_ioa                   detectron2_amd.structures (the product)
  detectron2.structures.Boxes / BoxMode / RotatedBoxes                the REFERENCE's own classes (containers, not hot path;
                                                                      RotatedBoxes' pairwise_iou calls the product's
                                                                      pairwise_iou_rotated)
  detectron2.projects.point_rend.point_features                       the REFERENCE's own file (pure torch)
  torchvision.ops.nms / box_iou                                       the product's nms / pairwise_iou (the reference's
                                                                      `detectron2.layers.nms` IS torchvision's: nms.py:6)
  detectron2.utils.testing.random_boxes / reload_script_model         restated (utils/testing.py:42-53,142-150)
  detectron2.utils.env.TORCH_VERSION, detectron2.config.get_cfg       the two RPN matcher defaults of config/defaults.py
  cv2.resize (INTER_LINEAR)                                           F.interpolate(bilinear, align_corners=False): the
                                                                      tests only halve images, where both are the 2x2 mean
  fvcore.common.benchmark.benchmark                                   no-op

The product has NO CPU path, while the reference's tests build most tensors on the CPU and compare "cpu" with
"cuda" results.  Every product callable bound here is wrapped by `on_device`: CPU tensors (and Boxes) are moved to the
GPU with differentiable copies, the HIP kernel runs, results go back to the CPU.  So the reference's *_cpu cases exercise
the same HIP kernels as its *_cuda cases; none of them runs a CPU implementation."""
import copy
import functools
import importlib.machinery
import importlib.util
import io
import os
import sys
import types

import torch

DEV = "cuda"


def _is_cpu_tensor(v):
    return isinstance(v, torch.Tensor) and v.device.type == "cpu"


def _move(v, state):
    if _is_cpu_tensor(v):
        state["moved"] = True
        return v.to(DEV)
    if hasattr(v, "tensor") and _is_cpu_tensor(getattr(v, "tensor")) and hasattr(v, "to"):  # Boxes / RotatedBoxes
        state["moved"] = True
        return v.to(DEV)
    if isinstance(v, (list, tuple)):
        return type(v)(_move(i, state) for i in v)
    return v


def _back(v):
    if isinstance(v, torch.Tensor):
        return v.cpu()
    if isinstance(v, (list, tuple)):
        return type(v)(_back(i) for i in v)
    return v


def on_device(fn):
    """CPU arguments -> GPU (differentiable), run the product's op, results -> CPU iff something was moved."""
    @functools.wraps(fn)
    def wrapper(*a, **k):
        state = {"moved": False}
        a = _move(a, state)
        k = {n: _move(v, state) for n, v in k.items()}
        out = fn(*a, **k)
        return _back(out) if state["moved"] else out

    return wrapper


def module_on_device(cls):
    """Subclass of a product nn.Module whose forward accepts CPU tensors (on_device); a module whose own parameters
    live on the CPU is run as a GPU copy."""
    def forward(self, *a, **k):
        impl = super(sub, self).forward
        if any(p.device.type == "cpu" for p in self.parameters()):
            clone = copy.deepcopy(self).to(DEV)
            impl = super(sub, clone).forward
        return on_device(impl)(*a, **k)

    sub = type(cls.__name__, (cls,), {"forward": forward, "__module__": cls.__module__})
    return sub


def _load(name, relpath):
    """A reference module by path: from /root/reference when it exists, else the bytecode oracle/build_ref.py staged."""
    from oracle import build_ref, ref

    path = os.path.join(ref.REF_ROOT, relpath)
    if os.path.exists(path):
        spec = importlib.util.spec_from_file_location(name, path)
    else:
        spec = importlib.util.spec_from_loader(
            name, importlib.machinery.SourcelessFileLoader(name, build_ref.pyc_path(relpath)))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


class Surface:
    """Context manager: installs the package tree, removes it again (and restores whatever was there)."""

    def __init__(self):
        self.saved = {}
        self.mods = {}

    def _mod(self, name, pkg=False):
        m = types.ModuleType(name)
        if pkg:
            m.__path__ = []
        self.mods[name] = m
        parent, _, leaf = name.rpartition(".")
        if parent in self.mods:
            setattr(self.mods[parent], leaf, m)
        return m

    def __enter__(self):
        import detectron2_amd.layers as L
        import detectron2_amd.modeling as M
        import detectron2_amd.structures as S

        for n in ("detectron2", "detectron2.layers", "detectron2.structures", "detectron2.modeling", "detectron2.utils",
                  "detectron2.projects", "detectron2.projects.point_rend", "torchvision", "fvcore", "fvcore.common"):
            self._mod(n, pkg=True)
        layers = self.mods["detectron2.layers"]
        for name in ("roi_align", "roi_align_rotated", "deform_conv", "modulated_deform_conv", "nms", "batched_nms",
                     "nms_rotated", "batched_nms_rotated", "paste_masks_in_image", "pairwise_iou_rotated"):
            setattr(layers, name, on_device(getattr(L, name)))
        # torch.jit.script needs the functions themselves (not a Python wrapper): the scriptability tests run with the
        # default device set to the GPU (test_gpu_reference_tests.py: DEFAULT_DEVICE_GPU) and get these
        layers.scriptable = {k: getattr(L, k) for k in ("batched_nms", "nms_rotated", "batched_nms_rotated", "nms")}
        for name in ("ROIAlign", "ROIAlignRotated", "DeformConv", "ModulatedDeformConv"):
            setattr(layers, name, module_on_device(getattr(L, name)))
        layers.cat = lambda ts, dim=0: ts[0] if len(ts) == 1 else torch.cat(ts, dim)          # wrappers.py:65-72
        layers.nonzero_tuple = lambda x: (x.unsqueeze(0) if x.dim() == 0 else x).nonzero().unbind(1)  # :158-169
        layers.shapes_to_tensor = lambda x, device=None: torch.as_tensor(x, device=device)   # :20-41 (eager branch)
        for sub, names in (("roi_align", ("ROIAlign", "roi_align")), ("roi_align_rotated", ("ROIAlignRotated",)),
                           ("rotated_boxes", ("pairwise_iou_rotated",))):
            m = self._mod("detectron2.layers." + sub)
            for n in names:
                setattr(m, n, getattr(layers, n))
        # structures: the reference's own containers around the product's IoU kernels
        sys.modules.update(self._install_prefix())
        boxes = _load("detectron2.structures.boxes", "detectron2/structures/boxes.py")
        boxes.pairwise_iou, boxes.pairwise_ioa = on_device(S.pairwise_iou), on_device(S.pairwise_ioa)
        boxes.pairwise_intersection = on_device(S.pairwise_intersection)
        self.mods["detectron2.structures.boxes"] = boxes
        rot = _load("detectron2.structures.rotated_boxes", "detectron2/structures/rotated_boxes.py")
        self.mods["detectron2.structures.rotated_boxes"] = rot
        st = self.mods["detectron2.structures"]
        st.boxes, st.rotated_boxes = boxes, rot
        st.Boxes, st.BoxMode, st.RotatedBoxes = boxes.Boxes, boxes.BoxMode, rot.RotatedBoxes
        st.pairwise_iou, st.pairwise_ioa, st.pairwise_iou_rotated = boxes.pairwise_iou, boxes.pairwise_ioa, rot.pairwise_iou
        st.BitMasks = S.BitMasks
        # modeling
        pool = self._mod("detectron2.modeling.poolers")
        pool.ROIPooler = module_on_device(M.ROIPooler)
        mat = self._mod("detectron2.modeling.matcher")

        class Matcher(M.Matcher):
            def __call__(self, match_quality_matrix):
                return on_device(super().__call__)(match_quality_matrix)

        mat.Matcher = Matcher
        # utils / config
        testing = self._mod("detectron2.utils.testing")

        def random_boxes(num_boxes, max_coord=100, device=None):
            """utils/testing.py:42-53; device=None (the reference: "cpu") follows torch's default device, which the
            scriptability tests set to the GPU."""
            b = torch.rand(num_boxes, 4, device=device) * (max_coord * 0.5)
            b.clamp_(min=1.0)
            b[:, 2:] += b[:, :2]
            return b

        def reload_script_model(module):
            buf = io.BytesIO()
            torch.jit.save(module, buf)
            buf.seek(0)
            return torch.jit.load(buf)

        testing.random_boxes, testing.reload_script_model = random_boxes, reload_script_model
        self._mod("detectron2.utils.env").TORCH_VERSION = tuple(int(x) for x in torch.__version__.split(".")[:2])
        cfgm = self._mod("detectron2.config")

        def get_cfg():  # config/defaults.py: MODEL.RPN.IOU_THRESHOLDS / IOU_LABELS
            rpn = types.SimpleNamespace(IOU_THRESHOLDS=[0.3, 0.7], IOU_LABELS=[0, -1, 1])
            return types.SimpleNamespace(MODEL=types.SimpleNamespace(RPN=rpn))

        cfgm.get_cfg = get_cfg
        # third-party names the tests import
        tvo = self._mod("torchvision.ops")
        tvo.nms = on_device(L.nms)
        tvo.box_iou = on_device(S.pairwise_iou)
        self._mod("fvcore.common.benchmark").benchmark = lambda *a, **k: None
        cv2 = self._mod("cv2")
        cv2.INTER_LINEAR = 1

        def resize(img, dsize, interpolation=1):
            import numpy as np

            t = torch.from_numpy(np.ascontiguousarray(img))[None, None].float()
            out = torch.nn.functional.interpolate(t, size=(dsize[1], dsize[0]), mode="bilinear", align_corners=False)
            return out[0, 0].numpy()

        cv2.resize = resize
        sys.modules.update(self._install_prefix())
        pf = _load("detectron2.projects.point_rend.point_features", "projects/PointRend/point_rend/point_features.py")
        self.mods["detectron2.projects.point_rend.point_features"] = pf
        self.mods["detectron2.projects.point_rend"].point_features = pf
        sys.modules.update(self._install_prefix())
        return self

    def _install_prefix(self):
        for k in self.mods:
            if k not in self.saved:
                self.saved[k] = sys.modules.get(k)
        return dict(self.mods)

    def load_test_module(self, relpath):
        name = "_d2ref_test_" + relpath.replace("/", "_").replace(".py", "")
        mod = _load(name, relpath)
        sys.modules.pop(name, None)
        return mod

    def __exit__(self, *exc):
        for k, v in self.saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
        for k in ("detectron2.structures.boxes", "detectron2.structures.rotated_boxes",
                  "detectron2.projects.point_rend.point_features"):
            if self.saved.get(k) is None:
                sys.modules.pop(k, None)
