"""Compile the reference's OWN CPU ops into oracle/_ref/libd2ref.so (test infrastructure).

Sources are compiled where they lie under /root/reference (never copied); the only file of
ours is oracle/ref_binding.cpp.  The reference's build system (setup.py) is not run.  The
output is git-ignored but travels to the GPU box with the snapshot.  No-op (returns False)
when /root/reference is absent (e.g. on the GPU box, which uses the prebuilt file).
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("D2_REFERENCE_ROOT", "/root/reference")
CSRC = os.path.join(REF, "detectron2", "layers", "csrc")
OUT_DIR = os.path.join(HERE, "_ref")
OUT = os.path.join(OUT_DIR, "libd2ref.so")

SOURCES = [
    os.path.join(CSRC, "ROIAlignRotated", "ROIAlignRotated_cpu.cpp"),
    os.path.join(CSRC, "box_iou_rotated", "box_iou_rotated_cpu.cpp"),
    os.path.join(CSRC, "nms_rotated", "nms_rotated_cpu.cpp"),
    os.path.join(HERE, "ref_binding.cpp"),
]


# The reference's own PYTHON modules that tests run (the callers of the hot path: SURVEY 8 row g1 -- poolers.py,
# proposal_utils.py, mask_head.py, mask_ops.py, structures/masks.py and what they need).  Byte-compiled where they lie
# into oracle/_ref/py/*.pyc -- a build product like libd2ref.so, git-ignored, never the source -- so that the GPU box,
# which has no /root/reference, can load them (oracle/ref.py: _load_by_path falls back to the .pyc).
PY_MODULES = [
    "detectron2/layers/mask_ops.py", "detectron2/structures/boxes.py", "detectron2/structures/instances.py",
    "detectron2/structures/masks.py", "detectron2/modeling/matcher.py", "detectron2/modeling/sampling.py",
    "detectron2/modeling/box_regression.py", "detectron2/modeling/poolers.py",
    "detectron2/modeling/proposal_generator/proposal_utils.py", "detectron2/modeling/roi_heads/mask_head.py",
    "detectron2/modeling/meta_arch/dense_detector.py", "detectron2/modeling/roi_heads/fast_rcnn.py",
    "detectron2/layers/deform_conv.py", "detectron2/layers/wrappers.py",
    "detectron2/structures/rotated_boxes.py", "projects/PointRend/point_rend/point_features.py",
    # the reference's OWN unit tests of the hot-path operators (tests/test_gpu_reference_tests.py runs them, unmodified,
    # against this package's operator surface)
    "tests/layers/test_roi_align.py", "tests/layers/test_roi_align_rotated.py", "tests/layers/test_nms.py",
    "tests/layers/test_nms_rotated.py", "tests/layers/test_deformable.py", "tests/structures/test_rotated_boxes.py",
    "tests/structures/test_boxes.py", "tests/modeling/test_roi_pooler.py", "tests/modeling/test_matcher.py",
]
PY_OUT = os.path.join(OUT_DIR, "py")


def pyc_path(relpath):
    return os.path.join(PY_OUT, relpath.replace("/", "__") + "c")


def build_py(force=False):
    """-> True if every module's bytecode is in place."""
    if not os.path.isdir(os.path.join(REF, "detectron2")):
        return all(os.path.exists(pyc_path(m)) for m in PY_MODULES)
    import py_compile


    os.makedirs(PY_OUT, exist_ok=True)
    for m in PY_MODULES:
        src, dst = os.path.join(REF, m), pyc_path(m)
        if force or not os.path.exists(dst) or os.path.getmtime(dst) < os.path.getmtime(src):
            py_compile.compile(src, cfile=dst, doraise=True, invalidation_mode=py_compile.PycInvalidationMode.UNCHECKED_HASH)
    return True


# The reference's WHOLE Python package (detectron2/**/*.py, 156 files), byte-compiled where it lies into
# oracle/_ref/pkg/detectron2/**/*.pyc -- sourceless modules the standard import machinery loads once oracle/_ref/pkg is
# on sys.path.  tests/_reference_model.py imports the reference's GeneralizedRCNN / RetinaNet / DeformBottleneckBlock
# from it (SURVEY 8 row g: the models load unchanged); third-party packages the image lacks are stubbed there.
PKG_OUT = os.path.join(OUT_DIR, "pkg")


def build_pkg(force=False):
    """-> True if the bytecode tree exists."""
    root = os.path.join(REF, "detectron2")
    marker = os.path.join(PKG_OUT, "detectron2", "__init__.pyc")
    if not os.path.isdir(root):
        return os.path.exists(marker)
    import py_compile

    for d, _dirs, files in os.walk(root):
        if "csrc" in d.split(os.sep) or "configs" in d.split(os.sep):
            continue
        for f in files:
            if not f.endswith(".py"):
                continue
            src = os.path.join(d, f)
            dst = os.path.join(PKG_OUT, os.path.relpath(src, REF)) + "c"
            if force or not os.path.exists(dst) or os.path.getmtime(dst) < os.path.getmtime(src):
                os.makedirs(os.path.dirname(dst), exist_ok=True)
                py_compile.compile(src, cfile=dst, doraise=True, invalidation_mode=py_compile.PycInvalidationMode.UNCHECKED_HASH)
    return os.path.exists(marker)


DCN_OUT = os.path.join(OUT_DIR, "_d2ref_C.so")
DCN_SOURCES = [
    os.path.join(CSRC, "deformable", "deform_conv_cuda.cu"),
    os.path.join(CSRC, "deformable", "deform_conv_cuda_kernel.cu"),
    os.path.join(HERE, "ref_dcn_binding.cpp"),
]
# The two .cu files use five CUDA runtime names (deform_conv_cuda_kernel.cu:483-522 and its four twins); hipcc takes
# the <<<>>> launches as they are.  The include shims (oracle/ref_shim/, ours) map the two CUDA-only torch headers.
DCN_RENAMES = ["cudaStream_t=hipStream_t", "cudaError_t=hipError_t", "cudaGetLastError=hipGetLastError",
               "cudaSuccess=hipSuccess", "cudaGetErrorString=hipGetErrorString"]


def build_dcn(force=False, verbose=True):
    """The reference's OWN deformable-convolution kernels (csrc/deformable/*.cu: the only upstream implementation of
    DCN backward) compiled WHERE THEY LIE as HIP for gfx950 into oracle/_ref/_d2ref_C.so -- a CHECKER: a pybind11
    module with detectron2._C's five DCN entry points (oracle/ref_dcn_binding.cpp), under which the reference's own
    layers/deform_conv.py runs (oracle/ref.py: py_deform_conv).  No hipify pass (it writes next to its inputs): hipcc
    -x hip + five -D renames + two shim headers.  hipcc cross-compiles without a GPU.  -> True if the module exists."""
    if not os.path.isdir(CSRC):
        return os.path.exists(DCN_OUT)
    shims = [os.path.join(HERE, "ref_shim", "c10", "cuda", "CUDAGuard.h"),
             os.path.join(HERE, "ref_shim", "ATen", "cuda", "Atomic.cuh")]
    if os.path.exists(DCN_OUT) and not force:
        if os.path.getmtime(DCN_OUT) >= max(os.path.getmtime(s) for s in DCN_SOURCES + shims):
            return True
    import sysconfig

    import torch
    from torch.utils import cpp_extension

    os.makedirs(OUT_DIR, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    abi = int(torch._C._GLIBCXX_USE_CXX11_ABI)
    flags = ["-x", "hip", "--offload-arch=gfx950", "-O2", "-fPIC", "-std=c++17", "-w", "-DWITH_HIP",
             "-DTORCH_EXTENSION_NAME=_d2ref_C", f"-D_GLIBCXX_USE_CXX11_ABI={abi}"]
    flags += cpp_extension.COMMON_HIP_FLAGS + cpp_extension.COMMON_HIPCC_FLAGS
    flags += ["-D" + r for r in DCN_RENAMES]
    flags += ["-I" + os.path.join(HERE, "ref_shim"), "-I" + CSRC, "-I" + sysconfig.get_paths()["include"]]
    flags += ["-I" + p for p in cpp_extension.include_paths()]

    def cc(src):
        obj = os.path.join(OUT_DIR, "dcn_" + os.path.basename(src).rsplit(".", 1)[0] + ".o")
        cmd = [hipcc] + flags + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        return obj

    with ThreadPoolExecutor(max_workers=3) as ex:
        objs = list(ex.map(cc, DCN_SOURCES))
    libdir = os.path.join(os.path.dirname(torch.__file__), "lib")
    cmd = [hipcc, "-shared", "-o", DCN_OUT] + objs + [
        "-L" + libdir, "-ltorch", "-ltorch_cpu", "-ltorch_hip", "-lc10", "-lc10_hip", "-ltorch_python",
        "-Wl,-rpath," + libdir]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    for o in objs:
        os.remove(o)
    return True


def build(force=False, verbose=True):
    build_py(force)
    build_pkg(force)
    build_dcn(force, verbose)
    if not os.path.isdir(CSRC):
        return False
    if os.path.exists(OUT) and not force:
        newest = max(os.path.getmtime(s) for s in SOURCES)
        if os.path.getmtime(OUT) >= newest:
            return True
    import torch
    from torch.utils import cpp_extension

    os.makedirs(OUT_DIR, exist_ok=True)
    incs = cpp_extension.include_paths()
    abi = int(torch._C._GLIBCXX_USE_CXX11_ABI)
    # -O2, no -march: same instruction set (no FMA) as a stock build of the reference
    cflags = ["-O2", "-fPIC", "-std=c++17", f"-D_GLIBCXX_USE_CXX11_ABI={abi}", "-I" + CSRC]
    cflags += ["-I" + p for p in incs]
    objs = []

    def cc(src):
        obj = os.path.join(OUT_DIR, os.path.basename(src).replace(".cpp", ".o"))
        cmd = ["g++", "-c", src, "-o", obj] + cflags
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        return obj

    with ThreadPoolExecutor(max_workers=4) as ex:
        objs = list(ex.map(cc, SOURCES))
    libdir = os.path.join(os.path.dirname(torch.__file__), "lib")
    cmd = ["g++", "-shared", "-o", OUT] + objs + [
        "-L" + libdir, "-ltorch", "-ltorch_cpu", "-lc10", "-Wl,-rpath," + libdir]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    for o in objs:
        os.remove(o)
    return True


if __name__ == "__main__":
    ok = build(force="--force" in sys.argv)
    print("built" if ok else "reference tree not found; skipped", OUT, DCN_OUT)
