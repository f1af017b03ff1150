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
    "detectron2/modeling/meta_arch/dense_detector.py",
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


def build(force=False, verbose=True):
    build_py(force)
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
    print("built" if ok else "reference tree not found; skipped", OUT)
