"""Build libd2amd.so (HIP kernels + C ABI) for gfx950 with hipcc, in-tree.

    python -m detectron2_amd.build [--force] [--verbose]

hipcc cross-compiles without a GPU.  The .so lands in detectron2_amd/lib/ (git-ignored, but it
travels to the GPU box with the repo snapshot).  Objects are cached in detectron2_amd/lib/obj.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "lib")
OBJ_DIR = os.path.join(LIB_DIR, "obj")
LIB = os.path.join(LIB_DIR, "libd2amd.so")
ARCH = "gfx950"

# translation units; value = extra flags.  NMS / IoU / paste must match the CPU ops bit for bit:
# no FMA contraction (explicit fmaf only), IEEE-correct fp32 divide/sqrt (hipcc default).
SOURCES = {
    "api.hip": [],
    "iou.hip": ["-ffp-contract=off"],
    "nms.hip": ["-ffp-contract=off"],
    "paste_masks.hip": ["-ffp-contract=off"],
    "roi_align.hip": [],
    "roi_pool.hip": [],
    "roi_pool_rot.hip": [],
    "deform_conv.hip": [],
    "deform_conv_tc.hip": [],
    "dcn_bww_gemm.hip": [],
    "dcn_gemm.hip": [],
    "dcn_colpath.hip": [],
    "matcher.hip": ["-ffp-contract=off"],
    "label_sample.hip": ["-ffp-contract=off"],
    "subsample.hip": ["-ffp-contract=off"],
    "random_keys.hip": [],
    "rpn.hip": ["-ffp-contract=off"],
    "topk.hip": ["-ffp-contract=off"],
    "mask_targets.hip": ["-ffp-contract=off"],
    "mask_head.hip": [],
    "box_head.hip": ["-ffp-contract=off"],
    "polygon_masks.hip": ["-ffp-contract=off"],
    "layout.hip": [],
}
# -packed-fp32-ops: no v_pk_{fma,mul,add}_f32.  MEASURED (profiles/r04/LOG.md, "packed fp32 ops beside an MFMA kernel"):
# the DCN data-gradient kernel, whose consumer waves the compiler had vectorised into packed fp32 math, produced wrong
# d(offset) sums in lanes 48-63 of a wave (the last of the four 16-lane passes) in about every second call -- but only
# while ANOTHER kernel full of MFMAs (the weight-gradient GEMM, on a second stream) was running on the same CUs, never
# alone; without the packed ops 80 calls were bit-identical.  The hazard is not one the compiler's recogniser knows,
# so the library does without the instructions everywhere: any of its kernels may share a CU with somebody's GEMM.
# Same-box A/B of the two builds: headline 0.3999 / 0.4023 ms with, 0.4021 / 0.4001 without; dcn_r50 unchanged.
COMMON = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-fno-gpu-rdc", "-Wall",
          "-Wno-unused-function", "-Wno-unused-variable", "-fhip-fp32-correctly-rounded-divide-sqrt",
          "-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def _deps_mtime():
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hdrs.append(os.path.join(os.path.dirname(HERE), "include", "d2amd.h"))
    return max(os.path.getmtime(h) for h in hdrs)


def build(force=False, verbose=False):
    os.makedirs(OBJ_DIR, exist_ok=True)
    hipcc = _hipcc()
    hdr_m = _deps_mtime()
    srcs = [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]

    def cc(name):
        src = os.path.join(CSRC, name)
        obj = os.path.join(OBJ_DIR, name.replace(".hip", ".o"))
        if (not force and os.path.exists(obj)
                and os.path.getmtime(obj) >= max(os.path.getmtime(src), hdr_m)):
            return obj, False
        cmd = [hipcc, "-c", src, "-o", obj] + COMMON + SOURCES[name] + os.environ.get("D2AMD_EXTRA_FLAGS", "").split()
        if verbose:
            print(" ".join(cmd), flush=True)
        # (the host pass of hipcc reports the device-only target feature as unknown: not an error, filtered)
        r = subprocess.run(cmd, stderr=subprocess.PIPE, text=True)
        err = "\n".join(l for l in r.stderr.splitlines() if "'-packed-fp32-ops' is not a recognized feature" not in l)
        if err.strip():
            print(err, file=sys.stderr, flush=True)
        if r.returncode != 0:
            raise subprocess.CalledProcessError(r.returncode, cmd)
        return obj, True

    with ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        res = list(ex.map(cc, srcs))
    objs = [o for o, _ in res]
    if force or any(ch for _, ch in res) or not os.path.exists(LIB):
        cmd = [hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="--verbose" in sys.argv or "-v" in sys.argv))
