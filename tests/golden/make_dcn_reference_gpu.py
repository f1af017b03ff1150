"""Generates tests/golden/dcn_reference_gpu.npz: outputs of the REFERENCE's own deformable-convolution code --
detectron2/layers/deform_conv.py (_DeformConv / _ModulatedDeformConv, forward and backward) on top of the reference's
own kernels csrc/deformable/deform_conv_cuda.cu + deform_conv_cuda_kernel.cu, compiled as HIP for gfx950 by
oracle/build_ref.py:build_dcn() into oracle/_ref/_d2ref_C.so -- on the cases of tests/_dcn_cases.py.

The reference has no CPU implementation of DCN backward (deform_conv.py:90-91,210-211), so this has to run on a GPU:

    gpurun -- 'python tests/golden/make_dcn_reference_gpu.py gpurun_out/dcn_reference_gpu.npz'

and the result is copied into tests/golden/.  TEST INFRASTRUCTURE: nothing under detectron2_amd/ is imported here.
Small cases: fp32 inputs, every element of every result.  Full-size cases (BASELINE configs[4] shapes): inputs rounded
to bf16 AND to fp16 (then run through the reference's fp32 kernels: exact-input maths for the 16-bit product paths),
FULL_SAMPLES sampled elements per tensor + fp64 sum / sum of squares of the whole tensor."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))            # tests/
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))  # repo root

import _dcn_cases as dc  # noqa: E402
from oracle import ref   # noqa: E402


def main(out_path):
    assert torch.cuda.is_available(), "the reference's DCN kernels only exist for a GPU"
    m = ref.py_deform_conv()
    out = {"meta_torch": np.array(torch.__version__), "meta_device": np.array(torch.cuda.get_device_name(0))}
    for name in dc.SMALL:
        case = dc.make_small(name)
        res = dc.run_module(m.modulated_deform_conv, m.deform_conv, case, "cuda")
        out[f"small/{name}/checksum"] = np.float64(dc.input_checksum(case))
        for k, v in res.items():
            out[f"small/{name}/{k}"] = v.astype(np.float32)
        print(name, {k: float(np.abs(v).max()) for k, v in res.items()}, flush=True)
    for rounding, tag in ((torch.bfloat16, "bf16"), (torch.float16, "f16")):
        for name in dc.FULL:
            case = dc.make_full(name, rounding)
            res = dc.run_module(m.modulated_deform_conv, m.deform_conv, case, "cuda")
            out[f"full/{tag}/{name}/checksum"] = np.float64(dc.input_checksum(case))
            for k, v in res.items():
                idx = dc.sample_indices(name, k, v.size)
                flat = v.reshape(-1)
                out[f"full/{tag}/{name}/{k}/values"] = flat[idx].astype(np.float32)
                out[f"full/{tag}/{name}/{k}/stats"] = np.array(
                    [flat.astype(np.float64).sum(), (flat.astype(np.float64) ** 2).sum(), float(np.abs(flat).max())])
            print(tag, name, {k: float(np.abs(v).max()) for k, v in res.items()}, flush=True)
    os.makedirs(os.path.dirname(os.path.abspath(out_path)), exist_ok=True)
    np.savez_compressed(out_path, **out)
    print("wrote", out_path, os.path.getsize(out_path), "bytes")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(HERE, "dcn_reference_gpu.npz"))
