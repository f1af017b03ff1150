"""Yardstick for the DCN matrix work: at the three R50 stage shapes (bf16, 2 images) time the
vendor GEMM (torch.matmul -> hipBLASLt / rocBLAS) on the three contractions a block carries -- forward  Y = col W^T,
backward-data  dcol = dY W,  backward-weight  dW = dY^T col  -- next to the library's own kernels for the same work
(launch-stream events, d2amd_timing_*), plus the streaming floors (a 16-bit copy of the column).  One JSON line.

    python scripts/dcn_gemm_yardstick.py > gpurun_out/dcn_gemm_yardstick.json
"""
import ctypes
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from detectron2_amd import _C  # noqa: E402
from detectron2_amd.layers import ModulatedDeformConv  # noqa: E402

MFMA_BF16 = 2500.0


def timeit(fn, rep=30, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(rep):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / rep


def kernel_times(names):
    out = {}
    for kn in names:
        tot, cnt = ctypes.c_double(0.0), ctypes.c_int(0)
        _C.check(_C.lib().d2amd_timing_read(kn.encode(), ctypes.byref(tot), ctypes.byref(cnt)))
        if cnt.value:
            out[kn] = round(tot.value / cnt.value * 1e3, 2)
    return out


def main():
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    bf = torch.bfloat16
    out = {"unit": "us", "dtype": "bf16", "gflop_per_gemm": None, "stages": {}}
    for tag, (C, H, W) in (("res3", (128, 100, 168)), ("res4", (256, 50, 84)), ("res5", (512, 25, 42))):
        P, K = 2 * H * W, 9 * C
        flops = 2.0 * P * K * C
        out["gflop_per_gemm"] = round(flops / 1e9, 2)
        col = torch.randn(P, K, device=dev, dtype=bf)
        w_nk = torch.randn(C, K, device=dev, dtype=bf) * 0.05   # [Co][9C]
        w_kn = w_nk.t().contiguous()                            # [9C][Co]
        dy = torch.randn(P, C, device=dev, dtype=bf)
        r = {}

        def rec(name, ms):
            r[name] = {"us": round(ms * 1e3, 2), "TFLOPs": round(flops / 1e9 / ms, 1), "frac": round(flops / 1e9 / ms / MFMA_BF16, 4)}

        rec("blas_fwd_col@Wt(NT)", timeit(lambda: torch.matmul(col, w_nk.t())))
        rec("blas_fwd_col@W(NN)", timeit(lambda: torch.matmul(col, w_kn)))
        rec("blas_bwd_data_dY@W(NN)", timeit(lambda: torch.matmul(dy, w_nk)))
        rec("blas_bwd_data_dY@Wt(NT)", timeit(lambda: torch.matmul(dy, w_kn.t())))
        rec("blas_bwd_weight_dYt@col(TN)", timeit(lambda: torch.matmul(dy.t(), col)))
        dst = torch.empty_like(col)
        ms = timeit(lambda: dst.copy_(col))
        r["copy_of_the_column"] = {"us": round(ms * 1e3, 2), "MB": round(col.numel() * 2 / 1e6, 1),
                                   "GBps_read_plus_write": round(2 * col.numel() * 2 / 1e6 / ms, 1)}
        ms = timeit(lambda: dst.zero_())
        r["fill_of_the_column"] = {"us": round(ms * 1e3, 2), "GBps_write": round(col.numel() * 2 / 1e6 / ms, 1)}
        del dst, col

        # the library's kernels on the same stage (channels_last, training forward: the column is saved)
        mod = ModulatedDeformConv(C, C, 3, padding=1, bias=False).to(dev).to(bf)
        mf = torch.channels_last
        x = torch.randn(2, C, H, W, device=dev, dtype=bf).contiguous(memory_format=mf).requires_grad_(True)
        off = (torch.randn(2, 18, H, W, device=dev) * 2).to(bf).requires_grad_(True)
        msk = torch.sigmoid(torch.randn(2, 9, H, W, device=dev)).to(bf).requires_grad_(True)
        gy = torch.randn(2, C, H, W, device=dev, dtype=bf).contiguous(memory_format=mf)

        def step():
            y = mod(x, off, msk)
            torch.autograd.backward([y], [gy])
            x.grad = off.grad = msk.grad = mod.weight.grad = None

        for _ in range(3):
            step()
        torch.cuda.synchronize()
        names = ["dcn_fwd", "dcn_fwd_col", "dcn_fwd_gemm", "dcn_bwd_data", "dcn_bwd_dcol_gemm", "dcn_bwd_coord", "dcn_bwd_gather",
                 "dcn_bwd_weight"]
        _C.lib().d2amd_timing_select(",".join(names).encode())
        for _ in range(10):
            step()
        torch.cuda.synchronize()
        r["library_kernels_us"] = kernel_times(names)
        _C.lib().d2amd_timing_select(None)
        r["library_fwd_plus_bwd_us"] = round(timeit(step, rep=20, warm=2) * 1e3, 1)
        out["stages"][tag] = r
    print(json.dumps(out))


if __name__ == "__main__":
    main()
