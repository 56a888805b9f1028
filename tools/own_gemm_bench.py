#!/usr/bin/env python3
"""Times the learner's dense-layer kernels (csrc/qa_gemm.hip) against the library path they replace (torch.addmm + elu, mm,
bmm-slab weight gradient + qa_elu_backward_bias) on the reference's layer shapes at a 24,576-row minibatch.  Needs a GPU.
  python tools/own_gemm_bench.py [--rows 24576] [--json out.json]"""
import argparse
import ctypes as C
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quadrupedal_agility_amd import _capi                                   # noqa: E402
from quadrupedal_agility_amd.rsl_rl.algorithms import fused                 # noqa: E402


def timeit(fn, n=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3      # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=24576)
    ap.add_argument("--json", default=None)
    a = ap.parse_args()
    lib = _capi.load_library()
    lib.qa_gemm_force_config.argtypes = [C.c_int32]
    rows = a.rows
    shapes = [(671, 512), (512, 256), (256, 128), (101, 512), (128, 12), (128, 1), (57, 128), (128, 64), (64, 4), (29, 64), (64, 29), (800, 512)]
    out = []
    for k, n in shapes:
        x = torch.randn(rows, k, device="cuda"); w = torch.randn(n, k, device="cuda") / k ** 0.5; b = torch.randn(n, device="cuda")
        gy = torch.randn(rows, n, device="cuda"); yprev = torch.randn(rows, k, device="cuda")
        fl = 2.0 * rows * k * n
        row = {"in": k, "out": n, "gflop": fl / 1e9}
        y = torch.empty(rows, n, device="cuda")

        def lib_fwd():
            torch.addmm(b, x, w.t(), out=y); torch.nn.functional.elu(y, inplace=True)
        row["fwd_lib_us"] = timeit(lib_fwd)
        for cfg in (-1, 0, 1, 2):
            lib.qa_gemm_force_config(cfg)
            try:
                row[f"fwd_own_cfg{cfg}_us"] = timeit(lambda: fused.linear_forward_raw(x, w, b, 1, 1.0, out=y))
            except RuntimeError as e:
                row[f"fwd_own_cfg{cfg}_us"] = str(e)

        def lib_dx():
            g, _ = fused._elu_bwd(gy @ w, yprev, 1.0)
            return g
        row["dx_lib_us"] = timeit(lib_dx)
        for cfg in (-1, 0, 1, 2):
            lib.qa_gemm_force_config(cfg)
            row[f"dx_own_cfg{cfg}_us"] = timeit(lambda: fused.linear_backward_input_raw(gy, w, yprev, 1, 1.0))

        def lib_dw():
            if n <= 32:
                return fused._NarrowLinear.backward  # placeholder, timed below
            return fused.weight_grad(gy, x), gy.sum(0)
        if n > 32:
            row["dw_lib_us"] = timeit(lib_dw)
        else:
            xr = x.clone(); wr = w.clone().requires_grad_(True); br = b.clone().requires_grad_(True)
            def nar():
                wr.grad = None; br.grad = None
                fused._NarrowLinear.apply(xr, wr, br).backward(gy)
            row["dw_lib_us(narrow fwd+bwd)"] = timeit(nar)
        for cfg in (-1, 0, 1, 2, 3):
            lib.qa_gemm_force_config(cfg)
            row[f"dw_own_cfg{cfg}_us"] = timeit(lambda: fused.linear_backward_weight_raw(gy, x))
        lib.qa_gemm_force_config(-1)
        for key in list(row):
            if key.endswith("_us") and isinstance(row[key], float) and "narrow" not in key:
                row[key.replace("_us", "_tflops")] = round(fl / row[key] / 1e6, 1)
        out.append(row)
        print(json.dumps(row), flush=True)
    if a.json:
        json.dump(out, open(a.json, "w"), indent=1)


if __name__ == "__main__":
    main()
