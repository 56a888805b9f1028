#!/usr/bin/env python3
"""r4: the LDS-DMA variant of the dense-layer GEMMs (qa_gemm_dma_kernel, csrc/qa_gemm.hip) on the learner's wide products at a 24,576-row
minibatch: every tile configuration against the register-staged kernel and against the library path it would replace (TunableOp picks
loaded, + the separate ELU / ELU' + bias-sum passes the library needs), correctness of every configuration against fp64, and
interleaved rounds (cdna_hip_programming.md 5.4 rule 24: N variants x M rounds in one process, median and min reported).  Needs a GPU.
  python tools/gemm_dma_bench.py [--rows 24576] [--json out.json]"""
import argparse
import ctypes as C
import json
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quadrupedal_agility_amd import _capi                                   # noqa: E402
from quadrupedal_agility_amd.rsl_rl.algorithms import fused                 # noqa: E402

OLD = {"reg128x128": 0, "reg128x64": 1, "reg64x64": 2, "reg64x128": 3}
DMA = {"dma128x192": 10, "dma128x128": 11, "dma128x64": 12, "dma64x64": 13, "dma64x192": 14}


def time_round(fns, n=10):
    """one interleaved round: each variant n back-to-back launches between two events"""
    out = {}
    for name, fn in fns.items():
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n):
            fn()
        b.record(); b.synchronize()
        out[name] = a.elapsed_time(b) / n * 1e3
    return out


def bench(fns, rounds=7):
    for fn in fns.values():
        fn(); fn()
    torch.cuda.synchronize()
    rows = [time_round(fns) for _ in range(rounds)]
    return {k: {"median_us": round(statistics.median(r[k] for r in rows), 2), "min_us": round(min(r[k] for r in rows), 2)} for k in fns}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=24576)
    ap.add_argument("--json", default=None)
    a = ap.parse_args()
    from quadrupedal_agility_amd.rsl_rl.runners.on_policy_runner import enable_tuned_gemms
    print("tunableop:", enable_tuned_gemms())
    lib = _capi.load_library()
    lib.qa_gemm_force_config.argtypes = [C.c_int32]
    rows = a.rows
    res = {"rows": rows, "layers": []}
    for k, n in [(672, 512), (512, 256), (256, 128), (112, 512), (800, 512)]:
        torch.manual_seed(k + n)
        x = torch.randn(rows, k, device="cuda"); w = torch.randn(n, k, device="cuda") / k ** 0.5; b = torch.randn(n, device="cuda")
        gy = torch.randn(rows, n, device="cuda"); yprev = torch.randn(rows, k, device="cuda")
        y = torch.empty(rows, n, device="cuda")
        fl = 2.0 * rows * k * n
        entry = {"in": k, "out": n, "gflop": round(fl / 1e9, 2)}
        # ---- correctness of every configuration (fp64 reference on a row subset for the products with rows outputs, all of dW)
        sub = torch.arange(0, rows, 37, device="cuda")
        ref_y = torch.nn.functional.elu(x[sub].double() @ w.double().t() + b.double())
        ref_dx = (gy[sub].double() @ w.double()) * torch.where(yprev[sub].double() > 0, torch.ones((), device="cuda", dtype=torch.float64), yprev[sub].double() + 1.0)
        ref_dw, ref_db = gy.double().t() @ x.double(), gy.double().sum(0)
        errs = {}
        for name, cfg in {**OLD, **DMA}.items():
            lib.qa_gemm_force_config(cfg)
            yy = fused.linear_forward_raw(x, w, b, 1, 1.0)
            dx = fused.linear_backward_input_raw(gy, w, yprev, 1, 1.0)
            dw, db = fused.linear_backward_weight_raw(gy, x)
            torch.cuda.synchronize()
            errs[name] = {"fwd": float((yy[sub].double() - ref_y).abs().max()), "dx": float((dx[sub].double() - ref_dx).abs().max()),
                          "dw_rel": float(((dw.double() - ref_dw).abs().max() / ref_dw.abs().max())), "db_rel": float((db.double() - ref_db).abs().max() / ref_db.abs().max())}
        entry["max_abs_error_vs_fp64"] = errs
        bad = {n_: e for n_, e in errs.items() if e["fwd"] > 2e-4 or e["dx"] > 2e-4 or e["dw_rel"] > 2e-5 or e["db_rel"] > 2e-5}
        entry["correct"] = not bad
        if bad:
            print("WRONG:", json.dumps(bad))

        def force(cfg, fn):
            def run():
                lib.qa_gemm_force_config(cfg); fn()
            return run

        # ---- forward: act(x W^T + b)
        def lib_fwd():
            torch.addmm(b, x, w.t(), out=y); torch.nn.functional.elu(y, inplace=True)
        fns = {"lib_addmm+elu": lib_fwd, "lib_addmm_only": lambda: torch.addmm(b, x, w.t(), out=y)}
        for name, cfg in {**OLD, **DMA}.items():
            fns[name] = force(cfg, lambda: fused.linear_forward_raw(x, w, b, 1, 1.0, out=y))
        entry["forward"] = bench(fns)
        # ---- input gradient with the previous layer's ELU'
        def lib_dx():
            fused._elu_bwd(gy @ w, yprev, 1.0)
        fns = {"lib_mm+elu_bwd_bias": lib_dx, "lib_mm_only": lambda: gy @ w}
        for name, cfg in {**OLD, **DMA}.items():
            fns[name] = force(cfg, lambda: fused.linear_backward_input_raw(gy, w, yprev, 1, 1.0))
        entry["input_grad"] = bench(fns)
        # ---- weight + bias gradient
        fns = {"lib_bmm_slabs+slab_sum": lambda: fused.weight_grad(gy, x)}
        for name, cfg in {**OLD, **DMA}.items():
            fns[name] = force(cfg, lambda: fused.linear_backward_weight_raw(gy, x))
        entry["weight_grad"] = bench(fns)
        lib.qa_gemm_force_config(-1)
        for part in ("forward", "input_grad", "weight_grad"):
            for v in entry[part].values():
                v["tflops_at_median"] = round(fl / v["median_us"] / 1e6, 1)
        res["layers"].append(entry)
        print(json.dumps(entry), flush=True)
    if a.json:
        json.dump(res, open(a.json, "w"), indent=1)


if __name__ == "__main__":
    main()
