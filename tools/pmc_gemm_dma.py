#!/usr/bin/env python3
"""Counter target for the LDS-DMA GEMM kernel (run under rocprofv3 --pmc ...): the three products of the 24,576 x 672 x 512 layer on the DMA tiles that
timed best (forward 128x192, input gradient 64x64, weight gradient 128x128) and, for comparison, on the register-staged 64x64 tile; 12 launches each.
`summarize DIR`: per kernel name and counter, the mean over the launches (first two dropped)."""
import collections
import csv
import ctypes as C
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def summarize(d):
    acc = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            n = r.get("Kernel_Name", "")
            if "qa_gemm" in n:
                acc[(n.split("(")[0][:70], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (n, c), v in sorted(acc.items()):
        v = v[2:] if len(v) > 4 else v
        print(f"{n:72s} {c:28s} n={len(v):3d} mean={sum(v) / len(v):16.1f}")


def main():
    import torch
    from quadrupedal_agility_amd import _capi
    from quadrupedal_agility_amd.rsl_rl.algorithms import fused
    lib = _capi.load_library()
    lib.qa_gemm_force_config.argtypes = [C.c_int32]
    rows, k, n = 24576, 672, 512
    x = torch.randn(rows, k, device="cuda"); w = torch.randn(n, k, device="cuda") / k ** 0.5; b = torch.randn(n, device="cuda")
    gy = torch.randn(rows, n, device="cuda"); yprev = torch.randn(rows, k, device="cuda"); y = torch.empty(rows, n, device="cuda")
    for cfg_f, cfg_x, cfg_w in ((10, 13, 11), (2, 2, 2)):
        for _ in range(12):
            lib.qa_gemm_force_config(cfg_f); fused.linear_forward_raw(x, w, b, 1, 1.0, out=y)
        for _ in range(12):
            lib.qa_gemm_force_config(cfg_x); fused.linear_backward_input_raw(gy, w, yprev, 1, 1.0)
        for _ in range(12):
            lib.qa_gemm_force_config(cfg_w); fused.linear_backward_weight_raw(gy, x)
        torch.cuda.synchronize()
    lib.qa_gemm_force_config(-1)


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "summarize":
        summarize(sys.argv[2])
    else:
        main()
