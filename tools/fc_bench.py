"""Dev tool: the depth encoder's first Linear (62,400 -> 128) -- library (torch) vs csrc/qa_gemm.hip, forward / input gradient / weight gradient."""
import sys, os, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quadrupedal_agility_amd.rsl_rl.algorithms import fused


def t(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


out = {}
K, N = 62400, 128
w = torch.randn(N, K, device="cuda") * 0.01
bias = torch.randn(N, device="cuda")
for B in (256, 2048):
    x = torch.randn(B, K, device="cuda"); g = torch.randn(B, N, device="cuda")
    r = {}
    r["lib_fwd"] = t(lambda: torch.addmm(bias, x, w.t()))
    r["lib_dx"] = t(lambda: g @ w)
    r["lib_dw"] = t(lambda: g.t() @ x)
    r["own_fwd_nosplit"] = t(lambda: fused.linear_forward_raw(x, w, bias, 1))
    if hasattr(fused, "linear_forward_split_raw"):
        r["own_fwd_split"] = t(lambda: fused.linear_forward_split_raw(x, w, bias, 1))
        y = fused.linear_forward_split_raw(x, w, bias, 1); ref = torch.nn.functional.elu(torch.addmm(bias, x, w.t()))
        r["split_err"] = float((y - ref).abs().max())
    r["own_dx"] = t(lambda: fused.linear_backward_input_raw(g, w, None, 0))
    r["own_dw"] = t(lambda: fused.linear_backward_weight_raw(g, x))
    gf = 2.0 * B * K * N / 1e9
    r["gflop"] = gf
    out[B] = {k: round(v, 1) for k, v in r.items()}
    print(B, out[B], flush=True)
print(json.dumps(out))
