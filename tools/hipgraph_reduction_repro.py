#!/usr/bin/env python3
"""Stand-alone reproducer (no code of this repo): torch reductions that take the two-stage path return the result of an EARLIER execution
when replayed from a hipGraph on new data.  ROCm 7.2 image, torch 2.10.0+rocm7.0, gfx950 (MI355X).  profiles/r2_hipgraph_stale_reductions.md.

    python tools/hipgraph_reduction_repro.py

For every (op, shape): capture fn(x) into a graph, then 4x { overwrite x with fresh random data; eager fn(x); replay; compare }.
"stale" rows print the relative error of the replayed result against the eager one (~1 = unrelated to the new data)."""
import torch
torch.manual_seed(0)
def check(name, fn, *shape):
    x = torch.randn(*shape, device="cuda")
    fn(x)
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s): fn(x)                       # warm-up on a side stream, as the CUDA-graphs notes ask
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g): out = fn(x)
    worst = 0.0
    for r in range(4):
        x.copy_(torch.randn(*shape, device="cuda"))
        want = fn(x).clone()
        g.replay(); torch.cuda.synchronize()
        worst = max(worst, float((out - want).abs().max() / (want.abs().max() + 1e-30)))
    print(f"{name:22s} {str(shape):16s} {'STALE  rel err %.2g' % worst if worst > 1e-5 else 'ok'}")
for shape in [(61440, 30), (24576, 20), (18432, 10), (6144, 29), (6144, 256), (24576, 256), (24576, 1), (24576, 12), (98304, 19), (6144, 512), (196608, 64)]:
    check("sum(0)", lambda t: t.sum(0), *shape)
for shape in [(6144,), (24576,), (24576, 12), (98304, 19), (61440, 30)]:
    check("mean()", lambda t: t.mean(), *shape)
    check("square().sum()", lambda t: t.square().sum(), *shape)
check("norm(dim=1).mean()", lambda t: t.norm(p=2, dim=1).mean(), 6144, 29)
check("var(0)", lambda t: t.var(0), 24576, 45)
check("max()", lambda t: t.max(), 24576, 12)
ones = None
def mv(t): return torch.mv(t.t(), torch.ones(t.shape[0], device=t.device))
def mm(t): return (torch.ones(1, t.shape[0], device=t.device) @ t)[0]
for shape in [(61440, 30), (18432, 10)]:
    check("mv(t.T, ones)", mv, *shape); check("ones @ t", mm, *shape)
