import sys, torch
sys.path.insert(0, "/root/repo")
import torch.cuda.tunable as tunable
tunable.enable(True); tunable.tuning_enable(True); tunable.set_filename("/tmp/ld_tune.csv")
tunable.set_max_tuning_duration(30); tunable.set_max_tuning_iterations(100)
B = 24576
def timeit(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
W = torch.randn(512, 671, device="cuda"); b = torch.randn(512, device="cuda"); g = torch.randn(B, 512, device="cuda")
xa = torch.randn(B, 671, device="cuda")
buf = torch.zeros(B, 672, device="cuda"); buf[:, :671] = xa; xb = buf[:, :671]
Wp = torch.zeros(512, 672, device="cuda"); Wp[:, :671] = W
S = 8
for name, x in (("ld671", xa), ("ld672 view", xb)):
    t_f = timeit(lambda: torch.addmm(b, x, W.t()))
    t_w = timeit(lambda: torch.bmm(g.unflatten(0, (S, B // S)).transpose(1, 2), x.unflatten(0, (S, B // S))).sum(0))
    print(f"{name}: fwd {t_f:.1f} us, dW split {t_w:.1f} us")
t_f = timeit(lambda: torch.addmm(b, buf, Wp.t()))
t_w = timeit(lambda: torch.bmm(g.unflatten(0, (S, B // S)).transpose(1, 2), buf.unflatten(0, (S, B // S))).sum(0))
print(f"K=672 padded: fwd {t_f:.1f} us, dW split {t_w:.1f} us")
