"""Time the learner's fp32 GEMM shapes (forward, weight-gradient, input-gradient) as PyTorch launches them, with the
committed TunableOp picks loaded (dev tool).  Prints TFLOP/s per shape and the per-minibatch total."""
import sys
sys.path.insert(0, ".")
import torch
from quadrupedal_agility_amd.rsl_rl.runners.on_policy_runner import enable_tuned_gemms
if "--untuned" not in sys.argv:
    print("tuned:", enable_tuned_gemms())
B = 24576
layers = [("critic0", 671, 512, False), ("critic1", 512, 256, True), ("critic2", 256, 128, True), ("critic_head", 128, 1, True),
          ("actor0", 101, 512, True), ("actor1", 512, 256, True), ("actor2", 256, 128, True), ("actor_head", 128, 12, True),
          ("priv0", 29, 64, False), ("priv1", 64, 29, True), ("est0", 57, 128, False), ("est1", 128, 64, True), ("est2", 64, 4, True)]


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3      # us


tot = 0.0
for name, i, o, need_dx in layers:
    x = torch.randn(B, i, device="cuda")
    if name in ("priv0", "est0"):          # in the update these inputs are column slices of the (B, 671) observation rows
        x = torch.randn(B, 671, device="cuda")[:, 61:61 + i] if name == "priv0" else torch.randn(B, 671, device="cuda")[:, :i]
    w = torch.randn(o, i, device="cuda"); b = torch.randn(o, device="cuda"); g = torch.randn(B, o, device="cuda")
    fl = 2.0 * B * i * o
    t_f = timeit(lambda: torch.addmm(b, x, w.t()))
    t_w = timeit(lambda: g.t().mm(x))
    t_x = timeit(lambda: g.mm(w)) if need_dx else 0.0
    tot += t_f + t_w + t_x
    alt = []
    for S in (4, 8, 16, 32):        # dW with the row reduction split into S slabs through a batched GEMM, partials added
        alt.append((S, timeit(lambda: torch.bmm(g.unflatten(0, (S, B // S)).transpose(1, 2), x.unflatten(0, (S, B // S))).sum(0))))
    t_mv = timeit(lambda: torch.addmv(b.expand(B), x, w[0])) if o == 1 else 0.0
    print(f"{name:12s} {i:4d}->{o:4d}  fwd {t_f:7.1f} us {fl / t_f / 1e6:6.1f} TF/s | dW {t_w:7.1f} us {fl / t_w / 1e6:6.1f} TF/s | dx {t_x:7.1f} us {(fl / t_x / 1e6) if t_x else 0:6.1f} TF/s"
          f" | dW split " + " ".join(f"S{S}:{t:.1f}" for S, t in alt) + (f" | addmv fwd {t_mv:.1f}" if o == 1 else ""))
print(f"sum per minibatch {tot / 1e3:.2f} ms -> x20 = {tot * 20 / 1e3:.1f} ms/iter")
