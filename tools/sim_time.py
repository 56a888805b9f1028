import sys, torch
sys.path.insert(0, ".")
from tests.oracle_lib import go2_cfg
from quadrupedal_agility_amd.sim import QaSim
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
for dec, it in ((4, 4), (1, 4), (4, 1), (4, 8)):
    h = QaSim(go2_cfg(n, decimation=dec, solver_iterations=it)); h.reset_all()
    act = torch.randn(n, 12, device="cuda") * 0.3
    for _ in range(50): h.step(act)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200): h.step(act)
    e1.record(); torch.cuda.synchronize()
    print(f"N={n} decimation={dec} sweeps={it}: env_step {e0.elapsed_time(e1)/200*1000:.1f} us")
