import ctypes as C, sys, torch
sys.path.insert(0, ".")
from tests.oracle_lib import go2_cfg
from quadrupedal_agility_amd.sim import QaSim
n = 4096
h = QaSim(go2_cfg(n)); h.reset_all()
act = torch.randn(n, 12, device="cuda") * 0.3
for _ in range(50): h.step(act)
buf = torch.zeros(n // 16 * 32, dtype=torch.int64, device="cuda")
h.lib.qa_debug_set_profile_buffer.argtypes = [C.c_void_p, C.c_void_p]
h.lib.qa_debug_set_profile_buffer(h.h, buf.data_ptr())
acc = torch.zeros(10, dtype=torch.float64); K = 50
for _ in range(K):
    h.step(act); torch.cuda.synchronize()
    b = buf.view(-1, 32).cpu().double()[:, 16:27]
    acc += (b[:, 1:] - b[:, :-1]).mean(0)
names = ["kinematics+link inertia", "composite+F+L", "bias (RNEA)", "Linv,G,Schur quad-sum", "6x6 inverse", "unconstrained vel", "contact candidates", "rows", "warm start + PGS", "integrate+forces"]
for i in range(10): print(f"{names[i]:26s} {acc[i].item()/K:9.0f} cycles")
print("substep total", acc.sum().item() / K)
