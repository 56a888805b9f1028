"""Dev tool: s_memtime stamps of the env-step kernel phases (averaged over blocks and launches)."""
import ctypes as C, sys, torch
sys.path.insert(0, ".")
from tests.oracle_lib import go2_cfg
from quadrupedal_agility_amd.sim import QaSim
n = 4096
h = QaSim(go2_cfg(n)); h.reset_all()
act = torch.randn(n, 12, device="cuda") * 0.3
for _ in range(50): h.step(act)
buf = torch.zeros(n // 4 * 16, dtype=torch.int64, device="cuda")       # 16 slots per workgroup, up to n/4 workgroups (16 lanes per env)
h.lib.qa_debug_set_profile_buffer.argtypes = [C.c_void_p, C.c_void_p]
h.lib.qa_debug_set_profile_buffer(h.h, buf.data_ptr())
acc = torch.zeros(10, dtype=torch.float64)
K = 50
for _ in range(K):
    h.step(act); torch.cuda.synchronize()
    b = buf.view(-1, 16).cpu().double()
    b = b[b[:, 0] > 0]
    acc += (b[:, 1:11] - b[:, 0:10]).mean(0)
names = ["stage table", "action history", "load state", "4 substeps", "refresh/body pos", "post: derived+cmd", "rewards", "reset+stage", "obs head/noise", "scalar writes", "obs row writes"]
tot = acc.sum().item() / K
for i in range(10):
    print(f"{names[i]:22s} {acc[i].item()/K:10.0f} ticks  {100*acc[i].item()/K/tot:5.1f}%")
print("total ticks", tot, "(s_memtime; 100 MHz constant clock => 10 ns per tick)")
