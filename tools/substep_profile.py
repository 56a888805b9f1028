"""Dev tool: s_memtime stamps of the sections of one physics substep (the last of the 4).  Builds a private copy of
the library with -DQA_SUBPROF (the product library carries no stamps)."""
import ctypes as C, os, subprocess, sys, torch
sys.path.insert(0, ".")
from quadrupedal_agility_amd import _capi
_so = "/tmp/libqa_sim_subprof.so"
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize",
                       "-fno-hip-fp32-correctly-rounded-divide-sqrt", "-DQA_SUBPROF", *[a for a in sys.argv[1:] if a.startswith("-D")], "-shared", "-fPIC", "-o", _so,
                       "quadrupedal_agility_amd/csrc/qa_sim.hip", "quadrupedal_agility_amd/csrc/qa_learner.hip", "quadrupedal_agility_amd/csrc/qa_gemm.hip", "quadrupedal_agility_amd/csrc/qa_conv.hip", "quadrupedal_agility_amd/csrc/qa_policy.hip", "quadrupedal_agility_amd/csrc/qa_tsc.hip", "quadrupedal_agility_amd/csrc/qa_depth.hip"])
_capi.LIB_PATH = _so
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
    order = [0, 1, 2, 3, 5, 6, 7, 8, 4, 9, 10]          # stamp ids in program order
    t = b[:, order]
    acc += (t[:, 1:] - t[:, :-1]).mean(0)
names = ["kinematics+link inertia", "composite+F+L", "bias (RNEA)", "Linv,G,Schur,6x6 inverse", "unconstrained vel", "contact candidates", "rows", "warm start", "PGS sweeps", "integrate"]
for i in range(10): print(f"{names[i]:26s} {acc[i].item()/K:9.0f} cycles")
print("substep total", acc.sum().item() / K)
