"""Workload for the PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE): 60 fused env steps at N envs, plus a
calibration stream of known size (a 256 MiB float32 copy: 256 MiB read + 256 MiB written)."""
import sys
import torch
sys.path.insert(0, ".")
import os
from quadrupedal_agility_amd import _capi
if os.environ.get("QA_LIB"): _capi.LIB_PATH = os.environ["QA_LIB"]        # an alternative build of the library (A/B counter passes)
from tests.oracle_lib import go2_cfg
from quadrupedal_agility_amd.sim import QaSim
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
lean = int(sys.argv[2]) if len(sys.argv) > 2 else 3        # 3 = what a training run without AMP launches (qa_set_lean_exports); 0 = the reference's exports
h = QaSim(go2_cfg(n)); h.reset_all()
h.set_lean_exports(lean)
act = torch.randn(n, 12, device="cuda") * 0.3
for _ in range(60):
    h.step(act)
x = torch.empty(64 * 1024 * 1024, dtype=torch.float32, device="cuda").normal_()
for _ in range(5):
    y = x.clone()
torch.cuda.synchronize()
print("done")
