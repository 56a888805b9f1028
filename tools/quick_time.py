"""Quick timing of the fused env-step kernel (dev tool)."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import os
from quadrupedal_agility_amd import _capi
if os.environ.get("QA_LIB"): _capi.LIB_PATH = os.environ["QA_LIB"]
from tests.oracle_lib import go2_cfg
from quadrupedal_agility_amd.sim import QaSim
TERRAIN = "--terrain" in sys.argv
for n in (4096, 16384):
    q = go2_cfg(n)
    q.contact_slots = int(os.environ.get("QA_SLOTS", "2"))
    if TERRAIN:          # rough field, robots scattered over it
        q.terrain_type = 1; q.hf_rows, q.hf_cols, q.hf_hscale, q.hf_vscale, q.hf_border = 1600, 1600, 0.1, 0.005, 30.0
        q.reset_xy_jitter = 1.0
    h = QaSim(q)
    if TERRAIN:
        g = torch.Generator().manual_seed(0)
        h.t["HEIGHT_SAMPLES"].copy_(torch.randint(-8, 9, (1600, 1600), generator=g, dtype=torch.int16))
        h.t["ENV_ORIGINS"][:, :2] = torch.rand(n, 2, device="cuda") * 90.0 + 5.0
        h.t["ENV_ORIGINS"][:, 2] = 0.05
    h.reset_all()
    act = torch.randn(n, 12, device="cuda") * 0.3
    for _ in range(20): h.step(act)
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    K = 200
    for _ in range(K): h.step(act)
    ev1.record(); torch.cuda.synchronize()
    ms = ev0.elapsed_time(ev1) / K
    print(f"N={n}: {ms*1000:.1f} us/env-step launch, {n/ms*1000:.3e} env-steps/s, resets/step={h.t['RESET'].float().mean().item():.3f}")
