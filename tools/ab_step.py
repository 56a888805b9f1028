"""Dev tool (r5): the arena after K env steps from one reset, once per build of the library (QA_LIB), saved to .npy -- and the comparison of two
such files per tensor.   python tools/ab_step.py run OUT.npy [K] [N]   |   python tools/ab_step.py cmp A.npy B.npy"""
import os, sys
import numpy as np
sys.path.insert(0, ".")
if sys.argv[1] == "run":
    import torch
    from quadrupedal_agility_amd import _capi
    if os.environ.get("QA_LIB"): _capi.LIB_PATH = os.environ["QA_LIB"]
    from tests.oracle_lib import go2_cfg
    from quadrupedal_agility_amd.sim import QaSim
    K = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    n = int(sys.argv[4]) if len(sys.argv) > 4 else 64
    over = dict(kv.split("=") for kv in os.environ.get("AB_CFG", "").split(",") if kv)
    h = QaSim(go2_cfg(n, seed=1, **{k: int(v) for k, v in over.items()})); h.reset_all()
    g = torch.Generator().manual_seed(0)
    scale = float(os.environ.get("AB_ACT", "1.0"))
    for _ in range(K):
        h.step((torch.randn(n, 12, generator=g) * scale).cuda())
    torch.cuda.synchronize()
    out = {k: v.cpu().numpy() for k, v in h.t.items() if k in ("ROOT_STATES", "DOF_STATE", "FOOT_IMPULSE", "CONTACT_FORCES", "OBS", "REW")}
    np.save(sys.argv[2], out, allow_pickle=True)
else:
    a, b = np.load(sys.argv[2], allow_pickle=True).item(), np.load(sys.argv[3], allow_pickle=True).item()
    for k in a:
        d = np.abs(a[k].astype(np.float64) - b[k].astype(np.float64)).reshape(a[k].shape[0], -1).max(1)
        print(f"{k:16s} max {d.max():.3e}  median over envs {np.median(d):.3e}  envs above 1e-3: {(d > 1e-3).sum()} of {len(d)}")
    d = np.abs(a["DOF_STATE"] - b["DOF_STATE"]).reshape(len(a["DOF_STATE"]), -1).max(1)
    lo = np.array([-1.0472, -1.5708, -2.7227] * 4); hi = np.array([1.0472, 3.4907, -0.83776] * 4)      # URDF limits, hip / thigh / calf (front legs; rear thigh -0.5236 .. 4.5379)
    for e in np.argsort(-d)[:4]:
        q = a["DOF_STATE"][e].reshape(12, 2)[:, 0]
        cf = np.linalg.norm(a["CONTACT_FORCES"][e].reshape(19, 3), axis=1)
        print(f"env {e}: err {d[e]:.2e}; bodies in contact (scalar build): {np.nonzero(cf > 0)[0].tolist()}; joints within 0.2 rad of a (front-leg) stop: {np.nonzero((q - lo < 0.2) | (hi - q < 0.2))[0].tolist()}")
