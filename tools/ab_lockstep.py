"""Dev tool (r5): two builds of the library in ONE process, in lockstep: before every env step build B's arena is overwritten with build A's,
both take the same actions, and the step's outputs are compared -- the first step / env in which B leaves A is printed with that env's state.
  python tools/ab_lockstep.py A.so B.so [steps] [envs] [action scale]"""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from quadrupedal_agility_amd import _capi
from tests.oracle_lib import go2_cfg
from quadrupedal_agility_amd.sim import QaSim

K = int(sys.argv[3]) if len(sys.argv) > 3 else 12
n = int(sys.argv[4]) if len(sys.argv) > 4 else 256
scale = float(sys.argv[5]) if len(sys.argv) > 5 else 0.2
sims = []
for path in sys.argv[1:3]:
    _capi._LIB = None; _capi.LIB_PATH = path
    sims.append(QaSim(go2_cfg(n, seed=1)))
a, b = sims
a.reset_all()
g = torch.Generator().manual_seed(0)
LO = np.array([[-1.0472, -1.5708, -2.7227], [-1.0472, -1.5708, -2.7227], [-1.0472, -0.5236, -2.7227], [-1.0472, -0.5236, -2.7227]]).reshape(12)
HI = np.array([[1.0472, 3.4907, -0.83776], [1.0472, 3.4907, -0.83776], [1.0472, 4.5379, -0.83776], [1.0472, 4.5379, -0.83776]]).reshape(12)
NAMES = ("ROOT_STATES", "DOF_STATE", "FOOT_IMPULSE", "CONTACT_FORCES")
for k in range(K):
    b.arena.copy_(a.arena); b.global_step = a.global_step
    before = {nm: a.t[nm].cpu().numpy().copy() for nm in NAMES}
    act = (torch.randn(n, 12, generator=g) * scale).cuda()
    a.step(act); b.step(act)
    torch.cuda.synchronize()
    d = {nm: np.abs(a.t[nm].cpu().numpy().astype(np.float64) - b.t[nm].cpu().numpy()).reshape(n, -1).max(1) for nm in NAMES}
    bad = d["DOF_STATE"] > 1e-2
    print(f"step {k}: max diff " + " ".join(f"{nm} {d[nm].max():.2e}" for nm in NAMES) + f"; envs with DOF_STATE diff > 1e-2: {int(bad.sum())} of {n}")
    if bad.any():
        for e in np.argsort(-d["DOF_STATE"])[:3]:
            q = before["DOF_STATE"][e].reshape(12, 2)
            cf0 = np.linalg.norm(before["CONTACT_FORCES"][e].reshape(19, 3), axis=1)
            cfa = np.linalg.norm(a.t["CONTACT_FORCES"][e].cpu().numpy().reshape(19, 3), axis=1)
            cfb = np.linalg.norm(b.t["CONTACT_FORCES"][e].cpu().numpy().reshape(19, 3), axis=1)
            near = np.nonzero((q[:, 0] - LO < 0.2) | (HI - q[:, 0] < 0.2))[0].tolist()
            print(f"  env {e}: diff {d['DOF_STATE'][e]:.2e}; before: z {before['ROOT_STATES'][e][2]:.3f}, bodies in contact {np.nonzero(cf0 > 0)[0].tolist()}, joints within 0.2 of a stop {near}, "
                  f"max |qd| {np.abs(q[:, 1]).max():.2f}, foot impulse {np.round(before['FOOT_IMPULSE'][e], 3).tolist()}")
            print(f"    after A: contacts {np.nonzero(cfa > 0)[0].tolist()} max |qd| {np.abs(a.t['DOF_STATE'][e].cpu().numpy().reshape(12, 2)[:, 1]).max():.2f};  after B: contacts {np.nonzero(cfb > 0)[0].tolist()} max |qd| {np.abs(b.t['DOF_STATE'][e].cpu().numpy().reshape(12, 2)[:, 1]).max():.2f}")
            qa_, qb_ = a.t['DOF_STATE'][e].cpu().numpy().reshape(12, 2), b.t['DOF_STATE'][e].cpu().numpy().reshape(12, 2)
            print(f"    qd A {np.round(qa_[:, 1], 2).tolist()}\n    qd B {np.round(qb_[:, 1], 2).tolist()}")
        break
