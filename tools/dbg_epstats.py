import sys; sys.path.insert(0, ".")
import numpy as np, torch
from tests.oracle_lib import go2_cfg
from quadrupedal_agility_amd.sim import QaSim
from quadrupedal_agility_amd import _capi
n = 256
q = go2_cfg(n, seed=1)
h = QaSim(q); h.reset_all()
scale = np.array(list(q.reward_scale_dt))
print("scales", scale)
g = torch.Generator(device="cuda").manual_seed(0)
bad = 0
for k in range(6000):
    act = torch.randn(n, 12, device="cuda", generator=g) * (0.3 if k % 50 else 3.0)
    h.step(act)
    st = h.t["EPISODE_STATS"][(h.global_step - 1) & 1].cpu().numpy()
    es = h.t["EPISODE_SUMS"].cpu().numpy()
    wrong = (np.sign(es) * np.sign(scale)[:, None] < 0)
    if wrong.any() or (np.sign(st[:14]) * np.sign(scale) < 0).any() or not np.isfinite(es).all():
        r, e = np.nonzero(wrong)
        print("step", k, "stats", st, "wrong sums at (reward, env)", list(zip(r[:5], e[:5])), es[r[:5], e[:5]] if len(r) else None, "finite", np.isfinite(es).all())
        bad += 1
        if bad > 5: break
print("done, bad =", bad)
