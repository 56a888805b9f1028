"""Dev tool (GPU box): are HIP-vs-oracle outliers legitimate threshold flips?  For every env-step where the two
disagree beyond tolerance, re-run the ORACLE from a state perturbed by 1e-6 and measure how far the oracle moves
from itself.  A flip is 'explained' if the oracle's self-sensitivity is of the same order as the HIP deviation."""
import sys
import numpy as np, torch
sys.path.insert(0, ".")
from tests.oracle_lib import OracleSim, go2_cfg
from tests.test_hip_parity import make_pair, push_arena, env_mismatch, TOL

n, seed = 1000, 7
q, o, h = make_pair(n, seed=seed)
o2 = OracleSim(q)
rng = np.random.default_rng(seed)
o.reset_all(); o.t["EPISODE_LENGTH"][:] = rng.integers(0, 1000, n); o.global_step = 380
rows = []
for k in range(40):
    push_arena(o, h)
    pre = o.arena.copy()
    act = rng.normal(0, 1.0, (n, 12)).astype(np.float32)
    if k % 7 == 3: act *= 8.0
    o.step(act); h.step(torch.from_numpy(act).cuda()); torch.cuda.synchronize()
    bad = np.zeros(n, bool)
    for name in TOL: bad |= env_mismatch(name, h.t[name].cpu().numpy(), o.t[name], n)
    if bad.any():
        o2.arena[:] = pre; o2.global_step = o.global_step - 1
        o2.t["ROOT_STATES"][:, 7:13] += rng.normal(0, 1e-6, (n, 6)).astype(np.float32)
        o2.t["DOF_STATE"][:] += rng.normal(0, 1e-6, (n, 12, 2)).astype(np.float32)
        o2.step(act)
        for e in np.nonzero(bad)[0]:
            dh = np.abs(h.t["DOF_STATE"].cpu().numpy()[e] - o.t["DOF_STATE"][e]).max()
            do = np.abs(o2.t["DOF_STATE"][e] - o.t["DOF_STATE"][e]).max()
            cf_h = h.t["CONTACT_FORCES"].cpu().numpy()[e]; cf_o = o.t["CONTACT_FORCES"][e]
            nb_h = (np.abs(cf_h).sum(1) > 0).nonzero()[0].tolist(); nb_o = (np.abs(cf_o).sum(1) > 0).nonzero()[0].tolist()
            q_ = o.t["DOF_STATE"][e, :, 0]
            rows.append((k, e, dh, do, nb_h, nb_o))
print("step env  |hip-oracle|dof  |oracle(perturbed)-oracle|dof  bodies-in-contact hip / oracle")
for r in rows[:80]: print(r)
expl = sum(1 for r in rows if r[3] > 0.1 * r[2])
print(f"{len(rows)} outliers, {expl} explained by oracle self-sensitivity to a 1e-6 perturbation")
