#!/usr/bin/env python3
"""Where the per-step tolerances of tests/test_hip_parity.py come from.

Runs the single-step parity protocol (both sides start every step from the SAME arena; 40 steps x 1000 envs, the test's own
seeds / action mix) and, for every tensor of the arena, records the distribution of the error over env-steps, split by whether
the two sides ended the step with the SAME discrete state: identical reset flags, identical sets of bodies in contact
(non-zero rows of CONTACT_FORCES), identical sets of active foot rows (non-zero FOOT_IMPULSE triples).  For an env-step the error
of a tensor is the maximum over the env's elements of |hip - oracle|; distributions are reported as p50 / p90 / p99 / p99.9 / max.

  python tools/step_error_distribution.py [--out profiles/r4_step_error_distribution.json] [--envs 1000] [--steps 40] [--terrain]
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

FLOAT_TENSORS = ["ROOT_STATES", "DOF_STATE", "CONTACT_FORCES", "RIGID_BODY_POS", "TORQUES", "TORQUES_ORG", "LAST_DOF_VEL", "LAST_TORQUES_ORG", "LAST_ROOT_VEL",
                 "OBS", "OBS_DISC", "OBS_DISC_TERM", "COMMANDS", "LATENT_EPS", "REW", "EPISODE_SUMS", "FEET_FORCE", "FOOT_IMPULSE", "BASE_LIN_VEL",
                 "BASE_ANG_VEL", "PROJECTED_GRAVITY", "RPY"]
EXACT_TENSORS = ["ACTIONS", "LAST_ACTIONS", "ACTION_HISTORY", "LATENT_C", "RESET", "TIME_OUT", "EPISODE_LENGTH", "LAST_CONTACTS", "CONTACT_FILT"]


def pct(x):
    if len(x) == 0:
        return None
    q = np.quantile(x, [0.5, 0.9, 0.99, 0.999])
    return {"p50": float(q[0]), "p90": float(q[1]), "p99": float(q[2]), "p99.9": float(q[3]), "max": float(np.max(x)), "n": int(len(x))}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    ap.add_argument("--envs", type=int, default=1000)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--seed", type=int, default=7)
    ap.add_argument("--slots", type=int, default=2)
    a = ap.parse_args()
    import torch
    from tests.oracle_lib import OracleSim, go2_cfg
    from quadrupedal_agility_amd.sim import QaSim
    n = a.envs
    q = go2_cfg(n, seed=a.seed, contact_slots=a.slots)
    o, h = OracleSim(q), QaSim(q)
    rng = np.random.default_rng(a.seed)
    o.reset_all()
    o.t["EPISODE_LENGTH"][:] = rng.integers(0, 1000, n)
    o.global_step = 380
    err = {k: [] for k in FLOAT_TENSORS}
    from tests.test_hip_parity import TOL as TOL_
    viol = {}
    mag = {k: 0.0 for k in FLOAT_TENSORS}
    same_all, exact_bad = [], {k: 0 for k in EXACT_TENSORS}
    exact_bad_same = {k: 0 for k in EXACT_TENSORS}
    for k in range(a.steps):
        h.arena.copy_(torch.from_numpy(o.arena.copy()).to(h.arena.device)); h.global_step = o.global_step
        act = rng.normal(0, 1.0, (n, 12)).astype(np.float32)
        if k % 7 == 3:
            act *= 8.0
        o.step(act); h.step(torch.from_numpy(act).cuda()); torch.cuda.synchronize()
        g = {name: h.t[name].cpu().numpy() for name in FLOAT_TENSORS + EXACT_TENSORS}
        cf_g, cf_o = g["CONTACT_FORCES"].reshape(n, -1, 3), o.t["CONTACT_FORCES"].reshape(n, -1, 3)
        fi_g, fi_o = g["FOOT_IMPULSE"].reshape(n, -1, 3), o.t["FOOT_IMPULSE"].reshape(n, -1, 3)
        same = ((np.abs(cf_g).sum(-1) > 0) == (np.abs(cf_o).sum(-1) > 0)).all(1)
        same &= ((np.abs(fi_g).sum(-1) > 0) == (np.abs(fi_o).sum(-1) > 0)).all(1)
        same &= (g["RESET"].reshape(n) == o.t["RESET"].reshape(n))
        same_all.append(same)
        for name in FLOAT_TENSORS:
            x, y = g[name].astype(np.float64), o.t[name].astype(np.float64)
            if name == "EPISODE_SUMS":
                x, y = x.T, y.T
            err[name].append(np.abs(x - y).reshape(n, -1).max(1))
            if name in TOL_:
                viol.setdefault(name, []).append((np.abs(x - y) > TOL_[name][0] + TOL_[name][1] * np.abs(y)).reshape(n, -1).any(1))
            mag[name] = max(mag[name], float(np.abs(y).max()))
        for name in EXACT_TENSORS:
            bad = (g[name].reshape(n, -1) != o.t[name].reshape(n, -1)).any(1)
            exact_bad[name] += int(bad.sum()); exact_bad_same[name] += int((bad & same).sum())
    same = np.concatenate(same_all)
    # env-steps outside the test's tolerance table (tests/test_hip_parity.py: TOL = 3 x the p99.9 measured here), per tensor and as a union
    from tests.test_hip_parity import TOL
    outside, union = {}, np.zeros(same.size, bool)
    for name in FLOAT_TENSORS:
        if name in TOL:
            atol, rtol = TOL[name]
            bad = np.concatenate(viol[name])
            outside[name] = float(bad.mean()); union |= bad
    res = {"what": "per env-step error of the HIP env step against the CPU oracle from identical arenas (max over the env's elements of |hip - oracle|), "
                   "split by whether both sides ended the step in the same discrete state (reset flags, bodies in contact, active foot rows)",
           "envs": n, "steps": a.steps, "seed": a.seed, "contact_slots": a.slots, "env_steps": int(same.size), "same_discrete_state": int(same.sum()),
           "different_discrete_state": int((~same).sum()), "fraction_different": float((~same).mean()),
           "outside_tolerance": {"union_share_of_env_steps": float(union.mean()), "per_tensor": outside,
                                 "tolerances": {k: list(TOL_[k]) for k in FLOAT_TENSORS if k in TOL_}}, "tensors": {}, "integer_tensors": {}}
    for name in FLOAT_TENSORS:
        e = np.concatenate(err[name])
        res["tensors"][name] = {"max_abs_value": mag[name], "same_state": pct(e[same]), "different_state": pct(e[~same])}
    for name in EXACT_TENSORS:
        res["integer_tensors"][name] = {"env_steps_differing": exact_bad[name], "of_which_same_state": exact_bad_same[name]}
    txt = json.dumps(res, indent=1)
    if a.out:
        open(a.out, "w").write(txt)
    print(txt)


if __name__ == "__main__":
    main()
