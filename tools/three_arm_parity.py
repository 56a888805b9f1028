#!/usr/bin/env python3
"""r5: config 3's return-curve parity as THREE arms.
  fast    HIP env kernel + GPU learner (the product)                               this round's seeds
  hybrid  the oracle's physics on the host cores + the SAME GPU learner             this round's seeds (tools/hybrid_backend.py)
  cpu     the oracle's physics + torch-CPU learner                                  the 7 seeds of r4 (2 h each; their per-seed tail values and
                                                                                   transition times are in profiles/r4_return_curve_parity_cfg3_amp_1024x1000.json)
fast vs hybrid differ ONLY in the physics route (fp32 Schur + PGS kernel vs double dense solve); hybrid vs cpu differ ONLY in the learner's
arithmetic (HIP kernels / hipBLASLt / recorded steps vs torch on the CPU).  The statistics are the pre-registered ones of tools/merge_d2.py
(transition time: Mann-Whitney; share past the transition: Fisher; r6 amendment: Mann-Whitney on the per-seed tail values instead of the
+-10 % reading of the medians, which r5's own table showed flipping with the seed count), applied pair by pair.

usage: three_arm_parity.py OUT.json FAST_VS_HYBRID.json(merge_d2 output) R4_PROFILE.json"""
import json, statistics, sys
sys.path.insert(0, __file__.rsplit("/", 1)[0])
from merge_d2 import mann_whitney, fisher_exact, BAR      # noqa: E402

out, fh, r4 = sys.argv[1:4]
a, b = json.load(open(fh)), json.load(open(r4))
iters = a["iters"]
res = {"what": __doc__.split("usage")[0].strip(), "num_envs": a["num_envs"], "iters": iters,
       "fast_vs_hybrid (physics route; pre-registered statistics)": {"seeds": [len(a["hip_seeds"]), len(a["cpu_seeds"])], "transition": {k: a["transition"][k] for k in a["transition"] if "per" not in k and k not in ("hip", "cpu_oracle")},
                                                                     "verdict": a["verdict"]}}
hyb_T, cpu_T = a["transition"]["cpu_oracle"], b["transition"]["cpu_oracle"]      # in a's field names the SECOND arm ("cpu_oracle") is the hybrid arm
u, p = mann_whitney(hyb_T, cpu_T)
past_h, past_c = sum(t <= iters for t in hyb_T), sum(t <= iters for t in cpu_T)
pf = fisher_exact(past_h, len(hyb_T), past_c, len(cpu_T))
med = {}
horizon_a, horizon_b = a["at_iteration"][str(iters)], b["at_iteration"][str(iters)]
for tag in BAR:
    h = horizon_a[tag]["cpu_per_seed"] if "cpu_per_seed" in horizon_a[tag] else horizon_a[tag]["cpu_oracle_per_seed"]
    c = horizon_b[tag]["cpu_per_seed"] if "cpu_per_seed" in horizon_b[tag] else horizon_b[tag]["cpu_oracle_per_seed"]
    mh, mc = statistics.median(h), statistics.median(c)
    med[tag] = {"tail_values_mann_whitney_p": mann_whitney(h, c)[1], "hybrid_median": mh, "cpu_median": mc, "median_rel_diff": (mh - mc) / abs(mc), "hybrid_mean": statistics.mean(h), "cpu_mean": statistics.mean(c),
                "mean_rel_diff": (statistics.mean(h) - statistics.mean(c)) / abs(statistics.mean(c)), "within_bar": abs(mh - mc) / abs(mc) <= BAR[tag]}
res["hybrid_vs_cpu (learner arithmetic; the CPU arm is r4's 7 seeds)"] = {
    "seeds": [len(hyb_T), len(cpu_T)], "transition": {"hybrid": sorted(hyb_T), "cpu": sorted(cpu_T), "mann_whitney_u": u, "mann_whitney_p": p, "same": p >= 0.05},
    "past_transition": {"hybrid": [past_h, len(hyb_T)], "cpu": [past_c, len(cpu_T)], "fisher_p": pf, "same": pf >= 0.05}, "medians_at_horizon": med,
    "pass": bool(p >= 0.05 and pf >= 0.05 and all(v["tail_values_mann_whitney_p"] >= 0.05 for v in med.values())),
    "registration": "r6 amendment of tools/merge_d2.py: transition time + share past the transition + Mann-Whitney on the tail values",
    "pass_under_the_r4_r5_registration (median clause; superseded)": bool(p >= 0.05 and pf >= 0.05 and all(v["within_bar"] for v in med.values()))}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1)[:6000])
