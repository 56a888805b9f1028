"""Merge the per-seed runs of the return-curve / terminal-reward parity at the stated horizon (1,000 iterations; d2 of VERDICT r2) into ONE
profile: HIP fast path (env kernel + GPU learner, recorded launches) vs CPU oracle + CPU torch learner, same seeds, same config.
usage: merge_d2.py OUT.json LABEL HIP_JSON... -- CPU_JSON...        (each a `return_curve_parity.py --side gpu|cpu --seeds S` output)"""
import json, statistics, sys

out, label = sys.argv[1], sys.argv[2]
rest = sys.argv[3:]
k = rest.index("--")
hip_files, cpu_files = rest[:k], rest[k + 1:]
TAGS = ["Train/mean_reward", "Train/mean_reward_t", "Train/mean_reward_i", "Train/mean_episode_length", "Episode/rew_tracking_lin_vel",
        "Episode/rew_tracking_ang_vel", "Episode/rew_torques", "Episode/rew_dof_error", "Episode/rew_collision"]
BAR = {"Train/mean_reward": 0.10, "Train/mean_reward_i": 0.10, "Train/mean_episode_length": 0.10}     # north_star / VERDICT r2 item 6: +-10 %


def load(files):
    rows, meta = [], None
    for f in files:
        d = json.load(open(f))
        meta = meta or {k: d[k] for k in ("num_envs", "iters", "amp")}
        assert {k: d[k] for k in ("num_envs", "iters", "amp")} == meta, f
        rows += d["rows"]
    return rows, meta


def tail(xs, end, k=10):
    xs = xs[max(0, end - k):end]
    return sum(xs) / max(len(xs), 1)


hip, mh = load(hip_files)
cpu, mc = load(cpu_files)
assert mh == mc, (mh, mc)
iters = mh["iters"]
res = {"what": label, "num_envs": mh["num_envs"], "iters": iters, "amp": mh["amp"], "hip_seeds": [r["seed"] for r in hip], "cpu_seeds": [r["seed"] for r in cpu],
       "hip_env_steps_per_s": [round(r["env_steps_per_s"]) for r in hip], "cpu_env_steps_per_s": [round(r["env_steps_per_s"]) for r in cpu],
       "statistic": "mean of the last 10 logged values before the checkpoint, per seed; rel_diff = (mean over HIP seeds - mean over CPU seeds) / |CPU mean|",
       "at_iteration": {}}
for ck in [c for c in (250, 500, 750, 1000) if c <= iters]:
    summ = {}
    for tag in TAGS:
        hs = [tail(r["curves"][tag], ck) for r in hip if tag in r["curves"] and len(r["curves"][tag]) >= ck]
        cs = [tail(r["curves"][tag], ck) for r in cpu if tag in r["curves"] and len(r["curves"][tag]) >= ck]
        if hs and cs:
            a, b = statistics.mean(hs), statistics.mean(cs)
            e = {"hip_mean": a, "cpu_oracle_mean": b, "rel_diff": (a - b) / (abs(b) + 1e-12), "hip_median": statistics.median(hs), "cpu_oracle_median": statistics.median(cs),
                 "cpu_seeds_rank_among_hip_seeds": [sum(1 for h in hs if h < c) / len(hs) for c in cs],        # 0.5 = the HIP median
                 "hip_per_seed": hs, "cpu_per_seed": cs}
            if len(hs) > 1 and len(cs) > 1:
                se = (statistics.variance(hs) / len(hs) + statistics.variance(cs) / len(cs)) ** 0.5
                e["rel_diff_standard_error"] = se / (abs(b) + 1e-12)
            if tag in BAR and ck == iters:
                e["bar"] = BAR[tag]; e["pass"] = abs(e["rel_diff"]) <= BAR[tag]
            summ[tag] = e
    res["at_iteration"][str(ck)] = summ
final = res["at_iteration"][str(iters)]
res["verdict"] = {t: ("pass" if final[t]["pass"] else "FAIL") + f" ({100 * final[t]['rel_diff']:+.1f} %)" for t in BAR if t in final}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps({"verdict": res["verdict"], "final": {t: [round(final[t]["hip_mean"], 4), round(final[t]["cpu_oracle_mean"], 4)] for t in final}}, indent=1))
