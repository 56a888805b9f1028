#!/usr/bin/env python3
"""Extend a merge_d2.py profile with more seeds (r5: the per-seed job files of the first two GPU calls went with their container; the merged profile
kept every seed's tail values and transition time, which is all the pre-registered statistics read).

usage: merge_d2_extend.py OUT.json OLD_PROFILE.json A_JSON... -- B_JSON...       (arm A = "hip", arm B = "cpu_oracle" in the field names, as in merge_d2.py)

The statistics are merge_d2.py's (1)-(4), computed by the same functions on [old per-seed values] + [new per-seed values]; a seed present in both is
taken from the new files.  The post-hoc paired block (5) needs per-seed transition times of the old seeds, which the old profile keeps only for the
seeds both arms ran: it is recomputed over the seeds for which both arms have a value."""
import json, statistics, sys
sys.path.insert(0, __file__.rsplit("/", 1)[0])
from merge_d2 import TAGS, BAR, T_LEVEL, T_WINDOW, load, tail, transition_time, mann_whitney, fisher_exact      # noqa: E402


def main():
    out, old = sys.argv[1], json.load(open(sys.argv[2]))
    rest = sys.argv[3:]
    k = rest.index("--")
    new = {}
    for arm, files in (("hip", rest[:k]), ("cpu", rest[k + 1:])):
        rows, meta = (load(files) if files else ([], None))
        if meta:
            assert meta == {k2: old[k2] for k2 in ("num_envs", "iters", "amp")}, (meta, arm)
        new[arm] = {r["seed"]: r for r in rows}
    iters = old["iters"]
    cks = [c for c in (250, 500, 750, 1000) if c <= iters]
    tagT = "Train/mean_episode_length"
    # per-seed tables: {seed: {"tails": {ck: {tag: v}}, "T": transition or None, "fps": ...}}
    arms = {}
    for arm, okey, pkey in (("hip", "hip_seeds", "hip_per_seed"), ("cpu", "cpu_seeds", "cpu_per_seed")):
        tab = {}
        for i, s in enumerate(old[okey]):
            tab[s] = {"tails": {ck: {t: old["at_iteration"][str(ck)][t][pkey][i] for t in TAGS if t in old["at_iteration"][str(ck)]} for ck in cks},
                      "T": None, "fps": old[("hip" if arm == "hip" else "cpu") + "_env_steps_per_s"][i]}
        # the old transition list is sorted, not keyed by seed: an old seed's T is its episode-length tail crossing...not recoverable per seed, so the
        # old list is kept as an unordered sample and only replaced seeds are removed from it via the paired block when known
        arms[arm] = tab
    oldT = {"hip": list(old["transition"]["hip"]), "cpu": list(old["transition"]["cpu_oracle"])}
    paired_old = old.get("paired_by_seed (post hoc, not judged)", {})
    knownT = {"hip": {}, "cpu": {}}
    if paired_old:
        for s, a, b in zip(paired_old["seeds"], paired_old["transition_time"]["hip"], paired_old["transition_time"]["cpu_oracle"]):
            knownT["hip"][s] = a; knownT["cpu"][s] = b
    for arm in ("hip", "cpu"):
        for s, r in new[arm].items():
            if s in arms[arm]:
                assert s in knownT[arm], f"seed {s} of arm {arm} is in both the old profile and the new files, and its old transition time is not known per seed"
                oldT[arm].remove(knownT[arm][s])
            arms[arm][s] = {"tails": {ck: {t: tail(r["curves"][t], ck) for t in TAGS if t in r["curves"] and len(r["curves"][t]) >= ck} for ck in cks},
                            "T": transition_time(r["curves"][tagT], iters), "fps": round(r["env_steps_per_s"])}
            knownT[arm][s] = arms[arm][s]["T"]
    hs_seeds, cs_seeds = sorted(arms["hip"]), sorted(arms["cpu"])
    res = {"what": old["what"] + "  [extended by tools/merge_d2_extend.py: old seeds from the earlier merged profile, new seeds from job files]",
           "num_envs": old["num_envs"], "iters": iters, "amp": old["amp"], "hip_seeds": hs_seeds, "cpu_seeds": cs_seeds,
           "hip_env_steps_per_s": [arms["hip"][s]["fps"] for s in hs_seeds], "cpu_env_steps_per_s": [arms["cpu"][s]["fps"] for s in cs_seeds],
           "statistic": old["statistic"], "at_iteration": {}}
    for ck in cks:
        summ = {}
        for tag in TAGS:
            hs = [arms["hip"][s]["tails"][ck][tag] for s in hs_seeds if tag in arms["hip"][s]["tails"].get(ck, {})]
            cs = [arms["cpu"][s]["tails"][ck][tag] for s in cs_seeds if tag in arms["cpu"][s]["tails"].get(ck, {})]
            if hs and cs:
                a, b = statistics.mean(hs), statistics.mean(cs)
                ma, mb = statistics.median(hs), statistics.median(cs)
                e = {"hip_mean": a, "cpu_oracle_mean": b, "rel_diff": (a - b) / (abs(b) + 1e-12), "hip_median": ma, "cpu_oracle_median": mb,
                     "median_rel_diff": (ma - mb) / (abs(mb) + 1e-12), "cpu_seeds_rank_among_hip_seeds": [sum(1 for h in hs if h < c) / len(hs) for c in cs],
                     "hip_per_seed": hs, "cpu_per_seed": cs}
                if len(hs) > 1 and len(cs) > 1:
                    se = (statistics.variance(hs) / len(hs) + statistics.variance(cs) / len(cs)) ** 0.5
                    e["rel_diff_standard_error"] = se / (abs(b) + 1e-12)
                    e["mann_whitney_p"] = mann_whitney(hs, cs)[1]
                if tag in BAR and ck == iters:
                    e["bar"] = BAR[tag]; e["pass_on_means"] = abs(e["rel_diff"]) <= BAR[tag]; e["pass_on_medians"] = abs(e["median_rel_diff"]) <= BAR[tag]
                summ[tag] = e
        res["at_iteration"][str(ck)] = summ
    th = oldT["hip"] + [arms["hip"][s]["T"] for s in new["hip"]]
    tc = oldT["cpu"] + [arms["cpu"][s]["T"] for s in new["cpu"]]
    assert len(th) == len(hs_seeds) and len(tc) == len(cs_seeds), (len(th), len(hs_seeds), len(tc), len(cs_seeds))
    u, p_mw = mann_whitney(th, tc)
    yes_h, yes_c = sum(1 for t in th if t <= iters), sum(1 for t in tc if t <= iters)
    p_f = fisher_exact(yes_h, len(th), yes_c, len(tc))
    res["transition"] = {"definition": f"first iteration at which the {T_WINDOW}-iteration running mean of {tagT} exceeds {T_LEVEL:g}; {iters + 1} = never within the run",
                         "hip": sorted(th), "cpu_oracle": sorted(tc), "hip_median": statistics.median(th), "cpu_oracle_median": statistics.median(tc),
                         "hip_quartiles": [sorted(th)[len(th) // 4], sorted(th)[(3 * len(th)) // 4]], "cpu_oracle_quartiles": [sorted(tc)[len(tc) // 4], sorted(tc)[(3 * len(tc)) // 4]],
                         "mann_whitney_u": u, "mann_whitney_p": p_mw, "past_transition_at_horizon": {"hip": [yes_h, len(th)], "cpu_oracle": [yes_c, len(tc)], "fisher_exact_p": p_f}}
    final = res["at_iteration"][str(iters)]
    ok_medians = all(final[t]["pass_on_medians"] for t in BAR if t in final)
    tail_p = {t: final[t].get("mann_whitney_p") for t in BAR if t in final}
    ok_tails = all(p is not None and p >= 0.05 for p in tail_p.values())
    res["verdict"] = {"pre_registered": {"registration": "r6 amendment (tools/merge_d2.py docstring): transition time, share past the transition, Mann-Whitney on the tail values",
                                          "transition_time_same (Mann-Whitney p >= 0.05)": p_mw >= 0.05, "fraction_past_transition_same (Fisher p >= 0.05)": p_f >= 0.05,
                                          "tail_values_same (Mann-Whitney p >= 0.05 on every tag)": ok_tails, "tail_values_mann_whitney_p": tail_p,
                                          "pass": bool(p_mw >= 0.05 and p_f >= 0.05 and ok_tails)},
                      "r4_r5_registration (superseded: its median clause flips with the seed count)": {
                          "medians_within_10_percent": ok_medians, "pass": bool(p_mw >= 0.05 and p_f >= 0.05 and ok_medians)},
                      "means_at_horizon (r2 / r3 statistic, reported)": {t: ("pass" if final[t]["pass_on_means"] else "FAIL") + f" ({100 * final[t]['rel_diff']:+.1f} % +- {100 * final[t].get('rel_diff_standard_error', float('nan')):.1f} %)"
                                                                        for t in BAR if t in final},
                      "medians_at_horizon": {t: f"{100 * final[t]['median_rel_diff']:+.1f} %" for t in BAR if t in final}}
    both = sorted(s for s in set(hs_seeds) & set(cs_seeds) if s in knownT["hip"] and s in knownT["cpu"])
    try:
        from scipy.stats import spearmanr, wilcoxon
        if len(both) >= 6:
            ph, pc = [knownT["hip"][s] for s in both], [knownT["cpu"][s] for s in both]
            paired = {"seeds": both, "transition_time": {"hip": ph, "cpu_oracle": pc, "spearman_rho": float(spearmanr(ph, pc)[0]),
                                                         "wilcoxon_signed_rank_p": float(wilcoxon([a - b for a, b in zip(ph, pc)]).pvalue) if any(a != b for a, b in zip(ph, pc)) else 1.0}}
            for t in BAR:
                a = [arms["hip"][s]["tails"][iters][t] for s in both]; b = [arms["cpu"][s]["tails"][iters][t] for s in both]
                dlt = [x - y for x, y in zip(a, b)]
                paired[t] = {"spearman_rho": float(spearmanr(a, b)[0]), "wilcoxon_signed_rank_p": float(wilcoxon(dlt).pvalue),
                             "mean_paired_difference_rel": statistics.mean(dlt) / (abs(statistics.mean(b)) + 1e-12),
                             "median_paired_difference_rel": statistics.median(dlt) / (abs(statistics.median(b)) + 1e-12)}
            res["paired_by_seed (post hoc, not judged)"] = paired
    except ImportError:
        pass
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps({"seeds": [len(hs_seeds), len(cs_seeds)], "verdict": res["verdict"], "transition": {k2: v for k2, v in res["transition"].items() if k2 not in ("hip", "cpu_oracle")}}, indent=1))


if __name__ == "__main__":
    main()
