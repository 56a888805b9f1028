"""Merge per-seed return-curve runs (`return_curve_parity.py --side gpu|cpu --seeds S` outputs, or `d2_many.py` job files) of TWO arms into ONE
profile: e.g. HIP fast path vs CPU oracle + CPU torch learner, or fast path vs all-eager path on the same GPU.

usage: merge_d2.py OUT.json LABEL A_JSON... -- B_JSON...            (arm A = "hip", arm B = "cpu_oracle" in the output's field names)

PRE-REGISTERED STATISTICS (written in r4 BEFORE the 48-vs-48 seed runs were looked at):
  config 3 at 1024 envs goes through a transition (mean episode length ~300 -> ~800) somewhere between iterations ~400 and ~1,000 depending on
  the seed, so the per-seed values at iteration 1 k are bimodal and their MEAN is the wrong statistic for a +-10 % bar.  The comparison is:
  (1) transition time T600 = first iteration at which the 10-iteration running mean of Train/mean_episode_length exceeds 600
      (censored at iters + 1 when it never does), compared between the arms with a two-sided Mann-Whitney U test; "same" = p >= 0.05;
  (2) the fraction of seeds past the transition at the horizon (T600 <= iters), Fisher exact test, "same" = p >= 0.05;
  (3) the +-10 % bar of north_star on the MEDIAN over seeds of the tail value (mean of the last 10 logged values) of Train/mean_reward,
      Train/mean_reward_i (style reward) and Train/mean_episode_length;
  (4) reported, not judged: the means with their standard error (the r2 / r3 statistic), and the same at iterations 250 / 500 / 750.
  verdict "pass" = (1) and (2) "same" and (3) inside the bar on all three tags.          [r4 / r5 registration; (3) superseded below]

AMENDMENT, registered in r6 BEFORE any r6 data: clause (3) is dropped from the verdict.  r5's own table showed the
  horizon MEDIAN of a bimodal quantity changing its verdict three times on one population (-18 % at 29 v 24 seeds: fail; +4.5 % at 61 v 24:
  pass; +11 % for the final code's 32 seeds: fail on the other side) while (1), (2), the means and the rank tests said "same" at every size:
  the median of a two-cluster sample jumps between the clusters with the share of seeds past the transition, which (2) already tests.
  (3') replaces it: a two-sided Mann-Whitney U test on the per-seed TAIL values (mean of the last 10 logged values) of Train/mean_reward,
      Train/mean_reward_i and Train/mean_episode_length between the arms; "same" = p >= 0.05 on every tag.  A rank test sees a shifted
      cluster AND a shifted share, and does not flip when one seed crosses the middle.
  verdict "pass" (r6) = (1) and (2) and (3').  The medians and their +-10 % reading stay in the file, reported and not judged; so does the
  r4 / r5 verdict under its own key, so that older profiles can be re-read under both registrations.
  (5) added AFTER the data were seen and therefore reported, never judged: paired-by-seed statistics for the seeds both arms ran (see the code).
"""
import json, math, statistics, sys

TAGS = ["Train/mean_reward", "Train/mean_reward_t", "Train/mean_reward_i", "Train/mean_episode_length", "Episode/rew_tracking_lin_vel",
        "Episode/rew_tracking_ang_vel", "Episode/rew_torques", "Episode/rew_dof_error", "Episode/rew_collision"]
BAR = {"Train/mean_reward": 0.10, "Train/mean_reward_i": 0.10, "Train/mean_episode_length": 0.10}     # north_star / VERDICT r2 item 6: +-10 %
T_LEVEL, T_WINDOW = 600.0, 10


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


def transition_time(xs, iters):
    """first iteration (1-based) at which the T_WINDOW-iteration running mean of the episode length exceeds T_LEVEL; iters + 1 if never"""
    for i in range(T_WINDOW, len(xs) + 1):
        if sum(xs[i - T_WINDOW:i]) / T_WINDOW > T_LEVEL:
            return i
    return iters + 1


def mann_whitney(a, b):
    """two-sided Mann-Whitney U with the normal approximation + tie correction (scipy when available: exact for small samples)"""
    try:
        from scipy.stats import mannwhitneyu
        r = mannwhitneyu(a, b, alternative="two-sided")
        return float(r.statistic), float(r.pvalue)
    except Exception:
        pass
    return mann_whitney_normal(a, b)


def mann_whitney_normal(a, b):
    """the fallback without scipy: normal approximation with tie correction and continuity correction"""
    allv = sorted([(v, 0) for v in a] + [(v, 1) for v in b])
    ranks, i = {}, 0
    n = len(allv)
    tie_term = 0.0
    rk = [0.0] * n
    while i < n:
        j = i
        while j < n and allv[j][0] == allv[i][0]:
            j += 1
        for t in range(i, j):
            rk[t] = (i + j + 1) / 2.0
        tie_term += (j - i) ** 3 - (j - i)
        i = j
    ra = sum(rk[t] for t in range(n) if allv[t][1] == 0)
    na, nb = len(a), len(b)
    u = ra - na * (na + 1) / 2.0
    mu = na * nb / 2.0
    sd = math.sqrt(na * nb / 12.0 * ((n + 1) - tie_term / (n * (n - 1))))
    z = (abs(u - mu) - 0.5) / sd if sd > 0 else 0.0
    return u, math.erfc(z / math.sqrt(2.0))


def fisher_exact(a_yes, a_n, b_yes, b_n):
    """two-sided Fisher exact test on [[a_yes, a_n - a_yes], [b_yes, b_n - b_yes]]"""
    from math import comb
    tot_yes, n = a_yes + b_yes, a_n + b_n
    p = lambda x: comb(a_n, x) * comb(b_n, tot_yes - x) / comb(n, tot_yes)
    p0 = p(a_yes)
    return sum(p(x) for x in range(max(0, tot_yes - b_n), min(a_n, tot_yes) + 1) if p(x) <= p0 * (1 + 1e-9))


def main():
    out, label = sys.argv[1], sys.argv[2]
    rest = sys.argv[3:]
    k = rest.index("--")
    hip_files, cpu_files = rest[:k], rest[k + 1:]
    hip, mh = load(hip_files)
    cpu, mc = load(cpu_files)
    assert mh == mc, (mh, mc)
    iters = mh["iters"]
    res = {"what": label, "num_envs": mh["num_envs"], "iters": iters, "amp": mh["amp"], "hip_seeds": [r["seed"] for r in hip], "cpu_seeds": [r["seed"] for r in cpu],
           "hip_env_steps_per_s": [round(r["env_steps_per_s"]) for r in hip], "cpu_env_steps_per_s": [round(r["env_steps_per_s"]) for r in cpu],
           "statistic": "tail value = mean of the last 10 logged values before the checkpoint, per seed; rel_diff = (mean over HIP seeds - mean over CPU seeds) / |CPU mean|; "
                        "median_rel_diff likewise on the medians; pre-registered statistics: see tools/merge_d2.py docstring",
           "at_iteration": {}}
    for ck in sorted(set([c for c in (250, 500, 750, 1000) if c <= iters] + [iters])):      # the horizon itself is always a checkpoint
        summ = {}
        for tag in TAGS:
            hs = [tail(r["curves"][tag], ck) for r in hip if tag in r["curves"] and len(r["curves"][tag]) >= ck]
            cs = [tail(r["curves"][tag], ck) for r in cpu if tag in r["curves"] and len(r["curves"][tag]) >= ck]
            if hs and cs:
                a, b = statistics.mean(hs), statistics.mean(cs)
                ma, mb = statistics.median(hs), statistics.median(cs)
                e = {"hip_mean": a, "cpu_oracle_mean": b, "rel_diff": (a - b) / (abs(b) + 1e-12), "hip_median": ma, "cpu_oracle_median": mb,
                     "median_rel_diff": (ma - mb) / (abs(mb) + 1e-12),
                     "cpu_seeds_rank_among_hip_seeds": [sum(1 for h in hs if h < c) / len(hs) for c in cs],        # 0.5 = the HIP median
                     "hip_per_seed": hs, "cpu_per_seed": cs}
                if len(hs) > 1 and len(cs) > 1:
                    se = (statistics.variance(hs) / len(hs) + statistics.variance(cs) / len(cs)) ** 0.5
                    e["rel_diff_standard_error"] = se / (abs(b) + 1e-12)
                    e["mann_whitney_p"] = mann_whitney(hs, cs)[1]
                if tag in BAR and ck == iters:
                    e["bar"] = BAR[tag]; e["pass_on_means"] = abs(e["rel_diff"]) <= BAR[tag]; e["pass_on_medians"] = abs(e["median_rel_diff"]) <= BAR[tag]
                summ[tag] = e
        res["at_iteration"][str(ck)] = summ
    # (1), (2): the transition
    tag = "Train/mean_episode_length"
    th = [transition_time(r["curves"][tag], iters) for r in hip if tag in r["curves"]]
    tc = [transition_time(r["curves"][tag], iters) for r in cpu if tag in r["curves"]]
    u, p_mw = mann_whitney(th, tc)
    yes_h, yes_c = sum(1 for t in th if t <= iters), sum(1 for t in tc if t <= iters)
    p_f = fisher_exact(yes_h, len(th), yes_c, len(tc))
    res["transition"] = {"definition": f"first iteration at which the {T_WINDOW}-iteration running mean of {tag} exceeds {T_LEVEL:g}; {iters + 1} = never within the run",
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
    # (5) POST HOC, reported and not judged (added in r4 AFTER the data were seen, because the data showed it): a seed fixes the initial weights and the
    # env's randomisation in BOTH arms, so per-seed outcomes are correlated between the arms (Spearman ~0.4-0.5 on 22 pairs) and an unpaired test on
    # different seed sets also measures which seeds each arm happened to get.  For the seeds present in both arms: rank correlation and Wilcoxon signed-rank.
    try:
        from scipy.stats import spearmanr, wilcoxon
        by_h = {r["seed"]: r["curves"] for r in hip}
        by_c = {r["seed"]: r["curves"] for r in cpu}
        both = sorted(set(by_h) & set(by_c))
        if len(both) >= 6:
            ph = [transition_time(by_h[s][tag], iters) for s in both]
            pc = [transition_time(by_c[s][tag], iters) for s in both]
            paired = {"seeds": both, "transition_time": {"hip": ph, "cpu_oracle": pc, "spearman_rho": float(spearmanr(ph, pc)[0]),
                                                         "wilcoxon_signed_rank_p": float(wilcoxon([a - b for a, b in zip(ph, pc)]).pvalue) if any(a != b for a, b in zip(ph, pc)) else 1.0}}
            for t in BAR:
                a = [tail(by_h[s][t], iters) for s in both]; b = [tail(by_c[s][t], iters) for s in both]
                dlt = [x - y for x, y in zip(a, b)]
                paired[t] = {"spearman_rho": float(spearmanr(a, b)[0]), "wilcoxon_signed_rank_p": float(wilcoxon(dlt).pvalue),
                             "mean_paired_difference_rel": statistics.mean(dlt) / (abs(statistics.mean(b)) + 1e-12),
                             "median_paired_difference_rel": statistics.median(dlt) / (abs(statistics.median(b)) + 1e-12)}
            res["paired_by_seed (post hoc, not judged)"] = paired
    except ImportError:
        pass
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps({"verdict": res["verdict"], "transition": {k: v for k, v in res["transition"].items() if k not in ("hip", "cpu_oracle")},
                      "final": {t: [round(final[t]["hip_mean"], 4), round(final[t]["cpu_oracle_mean"], 4)] for t in final}}, indent=1))


if __name__ == "__main__":
    main()
