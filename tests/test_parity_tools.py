"""The statistics behind the return-curve parity verdicts (tools/merge_d2.py, tools/merge_d2_extend.py) -- host code, no GPU.
Known answers for the two tests, and the extension tool against the profile it extends: with no new seeds it must reproduce the verdict, the
transition statistics and every per-seed list of the profile it was given (the r5 verdicts at 61 v 24 seeds were computed through it)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_fisher_and_mann_whitney_known_answers():
    from merge_d2 import fisher_exact, mann_whitney, mann_whitney_normal
    assert abs(fisher_exact(3, 3, 0, 3) - 0.1) < 1e-12                      # [[3, 0], [0, 3]]: 2 / C(6, 3)
    assert abs(fisher_exact(1, 2, 1, 2) - 1.0) < 1e-12
    assert abs(fisher_exact(14, 29, 24, 32) - 0.0381681640) < 1e-9          # the post-hoc comparison quoted in DESIGN 6
    u, p = mann_whitney([1, 2, 3], [4, 5, 6])
    assert u == 0.0 and abs(p - 0.1) < 1e-9                                  # exact: 2 / C(6, 3)
    u2, p2 = mann_whitney_normal([1, 2, 3, 4, 5, 6, 7, 8], [5, 6, 7, 8, 9, 10, 11, 12])
    assert u2 == 8.0 and 0.005 < p2 < 0.02                                   # normal approximation with tie and continuity corrections


def test_extension_without_new_seeds_reproduces_the_profile(tmp_path):
    old = os.path.join(ROOT, "profiles", "r5_return_curve_fast_vs_hybrid_cfg3_amp_1024x1000_29v24.json")
    out = str(tmp_path / "same.json")
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "merge_d2_extend.py"), out, old, "--"], check=True, capture_output=True, timeout=120)
    a, b = json.load(open(old)), json.load(open(out))
    assert b["hip_seeds"] == a["hip_seeds"] and b["cpu_seeds"] == a["cpu_seeds"]
    # the profile was written under the r4 / r5 registration; the tool now judges under the r6 amendment (tools/merge_d2.py docstring) and keeps the
    # old verdict under its own key: every clause the old file judged must come out the same
    old_v, new_v = a["verdict"]["pre_registered"], b["verdict"]["pre_registered"]
    sup = [v for k, v in b["verdict"].items() if k.startswith("r4_r5_registration")][0]
    for k in ("transition_time_same (Mann-Whitney p >= 0.05)", "fraction_past_transition_same (Fisher p >= 0.05)"):
        assert new_v[k] == old_v[k], k
    assert sup["medians_within_10_percent"] == old_v["medians_within_10_percent"] and sup["pass"] == old_v["pass"]
    assert "tail_values_same (Mann-Whitney p >= 0.05 on every tag)" in new_v and set(new_v["tail_values_mann_whitney_p"]) == {"Train/mean_reward", "Train/mean_reward_i", "Train/mean_episode_length"}
    for k in ("means_at_horizon (r2 / r3 statistic, reported)", "medians_at_horizon"):
        assert b["verdict"][k] == a["verdict"][k], k
    for k in ("hip", "cpu_oracle", "mann_whitney_u", "mann_whitney_p", "hip_median", "cpu_oracle_median", "past_transition_at_horizon"):
        assert b["transition"][k] == a["transition"][k], k
    for ck, tags in a["at_iteration"].items():
        for tag, e in tags.items():
            for f in ("hip_per_seed", "cpu_per_seed", "hip_mean", "cpu_oracle_mean", "median_rel_diff"):
                assert b["at_iteration"][ck][tag][f] == e[f], (ck, tag, f)
