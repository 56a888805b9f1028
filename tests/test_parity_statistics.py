"""The statistics behind the return-curve parity verdicts (tools/merge_d2.py) against scipy and hand-checked cases: the verdict files under
profiles/ are only as good as these four functions."""
import importlib.util
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load():
    spec = importlib.util.spec_from_file_location("merge_d2", os.path.join(ROOT, "tools", "merge_d2.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_transition_time_definition():
    m = _load()
    xs = [300.0] * 50 + [700.0] * 50
    # running mean of 10 values exceeds 600 once 8 of the window are 700: (2 * 300 + 8 * 700) / 10 = 620 -> iteration 58; 7 of 10 = 580
    assert m.transition_time(xs, 100) == 58
    assert m.transition_time([300.0] * 100, 100) == 101            # censored: never
    assert m.transition_time([700.0] * 100, 100) == 10             # first full window


def test_fisher_exact_matches_scipy():
    from scipy.stats import fisher_exact
    m = _load()
    for a, an, b, bn in [(3, 12, 0, 16), (26, 40, 6, 10), (15, 22, 17, 22), (0, 5, 5, 5), (4, 4, 26, 40)]:
        ref = fisher_exact([[a, an - a], [b, bn - b]])[1]
        assert abs(m.fisher_exact(a, an, b, bn) - ref) < 1e-12, (a, an, b, bn)


def test_mann_whitney_matches_scipy_with_ties_and_censoring():
    from scipy.stats import mannwhitneyu
    m = _load()
    rng = np.random.default_rng(0)
    for _ in range(5):
        a = list(rng.integers(400, 1002, 40).clip(max=1001))          # censored values tie at 1001
        b = list(rng.integers(400, 1002, 22).clip(max=1001))
        u, p = m.mann_whitney(a, b)
        r = mannwhitneyu(a, b, alternative="two-sided")
        assert abs(u - r.statistic) < 1e-9 and abs(p - r.pvalue) < 1e-12
        u2, p2 = m.mann_whitney_normal(a, b)                             # the no-scipy fallback = scipy's asymptotic method
        r2 = mannwhitneyu(a, b, alternative="two-sided", method="asymptotic", use_continuity=True)
        assert abs(u2 - r2.statistic) < 1e-9 and abs(p2 - r2.pvalue) < 1e-9


def test_merge_end_to_end_on_synthetic_runs(tmp_path):
    """two arms drawn from the SAME bimodal process pass; an arm whose transition is 300 iterations later fails on the pre-registered criteria"""
    rng = np.random.default_rng(1)

    def run(seed, shift):
        t0 = int(rng.integers(450, 1100)) + shift
        ln = [300.0 + rng.normal(0, 5) if i < t0 else 800.0 + rng.normal(0, 5) for i in range(1000)]
        cur = {"Train/mean_episode_length": ln, "Train/mean_reward": [x / 150.0 for x in ln], "Train/mean_reward_i": [x / 400.0 for x in ln]}
        return {"seed": seed, "env_steps_per_s": 1.0, "curves": cur}

    def arm(name, seeds, shift):
        f = tmp_path / f"{name}.json"
        json.dump({"num_envs": 1024, "iters": 1000, "amp": True, "rows": [run(s, shift) for s in seeds]}, open(f, "w"))
        return str(f)

    a, b, c = arm("a", range(1, 41), 0), arm("b", range(1, 41), 0), arm("c", range(1, 41), 300)
    for other, expect in ((b, True), (c, False)):
        out = tmp_path / "o.json"
        subprocess.run([sys.executable, os.path.join(ROOT, "tools", "merge_d2.py"), str(out), "synthetic", a, "--", other], check=True, capture_output=True)
        d = json.load(open(out))
        assert d["verdict"]["pre_registered"]["pass"] is expect, d["verdict"]
        assert len(d["transition"]["hip"]) == 40
        assert "paired_by_seed (post hoc, not judged)" in d
