#!/usr/bin/env python3
"""Is the recorded update path (hipGraph replays of the PPO / discriminator / DAgger steps, discriminator chain beside the PPO chain) the SAME
computation as the eager update path?  (the question behind config 3's return-curve difference.)

Both paths are deterministic given the seed (r3 / r4: twenty fast-path seeds reproduce bit for bit across boxes and under 16-way contention),
so the question has a sharper answer than a two-sample test over seeds: run ONE seed's whole 1,000-iteration training in both modes -- the
eager mode on exactly the sample tables the recorded path draws (`eager_from_tables`: same generator calls) -- and compare the training state
after EVERY iteration: float64 sums of the policy / estimator / discriminator parameters, the three discriminator optimisers' moments, the
normaliser, the replay ring, the prior, the logged losses, the rollout's observations and rewards, the generator state.  Equal for all
iterations = the recorded path IS the eager path on these seeds (no stale reduction, no race, no lost update); a first differing iteration
names where to look.

  python tools/recorded_vs_eager_checksums.py --seeds 3 6 1 --iters 1000 --out gpurun_out/r4c/recorded_vs_eager.json
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(seed, iters, num_envs, mode, out_dir):
    cs = os.path.join(out_dir, f"checksums_{mode}_s{seed}.json")
    env = dict(os.environ, QA_PARITY_CHECKSUMS=cs)
    env["QA_PARITY_TABLES_IN_WARMUP"] = "1"        # both modes draw the discriminator's samples through the tables from the first update on
    if mode.startswith("eager"):
        env["QA_PARITY_EAGER_TABLES"] = "1"
    curves = os.path.join(out_dir, f"curves_{mode}_s{seed}.json")
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "return_curve_parity.py"), "--side", "gpu", "--amp", "--num_envs", str(num_envs), "--iters", str(iters),
                        "--seeds", str(seed), "--out", curves], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, stdin=subprocess.DEVNULL)
    if r.returncode != 0:
        print(r.stdout[-2000:])
        raise SystemExit(f"{mode} seed {seed} failed")
    return json.load(open(cs)), json.load(open(curves)), time.time() - t0


def flat(row):
    out = {}
    for k, v in row.items():
        if isinstance(v, list):
            def walk(prefix, x):
                if isinstance(x, list):
                    for i, y in enumerate(x):
                        walk(f"{prefix}[{i}]", y)
                else:
                    out[prefix] = x
            walk(k, v)
        else:
            out[k] = v
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, nargs="+", default=[3, 6, 1])
    ap.add_argument("--iters", type=int, default=1000)
    ap.add_argument("--num_envs", type=int, default=1024)
    ap.add_argument("--out", required=True)
    ap.add_argument("--keep_rows_around", type=int, default=0, help="keep both modes' full checksum rows for the 6 iterations around this one (diagnostics)")
    ap.add_argument("--eager_twice_iters", type=int, default=0, help="also run the EAGER mode twice for this many iterations and report whether it repeats itself "
                    "(is the eager path deterministic run to run? the recorded path is)")
    a = ap.parse_args()
    out_dir = os.path.dirname(os.path.abspath(a.out))
    os.makedirs(out_dir, exist_ok=True)
    res = {"what": "training state after every iteration, recorded update path (the product's default: hipGraph replays, discriminator chain beside the PPO chain) "
                   "vs eager update steps on the same sample tables; BASELINE config 3, float64 checksums", "num_envs": a.num_envs, "iters": a.iters, "seeds": {}}
    for seed in a.seeds:
        rec, cur_r, t_r = run(seed, a.iters, a.num_envs, "recorded", out_dir)
        eag, cur_e, t_e = run(seed, a.iters, a.num_envs, "eager", out_dir)
        n = min(len(rec), len(eag))
        first, fields, worst = None, [], 0.0
        for i in range(n):
            fr, fe = flat(rec[i]), flat(eag[i])
            bad = [k for k in fr if fr[k] != fe.get(k)]
            if bad and first is None:
                first, fields = i + 1, bad[:12]
            for k in bad:
                if isinstance(fr[k], float) and isinstance(fe.get(k), float):
                    worst = max(worst, abs(fr[k] - fe[k]) / (abs(fe[k]) + 1e-30))
        # state (what the next iteration depends on) separately from read-outs (the update's returned mean losses: the recorded path accumulates
        # them on the device, the eager path in Python floats -- a logged number, not state), and the size of the state difference over time
        is_state = lambda k: not k.startswith("losses")
        first_state, state_fields = None, []
        rel_at = {}
        for i in range(n):
            fr, fe = flat(rec[i]), flat(eag[i])
            bad = [k for k in fr if is_state(k) and fr[k] != fe.get(k)]
            if bad and first_state is None:
                first_state, state_fields = i + 1, bad[:12]
            if i + 1 in (1, 2, 5, 10, 20, 50, 100, 200, 500, 1000, n):
                rel_at[str(i + 1)] = {k: abs(fr[k] - fe[k]) / (abs(fe[k]) + 1e-30) for k in ("policy", "estimator", "disc", "norm", "obs", "rewards") if k in fr and k in fe}
        if a.keep_rows_around > 0:
            lo = max(0, (first_state or a.keep_rows_around) - 4)
            res.setdefault("rows_around_first_state_difference", {})[str(seed)] = {"from_iteration": lo + 1, "recorded": [flat(r) for r in rec[lo:lo + 6]], "eager": [flat(r) for r in eag[lo:lo + 6]]}
        tail = lambda c, tag: sum(c["rows"][0]["curves"][tag][-10:]) / 10
        res["seeds"][str(seed)] = {"iterations_compared": n, "first_differing_iteration": first, "fields_differing_there": fields, "largest_relative_checksum_difference": worst,
                                   "first_iteration_with_a_state_difference": first_state, "state_fields_differing_there": state_fields,
                                   "relative_checksum_difference_at_iteration": rel_at,
                                   "identical_for_all_iterations": first is None and n == a.iters,
                                   "episode_length_at_horizon": {"recorded": tail(cur_r, "Train/mean_episode_length"), "eager": tail(cur_e, "Train/mean_episode_length")},
                                   "mean_reward_at_horizon": {"recorded": tail(cur_r, "Train/mean_reward"), "eager": tail(cur_e, "Train/mean_reward")},
                                   "wall_s": {"recorded": round(t_r, 1), "eager": round(t_e, 1)}}
        print(json.dumps({seed: res["seeds"][str(seed)]}), flush=True)
        for mode in ("recorded", "eager"):           # the checksum files are large: keep the verdict, drop the raw rows
            for kind in ("checksums", "curves"):
                try:
                    os.remove(os.path.join(out_dir, f"{kind}_{mode}_s{seed}.json"))
                except OSError:
                    pass
    if a.eager_twice_iters > 0:
        seed = a.seeds[0]
        e1, _, _ = run(seed, a.eager_twice_iters, a.num_envs, "eager", out_dir)
        e2, _, _ = run(seed, a.eager_twice_iters, a.num_envs, "eager_again", out_dir)
        r1, _, _ = run(seed, a.eager_twice_iters, a.num_envs, "recorded", out_dir)
        r2, _, _ = run(seed, a.eager_twice_iters, a.num_envs, "recorded_again", out_dir)
        def first_diff(x, y, state_only):
            for i in range(min(len(x), len(y))):
                fx, fy = flat(x[i]), flat(y[i])
                bad = [k for k in fx if fx[k] != fy.get(k) and (not state_only or not k.startswith("losses"))]
                if bad:
                    return {"iteration": i + 1, "fields": bad[:8]}
            return None
        res["run_to_run"] = {"seed": seed, "iters": a.eager_twice_iters,
                             "eager_vs_eager_first_difference": first_diff(e1, e2, False), "recorded_vs_recorded_first_difference": first_diff(r1, r2, False),
                             "recorded_vs_eager_first_state_difference": first_diff(r1, e1, True), "recorded_vs_eager_first_difference_any_field": first_diff(r1, e1, False)}
        print(json.dumps(res["run_to_run"]), flush=True)
    json.dump(res, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
