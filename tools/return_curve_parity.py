#!/usr/bin/env python3
"""Return-curve parity (BASELINE.json north_star): the same training run on (a) the HIP env + GPU learner and
(b) the CPU oracle env + CPU torch learner, identical seeds/config, compared on Train/mean_reward(_t) and mean
episode length.  Action noise comes from different torch generators on the two devices, so the comparison is
statistical (bands over the last iterations), not per-step; per-step state parity is tests/test_hip_parity.py.

  python tools/return_curve_parity.py --num_envs 512 --iters 60 --seeds 1 2 3     (needs a GPU; ~2 min)
"""
import argparse
import json
import os
import statistics
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(device, num_envs, iters, seed, amp=False):
    import torch
    from quadrupedal_agility_amd.legged_gym.envs import task_registry
    from quadrupedal_agility_amd.legged_gym.envs.go2.go2_locomotion_config import Go2LocomotionCfg, Go2LocomotionCfgAlgo
    from quadrupedal_agility_amd.legged_gym.utils import get_args
    from quadrupedal_agility_amd.legged_gym.utils.cfg_to_c import make_qa_config
    cfg = Go2LocomotionCfg(); cfg.env.num_envs = num_envs; cfg.terrain.mesh_type = "plane"; cfg.env.mocap_state_init = bool(amp); cfg.seed = seed
    t = Go2LocomotionCfgAlgo(); t.runner.amp_enabled = bool(amp); t.seed = seed; t.runner.save_interval = 10 ** 9
    torch.manual_seed(seed)
    log_root = None if os.environ.get("QA_PARITY_NO_LOG") == "1" else (os.environ.get("QA_PARITY_LOG_ROOT") or tempfile.mkdtemp(prefix="qa_parity_"))
    if device == "cpu":
        if os.environ.get("QA_CPU_THREADS"):
            torch.set_num_threads(int(os.environ["QA_CPU_THREADS"]))
        from tests.oracle_backend import OracleBackend
        args = get_args(["--device", "cpu"])
        env, _ = task_registry.make_env("go2_locomotion", args=args, env_cfg=cfg, backend=(OracleBackend if amp else OracleBackend(make_qa_config(cfg, seed=seed))))
    elif device == "hybrid":
        # r5: the oracle's physics on the host cores under the PRODUCT's GPU learner (tools/hybrid_backend.py): the arm that separates
        # "physics route" from "learner arithmetic" in the comparison with the all-CPU arm
        os.environ["QA_ROLLOUT_GRAPH"] = "0"
        from tools.hybrid_backend import HybridBackend
        args = get_args(["--device", "gpu"])
        env, _ = task_registry.make_env("go2_locomotion", args=args, env_cfg=cfg, backend=(HybridBackend if amp else HybridBackend(make_qa_config(cfg, seed=seed))))
    else:
        args = get_args(["--device", "gpu"])
        env, _ = task_registry.make_env("go2_locomotion", args=args, env_cfg=cfg)
    runner, _ = task_registry.make_alg_runner(env, name="go2_locomotion", args=args, train_cfg=t, log_root=log_root)
    if os.environ.get("QA_PARITY_EAGER") == "1":          # GPU side without recorded launches / stream overlap (bisecting a difference)
        runner.alg.use_update_graph = False
        runner.alg.overlap_updates = False
    if os.environ.get("QA_PARITY_EAGER_TABLES") == "1":   # eager update steps on EXACTLY the samples the recorded path draws (same generator calls): bit-comparable with it
        runner.alg.use_update_graph = False
        runner.alg.overlap_updates = False
        runner.alg.eager_from_tables = True
    if os.environ.get("QA_PARITY_TABLES_IN_WARMUP") == "1":   # the recorded path's ONE eager warm-up update samples through the tables too (it uses the generators otherwise)
        runner.alg.eager_from_tables = True
    if os.environ.get("QA_PARITY_EAGER_UPDATE") == "1":
        runner.alg.use_update_graph = False
    if os.environ.get("QA_PARITY_NO_DISC_GRAPH") == "1":
        runner.alg._disc_graph = False
    if os.environ.get("QA_PARITY_NO_AC_GRAPH") == "1":
        runner.alg._ac_graph = False
    if os.environ.get("QA_PARITY_CHECKSUMS"):             # per-iteration state checksums (diffing two deterministic variants)
        a, rows, orig = runner.alg, [], runner.alg.update
        def update_and_checksum(*args, **kw):
            r = orig(*args, **kw)
            cs = lambda ps: float(torch.cat([p.detach().flatten() for p in ps]).double().sum())
            rows.append(dict(policy=cs(a.actor_critic.parameters()), estimator=cs(a.estimator.parameters()),
                             disc=cs(a.disc.parameters()) if amp else 0.0, norm=float(a.disc_normalizer.mean.double().sum()) if amp else 0.0,
                             ring=[int(a.disc_storage.num_samples), int(a.disc_storage.step)] if amp else [],
                             ring_sum=float(a.disc_storage.states.double().sum()) if amp else 0.0,
                             disc_adam=[[float(sum(st[k].double().sum() for st in o.state.values())) for k in ("exp_avg", "exp_avg_sq", "step")]
                                        for o in (a.optim_d, a.optim_q_eps, a.optim_q_c)] if amp else [],
                             prior=float(env.prior_parameters.double().sum()) if amp else 0.0, losses=[float(v) for v in r] if amp else [],
                             disc_each=[float(p.detach().double().sum()) for p in a.disc.parameters()] if amp else [],
                             obs=float(a.storage.observations.double().sum()), rewards=float(a.storage.rewards.double().sum()),
                             policy_each=[float(p.detach().double().sum()) for p in a.actor_critic.parameters()],       # which tensor of the policy moves first
                             lr_ac=float(a.lr_ac), std=[float(v) for v in a.actor_critic.std.detach().flatten().tolist()] if hasattr(a.actor_critic, "std") else [],
                             ac_adam=[float(sum(st[k].double().sum() for st in a.optim_ac.state.values())) for k in ("exp_avg", "exp_avg_sq", "step")],
                             ))      # (the generator state is NOT read: get_rng_state() on a graph-registered generator may move its offset)
            if len(rows) % 50 == 0 or len(rows) >= iters:
                json.dump(rows, open(os.environ["QA_PARITY_CHECKSUMS"], "w"))
            return r
        a.update = update_and_checksum
    t0 = time.time()
    runner.learn(iters, init_at_random_ep_len=True)
    wall = time.time() - t0
    curves = {}
    path = os.path.join(runner.log_dir or "/nonexistent", "scalars.jsonl")
    if os.path.exists(path):
        for line in open(path):
            r = json.loads(line)
            curves.setdefault(r["tag"], []).append(r["value"])
    return curves, wall, num_envs * 24 * iters / wall


def tail_mean(xs, k=10):
    xs = xs[-k:]
    return sum(xs) / max(len(xs), 1)


def merge(gpu_json, cpu_json, out):
    g, c = json.load(open(gpu_json)), json.load(open(cpu_json))
    assert (g["num_envs"], g["iters"]) == (c["num_envs"], c["iters"])
    tags = ["Train/mean_reward", "Train/mean_reward_t", "Train/mean_reward_i", "Train/mean_episode_length", "Episode/rew_tracking_lin_vel",
            "Episode/rew_tracking_ang_vel", "Episode/rew_torques", "Episode/rew_dof_error", "Episode/rew_collision"]
    summary = {}
    for tag in tags:
        hs = [tail_mean(r["curves"][tag]) for r in g["rows"] if tag in r["curves"]]
        cs = [tail_mean(r["curves"][tag]) for r in c["rows"] if tag in r["curves"]]
        if hs and cs:
            mh, mc = statistics.mean(hs), statistics.mean(cs)
            summary[tag] = {"hip_mean": mh, "cpu_oracle_mean": mc, "rel_diff": (mh - mc) / (abs(mc) + 1e-12),
                            "hip_per_seed": hs, "cpu_per_seed": cs}
    res = {"num_envs": g["num_envs"], "iters": g["iters"], "seeds": [r["seed"] for r in g["rows"]],
           "hip_env_steps_per_s": [r["env_steps_per_s"] for r in g["rows"]], "cpu_env_steps_per_s": [r["env_steps_per_s"] for r in c["rows"]],
           "summary": summary}
    print(json.dumps(res, indent=1))
    if out:
        json.dump(res, open(out, "w"), indent=1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--num_envs", type=int, default=512)
    ap.add_argument("--iters", type=int, default=60)
    ap.add_argument("--seeds", type=int, nargs="+", default=[1, 2, 3])
    ap.add_argument("--out", type=str, default=None)
    ap.add_argument("--side", choices=["both", "gpu", "cpu", "hybrid"], default="both",
                    help="run one side only (the CPU-oracle side needs no GPU) and merge later with --merge")
    ap.add_argument("--merge", nargs=2, default=None, metavar=("GPU_JSON", "CPU_JSON"))
    ap.add_argument("--amp", action="store_true", help="BASELINE config 3: discriminator on, mocap-state resets (baked real clips)")
    a = ap.parse_args()
    if a.merge:
        return merge(a.merge[0], a.merge[1], a.out)
    if a.side != "both":
        rows = []
        for seed in a.seeds:
            cur, wall, fps = run(a.side, a.num_envs, a.iters, seed, a.amp)
            rows.append({"seed": seed, "env_steps_per_s": fps, "curves": {k: v for k, v in cur.items() if k.startswith(("Train/", "Episode/", "Loss"))}})
            print(a.side, "seed", seed, "done in", round(wall, 1), "s", flush=True)
        json.dump({"side": a.side, "amp": bool(a.amp), "num_envs": a.num_envs, "iters": a.iters, "rows": rows}, open(a.out, "w"))
        return
    tags = ["Train/mean_reward", "Train/mean_reward_t", "Train/mean_episode_length", "Episode/rew_tracking_lin_vel", "Episode/rew_torques"]
    rows = []
    for seed in a.seeds:
        g, wg, fg = run("gpu", a.num_envs, a.iters, seed)
        c, wc, fc = run("cpu", a.num_envs, a.iters, seed)
        row = {"seed": seed, "gpu_env_steps_per_s": fg, "cpu_env_steps_per_s": fc}
        for tag in tags:
            if tag in g and tag in c:
                row[tag] = {"hip": tail_mean(g[tag]), "cpu_oracle": tail_mean(c[tag])}
        rows.append(row)
        print(json.dumps(row))
    summary = {}
    for tag in tags:
        hs = [r[tag]["hip"] for r in rows if tag in r]; cs = [r[tag]["cpu_oracle"] for r in rows if tag in r]
        if hs:
            mh, mc = statistics.mean(hs), statistics.mean(cs)
            summary[tag] = {"hip_mean": mh, "cpu_mean": mc, "rel_diff": (mh - mc) / (abs(mc) + 1e-12),
                            "hip_range": [min(hs), max(hs)], "cpu_range": [min(cs), max(cs)]}
    out = {"num_envs": a.num_envs, "iters": a.iters, "seeds": a.seeds, "per_seed": rows, "summary": summary}
    print(json.dumps(out["summary"], indent=1))
    if a.out:
        json.dump(out, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
