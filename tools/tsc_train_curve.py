#!/usr/bin/env python3
"""r6: the task-level teacher's training curve for one seed (BASELINE configs[3]'s per-GPU share: 1024 envs, the bench's env / runner construction), logged
through the runner's own `_log` into a small JSON -- the whole-training check of the task-level learner's chain steps against autograd steps
(QA_TRAIN_CHAIN=0), as tools/d2_many.py is for the behaviour-level learner.

  python tools/tsc_train_curve.py --seed 3 --iters 300 --num_envs 1024 --out gpurun_out/x/chain_s3.json
  python tools/tsc_train_curve.py merge OUT.json A_*.json -- B_*.json        (arm A vs arm B: tail means, Mann-Whitney on the per-seed tail values)"""
import json, os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
TAGS = ("Train/mean_reward", "Train/mean_episode_length", "Train/success_rate")

if len(sys.argv) > 1 and sys.argv[1] == "merge":
    import statistics
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from merge_d2 import mann_whitney
    out, rest = sys.argv[2], sys.argv[3:]
    a, b = rest[:rest.index("--")], rest[rest.index("--") + 1:]
    A, B = [json.load(open(f)) for f in a], [json.load(open(f)) for f in b]
    res = {"what": __doc__.split("python")[0].strip(), "num_envs": A[0]["num_envs"], "iters": A[0]["iters"], "arm_a": {"env": A[0]["arm_env"], "seeds": [r["seed"] for r in A]},
           "arm_b": {"env": B[0]["arm_env"], "seeds": [r["seed"] for r in B]}, "tail = mean of the last 20 logged values": {}}
    for tag in TAGS:
        ta, tb = [r["tail"][tag] for r in A], [r["tail"][tag] for r in B]
        u, p = mann_whitney(ta, tb)
        res["tail = mean of the last 20 logged values"][tag] = {"a_per_seed": ta, "b_per_seed": tb, "a_mean": statistics.mean(ta), "b_mean": statistics.mean(tb),
                                                               "a_median": statistics.median(ta), "b_median": statistics.median(tb), "mann_whitney_p": p, "same (p >= 0.05)": bool(p >= 0.05)}
    res["finite"] = all(r["finite"] for r in A + B)
    res["iteration_ms"] = {"a": statistics.mean(r["ms_per_iteration"] for r in A), "b": statistics.mean(r["ms_per_iteration"] for r in B)}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps({k: ({kk: vv for kk, vv in v.items() if "per_seed" not in kk} if isinstance(v, dict) else v) for k, v in res["tail = mean of the last 20 logged values"].items()}, indent=1))
    print("iteration ms", res["iteration_ms"], "finite", res["finite"])
    sys.exit(0)

import argparse, time
ap = argparse.ArgumentParser()
ap.add_argument("--seed", type=int, default=1); ap.add_argument("--iters", type=int, default=300); ap.add_argument("--num_envs", type=int, default=1024)
ap.add_argument("--out", required=True)
args = ap.parse_args()
import torch
from quadrupedal_agility_amd.legged_gym.utils.cfg_to_c import class_to_dict
from quadrupedal_agility_amd.tsc.legged_gym.envs.base import legged_robot as lr
from quadrupedal_agility_amd.tsc.legged_gym.envs.go2.go2_agility_config import Go2AgilityCfg, Go2AgilityCfgPPO
from quadrupedal_agility_amd.tsc.rsl_rl.runners import OnPolicyRunner


class Collect:
    def __init__(self):
        self.rows = {}

    def add_scalar(self, tag, value, it):
        if tag in TAGS:
            self.rows.setdefault(tag, []).append(float(value))

    def flush(self): pass
    def close(self): pass


cfg = Go2AgilityCfg()
cfg.env.num_envs, cfg.seed, cfg.course_seed = args.num_envs, args.seed, 1          # the same course for every seed; the seed moves the draws and the initial weights
d = cfg.domain_rand
d.randomize_base_mass = d.randomize_base_com = d.push_robots = True
cfg.obstacle.randomize_start = True
cfg.depth.use_camera = False
tcfg = class_to_dict(Go2AgilityCfgPPO())
tcfg["depth_encoder"]["if_depth"] = False
tcfg["runner"]["save_interval"] = 10 ** 9
torch.manual_seed(args.seed)
env = lr.LeggedRobot(cfg, sim_device="cuda:0")
log_dir = tempfile.mkdtemp(prefix="qa_tsc_curve_")
runner = OnPolicyRunner(env, tcfg, log_dir=log_dir, device="cuda:0")
runner.save = lambda *a, **k: None
runner.writer = Collect()
torch.manual_seed(args.seed + 104729)
t0 = time.time()
runner.learn(args.iters, init_at_random_ep_len=True)
torch.cuda.synchronize()
wall = time.time() - t0
rows = runner.writer.rows
tail = {t: (sum(rows[t][-20:]) / len(rows[t][-20:]) if rows.get(t) else float("nan")) for t in TAGS}
finite = all(torch.isfinite(v).all().item() for v in runner.alg.actor_critic.state_dict().values())
json.dump({"seed": args.seed, "iters": args.iters, "num_envs": args.num_envs, "arm_env": {k: os.environ[k] for k in ("QA_TRAIN_CHAIN",) if k in os.environ},
           "tail": tail, "curves": {t: [round(v, 4) for v in rows.get(t, [])][::5] for t in TAGS}, "finite": finite, "ms_per_iteration": wall / args.iters * 1e3},
          open(args.out, "w"))
print("seed", args.seed, "tail", tail, "finite", finite, f"{wall / args.iters * 1e3:.1f} ms/iteration")
