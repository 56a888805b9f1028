"""Dev tool: a few hundred iterations at 4096 envs per configuration; prints reward / episode-length checkpoints
(evidence that the recorded-launch training loop learns and stays finite at scale)."""
import json, os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from quadrupedal_agility_amd.legged_gym.envs import task_registry
from quadrupedal_agility_amd.legged_gym.envs.go2.go2_locomotion_config import Go2LocomotionCfg, Go2LocomotionCfgAlgo
from quadrupedal_agility_amd.legged_gym.utils import get_args

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 300
out = {}
for name, amp, mesh in (("cfg2_plane", False, "plane"), ("cfg2_trimesh", False, "trimesh"), ("cfg3_amp_plane", True, "plane")):
    cfg = Go2LocomotionCfg(); cfg.env.num_envs = 4096; cfg.terrain.mesh_type = mesh; cfg.env.mocap_state_init = amp; cfg.seed = 1
    t = Go2LocomotionCfgAlgo(); t.runner.amp_enabled = amp; t.runner.save_interval = 10 ** 9
    torch.manual_seed(1)
    args = get_args(["--device", "gpu"])
    env, _ = task_registry.make_env("go2_locomotion", args=args, env_cfg=cfg)
    log_root = tempfile.mkdtemp(prefix="qa_sanity_")
    runner, _ = task_registry.make_alg_runner(env, name="go2_locomotion", args=args, train_cfg=t, log_root=log_root)
    t0 = time.time(); runner.learn(iters, init_at_random_ep_len=True); torch.cuda.synchronize(); wall = time.time() - t0
    cur = {}
    for line in open(os.path.join(runner.log_dir, "scalars.jsonl")):
        r = json.loads(line); cur.setdefault(r["tag"], []).append(r["value"])
    ck = [k for k in (10, 50, 100, 200, 300, 500, 1000) if k <= iters]
    mean = lambda xs, k: sum(xs[max(0, k - 10):k]) / len(xs[max(0, k - 10):k])
    finite = all(torch.isfinite(v).all().item() for v in runner.alg.actor_critic.state_dict().values())
    out[name] = {"iterations": iters, "wall_s": round(wall, 1), "env_steps_per_s": round(4096 * 24 * iters / wall),
                 "mean_reward": {str(k): round(mean(cur["Train/mean_reward"], k), 3) for k in ck},
                 "mean_episode_length": {str(k): round(mean(cur["Train/mean_episode_length"], k), 1) for k in ck},
                 "weights_finite": finite, "lr_ac_final": float(runner.alg.lr_ac)}
    print(name, json.dumps(out[name]), flush=True)
    del env, runner
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "train_sanity.json"), "w"), indent=1)
