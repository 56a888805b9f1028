"""Dev tool: TunableOp picks for the task-level learner's fp32 GEMM shapes (hybrid PPO on 800-wide observations, minibatches of
6 n envs rows).  Eager iterations with tuning enabled; the committed picks are loaded first and kept, new shapes are tuned, the
union goes to gpurun_out/tunableop_gfx950_tsc.csv.  usage: tune_gemms_tsc.py N [N ...]   (envs per GPU to cover)"""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch.cuda.tunable as tunable
from quadrupedal_agility_amd.legged_gym.utils.cfg_to_c import class_to_dict
from quadrupedal_agility_amd.rsl_rl.runners import on_policy_runner as opr
from quadrupedal_agility_amd.tsc.legged_gym.envs.base import legged_robot as lr
from quadrupedal_agility_amd.tsc.legged_gym.envs.go2.go2_agility_config import Go2AgilityCfg, Go2AgilityCfgPPO
from quadrupedal_agility_amd.tsc.rsl_rl.runners import OnPolicyRunner
out = os.path.join(ROOT, "gpurun_out", f"tunableop_gfx950_tsc_{sys.argv[1]}.csv")
os.makedirs(os.path.dirname(out), exist_ok=True)
opr.enable_tuned_gemms()
tunable.enable(True)
tunable.tuning_enable(True)
tunable.set_filename(out)
tunable.set_max_tuning_duration(15)
tunable.set_max_tuning_iterations(50)
for n in [int(a) for a in sys.argv[1:]]:
    t0 = time.time()
    cfg = Go2AgilityCfg(); cfg.env.num_envs, cfg.seed = n, 1
    cfg.obstacle.randomize_start = True
    env = lr.LeggedRobot(cfg, sim_device="cuda:0")
    runner = OnPolicyRunner(env, class_to_dict(Go2AgilityCfgPPO()), log_dir=None, device="cuda:0")
    runner.learn(1, init_at_random_ep_len=True)        # iteration 0 = a DAgger (history-encoder) iteration too
    runner.learn(1)
    torch.cuda.synchronize()
    (tunable.write_file(out) if hasattr(tunable, "write_file") else None)
    print(f"n={n}: tuned in {time.time() - t0:.0f} s, {len(tunable.get_results())} picks", flush=True)
    del runner, env
