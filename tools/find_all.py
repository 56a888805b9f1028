import sys, torch
sys.path.insert(0, ".")
from torch.profiler import profile, ProfilerActivity
from quadrupedal_agility_amd.legged_gym.envs import task_registry
from quadrupedal_agility_amd.legged_gym.envs.go2.go2_locomotion_config import Go2LocomotionCfg, Go2LocomotionCfgAlgo
from quadrupedal_agility_amd.legged_gym.utils import get_args
cfg = Go2LocomotionCfg(); cfg.env.num_envs = 4096; cfg.terrain.mesh_type = "plane"; cfg.env.mocap_state_init = False; cfg.seed = 1
t = Go2LocomotionCfgAlgo(); t.runner.amp_enabled = False
args = get_args(["--device", "gpu"])
env, _ = task_registry.make_env("go2_locomotion", args=args, env_cfg=cfg)
runner, _ = task_registry.make_alg_runner(env, name="go2_locomotion", args=args, train_cfg=t, log_root=None)
runner.learn(2, init_at_random_ep_len=True)
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    runner.learn(1)
import collections
cnt = collections.Counter()
for e in prof.events():
    if e.name in ("aten::all", "aten::any", "aten::isfinite", "aten::item", "aten::_local_scalar_dense", "aten::nonzero"):
        st = [s for s in (e.stack or []) if "quadrupedal" in s or "torch/" in s][:6]
        cnt[(e.name, tuple(st))] += 1
for k, v in cnt.most_common(12):
    print(v, k[0]); [print("     ", s) for s in k[1]]
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=30, max_name_column_width=60))
