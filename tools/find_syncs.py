import sys, torch, warnings
sys.path.insert(0, ".")
from quadrupedal_agility_amd.legged_gym.envs import task_registry
from quadrupedal_agility_amd.legged_gym.envs.go2.go2_locomotion_config import Go2LocomotionCfg, Go2LocomotionCfgAlgo
from quadrupedal_agility_amd.legged_gym.utils import get_args
cfg = Go2LocomotionCfg(); cfg.env.num_envs = 4096; cfg.terrain.mesh_type = "plane"; cfg.env.mocap_state_init = False; cfg.seed = 1
t = Go2LocomotionCfgAlgo(); t.runner.amp_enabled = "--amp" in sys.argv
args = get_args(["--device", "gpu"])
env, _ = task_registry.make_env("go2_locomotion", args=args, env_cfg=cfg)
runner, _ = task_registry.make_alg_runner(env, name="go2_locomotion", args=args, train_cfg=t, log_root=None)
runner.learn(2, init_at_random_ep_len=True)
import traceback, collections
seen = collections.Counter()
def hook(message, category, filename, lineno, file=None, line=None):
    st = [f"{f.filename.split('/')[-1]}:{f.lineno}:{f.name}" for f in traceback.extract_stack()[:-1] if "quadrupedal" in f.filename or "optim" in f.filename or "clip_grad" in f.filename]
    seen[tuple(st[-4:])] += 1
warnings.showwarning = hook
warnings.simplefilter("always")
torch.cuda.set_sync_debug_mode("warn")
runner.learn(1)
torch.cuda.set_sync_debug_mode("default")
for k, v in seen.most_common(15): print(v, k)
