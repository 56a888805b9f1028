"""CLI / seeding / checkpoint-path helpers (API of bbc/legged_gym/utils/helpers.py).

`get_args` is plain argparse: the reference delegates to isaacgym.gymutil.parse_arguments, which
also injects --sim_device/--pipeline/--physics_engine/...; the ones that still mean something
here (--sim_device, --rl_device, --headless, --num_threads) are accepted, the rest are ignored.
"""
import argparse
import os
import random

import numpy as np
import torch

from .cfg_to_c import class_to_dict  # noqa: F401  (re-exported like the reference)


def update_class_from_dict(obj, d):
    for key, val in d.items():
        attr = getattr(obj, key, None)
        if isinstance(attr, type):
            update_class_from_dict(attr, val)
        else:
            setattr(obj, key, val)


def set_seed(seed):
    if seed == -1:
        seed = np.random.randint(0, 10000)
    print("Random seed: {}".format(seed))
    random.seed(seed)
    np.random.seed(seed)
    os.environ["PYTHONHASHSEED"] = str(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
    return seed


def parse_sim_params(args, cfg):
    """The reference fills a gymapi.SimParams; this path only needs dt and the pipeline flag."""
    from quadrupedal_agility_amd.legged_gym.envs.base.legged_robot import _SimParamsView
    return _SimParamsView(cfg["sim"]["dt"], use_gpu_pipeline=getattr(args, "use_gpu_pipeline", True))


def get_load_path(root, load_run=-1, checkpoint=-1):
    try:
        runs = sorted(r for r in os.listdir(root) if r != "exported")
        last_run = os.path.join(root, runs[-1])
    except Exception:
        raise ValueError("No runs in this directory: " + root)
    load_run = last_run if load_run in (-1, "-1") else os.path.join(root, str(load_run))
    if checkpoint == -1:
        models = sorted((f for f in os.listdir(load_run) if "model" in f), key=lambda m: "{0:0>15}".format(m))
        model = models[-1]
    else:
        model = "model_{}.pt".format(checkpoint)
    return os.path.join(load_run, model)


_ENV_OVERRIDES = {
    "num_envs": ("env", "num_envs"), "action_scale": ("control", "action_scale"),
    "tracking_lin_vel": ("rewards.scales", "tracking_lin_vel"), "tracking_ang_vel": ("rewards.scales", "tracking_ang_vel"),
    "jump_up_height": ("rewards.scales", "jump_up_height"), "locomotion_height": ("rewards.scales", "locomotion_height"),
    "disc_history_len": ("env", "disc_history_len"), "disc_obs_len": ("env", "disc_obs_len"),
    "obs_disc_weight_step": ("env", "obs_disc_weight_step"),
    "task_obs_weight_decay_steps": ("normalization", "task_obs_weight_decay_steps"),
}
_TRAIN_OVERRIDES = {
    "max_iterations": ("runner", "max_iterations"), "experiment_name": ("runner", "experiment_name"),
    "load_run": ("runner", "load_run"), "checkpoint": ("runner", "checkpoint"),
    "reward_i_coef": ("runner", "reward_i_coef"), "reward_us_coef": ("runner", "reward_us_coef"),
    "reward_ss_coef": ("runner", "reward_ss_coef"), "reward_t_coef": ("runner", "reward_t_coef"),
    "experiment_idx": ("runner", "experiment_idx"), "us_coef": ("algorithm", "us_coef"), "ss_coef": ("algorithm", "ss_coef"),
    "lr_ac": ("algorithm", "lr_ac"), "lr_q": ("algorithm", "lr_q"), "disc_grad_penalty": ("algorithm", "disc_grad_penalty"),
    "lr_disc": ("algorithm", "lr_disc"), "disc_loss_function": ("algorithm", "disc_loss_function"),
    "bounds_loss_coef": ("algorithm", "bounds_loss_coef"),
}


def _node(cfg, dotted):
    for part in dotted.split("."):
        cfg = getattr(cfg, part)
    return cfg


def update_cfg_from_args(env_cfg, cfg_train, args):
    """Same override table as helpers.py:102-168."""
    if env_cfg is not None:
        for arg, (node, field) in _ENV_OVERRIDES.items():
            v = getattr(args, arg, None)
            if v is not None:
                setattr(_node(env_cfg, node), field, int(v) if arg in ("disc_history_len", "disc_obs_len") else v)
        if getattr(args, "terrain", None) is not None:
            env_cfg.terrain.mesh_type = args.terrain
        if getattr(args, "no_mocap_init", False):
            env_cfg.env.mocap_state_init = False
    if cfg_train is not None:
        if getattr(args, "seed", None) is not None:
            cfg_train.seed = args.seed
        if getattr(args, "resume", False):
            cfg_train.runner.resume = True
            cfg_train.algorithm.priv_reg_coef_schedual = cfg_train.algorithm.priv_reg_coef_schedual_resume
        for arg, (node, field) in _TRAIN_OVERRIDES.items():
            v = getattr(args, arg, None)
            if v is not None:
                setattr(_node(cfg_train, node), field, v)
        if getattr(args, "no_amp", False):
            cfg_train.runner.amp_enabled = False
    return env_cfg, cfg_train


def get_args(argv=None):
    p = argparse.ArgumentParser(description="GAIL")
    p.add_argument("--task", type=str, default="go2_locomotion")
    p.add_argument("--resume", action="store_true", default=False)
    p.add_argument("--experiment_name", type=str)
    p.add_argument("--load_run", type=str, default="-1")
    p.add_argument("--checkpoint", type=int)
    p.add_argument("--headless", type=bool, default=True)
    p.add_argument("--horovod", action="store_true", default=False, help="accepted and ignored, as in the reference")
    p.add_argument("--device", type=str, default="gpu")
    p.add_argument("--device_id", type=int, default=0)
    p.add_argument("--num_envs", type=int)
    p.add_argument("--seed", type=int)
    p.add_argument("--max_iterations", type=int)
    for name in ("reward_i_coef", "reward_us_coef", "reward_ss_coef", "reward_t_coef", "us_coef", "ss_coef", "lr_ac",
                 "lr_disc", "lr_q", "disc_grad_penalty", "action_scale", "tracking_lin_vel", "tracking_ang_vel",
                 "jump_up_height", "locomotion_height", "bounds_loss_coef", "disc_history_len", "disc_obs_len",
                 "obs_disc_weight_step"):
        p.add_argument("--" + name, type=float)
    p.add_argument("--disc_loss_function", type=str)
    p.add_argument("--task_obs_weight_decay_steps", type=int)
    p.add_argument("--experiment_idx", type=int, default=-1)
    # gymutil's standard flags that still mean something + this build's additions
    p.add_argument("--sim_device", type=str, default=None)
    p.add_argument("--rl_device", type=str, default=None)
    p.add_argument("--num_threads", type=int, default=0)
    p.add_argument("--physics_engine", default="qa")
    p.add_argument("--terrain", type=str, default=None, help="override terrain.mesh_type: plane | heightfield | trimesh")
    p.add_argument("--no_amp", action="store_true", help="disable the discriminator (BASELINE config 2)")
    p.add_argument("--no_mocap_init", action="store_true", help="reset from the default pose instead of mocap frames")
    p.add_argument("--log_root", type=str, default="default")
    args, _ = p.parse_known_args(argv)
    dev = "cuda:{}".format(args.device_id) if args.device == "gpu" else args.device
    args.rl_device = args.rl_device or dev
    args.sim_device = args.sim_device or dev
    args.compute_device_id = args.sim_device_id = args.device_id
    args.use_gpu_pipeline = args.device == "gpu"
    return args


# ------------------------------------------------------------------ deployment export
class PolicyExporter(torch.nn.Module):
    """The deployed policy as one scriptable module: obs (B, 671) -> actions (B, 12), the arithmetic of
    ActorCritic.act_inference(obs, hist_encoding=True) (bbc/rsl_rl/modules/actor_critic.py:203-214): latent from the
    history encoder over obs[:, 90:660], actor trunk on [prop | explicit | latent | command].

    The reference's `export_policy_as_jit` (bbc/legged_gym/utils/helpers.py:233-242) copies `actor_critic.actor`, an
    attribute the BBC ActorCritic does not have (it has actor_trunk/actor_head), so it cannot export this model; same
    entry point and output file (`policy_1.pt`) here, over the modules that exist."""

    def __init__(self, actor_critic):
        super().__init__()
        import copy
        ac = actor_critic
        he = ac.history_encoder
        self.frame = copy.deepcopy(he.encoder)                       # Linear(57, 30) + act
        self.convs = copy.deepcopy(he.conv_layers)                   # Conv1d stack + act + Flatten, channels-first
        self.out = copy.deepcopy(he.linear_output)
        self.trunk = copy.deepcopy(ac.actor_trunk)
        self.head = copy.deepcopy(ac.actor_head)
        self.num_prop, self.num_explicit, self.num_latent = int(ac.num_prop), int(ac.num_explicit), int(ac.num_latent)
        self.num_hist = int(ac.num_hist)

    def forward(self, obs: torch.Tensor) -> torch.Tensor:
        a = self.num_prop
        b = a + self.num_explicit
        c = b + self.num_latent
        d = c + self.num_hist * self.num_prop
        hist = obs[:, c:d].reshape(-1, self.num_prop)
        x = self.frame(hist).reshape(obs.shape[0], self.num_hist, -1).permute(0, 2, 1)       # (B, C, T)
        latent = self.out(self.convs(x))
        x = torch.cat([obs[:, :a], obs[:, a:b], latent, obs[:, d:]], dim=-1)
        return self.head(self.trunk(x))


def export_policy_as_jit(actor_critic, path):
    os.makedirs(path, exist_ok=True)
    path = os.path.join(path, "policy_1.pt")
    model = PolicyExporter(actor_critic).to("cpu").eval()
    scripted = torch.jit.script(model)
    scripted.save(path)
    return path
