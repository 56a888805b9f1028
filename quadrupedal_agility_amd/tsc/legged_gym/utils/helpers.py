"""Command line and config plumbing of the task-level tree (tsc/legged_gym/utils/helpers.py:70-256): the same flags with the same
meaning (`--use_camera` switches the env to the depth camera, its 256 envs, and the runner to `learn_vision`; `--resume` /
`--resumeid` / `--exptid` / `--proj_name` name the run directories), argparse instead of `gymutil.parse_arguments`."""
import argparse
import os

from quadrupedal_agility_amd.legged_gym.utils.helpers import class_to_dict, set_seed  # noqa: F401  (same functions in both trees)

LEGGED_GYM_ROOT_DIR = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

_ENV_OVERRIDES = {"tracking_goal_vel": ("rewards.scales", "tracking_goal_vel"), "tracking_yaw": ("rewards.scales", "tracking_yaw"),
                  "action_hl_rate": ("rewards.scales", "action_hl_rate"), "reach_goal": ("rewards.scales", "reach_goal"),
                  "termination": ("rewards.scales", "termination"), "target_lin_vel": ("rewards", "target_lin_vel"),
                  "curr_threshold": ("obstacle", "curr_threshold")}
_TRAIN_OVERRIDES = {"max_iterations": ("runner", "max_iterations"), "experiment_name": ("runner", "experiment_name"),
                    "run_name": ("runner", "run_name"), "load_run": ("runner", "load_run"), "checkpoint": ("runner", "checkpoint"),
                    "reward_i_coef": ("runner", "reward_i_coef"), "reward_t_coef": ("runner", "reward_t_coef")}


def _node(cfg, path):
    for part in path.split("."):
        cfg = getattr(cfg, part)
    return cfg


def get_load_path(root, load_run=-1, checkpoint=-1, model_name_include="model"):
    """:70-96 -- `root` may be a 6-character run prefix; the newest `model*.pt` unless a checkpoint number is given"""
    if not os.path.isdir(root):
        cand, parent = os.path.basename(root), os.path.dirname(root)
        for name in sorted(os.listdir(parent)):
            if os.path.isdir(os.path.join(parent, name)) and len(name) >= 6 and name[:6] == cand:
                root = os.path.join(parent, name)
    if checkpoint == -1:
        models = sorted((f for f in os.listdir(root) if model_name_include in f), key=lambda m: "{0:0>15}".format(m))
        model = models[-1]
    else:
        model = "model_{}.pt".format(checkpoint)
    return os.path.join(root, model)


def update_cfg_from_args(env_cfg, cfg_train, args):
    """:99-186"""
    if env_cfg is not None:
        if args.use_camera:
            env_cfg.depth.use_camera = True
        if env_cfg.depth.use_camera and args.headless:          # camera runs use the camera's env count
            env_cfg.env.num_envs = env_cfg.depth.camera_num_envs
        if args.num_envs is not None:
            env_cfg.env.num_envs = args.num_envs
        if args.seed is not None:
            env_cfg.seed = args.seed
        for flag in ("randomize_base_mass", "randomize_base_com", "push_robots"):
            if getattr(args, flag):
                setattr(env_cfg.domain_rand, flag, True)
        for arg, (node, field) in _ENV_OVERRIDES.items():
            v = getattr(args, arg, None)
            if v is not None:
                setattr(_node(env_cfg, node), field, v)
        # store_true flags with default False: the reference's `is not None` test always fires, i.e. the command line DECIDES these two
        env_cfg.obstacle.randomize_start = bool(args.randomize_start)
        env_cfg.obstacle.curriculum = bool(args.curriculum)
    if cfg_train is not None:
        if args.seed is not None:
            cfg_train.seed = args.seed
        if args.use_camera:
            cfg_train.depth_encoder.if_depth = True
        if args.resume:
            cfg_train.estimator.load_estimator_bbc = False
            cfg_train.runner.resume = True
            cfg_train.algorithm.priv_reg_coef_schedual = cfg_train.algorithm.priv_reg_coef_schedual_resume
        for arg, (node, field) in _TRAIN_OVERRIDES.items():
            v = getattr(args, arg, None)
            if v is not None:
                setattr(_node(cfg_train, node), field, v)
    return env_cfg, cfg_train


def get_args(argv=None):
    """:189-245"""
    p = argparse.ArgumentParser(description="RL Policy")
    p.add_argument("--task", type=str, default="go2")
    p.add_argument("--resume", action="store_true", default=False)
    p.add_argument("--resumeid", type=str)
    p.add_argument("--experiment_name", type=str)
    p.add_argument("--run_name", type=str)
    p.add_argument("--proj_name", type=str, default="agility")
    p.add_argument("--load_run", type=str)
    p.add_argument("--checkpoint", type=int, default=-1)
    p.add_argument("--exptid", type=str, default="run")
    p.add_argument("--debug", action="store_true", default=False)
    p.add_argument("--headless", action="store_true", default=False)
    p.add_argument("--rl_device", type=str, default=None)
    p.add_argument("--sim_device", type=str, default=None)
    p.add_argument("--num_envs", type=int)
    p.add_argument("--seed", type=int)
    p.add_argument("--max_iterations", type=int)
    p.add_argument("--device", type=str, default="gpu")
    p.add_argument("--device_id", type=int, default=0)
    p.add_argument("--rows", type=int)
    p.add_argument("--cols", type=int)
    p.add_argument("--use_camera", action="store_true", default=False, help="render camera for distillation")
    for name in ("tracking_goal_vel", "tracking_yaw", "action_hl_rate", "termination", "reach_goal", "reward_i_coef", "reward_t_coef",
                 "curr_threshold", "target_lin_vel"):
        p.add_argument("--" + name, type=float)
    for name in ("randomize_start", "curriculum", "randomize_base_mass", "randomize_base_com", "push_robots"):
        p.add_argument("--" + name, action="store_true", default=False)
    p.add_argument("--physics_engine", default="qa")
    p.add_argument("--bbc_path", type=str, default=None, help="behaviour-level model.pt (default: <root>/<runner.bbc_path>)")
    p.add_argument("--log_root", type=str, default="default")
    args, _ = p.parse_known_args(argv)
    dev = "cuda:{}".format(args.device_id) if args.device == "gpu" else args.device
    args.rl_device = args.rl_device or dev
    args.sim_device = args.sim_device or dev
    args.compute_device_id = args.sim_device_id = args.device_id
    args.use_gpu_pipeline = args.device == "gpu"
    return args
