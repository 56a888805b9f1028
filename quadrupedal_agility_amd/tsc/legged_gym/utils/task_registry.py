"""`task_registry` of the task-level tree (tsc/legged_gym/utils/task_registry.py:14-166): `make_env` / `make_alg_runner` with the
reference's resume logic (`--resume`, `--resumeid`, run-prefix matching), the frozen behaviour controller loaded from
`runner.bbc_path`, `reset_std` unless `policy.continue_from_last_std`.  Under `torch.distributed` the job's envs are split over the
ranks, as in the behaviour-level tree."""
import os

import torch

from .helpers import LEGGED_GYM_ROOT_DIR, class_to_dict, get_args, get_load_path, set_seed, update_cfg_from_args


class TaskRegistry:
    def __init__(self):
        self.task_classes, self.env_cfgs, self.train_cfgs = {}, {}, {}

    def register(self, name, task_class, env_cfg, train_cfg):
        self.task_classes[name], self.env_cfgs[name], self.train_cfgs[name] = task_class, env_cfg, train_cfg

    def get_task_class(self, name):
        return self.task_classes[name]

    def get_cfgs(self, name):
        train_cfg, env_cfg = self.train_cfgs[name], self.env_cfgs[name]
        env_cfg.seed = train_cfg.seed
        return env_cfg, train_cfg

    @staticmethod
    def _shard_over_ranks(env_cfg):
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()) or getattr(env_cfg.env, "num_envs_global", 0):
            return
        world, rank = dist.get_world_size(), dist.get_rank()
        total = int(env_cfg.env.num_envs)
        if total % world:
            raise ValueError(f"num_envs {total} is not divisible by {world} ranks")
        env_cfg.env.num_envs = total // world
        env_cfg.env.env_id_offset, env_cfg.env.num_envs_global = rank * (total // world), total
        env_cfg.course_seed = int(getattr(env_cfg, "seed", 1))          # ONE course for the job; each rank builds its envs of it (Obstacle skip_envs = env_id_offset)

    def make_env(self, name, args=None, env_cfg=None, **kwargs):
        if args is None:
            args = get_args()
        if name not in self.task_classes:
            raise ValueError(f"Task with name: {name} was not registered")
        if env_cfg is None:
            env_cfg, _ = self.get_cfgs(name)
        env_cfg, _ = update_cfg_from_args(env_cfg, None, args)
        set_seed(env_cfg.seed)
        self._shard_over_ranks(env_cfg)
        env = self.task_classes[name](cfg=env_cfg, sim_params={"sim": class_to_dict(env_cfg.sim)}, physics_engine=args.physics_engine,
                                      sim_device=args.sim_device, headless=args.headless, **kwargs)
        return env, env_cfg

    def make_alg_runner(self, env, name=None, args=None, train_cfg=None, log_root="default", **kwargs):
        from quadrupedal_agility_amd.tsc.rsl_rl.runners import OnPolicyRunner
        if args is None:
            args = get_args()
        if train_cfg is None:
            if name is None:
                raise ValueError("Either 'name' or 'train_cfg' must be not None")
            _, train_cfg = self.get_cfgs(name)
        elif name is not None:
            print(f"'train_cfg' provided -> Ignoring 'name={name}'")
        _, train_cfg = update_cfg_from_args(None, train_cfg, args)
        if log_root == "default":
            log_root = os.path.join(LEGGED_GYM_ROOT_DIR, "logs", args.proj_name, args.exptid)
        if log_root is not None:
            os.makedirs(log_root, exist_ok=True)
        want_dir = kwargs.pop("return_log_dir", False)
        runner = OnPolicyRunner(env, class_to_dict(train_cfg), log_root, device=args.rl_device, **kwargs)
        resume, resume_path = train_cfg.runner.resume, None
        if args.resumeid:
            log_root = os.path.join(LEGGED_GYM_ROOT_DIR, "logs", args.proj_name, args.resumeid)
            resume = True
        if resume:
            resume_path = get_load_path(log_root, load_run=train_cfg.runner.load_run, checkpoint=train_cfg.runner.checkpoint)
            runner.load(resume_path)
            if not train_cfg.policy.continue_from_last_std:
                runner.alg.actor_critic.reset_std(train_cfg.policy.init_noise_std, runner.env.num_actions_d * runner.env.num_actions_c,
                                                  device=runner.device)
        # the frozen behaviour controller, its estimator and the style discriminator
        bbc_path = getattr(args, "bbc_path", None) or os.path.join(LEGGED_GYM_ROOT_DIR, train_cfg.runner.bbc_path)
        if os.path.exists(bbc_path):
            runner.load_bbc(bbc_path)
        else:
            print(f"[task_registry] no behaviour-level checkpoint at {bbc_path}: the behaviour policy and the discriminator keep their initial weights")
        if want_dir:
            return runner, train_cfg, (os.path.dirname(resume_path) if resume_path else None)
        return runner, train_cfg


task_registry = TaskRegistry()
