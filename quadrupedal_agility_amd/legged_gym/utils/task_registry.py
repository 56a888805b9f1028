"""Task registry: name -> (env class, env cfg, train cfg); builds envs and runners
(API of bbc/legged_gym/utils/task_registry.py:15-148)."""
import os
from typing import Tuple

from quadrupedal_agility_amd.legged_gym import LEGGED_GYM_ROOT_DIR

from .helpers import class_to_dict, get_args, get_load_path, parse_sim_params, set_seed, update_cfg_from_args


class TaskRegistry:
    def __init__(self):
        self.task_classes, self.env_cfgs, self.train_cfgs = {}, {}, {}

    def register(self, name: str, task_class, env_cfg, train_cfg):
        self.task_classes[name], self.env_cfgs[name], self.train_cfgs[name] = task_class, env_cfg, train_cfg

    def get_task_class(self, name: str):
        return self.task_classes[name]

    def get_cfgs(self, name) -> Tuple[object, object]:
        env_cfg, train_cfg = self.env_cfgs[name], self.train_cfgs[name]
        env_cfg.seed = train_cfg.seed
        return env_cfg, train_cfg

    def make_env(self, name, args=None, env_cfg=None, backend=None):
        if args is None:
            args = get_args()
        if name not in self.task_classes:
            raise ValueError(f"Task with name: {name} was not registered")
        task_class = self.get_task_class(name)
        if env_cfg is None:
            env_cfg, _ = self.get_cfgs(name)
        env_cfg, _ = update_cfg_from_args(env_cfg, None, args)
        if not hasattr(env_cfg, "seed"):            # a hand-built config: take the task's training seed, as get_cfgs does
            env_cfg.seed = self.train_cfgs[name].seed
        self._shard_over_ranks(env_cfg)
        set_seed(env_cfg.seed)
        sim_params = parse_sim_params(args, {"sim": class_to_dict(env_cfg.sim)})
        env = task_class(cfg=env_cfg, sim_params=sim_params, physics_engine=args.physics_engine, sim_device=args.sim_device,
                         headless=args.headless, **({"backend": backend} if backend is not None else {}))
        return env, env_cfg

    @staticmethod
    def _shard_over_ranks(env_cfg):
        """train.py under `torch.distributed.run` (one process per GPU): the job's `num_envs` envs are split over the ranks,
        rank r owns [r N/W, (r+1) N/W).  Random draws stay keyed by the GLOBAL env id (qa_config.env_id_offset), so the
        env side of the job does not depend on the world size.  A config that already carries an offset (bench.py, tests) is left alone."""
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()) or hasattr(env_cfg.env, "env_id_offset"):
            return
        world, rank = dist.get_world_size(), dist.get_rank()
        if world > 1:
            total = int(env_cfg.env.num_envs)
            if total % world:
                raise ValueError(f"num_envs {total} does not split over {world} ranks")
            env_cfg.env.num_envs = total // world
            env_cfg.env.env_id_offset, env_cfg.env.num_envs_global = rank * (total // world), total

    def make_alg_runner(self, env, name=None, args=None, train_cfg=None, log_root="default"):
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
            log_root = os.path.join(LEGGED_GYM_ROOT_DIR, "logs", train_cfg.runner.experiment_name)
        log_dir = None if log_root is None else os.path.join(log_root, "{}".format(int(train_cfg.runner.experiment_idx)))
        from quadrupedal_agility_amd.rsl_rl.runners import OnPolicyRunner  # noqa: F401  (resolved by name below)
        runner_class = eval(train_cfg.runner_class_name)
        runner = runner_class(env, class_to_dict(train_cfg), log_dir, device=args.rl_device)
        if train_cfg.runner.resume:
            resume_path = get_load_path(log_root, load_run=train_cfg.runner.load_run, checkpoint=train_cfg.runner.checkpoint)
            print(f"Loading model from: {resume_path}")
            runner.load(resume_path)
        return runner, train_cfg


task_registry = TaskRegistry()
