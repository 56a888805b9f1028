"""python -m quadrupedal_agility_amd.legged_gym.scripts.train --task go2_locomotion [--num_envs N] [--max_iterations K]
Drop-in for bbc/legged_gym/scripts/train.py:13-21."""
from quadrupedal_agility_amd.legged_gym.envs import *  # noqa: F401,F403  (registers the tasks)
from quadrupedal_agility_amd.legged_gym.utils import get_args, task_registry


def train(args):
    env, env_cfg = task_registry.make_env(name=args.task, args=args)
    gail_runner, train_cfg = task_registry.make_alg_runner(env=env, name=args.task, args=args, log_root=args.log_root)
    gail_runner.learn(num_learning_iterations=train_cfg.runner.max_iterations, init_at_random_ep_len=True)


if __name__ == "__main__":
    train(get_args())
