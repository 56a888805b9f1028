"""python -m quadrupedal_agility_amd.legged_gym.scripts.play --task go2_locomotion [--load_run R] [--checkpoint K]
Drop-in for bbc/legged_gym/scripts/play.py:14-125: same config overrides, checkpoint resume, inference policy
(history-encoder latent), fixed gait/velocity command, state/reward logging.  There is no viewer in this build, so
the camera/frame-recording branches are gone and the loop length is a flag (`--num_steps`; the reference runs
1000 episodes' worth); `--export_policy` writes the TorchScript deployment module (helpers.export_policy_as_jit)."""
import argparse
import os
import time

import numpy as np
import torch

from quadrupedal_agility_amd.legged_gym.envs import *  # noqa: F401,F403
from quadrupedal_agility_amd.legged_gym.utils import Logger, export_policy_as_jit, get_args, task_registry


def play_overrides(env_cfg, train_cfg):
    """bbc/legged_gym/scripts/play.py:17-38"""
    env_cfg.env.num_envs = min(env_cfg.env.num_envs, 16)
    env_cfg.terrain.num_rows = 5
    env_cfg.terrain.num_cols = 5
    env_cfg.terrain.curriculum = False
    env_cfg.noise.add_noise = True
    env_cfg.domain_rand.randomize_friction = True
    env_cfg.domain_rand.randomize_base_mass = False
    env_cfg.domain_rand.randomize_base_com = False
    env_cfg.domain_rand.push_robots = False
    env_cfg.domain_rand.randomize_motor = True
    env_cfg.domain_rand.action_delay = True
    env_cfg.domain_rand.action_curr_step = [1]
    env_cfg.commands.curriculum = False
    env_cfg.commands.resampling_time = 1e10
    env_cfg.env.episode_length_s = 500.
    env_cfg.env.mocap_state_init = False
    env_cfg.env.recovery_init_prob = 0.
    env_cfg.env.root_height_obs = True
    train_cfg.runner.num_preload_transitions = 1
    train_cfg.policy.train_with_estimated_latent = True
    train_cfg.estimator.train_with_estimated_explicit = True
    return env_cfg, train_cfg


def play(args, num_steps=None, export_policy=False, realtime=False, gait=2, vel_x=2.0, log_root="default",
         stop_state_log=-1, robot_index=0):
    env_cfg, train_cfg = task_registry.get_cfgs(name=args.task)
    env_cfg, train_cfg = play_overrides(env_cfg, train_cfg)
    env, _ = task_registry.make_env(name=args.task, args=args, env_cfg=env_cfg)
    env.reset()
    obs = env.get_observations()
    train_cfg.runner.resume = True
    env_cfg.env.play_mode = True
    runner, train_cfg = task_registry.make_alg_runner(env=env, name=args.task, args=args, train_cfg=train_cfg, log_root=log_root)
    policy = runner.get_inference_policy(device=env.device)

    exported = None
    if export_policy:
        root = os.path.dirname(runner.log_dir) if runner.log_dir else "."
        exported = export_policy_as_jit(runner.alg.actor_critic, os.path.join(root, "exported", "policies"))
        print("Exported policy as jit script to: ", exported)

    logger = Logger(env.dt)
    stop_rew_log = env.max_episode_length + 1
    # fixed command: gait one-hot + forward velocity (play.py:64-68)
    env.latent_c[:, :] = 0
    env.commands[:, :] = 0
    env.latent_c[:, gait] = 1
    env.commands[:, 0] = vel_x

    n = int(num_steps) if num_steps is not None else 1000 * int(env.max_episode_length)
    for i in range(n):
        t0 = time.time()
        if not env_cfg.env.root_height_obs:
            obs[:, env_cfg.env.num_prop] = 0.0
        with torch.no_grad():
            actions = policy(obs.detach())
        obs, _, rews, dones, infos, _, _ = env.step(actions.detach())
        if i < stop_state_log:
            logger.log_states({
                "dof_pos_target": env.default_dof_pos[0, :].cpu().numpy() + env.cfg.control.action_scale * env.actions[robot_index, :].cpu().numpy(),
                "dof_pos": env.dof_pos[robot_index, :].cpu().numpy(),
                "dof_vel": env.dof_vel[robot_index, :].cpu().numpy(),
                "dof_torque": env.torques[robot_index, :].cpu().numpy(),
                "command_x": env.commands[robot_index, 0].item(),
                "command_y": env.commands[robot_index, 1].item(),
                "command_yaw": env.commands[robot_index, 2].item(),
                "base_vel_x": env.base_lin_vel[robot_index, 0].item(),
                "base_vel_y": env.base_lin_vel[robot_index, 1].item(),
                "base_vel_z": env.base_lin_vel[robot_index, 2].item(),
                "base_vel_yaw": env.base_ang_vel[robot_index, 2].item(),
                "contact_forces_z": env.contact_forces[robot_index, env.feet_indices, 2].cpu().numpy(),
            })
        elif i == stop_state_log:
            logger.plot_states()
            logger.plot_dof_pos()
        if 0 < i < stop_rew_log and infos.get("episode"):
            num_episodes = int(torch.sum(env.reset_buf).item())
            if num_episodes > 0:
                logger.log_rewards(infos["episode"], num_episodes)
        if realtime:
            left = env.dt - (time.time() - t0)
            if left > 0:
                time.sleep(left)
    return env, runner, logger, exported


def _main():
    ap = argparse.ArgumentParser(add_help=False)
    ap.add_argument("--num_steps", type=int, default=None)
    ap.add_argument("--export_policy", action="store_true")
    ap.add_argument("--realtime", action="store_true", help="pace the loop at env.dt like the reference's viewer loop")
    ap.add_argument("--gait", type=int, default=2, help="latent_c one-hot index (walk, pace, trot, canter, jump)")
    ap.add_argument("--vel_x", type=float, default=2.0)
    ap.add_argument("--log_states", type=int, default=-1, help="log robot 0 for this many steps, then dump/plot")
    extra, _ = ap.parse_known_args()
    args = get_args()
    _, _, logger, _ = play(args, num_steps=extra.num_steps, export_policy=extra.export_policy, realtime=extra.realtime,
                           gait=extra.gait, vel_x=extra.vel_x, log_root=args.log_root, stop_state_log=extra.log_states)
    logger.print_rewards()


if __name__ == "__main__":
    _main()
