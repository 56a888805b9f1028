"""tsc/legged_gym/scripts/play.py without the viewer: the reference's test-time overrides (:16-41), a resumed runner, and the
deployment loop -- estimator -> (depth encoder -> student actor | teacher actor) -> set_commands -> frozen behaviour policy ->
env.step -- for `--num_steps` steps; prints and returns the course success rate (`reach_goal` of the finished episodes).
`python -m quadrupedal_agility_amd.tsc.legged_gym.scripts.play --task go2 --exptid <run> [--use_camera] [--num_steps N]`"""
import statistics
from collections import deque

import torch
import torch.nn.functional as F


def play(args, num_steps=1000, env_kwargs=None, quiet=False):
    from quadrupedal_agility_amd.tsc.legged_gym.envs import task_registry
    import copy
    env_cfg, train_cfg = (copy.deepcopy(c) for c in task_registry.get_cfgs(name=args.task))      # the registry's objects stay as registered
    env_cfg.env.num_envs = 1024 if args.num_envs is None else args.num_envs
    env_cfg.env.episode_length_s = 40
    env_cfg.depth.angle = [0, 1]
    env_cfg.depth.depth_noise = 0.0
    env_cfg.noise.add_noise = False
    d = env_cfg.domain_rand
    d.randomize_friction, d.push_robots, d.randomize_base_mass, d.randomize_base_com, d.randomize_action = True, False, False, False, False
    env_cfg.obstacle.curriculum, env_cfg.obstacle.randomize_border = False, False
    env_cfg.env.next_goal_threshold = 0.45
    args.randomize_start, args.headless = True, True
    if args.use_camera and args.num_envs is None:
        args.num_envs = 256
    env, _ = task_registry.make_env(name=args.task, args=args, env_cfg=env_cfg, **(env_kwargs or {}))
    if hasattr(env, "bk"):
        env.bk._cfg.next_goal_threshold = 0.45
    train_cfg.runner.resume = True
    train_cfg.estimator.load_estimator_bbc = False
    runner, train_cfg = task_registry.make_alg_runner(env=env, name=args.task, args=args, train_cfg=train_cfg)
    policy = runner.get_inference_policy(device=env.device)
    policy_bbc = runner.get_inference_policy_bbc(device=env.device)
    estimator = runner.get_estimator_inference_policy(device=env.device)
    use_depth = bool(env.cfg.depth.use_camera)
    if use_depth:
        runner.depth_backbone.augment = None
        depth_encoder = runner.get_depth_encoder_inference_policy(device=env.device)
        depth_actor = runner.get_depth_actor_inference_policy(device=env.device)
    e = env.cfg.env
    n_pro, n_aux, n_yaw = e.n_proprio, e.n_auxiliary, e.n_delta_yaw
    n_lat = train_cfg.policy.scan_encoder_dims[-1]
    priv = runner.alg._priv_slice(True)
    obs, obs_bbc = env.get_observations().clone(), env.get_observations_bbc().clone()
    infos = {"depth": env.depth_buffer[:, -1].clone() if use_depth else None}
    reach_goal_buffer = deque(maxlen=1000)
    depth_latent = delta_yaw = obst_type = None
    with torch.no_grad():
        for i in range(num_steps):
            obs[:, priv] = estimator(obs[:, :runner.alg.num_prop])
            if use_depth:
                if infos["depth"] is not None:
                    obs_student = obs[:, :n_pro].clone()
                    obs_student[:, n_pro - n_aux:n_pro] = 0
                    out = depth_encoder(infos["depth"], obs_student)
                    depth_latent, delta_yaw, obst_type = out[:, :n_lat], out[:, n_lat:n_lat + n_yaw], out[:, n_lat + n_yaw:]
                obs[:, n_pro - n_aux:n_pro - n_aux + n_yaw] = 1.5 * delta_yaw
                obs[:, n_pro - n_aux + n_yaw:n_pro] = F.one_hot(torch.argmax(obst_type, dim=-1), num_classes=obst_type.shape[-1]).to(obs.dtype)
                emb = depth_actor(obs, hist_encoding=True, scandots_latent=depth_latent)
                actions = torch.cat([torch.argmax(depth_actor.actor_d(emb), dim=-1, keepdim=True).to(obs.dtype), depth_actor.actor_c(emb)], dim=-1)
            else:
                actions = policy(obs, hist_encoding=True, scandots_latent=None)
            obs_bbc[:, -(6 + env.dim_c):] = env.set_commands(actions)
            o, _, rews, dones, infos, _, _ = env.step(policy_bbc(obs_bbc, hist_encoding=True))
            obs, obs_bbc = o.clone(), env.get_observations_bbc().clone()
            done = dones != 0
            if bool(done.any()):
                reach_goal_buffer.extend(infos["reach_goal"][done].float().cpu().tolist())
            if not quiet and i % 50 == 0:
                sr = statistics.mean(reach_goal_buffer) if reach_goal_buffer else float("nan")
                print(f"step {i}: cmd_vx {env.commands[0, 0].item():.2f} actual_vx {env.base_lin_vel[0, 0].item():.2f} "
                      f"obstacle {int(env.cur_obstacle_types[0])} gait {int(torch.argmax(env.latent_c[0]))} success_rate {sr:.4f}")
    return statistics.mean(reach_goal_buffer) if reach_goal_buffer else None


if __name__ == "__main__":
    import argparse
    import sys
    from quadrupedal_agility_amd.tsc.legged_gym.utils.helpers import get_args
    ap = argparse.ArgumentParser(add_help=False)
    ap.add_argument("--num_steps", type=int, default=1000)
    extra, _ = ap.parse_known_args(sys.argv[1:])
    play(get_args(), num_steps=extra.num_steps)
