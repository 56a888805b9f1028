#!/usr/bin/env python3
"""tests/golden/tsc_env.npz: the reference's task-level env math run here on CPU -- LeggedRobot.set_commands
(tsc/legged_gym/envs/base/legged_robot.py:699-760) the goal / termination / reward part of post_physics_step (:226-273), and _get_heights (:1708-1755) +
compute_observations (:432-515) -- on a LeggedRobot built without Isaac Gym (object.__new__ + synthetic tensors), with the callbacks that need the simulator
(_post_physics_step_callback, everything of reset_idx but its goal/episode bookkeeping, get_observations_disc, update_depth_buffer, compute_observations, the gym refreshes) stubbed out.
Build container only (needs /root/reference).  The uniform action noise set_commands draws with torch's generator is
replaced by values stored in the fixture (torch_rand_float is patched), everything else is the reference's own arithmetic."""
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.dont_write_bytecode = True
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools", "ref_shims"))
tb = types.ModuleType("torch.utils.tensorboard"); tb.SummaryWriter = object; sys.modules["torch.utils.tensorboard"] = tb
for name in ("torchvision", "torchvision.transforms", "cv2", "skimage", "skimage.draw"):      # imported, never used here
    if name not in sys.modules:
        try:
            __import__(name)
        except ImportError:
            sys.modules[name] = types.ModuleType(name)
sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
sys.modules["skimage"].draw = sys.modules["skimage.draw"]
if not hasattr(sys.modules["skimage.draw"], "polygon"):
    sys.modules["skimage.draw"].polygon = lambda *a, **k: None
sys.path.insert(0, "/root/reference/tsc")
cwd = os.getcwd()
os.chdir("/root/reference/tsc/legged_gym/scripts")
import legged_gym.envs.base.legged_robot as ref_lr                                            # noqa: E402
from legged_gym.envs.go2.go2_agility_config import Go2AgilityCfg                              # noqa: E402
from legged_gym.utils.helpers import class_to_dict                                            # noqa: E402
os.chdir(cwd)

BODY_NAMES = ["base", "Head_upper", "Head_lower"] + [f"{l}_{p}" for l in ("FL", "FR", "RL", "RR") for p in ("hip", "thigh", "calf", "foot")]
N, STEPS = 64, 4
rng = np.random.default_rng(20250404)
T = lambda a, dt=torch.float32: torch.tensor(np.asarray(a), dtype=dt)          # noqa: E731


def make_env():
    cfg = Go2AgilityCfg()
    env = object.__new__(ref_lr.LeggedRobot)
    env.cfg = cfg
    env.device = "cpu"
    env.num_envs = N
    env.dt = cfg.control.decimation * cfg.sim.dt
    env.sim = None
    env.gym = types.SimpleNamespace(**{k: (lambda *a, **kw: None) for k in (
        "refresh_actor_root_state_tensor", "refresh_net_contact_force_tensor", "refresh_rigid_body_state_tensor",
        "refresh_force_sensor_tensor")})
    env.mocap_category_all = cfg.env.mocap_category_all
    env.dim_c = len(cfg.env.mocap_category_all)
    env.num_actions_c = cfg.env.num_actions_c
    env.category_mapping = {c: i for i, c in enumerate(cfg.env.mocap_category_all)}
    env.mocap_indices = torch.tensor([env.category_mapping[c] for c in cfg.env.mocap_category])
    env.command_ranges = class_to_dict(cfg.commands.ranges)
    env.reward_scales = class_to_dict(cfg.rewards.scales)
    env.max_episode_length_s = cfg.env.episode_length_s
    env.max_episode_length = np.ceil(env.max_episode_length_s / env.dt)
    env.obstacle = types.SimpleNamespace(last_goal_repeat=cfg.obstacle.last_goal_repeat, num_goals=cfg.obstacle.num_goals,
                                         cfg=cfg.obstacle)
    env.feet_indices = T([BODY_NAMES.index(n) for n in BODY_NAMES if cfg.asset.foot_name in n], torch.long)
    env.penalised_contact_indices = T([BODY_NAMES.index(n) for n in BODY_NAMES if any(p in n for p in cfg.asset.penalize_contacts_on)], torch.long)
    env.termination_contact_indices = T([BODY_NAMES.index(n) for n in BODY_NAMES if any(p in n for p in cfg.asset.terminate_after_contacts_on)], torch.long)
    env.gravity_vec = T([[0.0, 0.0, -1.0]]).repeat(N, 1)
    env.extras = {}
    env.common_step_counter = 0
    # callbacks that need the simulator
    env._post_physics_step_callback = lambda: None

    def reset_idx(ids):                      # the goal/episode bookkeeping of the reference's reset_idx (:376, :396-404), no simulator
        env.cur_goal_idx[ids] = 0
        env.reach_goal_timer[ids] = 0
        for key in env.episode_sums:
            env.episode_sums[key][ids] = 0.
        env.episode_length_buf[ids] = 0
    env.reset_idx = reset_idx
    env.get_observations_disc = lambda: torch.zeros(N, 1)
    env.update_depth_buffer = lambda: None
    env.compute_observations = lambda: None
    for name in ("last_actions", "actions", "last_dof_vel", "dof_vel", "last_torques_org", "torques_org"):
        setattr(env, name, torch.zeros(N, 12))
    env.viewer = None
    return env, cfg


def random_quats(n):
    rpy = rng.uniform(-0.6, 0.6, (n, 3)); rpy[:, 2] = rng.uniform(-np.pi, np.pi, n)
    rpy[: n // 12, 0] = rng.uniform(1.4, 1.7, n // 12)                    # some past the roll cut-off
    rpy[n // 12: n // 6, 1] = rng.uniform(1.35, 1.55, n // 6 - n // 12)   # and the pitch cut-off
    cr, sr, cp, sp, cy, sy = np.cos(rpy[:, 0] / 2), np.sin(rpy[:, 0] / 2), np.cos(rpy[:, 1] / 2), np.sin(rpy[:, 1] / 2), np.cos(rpy[:, 2] / 2), np.sin(rpy[:, 2] / 2)
    q = np.stack([sr * cp * cy - cr * sp * sy, cr * sp * cy + sr * cp * sy, cr * cp * sy - sr * sp * cy, cr * cp * cy + sr * sp * sy], 1)
    return q / np.linalg.norm(q, axis=1, keepdims=True)


def main():
    out = {}
    env, cfg = make_env()
    # ---------------------------------------------------------------- set_commands
    num_d = len(cfg.env.mocap_category)
    for case, resampling_time in (("every", cfg.commands.resampling_time), ("sparse", 0.1)):
        cfg.commands.resampling_time = resampling_time
        actions = rng.uniform(-1.5, 1.5, (N, 1 + num_d * env.num_actions_c)).astype(np.float32)
        actions[:, 0] = rng.integers(0, num_d, N)
        ep_len = rng.integers(0, 40, N)
        noise = rng.uniform(*cfg.domain_rand.action_noise, (N, 5)).astype(np.float32)
        commands0 = rng.uniform(-1, 1, (N, 5)).astype(np.float32)
        eps0 = rng.uniform(-1, 1, (N, 1)).astype(np.float32)
        c0 = np.eye(env.dim_c, dtype=np.float32)[rng.integers(0, env.dim_c, N)]
        env.episode_length_buf = T(ep_len, torch.long)
        env.commands, env.latent_eps, env.latent_c = T(commands0), T(eps0), T(c0)
        ref_lr.torch_rand_float = lambda lo, hi, shape, device=None: T(noise)
        nxt = env.set_commands(T(actions))
        out.update({f"cmd_{case}_{k}": v for k, v in dict(
            actions=actions, episode_length=ep_len, noise=noise, commands0=commands0, latent_eps0=eps0, latent_c0=c0,
            interval=np.int32(int(resampling_time / env.dt)), commands=env.commands.numpy(), latent_eps=env.latent_eps.numpy(),
            latent_c=env.latent_c.numpy(), next_commands=nxt.numpy()).items()})
    out["cmd_mocap_index"] = env.mocap_indices.numpy().astype(np.int32)
    out["cmd_vel_ranges"] = np.array([env.command_ranges[k] for k in ("lin_vel_x", "lin_vel_y", "ang_vel_yaw")], np.float32)
    out["cmd_jump_range"] = np.array(env.command_ranges["jump_height"], np.float32)
    out["cmd_height_range"] = np.array(env.command_ranges["locomotion_height"], np.float32)

    # ---------------------------------------------------------------- post_physics_step (goals, termination, rewards)
    env.dt = cfg.control.decimation * cfg.sim.dt
    env._prepare_reward_function()
    K, G = 6, cfg.obstacle.num_goals
    slots = K * G + cfg.obstacle.last_goal_repeat
    env_goals = np.zeros((N, slots, 3), np.float32)
    env_goals[:, :, 0] = np.arange(slots)[None, :] * 1.5 + rng.uniform(-0.2, 0.2, (N, slots))
    env_goals[:, :, 1] = rng.uniform(-1.0, 1.0, (N, slots))
    env_goals[:, K * G:, :] = env_goals[:, K * G - 1: K * G, :]           # the last goal, repeated
    env.env_goals = T(env_goals)
    env.obstacle_types = T(rng.integers(0, 6, (N, K)), torch.long)
    idx0 = rng.integers(0, slots - 1, N); idx0[:8] = slots - cfg.obstacle.last_goal_repeat - 1
    env.cur_goal_idx = T(idx0, torch.long)
    timer0 = rng.integers(0, 3, N).astype(np.float32); timer0[:8] = 2.0
    env.reach_goal_timer = T(timer0)
    env.cur_goals = env._gather_cur_goals(); env.next_goals = env._gather_cur_goals(future=1)
    ep0 = rng.integers(0, 1990, N); ep0[-6:] = int(env.max_episode_length) - 1
    env.episode_length_buf = T(ep0, torch.long)
    last_contacts0 = rng.random((N, 4)) < 0.3
    env.last_contacts = T(last_contacts0, torch.bool)
    rows, cols = 260, 140
    mask = rng.random((rows, cols)) < 0.25
    env.x_edge_mask = T(mask, torch.bool)
    env.commands = torch.zeros(N, 5)
    for name in ("base_quat", "base_lin_vel", "base_ang_vel", "projected_gravity"):
        setattr(env, name, torch.zeros(N, 4 if name == "base_quat" else 3))
    env.last_root_vel = torch.zeros(N, 6)
    env.rew_buf = torch.zeros(N)
    out.update(goal_env_goals=env_goals, goal_obstacle_types=env.obstacle_types.numpy(), goal_cur_goal_idx0=idx0,
               goal_timer0=timer0, goal_episode_length0=ep0, goal_last_contacts0=last_contacts0.astype(np.uint8),
               goal_x_edge_mask=mask.astype(np.uint8), goal_cur_goals0=env.cur_goals.numpy(), goal_next_goals0=env.next_goals.numpy(),
               goal_feet=env.feet_indices.numpy().astype(np.int32), goal_penalised=env.penalised_contact_indices.numpy().astype(np.int32),
               goal_termination=env.termination_contact_indices.numpy().astype(np.int32),
               goal_reward_names=np.array(env.reward_names + ["termination"]),
               goal_reward_scales=np.array([env.reward_scales[n] for n in env.reward_names + ["termination"]], np.float32),
               goal_scalars=np.array([cfg.env.reach_goal_delay / env.dt, cfg.env.next_goal_threshold, cfg.env.leave_goal_threshold,
                                      env.max_episode_length, cfg.rewards.target_lin_vel, cfg.obstacle.border_size,
                                      cfg.obstacle.horizontal_scale], np.float64),
               goal_ints=np.array([slots, cfg.obstacle.last_goal_repeat, G, K, rows, cols], np.int32))
    for use_camera in (0, 1):
        cfg.depth.use_camera = bool(use_camera)
        # the camera case continues from the state the first case left
        for t in range(STEPS):
            root = np.zeros((N, 13), np.float32)
            cur = env.cur_goals.numpy()
            spread = rng.choice([0.2, 1.5, 6.0], N, p=[0.4, 0.45, 0.15])[:, None]
            root[:, :2] = cur[:, :2] + rng.normal(0, 1, (N, 2)) * spread
            root[:, 2] = rng.uniform(0.2, 0.5, N); root[N // 2: N // 2 + 5, 2] = -0.3
            root[:, 3:7] = random_quats(N)
            root[:, 7:13] = rng.normal(0, 1.0, (N, 6))
            cf = rng.normal(0, 3.0, (N, 19, 3)).astype(np.float32) * (rng.random((N, 19, 1)) < 0.03)
            cf[:, [6, 10, 14, 18]] = rng.normal(0, 8.0, (N, 4, 3)) * (rng.random((N, 4, 1)) < 0.6)
            rb = np.zeros((N, 19, 13), np.float32)
            rb[:, :, :2] = root[:, None, :2] + rng.uniform(-0.4, 0.4, (N, 19, 2))
            hist = rng.normal(0, 0.7, (N, 5, 19)).astype(np.float32)
            hist[:, :, 0] = rng.integers(0, 3, (N, 5))
            env.root_states, env.contact_forces, env.rigid_body_states = T(root), T(cf), T(rb)
            env.action_hl_history_buf = T(hist) if t != 1 else None         # one step without the task-policy history
            env.post_physics_step()
            tag = f"goal_c{use_camera}_t{t}_"
            out.update({tag + k: v for k, v in dict(
                root_states=root, contact_forces=cf.astype(np.float32), rigid_body_states=rb,
                action_hl_history=hist if t != 1 else np.zeros(0, np.float32),
                episode_length=env.episode_length_buf.numpy().copy(), cur_goal_idx=env.cur_goal_idx.numpy().copy(),
                timer=env.reach_goal_timer.numpy().copy(), last_contacts=env.last_contacts.numpy().astype(np.uint8),
                contact_filt=env.contact_filt.numpy().astype(np.uint8), base_lin_vel=env.base_lin_vel.numpy().copy(),
                base_ang_vel=env.base_ang_vel.numpy().copy(), projected_gravity=env.projected_gravity.numpy().copy(),
                rpy=torch.stack([env.roll, env.pitch, env.yaw], 1).numpy(), target_pos_rel=env.target_pos_rel.numpy().copy(),
                next_target_pos_rel=env.next_target_pos_rel.numpy().copy(), target_yaw=env.target_yaw.numpy().copy(),
                next_target_yaw=env.next_target_yaw.numpy().copy(), reached_goal=env.reached_goal_ids.numpy().astype(np.uint8),
                cur_obstacle_type=env.cur_obstacle_types.numpy().copy(), reset_buf=env.reset_buf.numpy().astype(np.uint8),
                time_out_buf=env.time_out_buf.numpy().astype(np.uint8), reach_goal_cutoff=env.extras["reach_goal"].numpy().astype(np.uint8),
                rew_buf=env.rew_buf.numpy().copy(),
                episode_sums=np.stack([env.episode_sums[n].numpy() for n in env.reward_names + ["termination"]]),
                cur_goals=env.cur_goals.numpy().copy(), next_goals=env.next_goals.numpy().copy()).items()})

    # ---------------------------------------------------------------- _get_heights + compute_observations
    del env.compute_observations                      # back to the reference's own method
    env.obs_scales = cfg.normalization.obs_scales
    env.obstacle.proportions = list(range(6))
    env.obstacle.cfg = cfg.obstacle
    rows_h, cols_h = 320, 240
    hs = (rng.integers(0, 3, (rows_h, cols_h)) * rng.integers(0, 120, (rows_h, cols_h))).astype(np.int16)     # sparse boxes up to 0.6 m
    hs[100:140, :] = 60
    env.height_samples = T(hs, torch.int16)
    ref_lr.torch_rand_float = lambda lo, hi, shape, device=None: torch.zeros(shape)
    env.height_points = env._init_height_points()
    env.key_body_ids = env.feet_indices
    default = np.array([0.1, 0.8, -1.5, -0.1, 0.8, -1.5, 0.1, 1.0, -1.5, -0.1, 1.0, -1.5], np.float32)
    env.default_dof_pos = T(default[None])
    env.default_dof_pos_all = T(default[None] + 0.01)
    env.mass_params_tensor = T(rng.uniform(-1, 1, (N, 4)).astype(np.float32))
    env.friction_coeffs_tensor = T(rng.uniform(0.6, 2.0, (N, 1)).astype(np.float32))
    env.motor_strength = T(rng.uniform(0.8, 1.2, (2, N, 12)).astype(np.float32))
    env.obs_history_buf = T(rng.normal(0, 1, (N, 10, 57)).astype(np.float32))
    env.contact_buf = torch.zeros(N, cfg.env.contact_buf_len, 4)
    env.commands = T(rng.uniform(-1, 1, (N, 5)).astype(np.float32))
    env.latent_eps = T(rng.uniform(-1, 1, (N, 1)).astype(np.float32))
    env.latent_c = T(np.eye(5, dtype=np.float32)[rng.integers(0, 5, N)])
    env.delta_yaw, env.delta_next_yaw = torch.zeros(N), torch.zeros(N)
    cfg.depth.update_interval = 2                     # so that one of the steps carries the yaw errors over
    out.update(obs_height_samples=hs, obs_height_points=env.height_points.numpy().copy(), obs_default_dof_pos=default,
               obs_default_dof_pos_all=default + np.float32(0.01), obs_mass_params=env.mass_params_tensor.numpy(),
               obs_friction=env.friction_coeffs_tensor.numpy(), obs_motor_strength=env.motor_strength.numpy(),
               obs_history0=env.obs_history_buf.numpy().copy(), obs_commands=env.commands.numpy(), obs_latent_eps=env.latent_eps.numpy(),
               obs_latent_c=env.latent_c.numpy(),
               obs_scalars=np.array([cfg.obstacle.border_size, cfg.obstacle.horizontal_scale, cfg.obstacle.vertical_scale,
                                     env.obs_scales.lin_vel, env.obs_scales.ang_vel, env.obs_scales.dof_pos, env.obs_scales.dof_vel,
                                     0.7, 0.3, 1.5, 2.0,          # the four discriminator scales are 0 in the config: exercised non-zero
                                     cfg.normalization.clip_observations], np.float64))
    env.obs_scales.lin_vel_dist, env.obs_scales.ang_vel_dist, env.obs_scales.key_pos, env.obs_scales.foot_contact = 0.7, 0.3, 1.5, 2.0
    for t in range(3):
        root = np.zeros((N, 13), np.float32)
        root[:, 0] = rng.uniform(-4.0, 10.0, N); root[:, 1] = rng.uniform(-4.0, 6.0, N); root[:, 2] = rng.uniform(0.2, 1.4, N)
        root[:, 3:7] = random_quats(N)
        env.root_states = T(root)
        env.base_quat = env.root_states[:, 3:7]
        env.roll, env.pitch, env.yaw = ref_lr.euler_from_quaternion(env.base_quat)
        env.base_lin_vel, env.base_ang_vel = T(rng.normal(0, 1, (N, 3)).astype(np.float32)), T(rng.normal(0, 1, (N, 3)).astype(np.float32))
        if t == 2:
            env.base_lin_vel[:4] *= 150.0             # past the +-100 clip
        env.contact_filt = T(rng.random((N, 4)) < 0.5, torch.bool)
        env.dof_pos, env.dof_vel = T(default[None] + rng.normal(0, 0.3, (N, 12)).astype(np.float32)), T(rng.normal(0, 3, (N, 12)).astype(np.float32))
        env.action_history_buf = T(rng.normal(0, 1, (N, 8, 12)).astype(np.float32))
        rb = np.zeros((N, 19, 13), np.float32); rb[:, :, :3] = root[:, None, :3] + rng.uniform(-0.4, 0.4, (N, 19, 3))
        env.rigid_body_states = T(rb); env.rigid_body_pos = env.rigid_body_states[:, :, :3]
        env.cur_obstacle_types = T(rng.integers(0, 6, N), torch.long)
        env.target_yaw, env.next_target_yaw = T(rng.uniform(-np.pi, np.pi, N).astype(np.float32)), T(rng.uniform(-np.pi, np.pi, N).astype(np.float32))
        env.episode_length_buf = T(rng.integers(0, 6, N), torch.long)
        env.global_counter = t                        # with update_interval 2: steps 0 and 2 recompute the yaw errors, step 1 carries them over
        env.measured_heights = env._get_heights()     # what _post_physics_step_callback does when measure_heights is set
        env.compute_observations()
        tag = f"obs_t{t}_"
        out.update({tag + k: v for k, v in dict(
            root_states=root, rpy=torch.stack([env.roll, env.pitch, env.yaw], 1).numpy(), base_lin_vel=env.base_lin_vel.numpy().copy(),
            base_ang_vel=env.base_ang_vel.numpy().copy(), contact_filt=env.contact_filt.numpy().astype(np.uint8), dof_pos=env.dof_pos.numpy(),
            dof_vel=env.dof_vel.numpy(), action_history=env.action_history_buf.numpy(), rigid_body_states=rb,
            cur_obstacle_type=env.cur_obstacle_types.numpy(), target_yaw=env.target_yaw.numpy(), next_target_yaw=env.next_target_yaw.numpy(),
            episode_length=env.episode_length_buf.numpy(), update_yaw=np.int32(env.global_counter % cfg.depth.update_interval == 0),
            measured_heights=env.measured_heights.numpy().copy(), delta_yaw=env.delta_yaw.numpy().copy(),
            delta_next_yaw=env.delta_next_yaw.numpy().copy(), obs_buf=env.obs_buf.numpy().copy(), obs_bbc_buf=env.obs_bbc_buf.numpy().copy(),
            obs_disc_buf=env.obs_disc_buf.numpy().copy(), obs_history=env.obs_history_buf.numpy().copy()).items()})
    print("obs", out["obs_t0_obs_buf"].shape, out["obs_t0_obs_bbc_buf"].shape, out["obs_t0_obs_disc_buf"].shape,
          "scan hits", int((out["obs_t0_measured_heights"] > 0).sum()), "first-step envs", int((out["obs_t0_episode_length"] <= 1).sum()))
    path = os.path.join(ROOT, "tests", "golden", "tsc_env.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), "bytes,", len(out), "arrays; reward order", list(out["goal_reward_names"]))
    print("resets", [int(out[f"goal_c{c}_t{t}_reset_buf"].sum()) for c in (0, 1) for t in range(STEPS)],
          "reached", [int(out[f"goal_c{c}_t{t}_reached_goal"].sum()) for c in (0, 1) for t in range(STEPS)],
          "cutoff", [int(out[f"goal_c{c}_t{t}_reach_goal_cutoff"].sum()) for c in (0, 1) for t in range(STEPS)])


if __name__ == "__main__":
    main()
