"""Translate a (instantiated) LeggedRobotCfg tree into the POD `qa_config` of include/qa_sim.h."""
import math

import numpy as np

from quadrupedal_agility_amd import _capi


def class_to_dict(obj):
    """dir()-sorted dict view of a config node (same contract as bbc/legged_gym/utils/helpers.py:12-27:
    keys come out alphabetically, which is what fixes the reference's reward summation order)."""
    if not hasattr(obj, "__dict__"):
        return obj
    out = {}
    for key in dir(obj):
        if key.startswith("_"):
            continue
        val = getattr(obj, key)
        out[key] = [class_to_dict(v) for v in val] if isinstance(val, list) else class_to_dict(val)
    return out


def make_qa_config(cfg, seed=1, sim_dt=None, terrain=None):
    """`cfg.env.env_id_offset` / `cfg.env.num_envs_global` (optional): this process owns envs [offset, offset + num_envs) of a
    num_envs_global-env job -- random draws and spawn slots are keyed by the GLOBAL env id"""
    c = _capi.QaConfig()
    c.abi_version = _capi.QA_ABI_VERSION
    c.num_envs = int(cfg.env.num_envs)
    c.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    c.sim_dt = float(cfg.sim.dt if sim_dt is None else sim_dt)
    c.decimation = int(cfg.control.decimation)
    c.gravity_z = float(cfg.sim.gravity[2])
    qa = getattr(cfg.sim, "qa", None)
    c.solver_iterations = int(getattr(qa, "solver_iterations", 4))
    c.contact_slots = int(getattr(qa, "contact_slots", 2))
    # asset.self_collisions: 0 = ENABLED, 1 = disabled (the reference's bitwise filter flag, go2_locomotion_config.py:72)
    c.self_collision = int(getattr(cfg.asset, "self_collisions", 0) == 0)
    c.contact_offset = float(cfg.sim.physx.contact_offset)
    c.max_depenetration_velocity = float(cfg.sim.physx.max_depenetration_velocity)
    c.ground_friction = float(cfg.terrain.static_friction)
    if cfg.terrain.mesh_type in ("plane", None, "none"):
        c.terrain_type = 0
    elif cfg.terrain.mesh_type in ("heightfield", "trimesh"):
        if terrain is None:
            raise ValueError("a generated Terrain is needed for mesh_type 'heightfield' / 'trimesh'")
        c.terrain_type = 1
        c.hf_rows, c.hf_cols = int(terrain.tot_rows), int(terrain.tot_cols)
        c.hf_hscale, c.hf_vscale = float(cfg.terrain.horizontal_scale), float(cfg.terrain.vertical_scale)
        c.hf_border = float(cfg.terrain.border_size)
        c.reset_xy_jitter = 1.0          # custom_origins branch of _reset_root_states (legged_robot.py:622-625)
    else:
        raise ValueError("Terrain mesh type not recognised. Allowed types are [None, plane, heightfield, trimesh]")
    if cfg.control.control_type != "P":
        raise NotImplementedError("only control_type 'P' (legged_robot.py:563-570) is on the hot path")
    kp = _gain(cfg.control.stiffness)
    kd = _gain(cfg.control.damping)
    c.kp, c.kd = kp, kd
    c.action_scale = float(cfg.control.action_scale)
    c.hip_scale_reduction = float(getattr(cfg.control, "hip_scale_reduction", 1.0))
    c.clip_actions = float(cfg.normalization.clip_actions)
    for i, name in enumerate(_capi.DOF_NAMES):
        c.default_dof_pos[i] = float(cfg.init_state.default_joint_angles[name])
    dt = cfg.control.decimation * (cfg.sim.dt if sim_dt is None else sim_dt)   # python double, legged_robot.py:1139
    c.env_spacing = float(cfg.env.env_spacing)
    c.max_episode_length = int(np.ceil(cfg.env.episode_length_s / dt))
    c.resampling_steps = int(min(cfg.commands.resampling_time / dt, 2 ** 31 - 1))      # play.py sets 1e10 s = never
    c.push_interval = int(np.ceil(cfg.domain_rand.push_interval_s / dt))
    c.push_robots = int(bool(cfg.domain_rand.push_robots))
    c.max_push_vel_xy = float(cfg.domain_rand.max_push_vel_xy)
    c.reset_mode = 1 if cfg.env.mocap_state_init else 0
    for i in range(3):
        c.init_pos[i] = float(cfg.init_state.pos[i])
    ns, os_, lvl = cfg.noise.noise_scales, cfg.normalization.obs_scales, cfg.noise.noise_level
    c.add_noise = int(bool(cfg.noise.add_noise))
    c.noise_roll_pitch = float(getattr(ns, "roll_pitch", 0.0) * lvl)
    c.noise_ang_vel = float(ns.ang_vel * lvl * os_.ang_vel)
    c.noise_dof_pos = float(ns.dof_pos * lvl * os_.dof_pos)
    c.noise_dof_vel = float(ns.dof_vel * lvl * os_.dof_vel)
    c.noise_lin_vel = float(ns.lin_vel * lvl * os_.lin_vel)
    c.clip_obs = float(cfg.normalization.clip_observations)
    c.s_lin_vel, c.s_ang_vel = float(os_.lin_vel), float(os_.ang_vel)
    c.s_dof_pos, c.s_dof_vel = float(os_.dof_pos), float(os_.dof_vel)
    c.s_key_pos, c.s_foot_contact = float(os_.key_pos), float(os_.foot_contact)
    c.s_lin_vel_dist, c.s_ang_vel_dist = float(os_.lin_vel_dist), float(os_.ang_vel_dist)
    scales = class_to_dict(cfg.rewards.scales)
    known = set(_capi.REWARD_NAMES)
    for name, val in scales.items():
        if val != 0 and name not in known and name != "termination":
            raise NotImplementedError(f"reward term {name!r} has a non-zero scale but is not on the hot path")
    for i, name in enumerate(_capi.REWARD_NAMES):
        c.reward_scale_dt[i] = float(scales.get(name, 0.0) * dt)     # legged_robot.py:927-932
    c.only_positive_rewards = int(bool(cfg.rewards.only_positive_rewards))
    c.tracking_sigma = float(cfg.rewards.tracking_sigma)
    c.soft_dof_pos_limit = float(cfg.rewards.soft_dof_pos_limit)
    c.soft_dof_vel_limit = float(cfg.rewards.soft_dof_vel_limit)
    c.soft_torque_limit = float(cfg.rewards.soft_torque_limit)
    c.jump_goal = float(getattr(cfg.rewards, "jump_goal", 0.0))
    r = cfg.commands.ranges
    for g in range(_capi.NUM_GAITS):
        for k in range(2):
            c.lin_vel_x[g][k] = float(r.lin_vel_x[g][k])
            c.lin_vel_y[g][k] = float(r.lin_vel_y[g][k])
            c.ang_vel_yaw[g][k] = float(r.ang_vel_yaw[g][k])
    for k in range(2):
        c.jump_height[k] = float(r.jump_height[k])
        c.locomotion_height[k] = float(r.locomotion_height[k])
    c.lin_vel_x_clip = float(cfg.commands.lin_vel_x_clip)
    c.lin_vel_y_clip = float(cfg.commands.lin_vel_y_clip)
    c.ang_vel_yaw_clip = float(cfg.commands.ang_vel_yaw_clip)
    c.latent_temperature = 0.25
    d = cfg.domain_rand
    c.randomize_friction = int(bool(d.randomize_friction))
    c.randomize_base_mass = int(bool(d.randomize_base_mass))
    c.randomize_base_com = int(bool(getattr(d, "randomize_base_com", False)))
    c.randomize_motor = int(bool(getattr(d, "randomize_motor", False)))
    c.use_easi = int(bool(getattr(d, "use_easi", False)))
    for k in range(2):
        c.friction_range[k] = float(d.friction_range[k])
        c.added_mass_range[k] = float(d.added_mass_range[k])
        c.added_com_range[k] = float(getattr(d, "added_com_range", [0.0, 0.0])[k])
        c.motor_strength_range[k] = float(getattr(d, "motor_strength_range", [1.0, 1.0])[k])
    for k in range(6):
        c.easi_mean[k] = float(getattr(d, "easi_mean", [1.0] * 7)[k])
        c.easi_var[k] = float(getattr(d, "easi_var", [0.0] * 7)[k])
    c.num_mocap_frames = 0
    c.env_id_offset = int(getattr(cfg.env, "env_id_offset", 0))
    c.num_envs_global = int(getattr(cfg.env, "num_envs_global", 0))
    return c


def _gain(table):
    vals = {float(v) for k, v in table.items() if k in "joint" or "joint" in k}
    if len(vals) != 1:
        raise NotImplementedError("per-joint PD gains are not supported; Go2 uses one gain for every joint")
    return vals.pop()
