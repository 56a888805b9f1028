"""Host-side mirror of the task-level `LeggedRobot`'s command mapping and goal bookkeeping
(tsc/legged_gym/envs/base/legged_robot.py: set_commands :699-760, post_physics_step :226-273, _update_goals :204-224,
check_termination :322-346, compute_reward :412-430 with the rewards of :1779-1925, _get_heights :1708-1755, compute_observations
:432-515).  Attribute names are the reference's
(`commands`, `latent_eps`, `latent_c`, `cur_goal_idx`, `reach_goal_timer`, `target_yaw`, `reset_buf`, `rew_buf`, ...), so code
written against the reference env reads the same tensors.  All arithmetic runs in the HIP library (no CPU path)."""
import ctypes as C

import numpy as np
import torch

from ... import _capi


def coarse_depth_maps(height_samples, ceiling_samples, log2_block):
    """The depth ray-cast's acceleration structure (qa_tsc_depth_io.coarse_floor_max / coarse_ceiling_min): per block of 2^S x 2^S
    cells the highest height sample and the lowest ceiling sample among the (2^S + 1)^2 samples those cells touch."""
    import torch.nn.functional as F
    b = 1 << log2_block
    rows, cols = height_samples.shape
    cr, cc = ((rows - 2) >> log2_block) + 1, ((cols - 2) >> log2_block) + 1

    def pooled(m, sign, fill):
        x = F.pad(sign * m.to(torch.float32)[None, None], (0, cc * b + 1 - cols, 0, cr * b + 1 - rows), value=sign * fill)
        return (sign * F.max_pool2d(x, kernel_size=b + 1, stride=b))[0, 0].to(torch.int16).contiguous()
    return pooled(height_samples, 1.0, -32768.0), (pooled(ceiling_samples, -1.0, 32767.0) if ceiling_samples is not None else None)


class TaskLevelBookkeeping:
    """State and per-step calls of one rank's task-level envs.

    cfg: an object with the reference's config fields (`env.mocap_category(_all)`, `env.num_actions_c`, `env.reach_goal_delay`,
    `env.next_goal_threshold`, `env.leave_goal_threshold`, `env.episode_length_s`, `commands.ranges`, `commands.resampling_time`,
    `rewards.scales`, `rewards.target_lin_vel`, `obstacle.{num_goals,last_goal_repeat,border_size,horizontal_scale}`,
    `depth.use_camera`, `control.decimation`, `sim.dt`).
    env_goals (N, slots, 3), obstacle_types (N, K) long, x_edge_mask (rows, cols) bool; body index lists as in the reference's
    `feet_indices`, `penalised_contact_indices`, `termination_contact_indices`."""

    def __init__(self, cfg, env_goals, obstacle_types, x_edge_mask, feet_indices, penalised_contact_indices,
                 termination_contact_indices, num_bodies, device="cuda:0", lib=None):
        # `lib`: (library, symbol prefix) injected by the CPU tests, which drive this class with the oracle's twins on host
        # tensors; the product never passes it and loads the HIP library, which needs a GPU
        self._prefix = "qa_"
        if lib is not None:
            self.lib, self._prefix = lib
        else:
            self.lib = _capi.load_library()
            if not torch.cuda.is_available():
                raise RuntimeError("TaskLevelBookkeeping needs a GPU: quadrupedal_agility_amd has no CPU fallback")
        self.cfg, self.device = cfg, torch.device(device)
        dev = self.device
        self.dt = cfg.control.decimation * cfg.sim.dt
        self.env_goals = env_goals.to(dev, torch.float32).contiguous()
        self.obstacle_types = obstacle_types.to(dev, torch.long).contiguous()
        self.x_edge_mask = x_edge_mask.to(dev, torch.uint8).contiguous()
        n = self.num_envs = self.env_goals.shape[0]
        cats_all = list(cfg.env.mocap_category_all)
        self.dim_c = len(cats_all)
        self.num_actions_d, self.num_actions_c = len(cfg.env.mocap_category), cfg.env.num_actions_c
        self._mocap_index = np.array([cats_all.index(c) for c in cfg.env.mocap_category], np.int32)
        r = cfg.commands.ranges
        self._vel_ranges = np.ascontiguousarray([r.lin_vel_x, r.lin_vel_y, r.ang_vel_yaw], np.float32)
        self._jump_range = np.ascontiguousarray(r.jump_height, np.float32)
        self._height_range = np.ascontiguousarray(r.locomotion_height, np.float32)
        self.max_episode_length = float(np.ceil(cfg.env.episode_length_s / self.dt))

        z = lambda *s, dt=torch.float32: torch.zeros(*s, dtype=dt, device=dev)       # noqa: E731
        self.commands, self.latent_eps, self.latent_c = z(n, 5), z(n, 1), z(n, self.dim_c)
        self.next_commands = z(n, 6 + self.dim_c)
        self.episode_length_buf, self.cur_goal_idx = z(n, dt=torch.long), z(n, dt=torch.long)
        self.reach_goal_timer, self.last_contacts = z(n), z(n, 4, dt=torch.uint8)
        self.episode_sums_buf = z(len(_capi.TSC_REWARD_NAMES), n)
        self.episode_sums = {name: self.episode_sums_buf[i] for i, name in enumerate(_capi.TSC_REWARD_NAMES)}
        self.base_lin_vel, self.base_ang_vel, self.projected_gravity, self.rpy = z(n, 3), z(n, 3), z(n, 3), z(n, 3)
        self.contact_filt = z(n, 4, dt=torch.uint8)
        self.target_pos_rel, self.next_target_pos_rel, self.target_yaw, self.next_target_yaw = z(n, 2), z(n, 2), z(n), z(n)
        self.reached_goal_ids, self.cur_obstacle_types = z(n, dt=torch.uint8), z(n, dt=torch.long)
        self.reset_buf, self.time_out_buf, self.reach_goal_cutoff = (z(n, dt=torch.uint8) for _ in range(3))
        self.rew_buf = z(n)
        self.cur_goals, self.next_goals = self._gather_cur_goals(), self._gather_cur_goals(future=1)

        c = self._cfg = _capi.QaTscGoalCfg()
        c.num_envs, c.num_bodies = n, int(num_bodies)
        c.num_goal_slots, c.last_goal_repeat = self.env_goals.shape[1], cfg.obstacle.last_goal_repeat
        c.goals_per_obstacle, c.num_obstacles = cfg.obstacle.num_goals, self.obstacle_types.shape[1]
        c.mask_rows, c.mask_cols = self.x_edge_mask.shape
        c.use_camera = int(bool(cfg.depth.use_camera))
        for name, ids in (("termination", termination_contact_indices), ("penalised", penalised_contact_indices)):
            ids = [int(i) for i in ids]
            if len(ids) > _capi.TSC_MAX_BODY_IDS:
                raise ValueError(f"at most {_capi.TSC_MAX_BODY_IDS} {name} bodies")
            setattr(c, f"num_{name}_bodies", len(ids))
            for i, b in enumerate(ids):
                getattr(c, f"{name}_bodies")[i] = b
        for i, b in enumerate(feet_indices):
            c.feet_bodies[i] = int(b)
        c.reach_goal_delay_steps = cfg.env.reach_goal_delay / self.dt
        c.next_goal_threshold, c.leave_goal_threshold = cfg.env.next_goal_threshold, cfg.env.leave_goal_threshold
        c.max_episode_length, c.target_lin_vel = self.max_episode_length, cfg.rewards.target_lin_vel
        c.border_size, c.horizontal_scale = cfg.obstacle.border_size, cfg.obstacle.horizontal_scale
        self.reward_scales = {}
        for i, name in enumerate(_capi.TSC_REWARD_NAMES):                 # _prepare_reward_function :1107-1113: scale * dt
            self.reward_scales[name] = float(getattr(cfg.rewards.scales, name)) * self.dt
            c.reward_scales[i] = self.reward_scales[name]

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream) if self.device.type == "cuda" else None

    def _fn(self, name):
        return getattr(self.lib, self._prefix + name)

    def _check(self, rc, what):
        if rc != 0:
            err = self.lib.qa_last_error().decode() if hasattr(self.lib, "qa_last_error") else ""
            raise RuntimeError(f"{what} failed with code {rc}: {err}")

    def _gather_cur_goals(self, future=0):
        """legged_robot.py:646-648"""
        idx = (self.cur_goal_idx[:, None, None] + future).clamp(0, self.env_goals.shape[1] - 1).expand(-1, -1, 3)
        return self.env_goals.gather(1, idx).squeeze(1).contiguous()

    def set_commands(self, actions, action_noise=None):
        """legged_robot.py:699-760.  `action_noise` (N,5): the U(domain_rand.action_noise) draw, or None for no noise."""
        actions = actions.to(self.device, torch.float32).contiguous()
        interval = max(1, int(self.cfg.commands.resampling_time / self.dt))
        if action_noise is not None:
            action_noise = action_noise.to(self.device, torch.float32).contiguous()
        rc = self._fn("tsc_set_commands")(
            actions.data_ptr(), self.episode_length_buf.data_ptr(), self.num_envs, self.num_actions_d, self.num_actions_c, self.dim_c,
            interval, self._mocap_index.ctypes.data, self._vel_ranges.ctypes.data, self._jump_range.ctypes.data,
            self._height_range.ctypes.data, action_noise.data_ptr() if action_noise is not None else None, self.commands.data_ptr(),
            self.latent_eps.data_ptr(), self.latent_c.data_ptr(), self.next_commands.data_ptr(), self._stream())
        self._check(rc, "qa_tsc_set_commands")
        return self.next_commands

    def post_physics_step(self, root_states, contact_forces, rigid_body_states, action_hl_history_buf=None, want_ids=True):
        """The goal / termination / reward part of legged_robot.py:226-273 on the simulator's tensors; returns the ids to reset.
        The caller's reset then calls `reset_idx(env_ids)`."""
        keep = [t.to(self.device, torch.float32).contiguous() for t in (root_states, contact_forces, rigid_body_states)]
        io = _capi.QaTscGoalIo()
        io.root_states, io.contact_forces, io.rigid_body_states = (t.data_ptr() for t in keep)
        if action_hl_history_buf is not None:
            hist = action_hl_history_buf.to(self.device, torch.float32).contiguous()
            self._cfg.history_len, self._cfg.history_width = hist.shape[1], hist.shape[2]
            io.action_hl_history = hist.data_ptr()
        members = dict(env_goals=self.env_goals, obstacle_types=self.obstacle_types, x_edge_mask=self.x_edge_mask,
                       episode_length=self.episode_length_buf, cur_goal_idx=self.cur_goal_idx, reach_goal_timer=self.reach_goal_timer,
                       last_contacts=self.last_contacts, cur_goals=self.cur_goals, next_goals=self.next_goals,
                       episode_sums=self.episode_sums_buf, base_lin_vel=self.base_lin_vel, base_ang_vel=self.base_ang_vel,
                       projected_gravity=self.projected_gravity, rpy=self.rpy, contact_filt=self.contact_filt,
                       target_pos_rel=self.target_pos_rel, next_target_pos_rel=self.next_target_pos_rel, target_yaw=self.target_yaw,
                       next_target_yaw=self.next_target_yaw, reached_goal=self.reached_goal_ids, cur_obstacle_type=self.cur_obstacle_types,
                       reset_buf=self.reset_buf, time_out_buf=self.time_out_buf, reach_goal_cutoff=self.reach_goal_cutoff,
                       rew_buf=self.rew_buf)
        for name, t in members.items():
            setattr(io, name, t.data_ptr())
        self._check(self._fn("tsc_goal_step")(C.byref(self._cfg), C.byref(io), self._stream()), "qa_tsc_goal_step")
        self.roll, self.pitch, self.yaw = self.rpy[:, 0], self.rpy[:, 1], self.rpy[:, 2]
        return self.reset_buf.nonzero(as_tuple=False).flatten() if want_ids else None          # nonzero() is a host sync

    def init_observations(self, height_samples, height_points, default_dof_pos, default_dof_pos_all=None, key_body_ids=None):
        """Buffers and constants of compute_observations: the obstacle course's int16 height map, the reference's
        `height_points` tensor (N,132,3) or one shared (132,2|3) grid, the default joint angles."""
        dev, n, cfg = self.device, self.num_envs, self.cfg
        self.height_samples = height_samples.to(dev, torch.int16).contiguous()
        self.height_points = height_points.to(dev, torch.float32).contiguous()
        if self.height_points.shape[-2] != _capi.TSC_NUM_SCAN:
            raise ValueError(f"{_capi.TSC_NUM_SCAN} scan points expected")
        z = lambda *s: torch.zeros(*s, dtype=torch.float32, device=dev)       # noqa: E731
        self.obs_buf, self.obs_bbc_buf, self.obs_disc_buf = z(n, _capi.TSC_NUM_OBS), z(n, _capi.TSC_NUM_OBS_BBC), z(n, _capi.TSC_NUM_OBS_DISC)
        self.obs_history_buf, self.measured_heights = z(n, 10, 57), z(n, _capi.TSC_NUM_SCAN)
        self.delta_yaw, self.delta_next_yaw = z(n), z(n)
        c = self._ocfg = _capi.QaTscObsCfg()
        c.num_envs, c.num_bodies = n, self._cfg.num_bodies
        for i, b in enumerate(key_body_ids if key_body_ids is not None else list(self._cfg.feet_bodies)):
            c.key_bodies[i] = int(b)
        c.map_rows, c.map_cols = self.height_samples.shape
        c.root_height_obs = int(bool(cfg.env.root_height_obs))
        c.point_stride = self.height_points.shape[-1]
        c.points_env_stride = self.height_points.shape[-2] * self.height_points.shape[-1] if self.height_points.dim() == 3 else 0
        c.border_size, c.horizontal_scale, c.vertical_scale = cfg.obstacle.border_size, cfg.obstacle.horizontal_scale, cfg.obstacle.vertical_scale
        sc = cfg.normalization.obs_scales
        c.lin_vel, c.ang_vel, c.dof_pos, c.dof_vel = sc.lin_vel, sc.ang_vel, sc.dof_pos, sc.dof_vel
        c.lin_vel_dist, c.ang_vel_dist, c.key_pos, c.foot_contact = sc.lin_vel_dist, sc.ang_vel_dist, sc.key_pos, sc.foot_contact
        c.clip_observations = cfg.normalization.clip_observations
        d0 = [float(v) for v in torch.as_tensor(default_dof_pos).flatten()[:12]]
        d1 = [float(v) for v in torch.as_tensor(default_dof_pos_all).flatten()[:12]] if default_dof_pos_all is not None else d0
        for i in range(12):
            c.default_dof_pos[i], c.default_dof_pos_all[i] = d0[i], d1[i]

    def compute_observations(self, root_states, dof_pos, dof_vel, action_history_buf, rigid_body_states, mass_params_tensor,
                             friction_coeffs_tensor, motor_strength, update_yaw=True):
        """legged_robot.py:432-515 with the scan of :1708-1755, after `post_physics_step`; fills obs_buf (N,800), obs_bbc_buf (N,671),
        obs_disc_buf (N,49), measured_heights, and pushes obs_history_buf.  `update_yaw` = (global_counter % depth.update_interval == 0)."""
        f32 = lambda t: t.to(self.device, torch.float32)                       # noqa: E731
        hist = f32(action_history_buf)
        last = hist[:, -1]
        keep = dict(root_states=f32(root_states).contiguous(), dof_pos=f32(dof_pos).contiguous(), dof_vel=f32(dof_vel).contiguous(),
                    rigid_body_states=f32(rigid_body_states).contiguous(), mass_params=f32(mass_params_tensor).contiguous(),
                    friction=f32(friction_coeffs_tensor).contiguous(), motor_strength=f32(motor_strength).contiguous())
        if last.stride(-1) != 1:
            last = last.contiguous()
        self._ocfg.action_stride, self._ocfg.update_yaw = last.stride(0), int(bool(update_yaw))
        members = dict(keep, last_action=last, rpy=self.rpy, base_lin_vel=self.base_lin_vel, base_ang_vel=self.base_ang_vel,
                       contact_filt=self.contact_filt, cur_obstacle_type=self.cur_obstacle_types, target_yaw=self.target_yaw,
                       next_target_yaw=self.next_target_yaw, height_samples=self.height_samples, height_points=self.height_points,
                       commands=self.commands, latent_eps=self.latent_eps, latent_c=self.latent_c, episode_length=self.episode_length_buf,
                       delta_yaw=self.delta_yaw, delta_next_yaw=self.delta_next_yaw, obs_history=self.obs_history_buf,
                       measured_heights=self.measured_heights, obs_buf=self.obs_buf, obs_bbc_buf=self.obs_bbc_buf, obs_disc_buf=self.obs_disc_buf)
        io = _capi.QaTscObsIo()
        for name in _capi.TSC_OBS_IO_FIELDS:
            setattr(io, name, members[name].data_ptr())
        self._check(self._fn("tsc_observations")(C.byref(self._ocfg), C.byref(io), self._stream()), "qa_tsc_observations")
        return self.obs_buf

    def init_depth(self, height_samples, ceiling_samples, camera_pitch, seed=1, env_id_offset=0):
        """The depth camera of attach_camera (:1203-1226) and the `depth` config block: one camera per env on the trunk, pitched
        down by `camera_pitch` (N) radians; `depth_buffer` (N, buffer_len, 58, 87) is the reference's ring."""
        d, dev, n = self.cfg.depth, self.device, self.num_envs
        self._depth_maps = (height_samples.to(dev, torch.int16).contiguous(),
                            ceiling_samples.to(dev, torch.int16).contiguous() if ceiling_samples is not None else None)
        self.camera_pitch = camera_pitch.to(dev, torch.float32).contiguous()
        c = self._dcfg = _capi.QaTscDepthCfg()
        c.num_envs, c.seed, c.env_id_offset = n, int(seed), int(env_id_offset)
        c.width, c.height = int(d.original[0]), int(d.original[1])
        c.crop_top, c.crop_bottom, c.crop_left, c.crop_right = 1, 1, 10, 9            # crop_depth_image :170-172
        c.buffer_len = int(d.buffer_len)
        c.map_rows, c.map_cols = self._depth_maps[0].shape
        c.horizontal_fov_deg = float(d.horizontal_fov)
        for i in range(3):
            c.position[i] = float(d.position[i])
        c.near_clip, c.far_clip, c.depth_noise = float(d.near_clip), float(d.far_clip), float(d.depth_noise)
        ob = self.cfg.obstacle
        c.border_size, c.horizontal_scale, c.vertical_scale = ob.border_size, ob.horizontal_scale, ob.vertical_scale
        hc, wc = c.height - c.crop_top - c.crop_bottom, c.width - c.crop_left - c.crop_right
        if (wc, hc) != tuple(d.resized):
            raise ValueError(f"depth.resized {tuple(d.resized)} is not the cropped image {(wc, hc)}")
        self.depth_buffer = torch.zeros(n, c.buffer_len, hc, wc, dtype=torch.float32, device=dev)
        c.coarse_log2 = 3
        self._depth_coarse = coarse_depth_maps(self._depth_maps[0], self._depth_maps[1], c.coarse_log2)

    def update_depth_buffer(self, root_states, step):
        """update_depth_buffer + process_depth_image (:154-200) for all envs in one launch; `step` keys the noise draw"""
        rs = root_states.to(self.device, torch.float32).contiguous()
        io = _capi.QaTscDepthIo()
        io.root_states, io.camera_pitch, io.height_samples = rs.data_ptr(), self.camera_pitch.data_ptr(), self._depth_maps[0].data_ptr()
        io.ceiling_samples = self._depth_maps[1].data_ptr() if self._depth_maps[1] is not None else None
        io.episode_length, io.depth_buffer = self.episode_length_buf.data_ptr(), self.depth_buffer.data_ptr()
        io.coarse_floor_max = self._depth_coarse[0].data_ptr()
        io.coarse_ceiling_min = self._depth_coarse[1].data_ptr() if self._depth_coarse[1] is not None else None
        if torch.is_tensor(step) and step.is_cuda:           # read when the launch executes (recorded rollouts)
            io.step_dev = step.data_ptr()
        else:
            self._dcfg.step = int(step)
        self._check(self._fn("tsc_depth_update")(C.byref(self._dcfg), C.byref(io), self._stream()), "qa_tsc_depth_update")
        return self.depth_buffer

    def get_observations(self):
        return self.obs_buf

    def get_observations_bbc(self):
        return self.obs_bbc_buf

    def get_observations_disc(self):
        return self.obs_disc_buf

    def reset_idx(self, env_ids):
        """The goal / episode bookkeeping of the reference's reset_idx (:376, :396-404) and the re-gather of :272-273."""
        if len(env_ids):
            self.cur_goal_idx[env_ids] = 0
            self.reach_goal_timer[env_ids] = 0
            self.episode_sums_buf[:, env_ids] = 0
            self.episode_length_buf[env_ids] = 0
            self.cur_goals[env_ids] = self.env_goals[env_ids, 0]
            self.next_goals[env_ids] = self.env_goals[env_ids, 1]

    def reset_where(self, flags, start_goal_idx=None):
        """reset_idx's bookkeeping for the envs with flags != 0 (uint8 (N)), as masked in-place updates -- no index list, no host
        sync: goal index back to the start goal (0, or the first goal of the start obstacle with obstacle.randomize_start), dwell
        timer, episode sums and clock cleared, current / next goal re-gathered (:367-376, 396-404, 272-273)"""
        m = flags != 0
        start = torch.zeros_like(self.cur_goal_idx) if start_goal_idx is None else start_goal_idx
        self.cur_goal_idx.copy_(torch.where(m, start, self.cur_goal_idx))
        self.reach_goal_timer.mul_((~m).to(self.reach_goal_timer.dtype))
        self.episode_sums_buf.mul_((~m).to(self.episode_sums_buf.dtype).unsqueeze(0))
        self.episode_length_buf.mul_((~m).to(self.episode_length_buf.dtype))
        self.cur_goals.copy_(self._gather_cur_goals())
        self.next_goals.copy_(self._gather_cur_goals(future=1))
