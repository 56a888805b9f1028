"""`LeggedRobot` of the task-level (TSC) tree -- the agility-course env the hybrid task policy is trained in
(tsc/legged_gym/envs/base/legged_robot.py).  Same constructor, methods and attributes the task-level runner touches
(`step(actions_bbc, action_hl_history_buf)`, `set_commands`, `get_observations(_bbc|_disc)`, `num_obs`, `num_actions_d/c`,
`episode_length_buf`, `success_rate`, ...), built from four device pieces and no host-side arithmetic:

  * `qa_env_physics_step` (include/qa_sim.h): action history / delay / clip, 4 x (PD -> articulated-body substep with contact
    against the course's height field), refresh of root / dof / contact-force / rigid-body-state tensors   (:113-139, 231-234)
  * `qa_tsc_goal_step`: episode clock, goal dwell + advance, 7 termination causes, 8 rewards                    (:204-262, 322-346, 412-430)
  * `qa_tsc_reset` + `qa_simulate_if`: reset of the flagged envs and the reference's extra simulate on resets  (:348-410)
  * `qa_tsc_observations`: 132-point scan, the 800 / 671 / 49 rows, history push                                (:432-515, 1708-1755)

The course is `utils/obstacle.py::Obstacle` (bit-identical to the reference's generator).  Its int16 map is what the scan reads
AND -- this build's physics model, DESIGN.md section 3.2 -- the collision terrain: bars, A-frame, poles, see-saw, tyre and tunnel
are collided with as the 2.5-D surface the reference's scan map encodes, the border as 2 m walls.  What a height field cannot
hold is stated there (tunnel roof and upper tyre ring: no contact from above; see-saw: fixed at its initial tilt)."""
import numpy as np
import torch

from quadrupedal_agility_amd import _capi
from quadrupedal_agility_amd.legged_gym.utils.cfg_to_c import class_to_dict
from quadrupedal_agility_amd.tsc.legged_gym.task_level import TaskLevelBookkeeping
from quadrupedal_agility_amd.tsc.legged_gym.utils.obstacle import Obstacle

KEY_BODY_NAMES = ["FL_foot", "FR_foot", "RL_foot", "RR_foot"]


def make_qa_config(cfg, obstacle, seed=1):
    """qa_config for the physics-only use of the engine: simulator, PD controller, domain randomisation, the course as terrain.
    Everything the behaviour-level post-physics phase would read (rewards, commands, noise) stays zero: that phase never runs."""
    c = _capi.QaConfig()
    c.abi_version, c.num_envs, c.seed = _capi.QA_ABI_VERSION, int(cfg.env.num_envs), int(seed) & 0xFFFFFFFFFFFFFFFF
    c.sim_dt, c.decimation, c.gravity_z = float(cfg.sim.dt), int(cfg.control.decimation), float(cfg.sim.gravity[2])
    c.solver_iterations = int(getattr(getattr(cfg.sim, "qa", None), "solver_iterations", 4))
    c.contact_slots = int(getattr(getattr(cfg.sim, "qa", None), "contact_slots", 2))
    c.self_collision = int(getattr(cfg.asset, "self_collisions", 0) == 0)      # 0 = enabled (go2_agility_config.py:43)
    c.contact_offset = float(cfg.sim.physx.contact_offset)
    c.max_depenetration_velocity = float(cfg.sim.physx.max_depenetration_velocity)
    c.ground_friction = float(cfg.terrain.static_friction)
    c.terrain_type = 1
    c.hf_rows, c.hf_cols = int(obstacle.tot_rows), int(obstacle.tot_cols)
    c.hf_hscale, c.hf_vscale, c.hf_border = float(obstacle.horizontal_scale), float(obstacle.vertical_scale), float(cfg.obstacle.border_size)
    c.hf_ceiling = 1                                     # tunnel roof + upper arc of the tyre (obstacle.ceiling_raw -> CEILING_SAMPLES)
    # see-saw = 1-DoF revolute plank with joint damping, bar / tyre = 1-DoF prismatic bodies under the reference's position drive
    # (QA_T_OBST_DESC / QA_T_OBST_STATE; :792-794, :812-823, :1411-1427); 0 = the static shapes of the height map (the r2 model)
    c.articulated_obstacles = int(bool(getattr(cfg.obstacle, "articulated", True)))
    if cfg.control.control_type != "P":
        raise NotImplementedError("only control_type 'P' is on the hot path")
    kp = {float(v) for v in cfg.control.stiffness.values()}
    kd = {float(v) for v in cfg.control.damping.values()}
    if len(kp) != 1 or len(kd) != 1:
        raise NotImplementedError("one PD gain for every joint")
    c.kp, c.kd = kp.pop(), kd.pop()
    c.action_scale, c.hip_scale_reduction = float(cfg.control.action_scale), float(cfg.control.hip_scale_reduction)
    c.clip_actions = float(cfg.normalization.clip_actions)
    for i, name in enumerate(_capi.DOF_NAMES):
        c.default_dof_pos[i] = float(cfg.init_state.default_joint_angles[name])
    for i in range(3):
        c.init_pos[i] = float(cfg.init_state.pos[i])
    c.env_spacing, c.clip_obs = float(cfg.env.env_spacing), float(cfg.normalization.clip_observations)
    dt = cfg.control.decimation * cfg.sim.dt
    c.max_episode_length = int(np.ceil(cfg.env.episode_length_s / dt))
    c.resampling_steps, c.push_interval = 1 << 30, 1 << 30
    d = cfg.domain_rand
    c.randomize_friction, c.randomize_base_mass = int(bool(d.randomize_friction)), int(bool(d.randomize_base_mass))
    c.randomize_base_com, c.randomize_motor, c.use_easi = int(bool(d.randomize_base_com)), int(bool(d.randomize_motor)), 0
    for k in range(2):
        c.friction_range[k], c.added_mass_range[k] = float(d.friction_range[k]), float(d.added_mass_range[k])
        c.added_com_range[k], c.motor_strength_range[k] = float(d.added_com_range[k]), float(d.motor_strength_range[k])
    c.latent_temperature = 0.25
    c.export_body_state = 1
    # data parallel: this rank's envs are [env_id_offset, env_id_offset + num_envs) of a num_envs_global-env job; every Philox draw is
    # keyed by the global env id (the BBC tree's rule, cfg_to_c.py)
    c.env_id_offset = int(getattr(cfg.env, "env_id_offset", 0))
    c.num_envs_global = int(getattr(cfg.env, "num_envs_global", 0)) or int(cfg.env.num_envs)
    return c


class LeggedRobot:
    def __init__(self, cfg, sim_params=None, physics_engine=None, sim_device="cuda:0", headless=True, backend=None, bookkeeping_lib=None):
        self.cfg, self.sim_params, self.physics_engine, self.headless = cfg, sim_params, physics_engine, headless
        if cfg.terrain.mesh_type != "obstacle":
            raise NotImplementedError("the task-level env runs on the obstacle course (terrain.mesh_type 'obstacle')")
        self.mocap_category = cfg.env.mocap_category_all
        self.dim_c = len(self.mocap_category)
        self.dt = cfg.control.decimation * cfg.sim.dt
        self.obs_scales = cfg.normalization.obs_scales
        self.command_ranges = class_to_dict(cfg.commands.ranges)
        self.max_episode_length_s = cfg.env.episode_length_s
        self.max_episode_length = np.ceil(self.max_episode_length_s / self.dt)
        cfg.domain_rand.push_interval = np.ceil(cfg.domain_rand.push_interval_s / self.dt)
        self.success_rate, self.obst_curr_count, self.bar_jump_bias, self.tire_jump_bias = 0, 0, 0, 0
        self.num_envs = cfg.env.num_envs
        self.num_obs, self.num_obs_bbc = cfg.env.num_observations, cfg.env.num_observations_bbc
        self.num_privileged_obs, self.num_obs_disc = cfg.env.num_privileged_obs, cfg.env.num_obs_disc
        self.num_actions_d, self.num_actions_c, self.num_actions = cfg.env.num_actions_d, cfg.env.num_actions_c, cfg.env.num_actions_bbc
        self.num_bodies, self.num_dof = len(_capi.BODY_NAMES), 12
        seed = int(getattr(cfg, "seed", 1))

        # ---- course + engine
        # one course for the whole job: a rank builds its own envs of it (the draws of the envs before them are skipped, not re-keyed)
        self.obstacle = Obstacle(cfg.obstacle, self.num_envs, seed=getattr(cfg, "course_seed", seed), skip_envs=int(getattr(cfg.env, "env_id_offset", 0)))
        self.qcfg = make_qa_config(cfg, self.obstacle, seed=seed)
        if backend is None:
            from quadrupedal_agility_amd.sim import QaSim
            backend = QaSim(self.qcfg, sim_device)
        elif isinstance(backend, type):
            backend = backend(self.qcfg)             # a factory: the final qa_config (course size) is only known here
        self.sim = backend
        self.device = str(backend.device)
        dev = self.device
        t = self.sim.t
        self.height_samples = torch.from_numpy(np.ascontiguousarray(self.obstacle.height_field_raw)).to(dev)
        t["HEIGHT_SAMPLES"].copy_(self.height_samples)
        t["CEILING_SAMPLES"].copy_(torch.as_tensor(self.obstacle.ceiling_raw, dtype=torch.int16))
        self.x_edge_mask = torch.from_numpy(np.ascontiguousarray(self.obstacle.x_edge_mask)).to(dev)
        self.env_origins = t["ENV_ORIGINS"]
        self.env_origins.copy_(torch.from_numpy(self.obstacle.env_origins).to(dev, torch.float32))
        self.env_goals = torch.from_numpy(self.obstacle.flat_goals()).to(dev, torch.float32).contiguous()     # (N, 26, 3)
        self.obstacle_types = torch.from_numpy(self.obstacle.obstacle_types).to(dev, torch.long)
        self.obst_angs = torch.from_numpy(self.obstacle.frame_ang).to(dev, torch.float32).unsqueeze(0).expand(self.num_envs, -1).contiguous()
        self.custom_origins = True
        self._init_articulated_obstacles(seed)

        # ---- buffers = views of the arena (the reference's gym tensors)
        self.root_states = t["ROOT_STATES"]
        self.dof_state = t["DOF_STATE"].view(self.num_envs * 12, 2)
        self.dof_pos, self.dof_vel = t["DOF_STATE"][..., 0], t["DOF_STATE"][..., 1]
        self.base_quat = self.root_states[:, 3:7]
        self.contact_forces, self.rigid_body_states = t["CONTACT_FORCES"], t["RIGID_BODY_STATE"]
        self.torques, self.torques_org, self.actions = t["TORQUES"], t["TORQUES_ORG"], t["ACTIONS"]
        self.last_actions, self.last_dof_vel = t["LAST_ACTIONS"], t["LAST_DOF_VEL"]
        self.last_torques_org, self.last_root_vel = t["LAST_TORQUES_ORG"], t["LAST_ROOT_VEL"]
        self.action_history_buf = t["ACTION_HISTORY"]
        self.motor_strength, self.mass_params_tensor, self.friction_coeffs_tensor = t["MOTOR_STRENGTH"], t["MASS_PARAMS"], t["FRICTION"]
        q0 = [cfg.init_state.default_joint_angles[n] for n in _capi.DOF_NAMES]
        self.default_dof_pos = torch.tensor(q0, dtype=torch.float, device=dev).unsqueeze(0)
        self.default_dof_pos_all = self.default_dof_pos.clone()
        names = _capi.BODY_NAMES
        self.feet_indices = torch.tensor([i for i, n in enumerate(names) if cfg.asset.foot_name in n], device=dev)
        self.penalised_contact_indices = torch.tensor([i for i, n in enumerate(names) if any(k in n for k in cfg.asset.penalize_contacts_on)], device=dev)
        self.termination_contact_indices = torch.tensor([i for i, n in enumerate(names) if any(k in n for k in cfg.asset.terminate_after_contacts_on)], device=dev)
        self.key_body_ids = torch.tensor([names.index(n) for n in KEY_BODY_NAMES], device=dev)
        self.category_mapping = {c: i for i, c in enumerate(cfg.env.mocap_category_all)}
        self.mocap_indices = torch.tensor([self.category_mapping[c] for c in cfg.env.mocap_category], device=dev)

        # ---- goal / reward / observation bookkeeping (three HIP kernels behind one object)
        self.bk = TaskLevelBookkeeping(cfg, self.env_goals, self.obstacle_types, self.x_edge_mask, self.feet_indices.tolist(),
                                       self.penalised_contact_indices.tolist(), self.termination_contact_indices.tolist(), self.num_bodies,
                                       device=dev, lib=bookkeeping_lib)
        ys = torch.tensor(cfg.obstacle.measured_points_y, device=dev)
        xs = torch.tensor(cfg.obstacle.measured_points_x, device=dev)
        gx, gy = torch.meshgrid(xs, ys, indexing="ij")                                     # _init_height_points (:1686-1706)
        self.num_height_points = gx.numel()
        self.height_points = torch.stack([gx.flatten(), gy.flatten(), torch.zeros_like(gx.flatten())], dim=1).contiguous()
        self.bk.init_observations(self.height_samples, self.height_points, self.default_dof_pos, self.default_dof_pos_all, self.key_body_ids.tolist())
        self.reward_scales = dict(self.bk.reward_scales)
        self.episode_sums = self.bk.episode_sums
        if cfg.depth.use_camera:
            # attach_camera (:1203-1226): one camera per env on the trunk, pitched down by U(depth.angle) degrees; the draw is keyed
            # by the GLOBAL env id like every other per-env constant, so a sharded job has the cameras of the one-process job
            lo, hi = cfg.depth.angle
            n_glob, off = int(self.qcfg.num_envs_global) or self.num_envs, int(self.qcfg.env_id_offset)
            ang = np.random.default_rng([seed, 0xCA3E]).uniform(lo, hi, n_glob)[off:off + self.num_envs]
            self.bk.init_depth(self.height_samples, torch.as_tensor(self.obstacle.ceiling_raw, dtype=torch.int16),
                               torch.as_tensor(np.radians(ang), dtype=torch.float32), seed=seed, env_id_offset=off)
        self.extras = {}
        self.global_counter = self.total_env_steps_counter = self.common_step_counter = 0
        # common_step_counter's twin on the device: keys the reset / depth-noise draws and gates the push, so that a RECORDED rollout
        # (the same launches replayed every iteration) draws fresh numbers and pushes on the right steps
        self._step_dev = torch.zeros(1, dtype=torch.int64, device=dev)
        self.cur_obst_idx = torch.zeros(self.num_envs, dtype=torch.long, device=dev)
        self._start_xy = torch.zeros(self.num_envs, 2, device=dev)
        self._start_yaw = torch.zeros(self.num_envs, device=dev)
        self._episode_means = torch.zeros(len(_capi.TSC_REWARD_NAMES), device=dev)
        self._obs_disc_term = torch.zeros(self.num_envs, self.num_obs_disc, device=dev)
        self._all = torch.ones(self.num_envs, dtype=torch.uint8, device=dev)
        self.sync_reset_ids = True
        self.want_delta_yaw_ok = True     # extras["delta_yaw_ok"] every step, as the reference does (a runner that never reads it may switch it off)
        self.init_done = True

        # reset_idx(all); _resample_commands(all) is the reference's U(range) draw, superseded by the first set_commands
        self._reset(self._all, first=True)
        self.post_physics_step(None)

    # ------------------------------------------------------------------ the attributes the runner / policy read live in the bookkeeping object
    for _name in ("commands", "latent_eps", "latent_c", "cur_goal_idx", "reach_goal_timer", "cur_goals", "next_goals", "target_yaw", "next_target_yaw",
                  "delta_yaw", "delta_next_yaw", "base_lin_vel", "base_ang_vel", "projected_gravity", "rew_buf", "obs_buf", "obs_bbc_buf",
                  "obs_disc_buf", "obs_history_buf", "measured_heights", "cur_obstacle_types", "time_out_buf", "reach_goal_cutoff"):
        locals()[_name] = property(lambda self, _n=_name: getattr(self.bk, _n))
    del _name

    @property
    def reset_buf(self):
        return self.bk.reset_buf

    @property
    def depth_buffer(self):
        return self.bk.depth_buffer

    @property
    def privileged_obs_buf(self):
        return None                       # num_privileged_obs is None in the reference's config: the critic sees obs_buf

    @property
    def episode_length_buf(self):
        return self.bk.episode_length_buf

    @episode_length_buf.setter
    def episode_length_buf(self, value):   # the runner rebinds it (on_policy_runner.py:162-163)
        self.bk.episode_length_buf.copy_(value)

    # ------------------------------------------------------------------ articulated obstacles (see-saw, bar jump, tyre jump)
    def _init_articulated_obstacles(self, seed):
        """QA_T_OBST_DESC / QA_T_OBST_STATE from the course (tsc/legged_gym/envs/base/legged_robot.py:1411-1427): per env slot 0 = see-saw
        (revolute about its y axis through the pivot 0.26 m up, plank 3.0 x 0.6 m, damping U(1,10) N m s/rad drawn per GLOBAL env id), slot 1 =
        the jump bar (1.2 x 0.2 m, prismatic, vertical), slot 2 = the tyre ring (0.8 x 0.2 m, prismatic).  Origins and yaws are the
        obstacle frames the generator placed (`obstacle_origins`, `obstacle_yaws`)."""
        self.articulated = bool(self.qcfg.articulated_obstacles)
        if not self.articulated:
            return
        ob, n, dev = self.obstacle, self.num_envs, self.device
        kinds = {"seesaw": (0, _capi.OBST_SEESAW, 1.5, 0.3, 0.26), "bar_jump": (1, _capi.OBST_BAR, 0.1, 0.6, 0.0), "tire_jump": (2, _capi.OBST_TYRE, 0.1, 0.4, 0.0)}
        desc = np.zeros((n, 3, 8), dtype=np.float32)
        for name, (slot, kind, hx, hy, h0) in kinds.items():
            t = ob.obst_types.index(name)
            for i in range(n):
                j = int(np.nonzero(ob.obstacle_types[i] == t)[0][0])
                yaw = ob.obstacle_yaws[i, j]
                desc[i, slot] = [ob.obstacle_origins[i, j, 0], ob.obstacle_origins[i, j, 1], np.cos(yaw), np.sin(yaw), hx, hy, h0, kind]
        n_glob, off = int(self.qcfg.num_envs_global) or n, int(self.qcfg.env_id_offset)
        damping = np.random.default_rng([seed, 0x5EE5]).uniform(1.0, 10.0, n_glob)[off:off + n]       # :1413 np.random.uniform(1, 10) per see-saw
        state = np.zeros((n, 3, 4), dtype=np.float32)
        state[:, 0, 0], state[:, 0, 3] = ob.seesaw_dof_pos, damping
        self.sim.t["OBST_DESC"].copy_(torch.from_numpy(desc))
        self.obst_state = self.sim.t["OBST_STATE"]
        self.obst_state.copy_(torch.from_numpy(state))
        # order of the see-saw along the env's course: a robot that starts behind it meets the plank from the far side (:812-823)
        self._seesaw_order = (self.obstacle_types == ob.obst_types.index("seesaw")).int().argmax(dim=1)

    def _reset_articulated_obstacles(self, flags, any_reset):
        """:812-823 without a host sync: the resetting envs' see-saw goes to its rest tilt on the side the robot will meet (randomize_start:
        a start past the see-saw flips it); their bar / tyre keep their offsets (the reference writes the see-saw's position only); every
        env's obstacle velocities are zeroed when anyone resets (`self.obst_dof_vel[:] = 0.0`)"""
        if not self.articulated:
            return
        st = self.obst_state
        f = flags != 0
        rest = torch.full((self.num_envs,), float(self.obstacle.seesaw_dof_pos), device=self.device)
        if self.cfg.obstacle.randomize_start:
            rest = torch.where(self.cur_obst_idx > self._seesaw_order, -rest, rest)
        st[:, 0, 0] = torch.where(f, rest, st[:, 0, 0])
        st[:, :, 1] *= (1.0 - any_reset.to(torch.float32))

    # ------------------------------------------------------------------ API
    def set_commands(self, actions):
        """:699-760.  The U(action_noise) multiplier comes from torch's generator like the reference's torch_rand_float."""
        noise = None
        if self.cfg.domain_rand.randomize_action:
            lo, hi = self.cfg.domain_rand.action_noise
            noise = torch.empty(self.num_envs, 5, device=self.device).uniform_(lo, hi)      # one launch, the same draws as rand() * (hi - lo) + lo
        return self.bk.set_commands(actions, noise)

    def step(self, actions, action_hl_history_buf=None):
        a = actions.to(device=self.device, dtype=torch.float32).contiguous()
        self.action_hl_history_buf = action_hl_history_buf
        delay = int(self.cfg.domain_rand.action_delay_step) if self.cfg.domain_rand.action_delay else 0
        self.global_counter += 1
        self.total_env_steps_counter += 1
        self.sim.physics_step(a, delay)
        reset_env_ids, terminal = self.post_physics_step(action_hl_history_buf)
        # (only the vision student reads it: the teacher's runner switches it off -- two launches per env step of its recorded rollout)
        self.extras["delta_yaw_ok"] = (torch.abs(self.bk.delta_yaw) < 0.6) if self.want_delta_yaw_ok else None
        if self.cfg.depth.use_camera and self.global_counter % self.cfg.depth.update_interval == 0:
            self.extras["depth"] = self.bk.depth_buffer[:, -2]            # :145-146
        else:
            self.extras["depth"] = None
        return self.bk.obs_buf, self.privileged_obs_buf, self.bk.rew_buf, self.bk.reset_buf, self.extras, reset_env_ids, terminal

    def post_physics_step(self, action_hl_history_buf):
        """:226-296"""
        bk, cfg = self.bk, self.cfg
        self.common_step_counter += 1
        if self._glue_kernels():
            # r4: the counter's increment and the push as ONE launch (qa_tsc_push; Philox draws keyed by the step instead of torch's generator)
            lib = _capi.load_library()
            if getattr(self, "_ticket", None) is None:
                self._ticket = torch.zeros(1, dtype=torch.int32, device=self.device)
            d = cfg.domain_rand
            rc = lib.qa_tsc_push(self.root_states.data_ptr(), self.num_envs, self._step_dev.data_ptr(), self._ticket.data_ptr(),
                                 int(d.push_interval) if d.push_robots else 0, float(d.max_push_vel_xy), int(self.sim.cfg.seed) + 104729, int(self.sim.cfg.env_id_offset),
                                 torch.cuda.current_stream(self.device).cuda_stream)
            if rc != 0:
                raise RuntimeError(f"qa_tsc_push failed with code {rc}: {lib.qa_last_error().decode()}")
        else:
            self._step_dev.add_(1)
            if cfg.domain_rand.push_robots:
                self._push_robots()        # before the goal step: its rewards read the pushed world velocity, as the reference's do (:643-644)
        bk.post_physics_step(self.root_states, self.contact_forces, self.rigid_body_states, action_hl_history_buf, want_ids=False)
        self.extras["reach_goal"] = bk.reach_goal_cutoff.view(torch.bool)
        flags = bk.reset_buf
        prev_disc = bk.obs_disc_buf.clone()                                  # get_observations_disc() of the envs about to reset (:263)
        self._reset(flags)
        upd = self.global_counter % cfg.depth.update_interval == 0
        if cfg.depth.use_camera and upd:
            bk.update_depth_buffer(self.root_states, self._step_dev)                 # :275, after the resets, before the observations
        bk.compute_observations(self.root_states, self.dof_pos, self.dof_vel, self.action_history_buf, self.rigid_body_states,
                                self.mass_params_tensor, self.friction_coeffs_tensor, self.motor_strength, update_yaw=upd)
        # (the goal-step kernel writes the flags as 0 / 1 bytes: reinterpreting them as bool is free, `!= 0` is a launch)
        torch.where(flags.view(torch.bool).view(-1, 1) if flags.dtype == torch.uint8 else flags.view(-1, 1) != 0, prev_disc, bk.obs_disc_buf, out=self._obs_disc_term)
        if self.sync_reset_ids:
            env_ids = flags.nonzero(as_tuple=False).flatten()               # host sync, like the reference
            return env_ids, prev_disc[env_ids]
        return None, None

    def _reset(self, flags, first=False):
        """reset_idx (:348-410) of the flagged envs without a host sync: start pose, simulator state, bookkeeping, episode statistics,
        then the reference's extra simulate (all envs, last torques) if anything was reset"""
        bk, cfg = self.bk, self.cfg
        ob = cfg.obstacle
        n_goal = cfg.obstacle.num_goals
        glue = self._glue_kernels() and flags.is_cuda
        if glue:
            lib = _capi.load_library()
            if getattr(self, "_start_goal", None) is None:
                self._start_goal = torch.zeros(self.num_envs, dtype=torch.long, device=self.device)
            rc = lib.qa_tsc_start_pose(flags.data_ptr(), self.cur_obst_idx.data_ptr(), self.env_goals.data_ptr(), self.obst_angs.data_ptr(), self.num_envs,
                                       int(self.env_goals.shape[1]), int(self.obstacle_types.shape[1]), int(n_goal), int(bool(ob.randomize_start)),
                                       float(self.obstacle.frame_ang[0]), int(self.sim.cfg.seed) + 104729, self._step_dev.data_ptr(), int(self.sim.cfg.env_id_offset),
                                       self._start_xy.data_ptr(), self._start_yaw.data_ptr(), self._start_goal.data_ptr(), torch.cuda.current_stream(self.device).cuda_stream)
            if rc != 0:
                raise RuntimeError(f"qa_tsc_start_pose failed with code {rc}: {lib.qa_last_error().decode()}")
            start_goal = self._start_goal
        elif ob.randomize_start:
            draw = torch.randint(0, self.obstacle_types.shape[1], (self.num_envs,), device=self.device)
            self.cur_obst_idx.copy_(torch.where(flags != 0, draw, self.cur_obst_idx))
            start_goal = self.cur_obst_idx * n_goal
            self._start_xy.copy_(self.env_goals.gather(1, start_goal[:, None, None].expand(-1, 1, 3)).squeeze(1)[:, :2])
            self._start_yaw.copy_(self.obst_angs.gather(1, self.cur_obst_idx[:, None]).squeeze(1))
        else:
            start_goal = torch.zeros(self.num_envs, dtype=torch.long, device=self.device)
            self._start_xy.copy_(self.env_goals[:, 0, :2])
            self._start_yaw.fill_(float(self.obstacle.frame_ang[0]))
        e = cfg.env
        self.sim.tsc_reset(flags, self._start_xy, self._start_yaw,
                           e.rand_yaw_range if e.randomize_start_yaw else 0.0, e.rand_x_range if e.randomize_start_x else 0.0,
                           e.rand_y_range if e.randomize_start_y else 0.0, e.rand_pitch_range if (e.randomize_start_yaw and e.randomize_start_pitch) else 0.0,
                           self._step_dev if self._step_dev.is_cuda else self.common_step_counter)
        # extras["episode"]: mean episode sums of the resetting envs / episode length in seconds (:396-404), kept when nobody resets
        if flags.is_cuda:
            # ONE launch, fixed-order sums (qa_tsc_reset_stats): the OR of the flags decides the extra simulate below (control, not logging)
            # and this code is replayed from a hipGraph -- no torch reduction in it (profiles/r2_hipgraph_stale_reductions.md)
            if getattr(self, "_any_reset", None) is None:
                self._any_reset = torch.zeros(1, dtype=torch.uint8, device=self.device)
            lib = _capi.load_library()
            rc = lib.qa_tsc_reset_stats(flags.data_ptr(), bk.episode_sums_buf.data_ptr(), self.num_envs, bk.episode_sums_buf.shape[0],
                                        float(self.max_episode_length_s), self._episode_means.data_ptr(), self._any_reset.data_ptr(),
                                        torch.cuda.current_stream(self.device).cuda_stream)
            if rc != 0:
                raise RuntimeError(f"qa_tsc_reset_stats failed with code {rc}: {lib.qa_last_error().decode()}")
            any_reset = self._any_reset
        else:
            f = (flags != 0).to(torch.float32)
            cnt = f.sum()
            mean = (bk.episode_sums_buf * f).sum(dim=1) / torch.clamp(cnt, min=1.0) / self.max_episode_length_s
            self._episode_means.copy_(torch.where(cnt > 0, mean, self._episode_means))
            any_reset = (cnt > 0).to(torch.uint8).reshape(1)
        snap = self._episode_means.clone()
        self.extras["episode"] = {"rew_" + n: snap[i] for i, n in enumerate(_capi.TSC_REWARD_NAMES)}
        if cfg.env.send_timeouts:
            self.extras["time_outs"] = bk.time_out_buf.view(torch.bool)
        if glue:
            rc = lib.qa_tsc_reset_where(flags.data_ptr(), any_reset.data_ptr(), start_goal.data_ptr(), bk.cur_goal_idx.data_ptr(), bk.reach_goal_timer.data_ptr(),
                                        bk.episode_sums_buf.data_ptr(), int(bk.episode_sums_buf.shape[0]), bk.episode_length_buf.data_ptr(), self.env_goals.data_ptr(),
                                        int(self.env_goals.shape[1]), bk.cur_goals.data_ptr(), bk.next_goals.data_ptr(),
                                        self.obst_state.data_ptr() if self.articulated else None, float(self.obstacle.seesaw_dof_pos) if self.articulated else 0.0,
                                        self.cur_obst_idx.data_ptr(), self._seesaw_order.data_ptr() if (self.articulated and ob.randomize_start) else None,
                                        self.num_envs, torch.cuda.current_stream(self.device).cuda_stream)
            if rc != 0:
                raise RuntimeError(f"qa_tsc_reset_where failed with code {rc}: {lib.qa_last_error().decode()}")
        else:
            bk.reset_where(flags, start_goal)
            self._reset_articulated_obstacles(flags, any_reset)
        if not first:
            self.sim.simulate_if(None, any_reset)
        else:
            self.sim.simulate_if(None, torch.ones(1, dtype=torch.uint8, device=self.device))

    def _glue_kernels(self):
        """r4: the torch glue between the step's kernels (counter + push, start pose of the resetting envs, reset bookkeeping + obstacle reset) as
        three launches of its own (qa_tsc_push / _start_pose / _reset_where) on the GPU; QA_TSC_GLUE=0 keeps the torch ops"""
        if getattr(self, "_glue", None) is None:
            import os
            self._glue = bool(self._step_dev.is_cuda and os.environ.get("QA_TSC_GLUE", "1") != "0" and self.env_goals.is_contiguous() and self.obst_angs.is_contiguous())
        return self._glue

    def _push_robots(self):
        """:905-915, on the steps where common_step_counter % push_interval == 0 -- decided on the device (no host branch: the launch
        sequence of a step is the same every step)"""
        m = self.cfg.domain_rand.max_push_vel_xy
        now = (self._step_dev % int(self.cfg.domain_rand.push_interval)) == 0
        push = (torch.rand(self.num_envs, 2, device=self.device) * 2 - 1) * m
        self.root_states[:, 7:9] = torch.where(now, push, self.root_states[:, 7:9])

    def get_observations(self):
        return self.bk.obs_buf

    def get_observations_bbc(self):
        return self.bk.obs_bbc_buf

    def get_observations_disc(self):
        return self.bk.obs_disc_buf

    def get_privileged_observations(self):
        return self.privileged_obs_buf

    def get_history_observations(self):
        return self.bk.obs_history_buf

    @property
    def obs_disc_term_buf(self):
        """obs_disc_buf with the rows of the envs that reset this step replaced by their terminal (pre-reset) observation"""
        return self._obs_disc_term

    def reset(self):
        self._reset(self._all)
        self.post_physics_step(None)
        return self.bk.obs_buf, self.privileged_obs_buf

    def set_camera(self, position, lookat):
        pass

    def render(self, sync_frame_time=True):
        pass
