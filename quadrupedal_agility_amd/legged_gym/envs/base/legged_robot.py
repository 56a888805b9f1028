"""`LeggedRobot` -- host-side mirror of the reference env class for the Go2 BBC hot path.

Same constructor, methods and attributes as bbc/legged_gym/envs/base/legged_robot.py (the
learner/`train.py` seam, SURVEY.md 8b seam 2), but `step()` is ONE call into the HIP library
(`qa_env_step`, include/qa_sim.h) instead of ~300 eager torch launches around Isaac Gym.  All
buffers (`root_states`, `dof_pos`, `obs_buf`, ...) are zero-copy views of the engine's device
arena, exactly as the reference's are views of the gym tensors (legged_robot.py:757-770).

Nothing here computes physics, rewards or observations on the host: if the HIP library is not
built the constructor raises.
"""
import numpy as np
import ctypes as C

import torch

from quadrupedal_agility_amd import _capi
from quadrupedal_agility_amd.legged_gym.utils.cfg_to_c import class_to_dict, make_qa_config

KEY_BODY_NAMES = ["FL_foot", "FR_foot", "RL_foot", "RR_foot"]


class _SimParamsView:
    """The two fields of gymapi.SimParams this path reads (legged_robot.py:1139)."""

    def __init__(self, dt, use_gpu_pipeline=True):
        self.dt = dt
        self.use_gpu_pipeline = use_gpu_pipeline


class LeggedRobot:
    def __init__(self, cfg, sim_params=None, physics_engine=None, sim_device="cuda:0", headless=True, backend=None):
        self.cfg = cfg
        self.sim_params = sim_params if sim_params is not None and hasattr(sim_params, "dt") else _SimParamsView(cfg.sim.dt)
        self.physics_engine = physics_engine
        self.sim_device = sim_device
        self.headless = headless
        self.height_samples = None
        self.debug_viz = False
        self.init_done = False
        self.viewer = None
        self.mocap_category = self.cfg.env.mocap_category
        self.mocap_category_all = self.cfg.env.mocap_category_all
        self.num_mocap = len(self.mocap_category)
        self.dim_c = len(self.mocap_category_all)
        if self.num_mocap != self.dim_c:
            raise NotImplementedError("single-gait training (_set_latent_c, legged_robot.py:542-545) is not on the hot path")
        self._parse_cfg()

        self.num_envs = cfg.env.num_envs
        self.num_obs = cfg.env.num_obs
        self.num_obs_disc = cfg.env.num_obs_disc
        self.num_privileged_obs = cfg.env.num_privileged_obs
        self.num_actions = cfg.env.num_actions
        self.include_history_steps = cfg.env.include_history_steps
        if self.include_history_steps is not None:
            raise NotImplementedError("include_history_steps is None in the Go2 config (go2_locomotion_config.py:11)")
        if cfg.env.num_prop != _capi.NUM_PROP or cfg.env.history_len != _capi.HISTORY_LEN or self.num_obs_disc != _capi.NUM_OBS_DISC:
            raise NotImplementedError("observation layout is fixed by the kernels: 57 prop, 10 history frames, 49 disc")

        # ---- engine
        self.terrain = None
        if cfg.terrain.mesh_type in ("heightfield", "trimesh"):
            from quadrupedal_agility_amd.legged_gym.utils.terrain import Terrain
            self.terrain = Terrain(cfg.terrain, cfg.env.num_envs)          # legged_robot.py:353-355
        self.qcfg = make_qa_config(cfg, seed=getattr(cfg, "seed", 1), sim_dt=self.sim_params.dt, terrain=self.terrain)
        self._mocap_table = None
        if cfg.env.mocap_state_init:
            self._mocap_table = self._build_mocap_reset_table()
            self.qcfg.num_mocap_frames = int(self._mocap_table[0].shape[0])
        if backend is None:
            from quadrupedal_agility_amd.sim import QaSim
            backend = QaSim(self.qcfg, sim_device)
        elif isinstance(backend, type) or (callable(backend) and not hasattr(backend, "t")):
            backend = backend(self.qcfg)          # a factory: the final qa_config (mocap table size, terrain) is only known here
        self.sim = backend
        self.device = str(backend.device)
        if self._mocap_table is not None:
            self.sim.set_mocap(*self._mocap_table)
        self._init_buffers()
        self._install_terrain()
        self._prepare_reward_function()
        self.init_done = True

        self.key_body_ids = torch.tensor([_capi.BODY_NAMES.index(n) for n in KEY_BODY_NAMES], device=self.device)
        self.task_obs_weight_decay = self.cfg.normalization.task_obs_weight_decay
        self.task_obs_weight_decay_steps = self.cfg.normalization.task_obs_weight_decay_steps
        self.task_obs_weight = 1.0
        self.global_counter = 0
        self.common_step_counter = 0
        self.delay = 0
        self._delay_schedule = list(getattr(cfg.domain_rand, "action_curr_step", []))
        self.sync_reset_ids = True      # False: step() returns (None, None) for the last two outputs, no host sync
        self._step_ctr = None           # device-side step counter (graph-replayable stepping), see use_device_step_counter()

    # ------------------------------------------------------------------ config
    def _parse_cfg(self):
        self.dt = self.cfg.control.decimation * self.sim_params.dt
        self.obs_scales = self.cfg.normalization.obs_scales
        self.reward_scales = class_to_dict(self.cfg.rewards.scales)
        self.jump_goal_rwd = self.cfg.rewards.jump_goal
        self.command_ranges = class_to_dict(self.cfg.commands.ranges)
        self.command_curriculum_step = self.cfg.commands.curriculum_step
        if self.cfg.terrain.mesh_type not in ["heightfield", "trimesh"]:
            self.cfg.terrain.curriculum = False
        self.max_episode_length_s = self.cfg.env.episode_length_s
        self.max_episode_length = np.ceil(self.max_episode_length_s / self.dt)
        self.cfg.domain_rand.push_interval = np.ceil(self.cfg.domain_rand.push_interval_s / self.dt)

    def _prepare_reward_function(self):
        """Drop zero scales, multiply the rest by dt (legged_robot.py:922-946).  The terms themselves live in
        the kernel; this keeps `reward_scales` / `episode_sums` / `reward_names` as the runner's logger reads them."""
        for key in list(self.reward_scales.keys()):
            if self.reward_scales[key] == 0:
                self.reward_scales.pop(key)
            else:
                self.reward_scales[key] *= self.dt
        self.reward_names = [n for n in self.reward_scales if n != "termination"]
        assert self.reward_names == [n for n in _capi.REWARD_NAMES if n in self.reward_scales], self.reward_names
        sums = self.sim.t["EPISODE_SUMS"]
        self.episode_sums = {n: sums[_capi.REWARD_NAMES.index(n)] for n in self.reward_names}

    # ------------------------------------------------------------------ buffers = views of the arena
    def _init_buffers(self):
        t = self.sim.t
        self.root_states = t["ROOT_STATES"]
        self.dof_state = t["DOF_STATE"].view(self.num_envs * 12, 2)
        self.dof_pos = t["DOF_STATE"][..., 0]
        self.dof_vel = t["DOF_STATE"][..., 1]
        self.base_quat = self.root_states[:, 3:7]
        self.contact_forces = t["CONTACT_FORCES"]
        self.rigid_body_pos = t["RIGID_BODY_POS"]
        self.num_dof = 12
        self.num_bodies = len(_capi.BODY_NAMES)
        self.dof_names = list(_capi.DOF_NAMES)
        self.obs_buf = t["OBS"]
        self.privileged_obs_buf = t["OBS"]          # same values in the reference (legged_robot.py:321)
        self.obs_disc_buf = t["OBS_DISC"]
        self.obs_disc_term_buf = t["OBS_DISC_TERM"]
        self.rew_buf = t["REW"]
        self.reset_buf = t["RESET"]
        self.time_out_buf = t["TIME_OUT"].view(torch.bool)
        self._episode_length = t["EPISODE_LENGTH"]
        self.torques = t["TORQUES"]
        self.torques_org = t["TORQUES_ORG"]
        self.actions = t["ACTIONS"]
        self.last_actions = t["LAST_ACTIONS"]
        self.last_dof_vel = t["LAST_DOF_VEL"]
        self.last_root_vel = t["LAST_ROOT_VEL"]
        self.last_torques_org = t["LAST_TORQUES_ORG"]
        self.action_history_buf = t["ACTION_HISTORY"]
        self.obs_history_buf = t["OBS"][:, 90:660].unflatten(1, (_capi.HISTORY_LEN, _capi.NUM_PROP))   # (N,10,57) view into the obs rows
        self.commands = t["COMMANDS"]
        self.latent_eps = t["LATENT_EPS"]
        self.latent_c = t["LATENT_C"]
        self._prior_parameters = t["PRIOR_PARAMETERS"]
        self.base_lin_vel = t["BASE_LIN_VEL"]
        self.base_ang_vel = t["BASE_ANG_VEL"]
        self.projected_gravity = t["PROJECTED_GRAVITY"]
        self.feet_forces = t["FEET_FORCE"]
        self.contact_filt = t["CONTACT_FILT"].view(torch.bool)
        self.last_contacts = t["LAST_CONTACTS"].view(torch.bool)
        self.motor_strength = t["MOTOR_STRENGTH"]
        self.mass_params_tensor = t["MASS_PARAMS"]
        self.friction_coeffs_tensor = t["FRICTION"]
        self.env_origins = t["ENV_ORIGINS"]
        self.extras = {}
        dev = self.device
        q0 = [self.cfg.init_state.default_joint_angles[n] for n in _capi.DOF_NAMES]
        self.default_dof_pos = torch.tensor(q0, dtype=torch.float, device=dev).unsqueeze(0)
        self.p_gains = torch.full((12,), float(self.qcfg.kp), device=dev)
        self.d_gains = torch.full((12,), float(self.qcfg.kd), device=dev)
        # URDF limits (legged_robot.py:403-430), soft position limits
        lo = torch.tensor([-1.0472, -1.5708, -2.7227] * 2 + [-1.0472, -0.5236, -2.7227] * 2, device=dev)
        hi = torch.tensor([1.0472, 3.4907, -0.83776] * 2 + [1.0472, 4.5379, -0.83776] * 2, device=dev)
        m, r = (lo + hi) / 2, hi - lo
        s = self.cfg.rewards.soft_dof_pos_limit
        self.dof_pos_limits = torch.stack([m - 0.5 * r * s, m + 0.5 * r * s], dim=1)
        self.dof_vel_limits = torch.tensor([30.1, 30.1, 20.07] * 4, device=dev)
        self.torque_limits = torch.tensor([20.0, 20.0, 40.0] * 4, device=dev)
        self.feet_indices = torch.tensor([_capi.BODY_NAMES.index(n) for n in KEY_BODY_NAMES], device=dev)
        self.penalised_contact_indices = torch.tensor(
            [i for i, n in enumerate(_capi.BODY_NAMES) if any(k in n for k in self.cfg.asset.penalize_contacts_on)], device=dev)
        self.termination_contact_indices = torch.tensor(
            [i for i, n in enumerate(_capi.BODY_NAMES) if any(k in n for k in self.cfg.asset.terminate_after_contacts_on)], device=dev)
        self.hip_indices = torch.tensor([0, 3, 6, 9], device=dev)
        self.thigh_indices = torch.tensor([1, 4, 7, 10], device=dev)
        self.calf_indices = torch.tensor([2, 5, 8, 11], device=dev)
        self.prior_prob = torch.ones(self.dim_c, device=dev) / self.dim_c
        self.add_noise = self.cfg.noise.add_noise
        self.noise_scale_vec = self._get_noise_scale_vec(self.cfg)
        self._episode_means = torch.zeros(_capi.NUM_REWARDS, device=dev)

    def _get_noise_scale_vec(self, cfg):
        v = torch.zeros(cfg.env.num_obs + cfg.env.history_len * cfg.env.num_prop, device=self.device)
        q = self.qcfg
        v[:2] = q.noise_roll_pitch; v[2:5] = q.noise_ang_vel; v[5:17] = q.noise_dof_pos; v[17:29] = q.noise_dof_vel
        v[58:61] = q.noise_lin_vel
        return v

    # ------------------------------------------------------------------ terrain (legged_robot.py:958-993, 1109-1125, 1174-1228)
    def _install_terrain(self):
        self.custom_origins = self.terrain is not None
        dev = self.device
        y = torch.tensor(self.cfg.terrain.measured_points_y, device=dev)
        x = torch.tensor(self.cfg.terrain.measured_points_x, device=dev)
        gx, gy = torch.meshgrid(x, y, indexing="ij")
        self.num_height_points = gx.numel()
        self.height_points = torch.zeros(self.num_envs, self.num_height_points, 3, device=dev)
        self.height_points[:, :, 0] = gx.flatten()
        self.height_points[:, :, 1] = gy.flatten()
        if self.terrain is None:
            return
        t = self.terrain
        self.height_samples = torch.tensor(t.heightsamples).view(t.tot_rows, t.tot_cols).to(dev)
        self.sim.t["HEIGHT_SAMPLES"].copy_(self.height_samples)                 # gym.add_heightfield / add_triangle_mesh
        max_init_level = self.cfg.terrain.max_init_terrain_level if self.cfg.terrain.curriculum else self.cfg.terrain.num_rows - 1
        self.terrain_levels = torch.randint(0, max_init_level + 1, (self.num_envs,), device=dev)
        self.terrain_types = torch.div(torch.arange(self.num_envs, device=dev), (self.num_envs / self.cfg.terrain.num_cols),
                                       rounding_mode="floor").to(torch.long)
        self.max_terrain_level = self.cfg.terrain.num_rows
        self.terrain_origins = torch.from_numpy(t.env_origins).to(dev).to(torch.float)
        self.env_origins.copy_(self.terrain_origins[self.terrain_levels, self.terrain_types])

    def _get_heights(self, env_ids=None):
        """The reference's 187-point height scan (legged_robot.py:1190-1228) for callers that want all of it; the
        kernel itself only evaluates the one sample the BBC observation/reward use (SCAN_HEIGHT)."""
        from quadrupedal_agility_amd.legged_gym.utils.torch_jit_utils import quat_apply_yaw
        if self.terrain is None:
            return torch.zeros(self.num_envs, self.num_height_points, device=self.device)
        pts = quat_apply_yaw(self.base_quat.repeat(1, self.num_height_points), self.height_points) + self.root_states[:, :3].unsqueeze(1)
        pts = ((pts + self.cfg.terrain.border_size) / self.cfg.terrain.horizontal_scale).long()
        px = torch.clip(pts[:, :, 0].reshape(-1), 0, self.height_samples.shape[0] - 2)
        py = torch.clip(pts[:, :, 1].reshape(-1), 0, self.height_samples.shape[1] - 2)
        h = torch.min(torch.min(self.height_samples[px, py], self.height_samples[px + 1, py]), self.height_samples[px, py + 1])
        return h.view(self.num_envs, -1) * self.cfg.terrain.vertical_scale

    @property
    def measured_heights(self):
        return self._get_heights() if self.cfg.terrain.measure_heights else 0

    # attributes the learner REBINDS (on_policy_runner.py:123-125, gail.py:463-464): write through to the arena
    @property
    def episode_length_buf(self):
        return self._episode_length

    @episode_length_buf.setter
    def episode_length_buf(self, value):
        self._episode_length.copy_(value)

    @property
    def prior_parameters(self):
        return self._prior_parameters

    @prior_parameters.setter
    def prior_parameters(self, value):
        self._prior_parameters.copy_(value)

    # ------------------------------------------------------------------ mocap reset table
    def _build_mocap_reset_table(self):
        from quadrupedal_agility_amd.rsl_rl.datasets.motion_loader import MotionLoader
        loader = MotionLoader(device="cpu", motion_files_lb=self.cfg.env.motion_files_lb, motion_files_ulb=[],
                              mocap_category=self.mocap_category, time_between_frames=self.dt, mocap_state_init=True)
        self.mocap_source = loader.source
        return loader.reset_clip_table()

    # ------------------------------------------------------------------ API
    def reset(self):
        """reset_idx(all) then one zero-action step (legged_robot.py:67-76)."""
        self.sim.global_step = self.common_step_counter
        self.sim.reset_all()
        obs, priv, *_ = self.step(torch.zeros(self.num_envs, self.num_actions, device=self.device))
        return obs, priv

    def step(self, actions):
        """legged_robot.py:78-115 as one device launch."""
        if self.cfg.domain_rand.action_delay:
            if self.global_counter % self.cfg.domain_rand.delay_update_global_steps == 0 and len(self._delay_schedule):
                self.delay = int(self._delay_schedule.pop(0))
            delay = self.delay
        else:
            delay = 0
        self.global_counter += 1
        a = actions.to(device=self.device, dtype=torch.float32).contiguous()
        if self._step_ctr is not None:
            self.sim.step_dev(a, delay, self._step_ctr)       # counter read + incremented on the device
        else:
            self.sim.global_step = self.common_step_counter
            self.sim.step(a, delay)
        self.common_step_counter += 1
        self._fill_extras()
        if self.sync_reset_ids:
            env_ids = self.reset_buf.nonzero(as_tuple=False).flatten()          # host sync, like the reference
            terminal = self.obs_disc_term_buf[env_ids]
        else:
            env_ids, terminal = None, None
        return self.obs_buf, self.privileged_obs_buf, self.rew_buf, self.reset_buf, self.extras, env_ids, terminal

    def set_lean_exports(self, mask):
        """Training mode of the fused step (qa_set_lean_exports, include/qa_sim.h): 1 = the tensors only seam 1 / play / logging read (contact_forces,
        rigid_body_pos, torques, base_lin_vel, ..., all but the two newest action-history slots) are no longer refreshed; 3 = the discriminator
        observations neither (no AMP).  0 restores the reference's behaviour.  The runner switches it on for learn(); everything the learner
        consumes is bit-identical either way (tests/test_full_size_properties.py)."""
        self.sim.set_lean_exports(mask)

    def use_device_step_counter(self):
        """Keep `common_step_counter` ALSO in device memory so that recorded (hipGraph) rollouts can replay without
        host arguments; the host integer stays the authority for logging and is advanced by the caller on replay."""
        if self._step_ctr is None:
            self._step_ctr = torch.tensor([self.common_step_counter], dtype=torch.int64, device=self.device)
        return self._step_ctr

    def advance_host_counters(self, n):
        self.common_step_counter += n
        self.global_counter += n

    def steps_until_delay_change(self):
        """Number of steps that can be taken before the action-delay schedule fires again (legged_robot.py:87-93)."""
        if not self.cfg.domain_rand.action_delay or not self._delay_schedule:
            return 1 << 62
        period = self.cfg.domain_rand.delay_update_global_steps
        return (-self.global_counter) % period if self.global_counter % period else 0

    def _fill_extras(self):
        """extras['episode'] / extras['time_outs'] of reset_idx (legged_robot.py:229-240) without a host sync:
        means over the envs that reset this step; when none reset the previous values are kept, like the
        reference keeps the dict of the last reset."""
        # in place: `_episode_means` is a persistent buffer.  A recorded rollout bakes the ADDRESS of whatever tensor it
        # read first; rebinding the attribute to a fresh tensor every step let the pre-capture one be freed and reused,
        # and replays then read foreign data whenever no env reset in their first steps.
        stats = self.sim.t["EPISODE_STATS"]
        if stats.is_cuda and self._episode_means.is_contiguous():
            # ONE launch (qa_episode_means; r4) for what used to be ten eager ones per env step inside the recorded rollout.  Recorded
            # rollouts: the bin index follows the DEVICE step counter (a host parity baked into a recording is only right when
            # num_steps_per_env is even); the kernel accumulated into bin (step & 1) with step = counter - 1
            snap = torch.empty_like(self._episode_means)      # per-step values for the runner's per-iteration mean over the 24 steps
            lib = _capi.load_library()
            P = lambda t: C.c_void_p(t.data_ptr())
            rc = lib.qa_episode_means(P(stats), P(self._step_ctr) if self._step_ctr is not None else None, int(self.common_step_counter), _capi.NUM_REWARDS,
                                      float(self.max_episode_length_s), P(self._episode_means), P(snap), C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream))
            if rc != 0:
                raise RuntimeError(f"qa_episode_means failed with code {rc}: {lib.qa_last_error().decode()}")
        else:
            if self._step_ctr is not None:
                st = stats.index_select(0, (self._step_ctr - 1) & 1)[0]
            else:
                st = stats[(self.common_step_counter - 1) & 1]
            cnt = st[14]
            mean = st[:_capi.NUM_REWARDS] / torch.clamp(cnt, min=1.0) / self.max_episode_length_s
            self._episode_means.copy_(torch.where(cnt > 0, mean, self._episode_means))
            snap = self._episode_means.clone()
        self.extras["episode"] = {"rew_" + n: snap[_capi.REWARD_NAMES.index(n)] for n in self.reward_names}
        if self.cfg.env.send_timeouts:
            self.extras["time_outs"] = self.time_out_buf

    def get_observations(self):
        return self.obs_buf

    def get_privileged_observations(self):
        return self.privileged_obs_buf

    def get_disc_observations(self):
        return self.obs_disc_buf

    def set_camera(self, position, lookat):
        pass    # no viewer in this build (UI is out of scope)

    def render(self, sync_frame_time=True):
        pass
