#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by RUNNING THE REFERENCE'S OWN PYTHON on CPU.

Build container only (needs /root/reference); the GPU box only ever sees the .npz/.pt fixtures.
Recipe: put tools/ref_shims on sys.path (stand-ins for isaacgym, turtle,
pybullet_utils, tensorboard), put /root/reference/bbc on sys.path, import the reference modules,
drive them with synthetic tensors, dump inputs + the reference's outputs.

Fixtures:
  env_post_physics.npz   LeggedRobot.post_physics_step / check_termination / compute_reward /
                         reset_idx bookkeeping / compute_observations / compute_flat_key_pos
                         (legged_robot.py:124-331, 1231-1396) on states taken from real rollouts
  env_torques.npz        LeggedRobot._compute_torques (:547-579)
  gae.npz                RolloutStorage.compute_returns (rollout_storage.py:97-111)
  learner.pt             ActorCritic / Estimator / Discriminator forward, predict_disc_reward,
                         one update_actor_critic step and one update_ss_info_gail step (losses and
                         post-step weights) of SSInfoGAIL (gail.py:328-541)
  mocap.npz              MotionLoader.reorder and get_full_frame_at_time_batch on a labelled clip

Random draws: the reference mixes torch / numpy RNGs that cannot be matched by a counter-based
Philox.  For the env fixture the reference's sampling methods are patched to inject the values the
oracle drew (commands, latents, reset poses, pushes), so every DETERMINISTIC statement of the
reference -- including its stale-buffer semantics around resets -- is what the fixture pins.
"""
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/bbc"
GOLD = os.path.join(ROOT, "tests", "golden")
sys.dont_write_bytecode = True
sys.path.insert(0, ROOT)


def import_reference():
    sys.path.insert(0, os.path.join(ROOT, "tools", "ref_shims"))
    tb = types.ModuleType("torch.utils.tensorboard")
    tb.SummaryWriter = object
    sys.modules["torch.utils.tensorboard"] = tb
    sys.path.insert(0, REF)
    cwd = os.getcwd()
    os.chdir(os.path.join(REF, "legged_gym", "scripts"))          # the Go2 config globs mocap files relative to CWD
    import legged_gym.envs.base.legged_robot as ref_lr             # first, to avoid the circular import
    from legged_gym.envs.go2.go2_locomotion_config import Go2LocomotionCfg, Go2LocomotionCfgAlgo
    os.chdir(cwd)
    return ref_lr, Go2LocomotionCfg, Go2LocomotionCfgAlgo


BODY_NAMES = ["base", "Head_upper", "Head_lower"] + [f"{l}_{p}" for l in ("FL", "FR", "RL", "RR") for p in ("hip", "thigh", "calf", "foot")]


def make_ref_env(ref_lr, RefCfg, n, arena):
    """A reference LeggedRobot without Isaac Gym: attributes filled from an oracle arena (dict of numpy arrays)."""
    cfg = RefCfg()
    cfg.env.num_envs = n
    cfg.terrain.mesh_type = "plane"
    cfg.env.mocap_state_init = False
    cfg.noise.add_noise = False
    env = object.__new__(ref_lr.LeggedRobot)
    env.cfg = cfg
    env.device = "cpu"
    env.num_envs, env.num_actions, env.num_dof, env.num_bodies = n, 12, 12, 19
    env.sim_params = types.SimpleNamespace(dt=cfg.sim.dt)
    env.sim = None
    env.viewer = None
    env.debug_viz = False
    env.init_done = True
    env.enable_viewer_sync = False
    env.gym = types.SimpleNamespace(**{k: (lambda *a, **kw: None) for k in (
        "refresh_actor_root_state_tensor", "refresh_net_contact_force_tensor", "refresh_rigid_body_state_tensor",
        "refresh_dof_state_tensor", "set_actor_root_state_tensor", "set_dof_state_tensor_indexed",
        "set_actor_root_state_tensor_indexed")})
    env.mocap_category = cfg.env.mocap_category
    env.mocap_category_all = cfg.env.mocap_category_all
    env.num_mocap, env.dim_c = 5, 5
    env._parse_cfg()
    env.up_axis_idx = 2
    T = lambda x, dt=torch.float32: torch.tensor(np.array(x), dtype=dt)
    a = arena
    env.root_states = T(a["ROOT_STATES"])
    env.dof_state = T(a["DOF_STATE"]).reshape(n * 12, 2)
    env.dof_pos = env.dof_state.view(n, 12, 2)[..., 0]
    env.dof_vel = env.dof_state.view(n, 12, 2)[..., 1]
    env.base_quat = env.root_states[:, 3:7]
    env.contact_forces = T(a["CONTACT_FORCES"])
    rb = torch.zeros(n, 19, 13)
    rb[..., 0:3] = T(a["RIGID_BODY_POS"])
    env.rigid_body_state = rb.reshape(n * 19, 13)
    env.rigid_body_pos = env.rigid_body_state.view(n, 19, 13)[..., 0:3]
    env.common_step_counter = int(a["_step"])
    env.extras = {}
    env.gravity_vec = torch.tensor([0.0, 0.0, -1.0]).repeat(n, 1)
    env.forward_vec = torch.tensor([1.0, 0.0, 0.0]).repeat(n, 1)
    env.torques = T(a["TORQUES"]); env.torques_org = T(a["TORQUES_ORG"])
    env.p_gains = torch.full((12,), 40.0); env.d_gains = torch.full((12,), 1.0)
    env.actions = T(a["ACTIONS"]); env.last_actions = T(a["LAST_ACTIONS"])
    env.last_dof_vel = T(a["LAST_DOF_VEL"]); env.last_root_vel = T(a["LAST_ROOT_VEL"]); env.last_torques_org = T(a["LAST_TORQUES_ORG"])
    env.action_history_buf = T(a["ACTION_HISTORY"]); env.obs_history_buf = T(a["OBS"])[:, 90:660].reshape(n, 10, 57).clone()
    env.motor_strength = T(a["MOTOR_STRENGTH"])
    env.contact_buf = torch.zeros(n, 100, 4); env.contact_force_buf = torch.zeros(n, 100, 4)
    env.commands = T(a["COMMANDS"]); env.latent_eps = T(a["LATENT_EPS"]); env.latent_c = T(a["LATENT_C"])
    env.prior_parameters = T(a["PRIOR_PARAMETERS"]); env.prior_prob = torch.ones(5) / 5
    env.feet_air_time = torch.zeros(n, 4)
    env.last_contacts = T(a["LAST_CONTACTS"], torch.bool)
    env.base_lin_vel = torch.zeros(n, 3); env.base_ang_vel = torch.zeros(n, 3); env.projected_gravity = torch.zeros(n, 3)
    env.height_points = env._init_height_points()
    env.measured_heights = 0
    env.default_dof_pos = torch.tensor([0.0, 0.9, -1.8] * 4).unsqueeze(0)
    env.obs_buf = T(a["OBS"])[:, :] * 0
    env.obs_disc_buf = T(a["OBS_DISC"])
    env.rew_buf = torch.zeros(n)
    env.reset_buf = T(a["RESET"], torch.long)
    env.episode_length_buf = T(a["EPISODE_LENGTH"], torch.long)
    env.time_out_buf = torch.zeros(n, dtype=torch.bool)
    env.privileged_obs_buf = torch.zeros(n, 671)
    lo = torch.tensor([-1.0472, -1.5708, -2.7227] * 2 + [-1.0472, -0.5236, -2.7227] * 2)
    hi = torch.tensor([1.0472, 3.4907, -0.83776] * 2 + [1.0472, 4.5379, -0.83776] * 2)
    env.dof_pos_limits = torch.zeros(12, 2)
    for i in range(12):       # legged_robot.py:420-429 verbatim arithmetic (scalar fp32 tensor ops)
        env.dof_pos_limits[i, 0] = lo[i].item(); env.dof_pos_limits[i, 1] = hi[i].item()
        m = (env.dof_pos_limits[i, 0] + env.dof_pos_limits[i, 1]) / 2
        r = env.dof_pos_limits[i, 1] - env.dof_pos_limits[i, 0]
        env.dof_pos_limits[i, 0] = m - 0.5 * r * cfg.rewards.soft_dof_pos_limit
        env.dof_pos_limits[i, 1] = m + 0.5 * r * cfg.rewards.soft_dof_pos_limit
    env.dof_vel_limits = torch.tensor([30.1, 30.1, 20.07] * 4)
    env.torque_limits = torch.tensor([20.0, 20.0, 40.0] * 4)
    env.feet_indices = torch.tensor([BODY_NAMES.index(f"{l}_foot") for l in ("FL", "FR", "RL", "RR")])
    env.key_body_ids = env.feet_indices.clone()
    env.penalised_contact_indices = torch.tensor([i for i, b in enumerate(BODY_NAMES) if "thigh" in b or "calf" in b])
    env.termination_contact_indices = torch.tensor([i for i, b in enumerate(BODY_NAMES) if "base" in b or "hip" in b])
    env.hip_indices = torch.tensor([0, 3, 6, 9])
    env.mass_params_tensor = T(a["MASS_PARAMS"]); env.friction_coeffs_tensor = T(a["FRICTION"]).unsqueeze(-1)
    env.env_origins = T(a["ENV_ORIGINS"])
    env.base_init_state = torch.tensor(cfg.init_state.pos + cfg.init_state.rot + cfg.init_state.lin_vel + cfg.init_state.ang_vel)
    env.custom_origins = False
    env.noise_scale_vec = env._get_noise_scale_vec(cfg)
    env.add_noise = False
    env._prepare_reward_function()
    for i, name in enumerate(["action_rate", "collision", "delta_torques", "dof_acc", "dof_error", "dof_pos_limits", "dof_vel_limits",
                              "hip_pos", "jump_up_height", "locomotion_height", "torque_limits", "torques", "tracking_ang_vel",
                              "tracking_lin_vel"]):
        env.episode_sums[name] = T(a["EPISODE_SUMS"][i])
    env.task_obs_weight = 1.0
    env.global_counter = 0
    return env


def gen_env(ref_lr, RefCfg):
    from tests.oracle_lib import OracleSim, go2_cfg
    import ctypes as C
    n = 48
    q = go2_cfg(n, seed=11, add_noise=0)
    o = OracleSim(q)
    lib = o.lib
    lib.qo_debug_pre_physics.argtypes = [C.c_void_p, C.c_void_p, C.c_int32]
    lib.qo_debug_post_physics.argtypes = [C.c_void_p, C.c_int64]
    lib.qo_debug_torques.argtypes = [C.c_void_p] * 4
    rng = np.random.default_rng(11)
    o.reset_all()
    for _ in range(30):                               # settle into contact-rich states
        o.step(rng.normal(0, 0.6, (n, 12)).astype(np.float32))
    cases = []
    # case steps: 397..403 so that step+1 == 400 (push) is inside; episode lengths force time-outs and resampling
    o.global_step = 396
    o.t["EPISODE_LENGTH"][:] = rng.integers(5, 250, n)
    o.t["EPISODE_LENGTH"][0:6] = [999, 1000, 1001, 298, 299, 599]
    o.t["ROOT_STATES"][6, 2] = -7.0                   # fell out of the world -> time-out path
    for k in range(7):
        act = rng.normal(0, 1.5 if k % 2 else 0.5, (n, 12)).astype(np.float32)
        if k == 3:
            act[::3] *= 30                            # torque saturation / joint-limit terms
        step = int(o.global_step)
        assert lib.qo_debug_pre_physics(o.h, act.ctypes.data, 0) == 0
        mid = {k2: v.copy() for k2, v in o.t.items()}
        mid["_step"] = np.array(step)
        assert lib.qo_debug_post_physics(o.h, step) == 0
        out = {k2: v.copy() for k2, v in o.t.items()}
        o.global_step += 1
        # ---- the reference, with the oracle's random draws injected
        env = make_ref_env(ref_lr, RefCfg, n, mid)
        inj = {k2: torch.tensor(out[k2]) for k2 in ("COMMANDS", "LATENT_EPS", "LATENT_C", "ROOT_STATES", "DOF_STATE")}
        resampled, reset_ids = [], []

        def _cmd(self, env_ids):
            self.commands[env_ids] = inj["COMMANDS"][env_ids]; resampled.extend(env_ids.tolist())

        def _eps(self, env_ids):
            self.latent_eps[env_ids] = inj["LATENT_EPS"][env_ids]

        def _c(self, env_ids, temperature=0.25):
            self.latent_c[env_ids] = inj["LATENT_C"][env_ids]

        def _dofs(self, env_ids):
            d = inj["DOF_STATE"][env_ids]
            self.dof_pos[env_ids] = d[..., 0]; self.dof_vel[env_ids] = d[..., 1]; reset_ids.extend(env_ids.tolist())

        def _root(self, env_ids):
            self.root_states[env_ids] = inj["ROOT_STATES"][env_ids]

        def _push(self):
            self.root_states[:, 7:9] = inj["ROOT_STATES"][:, 7:9]
        for name, fn in (("_resample_commands", _cmd), ("_resample_latent_eps", _eps), ("_resample_latent_c", _c),
                         ("_reset_dofs", _dofs), ("_reset_root_states", _root), ("_push_robots", _push)):
            setattr(env, name, types.MethodType(fn, env))
        env_ids, terminal = env.post_physics_step()
        both = set(resampled) & set(reset_ids) - set(env_ids.tolist())
        periodic = [e for e in set(resampled) if e not in set(env_ids.tolist())]
        clash = [e for e in env_ids.tolist() if (int(mid["EPISODE_LENGTH"][e]) + 1) % 300 == 0]
        assert not clash, f"env {clash} resamples periodically and resets in the same step; change the seed"
        ref = {
            "obs": env.obs_buf.numpy(), "priv_obs": env.privileged_obs_buf.numpy(), "obs_disc": env.obs_disc_buf.numpy(),
            "obs_history": env.obs_history_buf.numpy(), "rew": env.rew_buf.numpy(), "reset": env.reset_buf.numpy(),
            "time_out": env.time_out_buf.numpy(), "episode_length": env.episode_length_buf.numpy(),
            "reset_env_ids": env_ids.numpy(), "terminal_disc": terminal.numpy(),
            "base_lin_vel": env.base_lin_vel.numpy(), "base_ang_vel": env.base_ang_vel.numpy(),
            "projected_gravity": env.projected_gravity.numpy(), "rpy": torch.stack([env.roll, env.pitch, env.yaw], 1).numpy(),
            "feet_force": env.feet_forces.numpy(), "contact_filt": env.contact_filt.numpy(), "last_contacts": env.last_contacts.numpy(),
            "last_actions": env.last_actions.numpy(), "last_dof_vel": env.last_dof_vel.numpy(), "last_root_vel": env.last_root_vel.numpy(),
            "last_torques_org": env.last_torques_org.numpy(), "action_history": env.action_history_buf.numpy(),
            "episode_sums": np.stack([env.episode_sums[k2].numpy() for k2 in env.reward_names]),
            "reward_names": np.array(env.reward_names),
            "extras_episode": np.array([float(env.extras["episode"]["rew_" + k2]) for k2 in env.reward_names]) if len(env_ids) else np.zeros(14),
            "commands": env.commands.numpy(), "root_states": env.root_states.numpy(),
        }
        case = {f"in_{k2}": v for k2, v in mid.items()}
        case.update({f"ref_{k2}": v for k2, v in ref.items()})
        cases.append(case)
        print(f"  env case step {step}: {len(env_ids)} resets, {len(periodic)} periodic resamples, push={(step + 1) % 400 == 0}")
    flat = {}
    for i, c in enumerate(cases):
        for k2, v in c.items():
            flat[f"c{i}_{k2}"] = v
    flat["num_cases"] = np.array(len(cases)); flat["num_envs"] = np.array(n); flat["seed"] = np.array(11)
    np.savez_compressed(os.path.join(GOLD, "env_post_physics.npz"), **flat)

    # ---- _compute_torques
    env = make_ref_env(ref_lr, RefCfg, n, mid)
    actions = rng.normal(0, 3.0, (n, 12)).astype(np.float32)
    tq = env._compute_torques(torch.tensor(actions).clone())
    np.savez_compressed(os.path.join(GOLD, "env_torques.npz"), actions=actions, dof_state=mid["DOF_STATE"],
                        motor_strength=mid["MOTOR_STRENGTH"], torques=tq.numpy(), torques_org=env.torques_org.numpy(), seed=np.array(11))


def gen_gae():
    from rsl_rl.storage.rollout_storage import RolloutStorage
    rng = np.random.default_rng(5)
    out = {}
    for i, (T, N) in enumerate([(24, 64), (24, 1000), (5, 3)]):
        st = RolloutStorage(N, T, [4], [4], [2], "cpu")
        st.rewards[:] = torch.tensor(rng.normal(0, 1, (T, N, 1)), dtype=torch.float32)
        st.values[:] = torch.tensor(rng.normal(0, 1, (T, N, 1)), dtype=torch.float32)
        st.dones[:] = torch.tensor(rng.random((T, N, 1)) < 0.08).byte()
        last = torch.tensor(rng.normal(0, 1, (N, 1)), dtype=torch.float32)
        st.compute_returns(last, 0.99, 0.95)
        out.update({f"c{i}_rewards": st.rewards.numpy(), f"c{i}_values": st.values.numpy(), f"c{i}_dones": st.dones.numpy(),
                    f"c{i}_last": last.numpy(), f"c{i}_returns": st.returns.numpy(), f"c{i}_advantages": st.advantages.numpy()})
    out["num_cases"] = np.array(3)
    np.savez_compressed(os.path.join(GOLD, "gae.npz"), **out)


def compact(sd, stride=97):
    """Post-step weights are pinned through a strided sample + two moments per tensor (keeps the fixture small)."""
    return {k: {"sample": v.flatten()[::stride].clone(), "sum": v.double().sum().item(), "abs_sum": v.double().abs().sum().item()}
            for k, v in sd.items()}


def gen_learner(RefCfg, RefAlgoCfg):
    """Reference networks + one PPO step + one discriminator step with everything seeded/fixed."""
    from legged_gym.utils.helpers import class_to_dict
    from rsl_rl.algorithms.discriminator import Discriminator
    from rsl_rl.algorithms.gail import SSInfoGAIL
    from rsl_rl.modules import ActorCritic, Estimator
    from rsl_rl.utils.utils import Normalizer
    torch.manual_seed(123)
    np.random.seed(123)
    cfg = RefCfg()
    tcfg = class_to_dict(RefAlgoCfg())
    env = types.SimpleNamespace(
        cfg=cfg, dim_c=5, num_obs_disc=49, num_envs=32, task_obs_weight_decay=True, task_obs_weight=0.7,
        latent_eps=torch.zeros(32, 1), latent_c=torch.zeros(32, 5), prior_parameters=torch.ones(5) / 5, dt=0.02)
    ac = ActorCritic(101, 671, 12, 57, 10, 4, 29, 11, **tcfg["policy"])
    est = Estimator(input_dim=57, output_dim=4, hidden_dims=tcfg["estimator"]["hidden_dims"])
    norm = Normalizer(98)
    norm.mean = np.random.normal(0, 0.3, 98); norm.var = np.random.uniform(0.5, 2.0, 98); norm.count = 1000.0
    disc = Discriminator(env, 98, 49, 5, 0.02, "MSELoss", None, 1.0, 0.01, 0.2, 0.2, 2, 2, 0.0, [512, 256], "cpu")
    init = {"actor_critic": {k: v.clone() for k, v in ac.state_dict().items()}, "estimator": {k: v.clone() for k, v in est.state_dict().items()},
            "disc": {k: v.clone() for k, v in disc.state_dict().items()}}
    B = 96
    g = torch.Generator().manual_seed(7)
    obs = torch.randn(B, 671, generator=g) * 0.5
    obs[:, 666:671] = torch.nn.functional.one_hot(torch.randint(0, 5, (B,), generator=g), 5).float()
    obs[:, 665] = torch.rand(B, generator=g) * 2 - 1
    actions = torch.randn(B, 12, generator=g)
    fx = {"init": init, "obs": obs, "actions": actions, "normalizer": {"mean": norm.mean.copy(), "var": norm.var.copy(), "count": norm.count},
          "policy_cfg": tcfg["policy"], "alg_cfg": tcfg["algorithm"], "estimator_cfg": tcfg["estimator"]}
    with torch.no_grad():
        fx["act_inference_priv"] = ac.act_inference(obs, hist_encoding=False)
        fx["act_inference_hist"] = ac.act_inference(obs, hist_encoding=True)
        fx["value"] = ac.evaluate(obs)
        ac.update_distribution(obs, False)
        fx["log_prob"] = ac.get_actions_log_prob(actions)
        fx["entropy"] = ac.entropy
        fx["estimator_out"] = est(obs[:, :57])
        fx["hist_latent"] = ac.infer_hist_latent(obs[:, 90:660])
        x = torch.randn(B, 98, generator=g)
        d, eps, c = disc(x)
        fx["disc_in"], fx["disc_d"], fx["disc_eps"], fx["disc_c"] = x, d, eps, c
        obs_disc = torch.randn(B, 2, 49, generator=g)
        reward_t = torch.rand(B, 1, generator=g) * 0.05
        r = disc.predict_disc_reward(reward_t, obs, obs_disc, normalizer=norm)
        fx["pdr_obs_disc"], fx["pdr_reward_t"], fx["pdr_out"] = obs_disc, reward_t, [t.clone() for t in r]
        disc.train()
    # ---- one update_actor_critic step
    ml = types.SimpleNamespace()
    alg = SSInfoGAIL(env, ac, disc, est, tcfg["estimator"], ml, norm, 2, 2, 49, 0.0, device="cpu", min_std=torch.full((12,), 0.05),
                     **tcfg["algorithm"])
    alg.priv_reg_counter = 1500                      # mid-ramp regulariser coefficient
    old_mu = torch.randn(B, 12, generator=g) * 0.2
    old_sigma = torch.ones(B, 12) * 0.9
    sample = (obs, obs, actions, torch.randn(B, 1, generator=g), torch.randn(B, 1, generator=g), torch.randn(B, 1, generator=g),
              torch.randn(B, 1, generator=g) - 12.0, old_mu, old_sigma, (None, None), None)
    fx["ppo_sample"] = [s.clone() if torch.is_tensor(s) else None for s in sample[:9]]
    losses = alg.update_actor_critic(sample)
    fx["ppo_losses"] = [l.detach().clone() if l.dim() == 0 else l.detach().mean().clone() for l in losses]
    fx["ppo_lr_after"] = alg.lr_ac
    fx["ppo_after"] = {"actor_critic": compact(ac.state_dict()), "estimator": compact(est.state_dict())}
    # ---- one update_ss_info_gail step
    pol = (torch.randn(B, 98, generator=g), torch.rand(B, 1, generator=g) * 2 - 1,
           torch.nn.functional.one_hot(torch.randint(0, 5, (B,), generator=g), 5).float())
    lb = (torch.randn(B, 98, generator=g), torch.randint(0, 5, (B,), generator=g))
    ulb = torch.randn(B, 98, generator=g)
    alg.info_max_coef_on = 0.3
    fx["disc_samples"] = {"policy": [t.clone() for t in pol], "lb": [t.clone() for t in lb], "ulb": ulb.clone()}
    dl = alg.update_ss_info_gail(pol, lb, ulb)
    fx["disc_losses"] = [l.detach().clone() for l in dl]
    fx["disc_after"] = compact(disc.state_dict())
    fx["prior_after"] = env.prior_parameters.clone()
    fx["std_after"] = ac.std.detach().clone()
    fx["normalizer_after"] = {"mean": norm.mean.copy(), "var": norm.var.copy(), "count": norm.count}
    fx["task_obs_weight"] = 0.7
    torch.save(fx, os.path.join(GOLD, "learner.pt"))


def gen_mocap():
    """The reference's reorder + batched frame blending on one labelled clip (input frames are a data file of the
    reference's dataset, truncated to 40 frames)."""
    import json
    from rsl_rl.datasets.motion_loader import MotionLoader
    path = os.path.join(REF, "mocap_data", "mocap_all_lb", "trot_1.json")
    js = json.load(open(path))
    raw = np.array(js["Frames"])[:40]
    clip = {"LoopMode": js["LoopMode"], "FrameDuration": js["FrameDuration"], "EnableCycleOffsetPosition": True,
            "EnableCycleOffsetRotation": True, "MotionWeight": js["MotionWeight"], "Frames": raw.tolist()}
    tmp = os.path.join(GOLD, "trot_clip40.json")
    json.dump(clip, open(tmp, "w"))
    np.random.seed(3)
    ml = MotionLoader("cpu", 0.02, mocap_state_init=True, motion_files_lb=[tmp], motion_files_ulb=[tmp],
                      mocap_category=["walk", "pace", "trot", "canter", "jump"])
    traj = np.zeros(64, dtype=np.int64)
    times = ml.traj_time_sample_batch(traj, labeled=True)
    frames = ml.get_full_frame_at_time_batch(traj, times, labeled=True)
    np.savez_compressed(os.path.join(GOLD, "mocap.npz"), reordered=ml.mocap_trajectory_full_lb[0].numpy(), times=times,
                        frames=frames.numpy(), frame_duration=np.array(js["FrameDuration"]))


def _main():
    os.makedirs(GOLD, exist_ok=True)
    ref_lr, RefCfg, RefAlgoCfg = import_reference()
    which = sys.argv[1:] or ["env", "gae", "learner", "mocap", "heights", "noise"]
    if "env" in which:
        gen_env(ref_lr, RefCfg)
    if "gae" in which:
        gen_gae()
    if "learner" in which:
        gen_learner(RefCfg, RefAlgoCfg)
    if "mocap" in which:
        gen_mocap()
    if "heights" in which:
        gen_heights(ref_lr, RefCfg)
    if "noise" in which:
        gen_noise(ref_lr, RefCfg)
    for f in sorted(os.listdir(GOLD)):
        print(f, os.path.getsize(os.path.join(GOLD, f)))


def gen_heights(ref_lr, RefCfg):
    """The reference's _get_heights (legged_robot.py:1190-1228) on a synthetic rough height field."""
    rng = np.random.default_rng(21)
    n, rows, cols = 64, 200, 180
    hs = (rng.integers(-40, 40, (rows, cols))).astype(np.int16)
    cfg = RefCfg(); cfg.env.num_envs = n; cfg.terrain.mesh_type = "trimesh"; cfg.terrain.border_size = 3.0
    env = object.__new__(ref_lr.LeggedRobot)
    env.cfg = cfg; env.device = "cpu"; env.num_envs = n
    env.terrain = types.SimpleNamespace(cfg=cfg.terrain)
    env.height_samples = torch.tensor(hs)
    env.height_points = env._init_height_points()
    root = np.zeros((n, 13), np.float32)
    root[:, 0] = rng.uniform(0.5, rows * 0.1 - 6.5, n); root[:, 1] = rng.uniform(0.5, cols * 0.1 - 6.5, n); root[:, 2] = 0.3
    q = rng.normal(size=(n, 4)); q /= np.linalg.norm(q, axis=1, keepdims=True); root[:, 3:7] = q
    env.root_states = torch.tensor(root); env.base_quat = env.root_states[:, 3:7]
    heights = env._get_heights()
    np.savez_compressed(os.path.join(GOLD, "heights.npz"), height_samples=hs, root_states=root, heights=heights.numpy(),
                        border=np.array(3.0), hscale=np.array(0.1), vscale=np.array(0.005))
    print("  heights golden:", heights.shape, float(heights.min()), float(heights.max()))


def gen_noise(ref_lr, RefCfg):
    """The reference's own _get_noise_scale_vec (legged_robot.py:721-740) for the Go2 config, and what one noisy
    compute_observations adds for injected uniforms: obs += (2u - 1) * noise_scale_vec (legged_robot.py:315-317)."""
    cfg = RefCfg()
    env = object.__new__(ref_lr.LeggedRobot)
    env.cfg = cfg; env.device = "cpu"
    env.obs_scales = cfg.normalization.obs_scales
    vec = env._get_noise_scale_vec(cfg)
    np.savez_compressed(os.path.join(GOLD, "noise_scale_vec.npz"), noise_scale_vec=vec.numpy(), add_noise=np.array(bool(env.add_noise)),
                        noise_level=np.array(cfg.noise.noise_level))
    print("  noise golden: non-zero entries", np.nonzero(vec.numpy())[0].tolist())


if __name__ == "__main__":
    _main()
