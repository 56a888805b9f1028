"""Seam 1: the REFERENCE'S OWN `LeggedRobot` class -- its `_init_buffers`, `step`, `post_physics_step`, `reset_idx`,
unmodified, imported from /root/reference -- running on this engine through `quadrupedal_agility_amd.seam1.QaGym`, the stand-in
for the Isaac Gym objects the env calls.  Build container only (skipped where /root/reference is absent, i.e. on the GPU box);
the engine behind the shim is the CPU oracle's twin of the C ABI, injected as `lib=(oracle, "qo_")`.

What it shows: the acquire_* views have the shapes and aliasing the reference expects (root (N,13), dof (N*12,2), net contact
force (N*19,3), rigid body state (N*19,13) viewed as (N, num_bodies, 13)); one `gym.simulate` = one `qa_simulate`; the env's
in-place resets through the views reach the engine; and the reference's env, stepped this way, computes the same rewards /
resets / observations as this build's fused step from the same state and actions (the physics underneath is the same model)."""
import os
import sys
import types

import numpy as np
import pytest
import torch

REF = "/root/reference/bbc"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="needs the reference checkout (build container only)")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def ref():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    sys.dont_write_bytecode = True
    from gen_golden import import_reference
    # an earlier test may have installed this package's `legged_gym` / `rsl_rl` aliases (install_reference_aliases): the
    # REFERENCE's modules must be the ones imported here; the aliases come back when this module is done
    mine = lambda k: k.split(".")[0] in ("legged_gym", "rsl_rl")
    saved = {k: sys.modules.pop(k) for k in [k for k in sys.modules if mine(k)]}
    path = list(sys.path)
    ref_lr, RefCfg, _ = import_reference()
    assert ref_lr.__file__.startswith(REF), ref_lr.__file__
    yield ref_lr, RefCfg
    for k in [k for k in sys.modules if mine(k)]:
        del sys.modules[k]
    sys.modules.update(saved)
    sys.path[:] = path


def build_reference_env(ref_lr, RefCfg, n, seed=3):
    from quadrupedal_agility_amd import _capi, seam1
    from quadrupedal_agility_amd.legged_gym.utils.cfg_to_c import make_qa_config
    from tests.oracle_lib import load_oracle
    cfg = RefCfg()
    cfg.env.num_envs = n
    cfg.terrain.mesh_type = "plane"; cfg.terrain.measure_heights = False
    cfg.env.mocap_state_init = False
    cfg.noise.add_noise = False
    cfg.domain_rand.push_robots = False
    qcfg = make_qa_config(cfg, seed=seed)
    qcfg.export_body_state = 1
    gym = seam1.QaGym(qcfg, device="cpu", lib=(load_oracle(), "qo_"))
    ref_lr.gymtorch = seam1.gymtorch                          # `from isaacgym import gymtorch` of the reference module
    env = object.__new__(ref_lr.LeggedRobot)
    env.cfg, env.device, env.gym, env.sim = cfg, "cpu", gym, gym.sim_handle
    env.sim_params = types.SimpleNamespace(dt=cfg.sim.dt)
    env.viewer, env.debug_viz, env.enable_viewer_sync, env.headless = None, False, False, True
    env.num_envs, env.num_actions, env.num_dof, env.num_bodies = n, 12, 12, 19
    env.num_obs, env.num_privileged_obs = cfg.env.num_obs, cfg.env.num_privileged_obs
    env.mocap_category, env.mocap_category_all = cfg.env.mocap_category, cfg.env.mocap_category_all
    env.num_mocap, env.dim_c = 5, 5
    env.up_axis_idx = 2
    env.dof_names = list(_capi.DOF_NAMES)
    env._parse_cfg()
    names = _capi.BODY_NAMES
    env.feet_indices = torch.tensor([names.index(f"{l}_foot") for l in ("FL", "FR", "RL", "RR")])
    env.key_body_ids = env.feet_indices.clone()
    env.penalised_contact_indices = torch.tensor([i for i, b in enumerate(names) if "thigh" in b or "calf" in b])
    env.termination_contact_indices = torch.tensor([i for i, b in enumerate(names) if "base" in b or "hip" in b])
    env.hip_indices = torch.tensor([0, 3, 6, 9])
    # what create_sim / _create_envs leave behind (asset properties), from the engine's own tables
    lo = torch.tensor([-1.0472, -1.5708, -2.7227] * 2 + [-1.0472, -0.5236, -2.7227] * 2)
    hi = torch.tensor([1.0472, 3.4907, -0.83776] * 2 + [1.0472, 4.5379, -0.83776] * 2)
    m, r = (lo + hi) / 2, hi - lo
    env.dof_pos_limits = torch.stack([m - 0.5 * r * cfg.rewards.soft_dof_pos_limit, m + 0.5 * r * cfg.rewards.soft_dof_pos_limit], dim=1)
    env.dof_vel_limits, env.torque_limits = torch.tensor([30.1, 30.1, 20.07] * 4), torch.tensor([20.0, 20.0, 40.0] * 4)
    env.env_origins = gym.view("ENV_ORIGINS")
    env.mass_params_tensor, env.friction_coeffs_tensor = gym.view("MASS_PARAMS"), gym.view("FRICTION").unsqueeze(-1)
    env.base_init_state = torch.tensor(cfg.init_state.pos + cfg.init_state.rot + cfg.init_state.lin_vel + cfg.init_state.ang_vel)
    env.custom_origins = False
    # the BaseTask buffers (base_task.py:60-77)
    env.obs_buf = torch.zeros(n, 671); env.privileged_obs_buf = torch.zeros(n, 671); env.obs_disc_buf = torch.zeros(n, cfg.env.num_obs_disc)
    env.rew_buf = torch.zeros(n); env.reset_buf = torch.ones(n, dtype=torch.long)
    env.episode_length_buf = torch.zeros(n, dtype=torch.long); env.time_out_buf = torch.zeros(n, dtype=torch.bool)
    env.extras = {}
    env._init_buffers()                                       # <- the reference's own code on the shim's tensors
    env.motor_strength = gym.view("MOTOR_STRENGTH")           # the engine's domain randomisation drives both sides
    env._prepare_reward_function()
    env.init_done = True
    env.task_obs_weight, env.global_counter = 1.0, 0
    env.delay = 0
    return env, gym, qcfg


def test_init_buffers_binds_the_engine_tensors(ref):
    ref_lr, RefCfg = ref
    env, gym, q = build_reference_env(ref_lr, RefCfg, 6)
    assert env.root_states.shape == (6, 13) and env.dof_state.shape == (72, 2) and env.contact_forces.shape == (6, 19, 3)
    assert env.rigid_body_state.shape == (6 * 19, 13) and env.rigid_body_pos.shape == (6, 19, 3) and env.rigid_body_rot.shape == (6, 19, 4)
    # aliases, not copies: a write through the env's view is what the engine steps from
    env.root_states[2, 2] = 0.777
    assert gym.view("ROOT_STATES")[2, 2] == pytest.approx(0.777)
    env.dof_pos[1, 3] = 0.25
    assert gym.view("DOF_STATE")[1, 3, 0] == pytest.approx(0.25)
    assert env.root_states.data_ptr() == gym.view("ROOT_STATES").data_ptr()


def test_reference_env_steps_on_this_engine_and_agrees_with_the_fused_step(ref):
    ref_lr, RefCfg = ref
    from tests.oracle_lib import OracleSim
    n = 24
    env, gym, q = build_reference_env(ref_lr, RefCfg, n)
    torch.manual_seed(0)
    env.reset_idx(torch.arange(n))                            # the reference's reset, writing through the views
    assert (env.root_states[:, 2] - 0.42).abs().max() < 1e-6 and (env.dof_vel == 0).all()
    # this build's fused step on a second engine instance, started from the very same state / latents / commands
    q2 = type(q).from_buffer_copy(q); q2.export_body_state = 0
    fused = OracleSim(q2)
    fused.arena[:] = 0
    for name in ("ROOT_STATES", "DOF_STATE", "MOTOR_STRENGTH", "MASS_PARAMS", "FRICTION", "ENV_ORIGINS", "BASE_INERTIA", "PRIOR_PARAMETERS"):
        fused.t[name][...] = gym.view(name).numpy()
    fused.t["COMMANDS"][...] = env.commands.numpy(); fused.t["LATENT_EPS"][...] = env.latent_eps.numpy(); fused.t["LATENT_C"][...] = env.latent_c.numpy()
    rng = np.random.default_rng(1)
    resets = 0
    for k in range(30):
        act = rng.normal(0, 0.6, (n, 12)).astype(np.float32)
        obs, priv, rew, reset, extras, ids, term = env.step(torch.from_numpy(act))         # reference code: 4 x (torques, gym.simulate), post_physics_step
        fused.global_step = k
        fused.step(act)
        alive = (env.reset_buf == 0).numpy() & (fused.t["RESET"] == 0)
        assert np.array_equal(env.reset_buf.numpy() != 0, fused.t["RESET"] != 0)
        assert np.allclose(env.root_states.numpy()[alive], fused.t["ROOT_STATES"][alive], atol=2e-5)
        assert np.allclose(env.dof_pos.numpy()[alive], fused.t["DOF_STATE"][alive][:, :, 0], atol=2e-5)
        assert np.allclose(rew.numpy(), fused.t["REW"], atol=2e-5, rtol=1e-4)
        assert np.allclose(obs.numpy()[alive][:, :61], fused.t["OBS"][alive][:, :61], atol=5e-5)          # proprioception + explicit privileged
        assert np.allclose(env.contact_forces.numpy()[alive], fused.t["CONTACT_FORCES"][alive], atol=2e-3, rtol=1e-4)
        assert np.allclose(env.rigid_body_pos.numpy()[alive], fused.t["RIGID_BODY_POS"][alive], atol=2e-5)
        resets += int((env.reset_buf != 0).sum())
        # envs that reset drew different random poses on the two sides (torch vs Philox): re-synchronise them
        for name, src in (("ROOT_STATES", env.root_states), ("DOF_STATE", env.dof_state.view(n, 12, 2))):
            fused.t[name][...] = src.numpy()
        fused.t["COMMANDS"][...] = env.commands.numpy(); fused.t["LATENT_EPS"][...] = env.latent_eps.numpy(); fused.t["LATENT_C"][...] = env.latent_c.numpy()
        fused.t["FOOT_IMPULSE"][...] = gym.view("FOOT_IMPULSE").numpy()
        fused.t["EPISODE_LENGTH"][...] = env.episode_length_buf.numpy()
    assert torch.isfinite(obs).all() and obs.shape == (n, 671)
