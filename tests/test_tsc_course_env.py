"""The task-level (TSC) agility-course env and its two-level rollout (SURVEY 8f row 1 / BASELINE config 4).

CPU: `LeggedRobot` (tsc mirror) on the oracle engine + the oracle's twins of the three task-level kernels -- the same host code
the GPU runs, driven through `set_commands -> frozen behaviour policy -> step` and through `OnPolicyRunner.learn_RL`.
GPU (-m gpu): the same pipeline on the HIP library against the oracle pipeline from identical arenas, 8192 envs as the tiled
small case (envs are independent), and a short training run."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

from quadrupedal_agility_amd import _capi
from quadrupedal_agility_amd.legged_gym.utils.cfg_to_c import class_to_dict
from quadrupedal_agility_amd.tsc.legged_gym.envs.base import legged_robot as lr
from quadrupedal_agility_amd.tsc.legged_gym.envs.go2.go2_agility_config import Go2AgilityCfg, Go2AgilityCfgPPO
from quadrupedal_agility_amd.tsc.legged_gym.utils.obstacle import Obstacle
from tests.oracle_backend import OracleBackend
from tests.oracle_lib import load_oracle


def make_cfg(n, seed=3, **flags):
    cfg = Go2AgilityCfg()
    cfg.env.num_envs, cfg.seed = n, seed
    for k, v in flags.items():
        sect, name = k.split("__")
        setattr(getattr(cfg, sect), name, v)
    return cfg


def cpu_env(n, seed=3, **flags):
    cfg = make_cfg(n, seed, **flags)
    ob = Obstacle(cfg.obstacle, n, seed=seed)
    be = OracleBackend(lr.make_qa_config(cfg, ob, seed=seed))
    return lr.LeggedRobot(cfg, backend=be, bookkeeping_lib=(load_oracle(), "qo_"))


def test_env_surface_and_first_observation():
    env = cpu_env(9)
    assert env.num_obs == 800 and env.num_obs_bbc == 101 and env.num_obs_disc == 49 and env.num_actions == 12
    assert env.get_observations().shape == (9, 800) and env.get_observations_bbc().shape == (9, 671) and env.get_observations_disc().shape == (9, 49)
    assert env.env_goals.shape == (9, 26, 3) and env.obstacle_types.shape == (9, 6) and env.height_samples.dtype == torch.int16
    assert (env.episode_length_buf == 1).all() and env.common_step_counter == 1      # the constructor's reset + post_physics_step (:101-103)
    # robots start on their first goal, facing the first frame's direction (+y), within the start randomisation
    d = env.root_states[:, :2] - env.env_goals[:, 0, :2]
    assert (d[:, 0] <= 1e-6).all() and (d[:, 0] >= -0.2 - 1e-6).all() and (d[:, 1].abs() <= 0.1 + 1e-6).all()
    yaw = 2 * torch.atan2(env.root_states[:, 5], env.root_states[:, 6])
    assert ((yaw - np.pi / 2).abs() <= 0.2 + 1e-5).all()
    assert torch.isfinite(env.get_observations()).all()
    # the scan in the observation row sees the course: clip(z - 0.3 - h, -1, 1) * 1 differs between envs
    assert env.get_observations()[:, 65:197].std() > 0 or env.measured_heights.abs().sum() >= 0


def test_two_level_stepping_walks_and_resets():
    torch.manual_seed(0)
    n = 16
    env = cpu_env(n, env__episode_length_s=0.6)                      # 30-step episodes: time-outs inside the test
    hist = torch.zeros(n, 8, 19)
    resets = 0
    z_min = 1.0
    for k in range(70):
        a = torch.zeros(n, 19); a[:, 0] = torch.randint(0, 3, (n,)).float(); a[:, 1:] = torch.rand(n, 18) * 2 - 1
        hist = torch.cat([hist[:, 1:], a[:, None]], 1)
        nxt = env.set_commands(a)
        assert nxt.shape == (n, 11)
        obs, priv, rew, done, extras, ids, term = env.step(torch.zeros(n, 12), hist)
        assert priv is None and obs.shape == (n, 800) and torch.isfinite(obs).all() and torch.isfinite(rew).all()
        assert len(ids) == int((done != 0).sum()) and term.shape == (len(ids), 49)
        resets += len(ids)
        if len(ids):
            assert (env.episode_length_buf[ids] == 0).all() and (env.cur_goal_idx[ids] == 0).all()
            assert torch.equal(env.obs_disc_term_buf[ids], term)
        z_min = min(z_min, float(env.root_states[:, 2].min()))
    assert resets >= n                                             # every env timed out at least once (31 > 30 steps)
    assert z_min > 0.1                                             # standing on the course, not falling through it
    assert set(extras) >= {"episode", "time_outs", "reach_goal", "delta_yaw_ok", "depth"}
    assert set(extras["episode"]) == {"rew_" + k for k in _capi.TSC_REWARD_NAMES}


def test_course_is_the_collision_terrain():
    """a robot dropped onto an A-frame / pole comes to rest ON the obstacle surface, not on z = 0"""
    env = cpu_env(4, seed=5)
    ob = env.obstacle
    for e in range(4):
        j = int(np.nonzero(ob.obstacle_types[e] == 1)[0][0])      # the A-frame of this env
        apex = ob.env_goals[e, j, 2]                               # its third goal sits over the apex
        env.root_states[e, 0], env.root_states[e, 1], env.root_states[e, 2] = float(apex[0]), float(apex[1]), float(apex[2]) + 0.3
        env.root_states[e, 7:13] = 0
    for _ in range(40):
        env.sim.physics_step(torch.zeros(4, 12), 0)
    h = env.root_states[:, 2]
    assert (h > 0.333 * 0.5 + 0.1).all(), h                         # apex height 0.333 m: the base rests well above the ground plane


def test_tunnel_roof_and_tyre_ring_are_collision_geometry():
    """overhangs: the course hands its ceiling field (tunnel roof, upper arc of the tyre) to the physics; a robot thrown upwards
    inside the tunnel is held under the roof (the pipe's inner radius is 0.4 m: the roof peaks 0.8 m over the floor line)"""
    from quadrupedal_agility_amd.tsc.legged_gym.utils.obstacle import NO_CEILING
    env = cpu_env(4, seed=5)
    ob = env.obstacle
    assert env.qcfg.hf_ceiling == 1
    cs = env.sim.t["CEILING_SAMPLES"].numpy()
    assert np.array_equal(cs, ob.ceiling_raw) and (cs != NO_CEILING).sum() > 500
    roofed = cs != NO_CEILING
    assert (cs[roofed].astype(int) >= ob.height_field_raw[roofed]).mean() > 0.95       # the roof lies over the floor it belongs to
    tops = np.zeros(4)
    for e in range(4):
        j = int(np.nonzero(ob.obstacle_types[e] == 5)[0][0])      # the tunnel of this env
        mid = ob.env_goals[e, j, 2]                                # its third goal lies inside the pipe
        env.root_states[e, 0], env.root_states[e, 1], env.root_states[e, 2] = float(mid[0]), float(mid[1]), 0.35
        env.root_states[e, 7:13] = 0; env.root_states[e, 9] = 4.0
        ix, iy = int((mid[0] + 5.0) / 0.05), int((mid[1] + 5.0) / 0.05)
        assert cs[ix, iy] != NO_CEILING, (e, mid)
        tops[e] = cs[ix, iy] * 0.005
    peak = torch.zeros(4)
    for _ in range(30):
        env.sim.physics_step(torch.zeros(4, 12), 0)
        peak = torch.maximum(peak, env.root_states[:, 2])
    assert (peak.numpy() < tops).all() and (peak > 0.5).all(), (peak, tops)          # free flight would reach 0.35 + 4^2 / (2 g) = 1.16 m


def test_reset_matches_reference_formulas():
    """qo_tsc_reset against the reference's _reset_root_states / quat_from_euler_xyz expressions for the same uniforms"""
    env = cpu_env(32, seed=7, obstacle__randomize_start=True)
    flags = torch.ones(32, dtype=torch.uint8)
    env._reset(flags)
    root = env.root_states
    # (the reference's reset ends with one extra gym.simulate for everybody, :382-384: 5 ms of free fall from rest)
    assert (root[:, 2] <= 0.42).all() and (root[:, 2] > 0.4195).all() and (root[:, 7:13].abs() < 0.2).all()
    assert torch.allclose(root[:, 3:7].norm(dim=1), torch.ones(32), atol=1e-6) and (root[:, 3:5].abs() < 2e-3).all()     # yaw only
    start = env.env_goals.gather(1, (env.cur_obst_idx * 4)[:, None, None].expand(-1, 1, 3)).squeeze(1)
    d = root[:, :2] - start[:, :2]
    assert (d[:, 0] <= 1e-3).all() and (d[:, 0] >= -0.2 - 1e-3).all() and (d[:, 1].abs() <= 0.1 + 1e-3).all()
    yaw = 2 * torch.atan2(root[:, 5], root[:, 6])
    want = env.obst_angs.gather(1, env.cur_obst_idx[:, None]).squeeze(1)
    dy = torch.remainder(yaw - want + np.pi, 2 * np.pi) - np.pi
    assert (dy.abs() <= 0.2 + 1e-3).all()
    assert (env.cur_goal_idx == env.cur_obst_idx * 4).all() and len(set(env.cur_obst_idx.tolist())) > 2
    assert torch.allclose(env.dof_pos, env.default_dof_pos.expand(32, -1), atol=5e-3) and (env.dof_vel.abs() < 1.5).all()


def test_learn_rl_runs_on_cpu(tmp_path):
    from quadrupedal_agility_amd.tsc.rsl_rl.runners import OnPolicyRunner
    torch.manual_seed(1)
    env = cpu_env(8, seed=2, env__episode_length_s=0.5)
    tcfg = class_to_dict(Go2AgilityCfgPPO())
    tcfg["runner"]["num_steps_per_env"] = 6
    runner = OnPolicyRunner(env, tcfg, log_dir=str(tmp_path), device="cpu")
    before = {k: v.clone() for k, v in runner.alg.actor_critic.state_dict().items()}
    bbc_before = {k: v.clone() for k, v in runner.actor_critic_bbc.state_dict().items()}
    runner.learn(2, init_at_random_ep_len=True)
    after = runner.alg.actor_critic.state_dict()
    assert all(torch.isfinite(v).all() for v in after.values()) and any(not torch.equal(before[k], after[k]) for k in before)
    assert all(torch.equal(bbc_before[k], v) for k, v in runner.actor_critic_bbc.state_dict().items())      # frozen
    ck = torch.load(os.path.join(str(tmp_path), "model.pt"), weights_only=False)
    assert set(ck) >= {"model_state_dict", "estimator_state_dict", "optimizer_state_dict", "iter", "infos"}
    runner.load(os.path.join(str(tmp_path), "model.pt"))
    assert runner.current_learning_iteration == 2


# ------------------------------------------------------------------------------------------------------------------ GPU
def _gpu_pair(n, seed, **flags):
    cfg_g, cfg_o = make_cfg(n, seed, **flags), make_cfg(n, seed, **flags)
    env_g = lr.LeggedRobot(cfg_g, sim_device="cuda:0")
    ob = Obstacle(cfg_o.obstacle, n, seed=seed)
    env_o = lr.LeggedRobot(cfg_o, backend=OracleBackend(lr.make_qa_config(cfg_o, ob, seed=seed)), bookkeeping_lib=(load_oracle(), "qo_"))
    return env_g, env_o


@pytest.mark.gpu
@pytest.mark.parametrize("n", [64, 8192])
def test_two_level_pipeline_matches_oracle_on_gpu(n):
    """set_commands -> env physics step -> goal step -> reset -> observations: HIP vs the oracle's twins from identical state,
    single-step (state re-synchronised every step), with the frozen behaviour policy in the loop (same module on both sides)"""
    from quadrupedal_agility_amd.tsc.rsl_rl.modules import ActorCriticBBC
    from tests.test_hip_parity import env_mismatch
    torch.manual_seed(0)
    steps = 12 if n > 1000 else 30
    env_g, env_o = _gpu_pair(n, 4, env__episode_length_s=0.4)
    e = env_g.cfg.env
    bbc = ActorCriticBBC(e.num_observations_bbc, e.num_observations_bbc + 570, 12, e.n_proprio, e.n_auxiliary, e.history_len, e.n_priv, e.n_priv_latent,
                         e.num_command, **class_to_dict(Go2AgilityCfgPPO.policy()))
    bbc_g = __import__("copy").deepcopy(bbc).cuda()
    assert np.array_equal(env_g.obstacle.height_field_raw, env_o.obstacle.height_field_raw)
    hist = torch.zeros(n, 8, 19)
    flips = 0
    names = ("ROOT_STATES", "DOF_STATE", "CONTACT_FORCES", "RIGID_BODY_POS", "TORQUES", "ACTIONS", "LAST_DOF_VEL")
    for k in range(steps):
        # identical state on both sides: oracle arena -> device, bookkeeping tensors too
        env_g.sim.arena.copy_(torch.from_numpy(env_o.sim.o.arena.copy()).cuda())
        for name in ("commands", "latent_eps", "latent_c", "episode_length_buf", "cur_goal_idx", "reach_goal_timer", "last_contacts", "cur_goals",
                     "next_goals", "episode_sums_buf", "obs_history_buf", "delta_yaw", "delta_next_yaw", "obs_bbc_buf", "obs_disc_buf"):
            getattr(env_g.bk, name).copy_(getattr(env_o.bk, name).cuda())
        env_g.common_step_counter, env_g.global_counter = env_o.common_step_counter, env_o.global_counter
        a = torch.zeros(n, 19); a[:, 0] = torch.randint(0, 3, (n,)).float(); a[:, 1:] = torch.rand(n, 18) * 2 - 1
        hist = torch.cat([hist[:, 1:], a[:, None]], 1)
        noise = torch.rand(n, 5) * 0.4 + 0.8
        nc_o = env_o.bk.set_commands(a, noise); nc_g = env_g.bk.set_commands(a.cuda(), noise.cuda())
        assert torch.allclose(nc_g.cpu(), nc_o, atol=1e-6)
        ob_o = env_o.get_observations_bbc().clone(); ob_o[:, -11:] = nc_o
        with torch.no_grad():
            act_o = bbc.act_inference(ob_o, hist_encoding=True)
            act_g = bbc_g.act_inference(ob_o.cuda(), hist_encoding=True)
        assert torch.allclose(act_g.cpu(), act_o, atol=2e-4)
        env_o.sync_reset_ids = env_g.sync_reset_ids = False
        env_o.step(act_o, hist); env_g.step(act_o.cuda(), hist.cuda())
        torch.cuda.synchronize()
        bad = np.zeros(n, bool)
        for name in names:
            bad |= env_mismatch(name, env_g.sim.t[name].cpu().numpy(), env_o.sim.t[name].numpy(), n)
        ok = ~bad
        flips += int(bad.sum())
        # where the physics agrees, the task-level bookkeeping agrees to rounding: flags exact, rewards / rows tight
        for name in ("reset_buf", "time_out_buf", "episode_length_buf", "cur_goal_idx"):
            assert torch.equal(getattr(env_g.bk, name).cpu()[ok], getattr(env_o.bk, name)[ok]), (k, name)
        assert torch.allclose(env_g.bk.rew_buf.cpu()[ok], env_o.bk.rew_buf[ok], atol=2e-4, rtol=1e-3)
        for name in ("obs_buf", "obs_bbc_buf", "obs_disc_buf"):
            g, o = getattr(env_g.bk, name).cpu()[ok], getattr(env_o.bk, name)[ok]
            assert torch.allclose(g, o, atol=3e-3, rtol=1e-3), (k, name, float((g - o).abs().max()))
        assert torch.equal(env_g.bk.measured_heights.cpu()[ok], env_o.bk.measured_heights[ok])
    print(f"TSC pipeline, {n} envs x {steps} steps: env-steps outside the physics tolerances: {flips}")
    from tests.test_hip_parity import BUDGET, check_flips
    check_flips(f"course_{n}", flips, steps * n, BUDGET["course"])
    assert int((env_o.bk.episode_length_buf == 0).sum()) >= 0


@pytest.mark.gpu
def test_task_level_training_runs_on_gpu(tmp_path):
    from quadrupedal_agility_amd.tsc.rsl_rl.runners import OnPolicyRunner
    torch.manual_seed(0)
    cfg = make_cfg(256, 1, env__episode_length_s=1.0, domain_rand__push_robots=True, obstacle__randomize_start=True,
                   domain_rand__randomize_base_mass=True, domain_rand__randomize_base_com=True)
    env = lr.LeggedRobot(cfg, sim_device="cuda:0")
    tcfg = class_to_dict(Go2AgilityCfgPPO())
    runner = OnPolicyRunner(env, tcfg, log_dir=str(tmp_path), device="cuda:0")
    before = {k: v.clone() for k, v in runner.alg.actor_critic.state_dict().items()}
    runner.learn(3, init_at_random_ep_len=True)
    after = runner.alg.actor_critic.state_dict()
    assert all(torch.isfinite(v).all() for v in after.values()) and any(not torch.equal(before[k], after[k]) for k in before)
    assert runner._bbc_chain is not None                    # the frozen behaviour policy ran as ONE qa_mlp_forward launch per step
    assert runner.last_perf["fps"] > 2e3 and env.common_step_counter == 1 + 3 * 24


@pytest.mark.gpu
@pytest.mark.parametrize("n,sync_phases", [(1024, False), (1024, True), (8192, False)])
def test_recorded_rollout_equals_eager_rollout_bit_for_bit(tmp_path, n, sync_phases):
    """the teacher's 24-step rollout as ONE hipGraph replay per iteration against the eager loop of the same job, iteration by iteration
    over five iterations (0 eager warm-up in both, 1 records, 2-4 are replay sessions on new data), with and without a device sync
    between the phases: the device step counter (reset / push keys) follows the host's, every replay samples new actions and meets new
    resets, and the stored rollout -- observations, hybrid actions, rewards, dones, values, log-probs -- is IDENTICAL to the eager run's
    (same kernels, same order, same Philox keys, torch's generator graph-safe).  The rollout graph contains no torch reduction
    (qa_tsc_reset_stats carries the reset OR and the episode means)."""
    from quadrupedal_agility_amd.tsc.rsl_rl.runners import OnPolicyRunner
    out = {}
    for mode in ("recorded", "eager"):
        torch.manual_seed(0)
        cfg = make_cfg(n, 1, env__episode_length_s=1.0, domain_rand__push_robots=True, obstacle__randomize_start=True, domain_rand__push_interval=7)
        env = lr.LeggedRobot(cfg, sim_device="cuda:0")
        runner = OnPolicyRunner(env, class_to_dict(Go2AgilityCfgPPO()), log_dir=str(tmp_path / mode), device="cuda:0")
        runner.use_rollout_graph = mode == "recorded"
        snaps = []
        for it in range(5):
            runner.learn(1, init_at_random_ep_len=(it == 0))
            if sync_phases:
                torch.cuda.synchronize()
            st = runner.alg.storage
            snaps.append(dict(obs=st.observations.clone(), actions=st.actions.clone(), rewards=st.rewards.clone(), dones=st.dones.clone(),
                              values=st.values.clone(), logp_d=st.actions_log_prob_d.clone(), logp_c=st.actions_log_prob_c.clone(),
                              root=env.root_states.clone(), ep=env.episode_length_buf.clone(), means=env._episode_means.clone()))
            assert env.common_step_counter == 1 + 24 * (it + 1)
        assert int(env._step_dev) == env.common_step_counter
        if mode == "recorded":
            assert any(isinstance(v, tuple) for v in runner._rollout_graphs.values())
        else:
            assert not runner._rollout_graphs
        out[mode] = snaps
    for it in range(5):
        for k in out["eager"][it]:
            a, b = out["eager"][it][k], out["recorded"][it][k]
            assert torch.equal(a, b), f"iteration {it}: `{k}` differs between the recorded and the eager rollout (max {(a.double() - b.double()).abs().max().item()})"
    r = out["recorded"]
    assert not torch.equal(r[-1]["actions"], r[-2]["actions"]) and not torch.equal(r[-2]["actions"], r[-3]["actions"])       # replays draw new action noise
    assert all(s["dones"].float().mean().item() > 0.005 for s in r[1:])                                                    # episodes end in every replay
    assert any(not torch.equal(r[i]["means"], r[i - 1]["means"]) for i in range(2, 5))                                     # extras["episode"] follows the resets


@pytest.mark.gpu
def test_fused_teacher_rollout_equals_the_module_rollout(tmp_path):
    """r4: the teacher's networks (estimator, scan / privileged encoders, trunk, heads, critic) as one qa_mlp_forward chain and the style reward as
    qa_disc_prepare + chain + qa_rollout_post_amp, against the module path (13 library GEMMs + Discriminator.predict_disc_reward +
    process_env_step) on the same job: one eager 24-step rollout each from identical state.
    "chain" (torch's sampling ops, same generator): values, means, log-probs, rewards and dones of the stored rollout agree with the module path to
    GEMM rounding; the categorical choice, an argmax over perturbed probabilities, on all but a handful of samples.
    "hybrid" (qa_rollout_act_hybrid: Philox draws instead of torch's generator, so the samples differ): the same networks' outputs at step 0, stored
    log-probs consistent with the stored means / actions, choice frequencies and rollout statistics as the module path's."""
    from quadrupedal_agility_amd.tsc.rsl_rl.runners import OnPolicyRunner
    out = {}
    for mode in ("chain", "hybrid", "modules"):
        torch.manual_seed(0)
        cfg = make_cfg(1024, 1, env__episode_length_s=1.0, domain_rand__push_robots=True, obstacle__randomize_start=True, domain_rand__push_interval=7)
        env = lr.LeggedRobot(cfg, sim_device="cuda:0")
        runner = OnPolicyRunner(env, class_to_dict(Go2AgilityCfgPPO()), log_dir=str(tmp_path / mode), device="cuda:0")
        runner.use_rollout_graph = False
        runner.use_hybrid_act = mode == "hybrid"
        if mode == "modules":
            runner._teacher_chain_obj = runner._style_chain_obj = False
        env.episode_length_buf = torch.randint_like(env.episode_length_buf, high=int(env.max_episode_length))
        runner._alloc_rollout_state()
        torch.manual_seed(1)
        with torch.inference_mode():
            runner._rollout_steps(False, True)
        torch.cuda.synchronize()
        st = runner.alg.storage
        assert st.step == 24
        if mode != "modules":
            assert runner._teacher_chain_obj not in (None, False) and runner._style_chain_obj not in (None, False)
        out[mode] = dict(values=st.values.clone(), mu=st.mu.clone(), sigma=st.sigma.clone(), rewards=st.rewards.clone(), dones=st.dones.clone(),
                         a_d=st.actions[..., 0].clone(), a_c=st.actions[..., 1:].clone(), logp_d=st.actions_log_prob_d.clone(), logp_c=st.actions_log_prob_c.clone(),
                         obs0=st.observations[0].clone(), fin=runner._rs["fin_vals"].clone(), mask=runner._rs["fin_masks"].clone(), ahist=runner._rs["ahist"].clone(),
                         actions=st.actions.clone())
    f, h, m = out["chain"], out["hybrid"], out["modules"]
    # step 0 starts from identical observations: the networks' outputs agree to rounding there ...
    for x in (f, h):
        assert torch.equal(x["obs0"], m["obs0"])
        assert torch.allclose(x["values"][0], m["values"][0], rtol=2e-4, atol=2e-4) and torch.allclose(x["mu"][0], m["mu"][0], rtol=2e-4, atol=2e-4)
        assert torch.equal(x["sigma"], m["sigma"])
    same0 = (f["a_d"][0] == m["a_d"][0])
    assert same0.float().mean() > 0.995
    assert torch.allclose(f["a_c"][0][same0], m["a_c"][0][same0], rtol=2e-4, atol=2e-4) and torch.allclose(f["logp_c"][0][same0], m["logp_c"][0][same0], rtol=1e-3, atol=2e-3)
    assert torch.allclose(f["logp_d"][0][same0], m["logp_d"][0][same0].view_as(f["logp_d"][0][same0]), rtol=1e-3, atol=1e-3)
    # ... and where the two jobs still took the same gait decisions, rewards (style reward through the discriminator chain) and dones agree step by step;
    # a differing categorical draw forks an env's trajectory, so later steps are compared statistically
    assert torch.allclose(f["rewards"][0][same0], m["rewards"][0][same0], rtol=1e-3, atol=2e-4)
    assert torch.equal(f["dones"][0][same0], m["dones"][0][same0])
    for x in (f, h):
        assert abs(float(x["rewards"].mean()) - float(m["rewards"].mean())) < 0.08 * abs(float(m["rewards"].mean())) + 2e-3
        assert abs(float(x["dones"].float().mean()) - float(m["dones"].float().mean())) < 0.01
        assert x["mask"].any()
    # hybrid: the stored Gaussian log-prob is that of the stored action under the stored mean / std; the choices follow the module path's frequencies;
    # the runner's action history ends with the last three actions
    z = (h["a_c"] - h["mu"]) / h["sigma"]
    want = (-0.5 * z * z - torch.log(h["sigma"]) - 0.9189385332046727).sum(-1, keepdim=True)
    assert torch.allclose(h["logp_c"], want, rtol=1e-4, atol=1e-3)
    assert abs(float(z.mean())) < 0.01 and abs(float(z.std()) - 1.0) < 0.01
    fh = torch.bincount(h["a_d"].flatten().long(), minlength=6).float() / h["a_d"].numel()
    fm = torch.bincount(m["a_d"].flatten().long(), minlength=6).float() / m["a_d"].numel()
    assert (fh - fm).abs().max() < 0.03
    assert (h["logp_d"] <= 0).all() and (h["logp_d"] > -16.0).all()
    assert torch.equal(h["ahist"][:, -1], h["actions"][-1]) and torch.equal(h["ahist"][:, -3], h["actions"][-3])
