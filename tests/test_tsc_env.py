"""SURVEY 8a row a18, env side: the task-level command mapping and goal / termination / reward bookkeeping.
CPU: the oracle's qo_tsc_* against the reference's own outputs (tests/golden/tsc_env.npz).
GPU: the HIP qa_tsc_* against the same goldens and against the oracle, through the C ABI."""
import numpy as np
import pytest
import torch

from quadrupedal_agility_amd import _capi
from tests import tsc_env_protocol as proto
from tests.oracle_lib import load_oracle


def _check_commands(out, fx, case, tol):
    for k in ("commands", "latent_eps", "latent_c", "next_commands"):
        np.testing.assert_allclose(out[k], fx[f"cmd_{case}_{k}"], rtol=tol, atol=tol, err_msg=f"{case} {k}")


@pytest.mark.parametrize("case", ["every", "sparse"])
def test_oracle_set_commands_matches_the_reference(case):
    fx = proto.load_fixture()
    out = proto.run_set_commands(proto.NumpyBackend(load_oracle()), fx, case)
    _check_commands(out, fx, case, 1e-6)
    if case == "sparse":                      # envs off the resampling step keep their commands (times the noise)
        off = fx["cmd_sparse_episode_length"] % int(fx["cmd_sparse_interval"]) != 0
        assert off.any() and (~off).any()
        np.testing.assert_allclose(out["commands"][off], (fx["cmd_sparse_commands0"] * fx["cmd_sparse_noise"])[off], rtol=1e-6)
        np.testing.assert_array_equal(out["latent_c"][off], fx["cmd_sparse_latent_c0"][off])


def test_oracle_goal_steps_match_the_reference():
    fx = proto.load_fixture()
    results = proto.run_goal_steps(proto.NumpyBackend(load_oracle()), fx)
    proto.compare_goal_steps(results, fx)
    # the fixture exercises every branch
    flags = {k: sum(int(r[k].sum()) for r in results) for k in ("reset_buf", "time_out_buf", "reach_goal_cutoff", "reached_goal")}
    assert all(v > 0 for v in flags.values()), flags
    assert any((r["rew_buf"] < 0).any() for r in results) and any((r["rew_buf"] > 0).any() for r in results)


def test_oracle_rejects_bad_arguments():
    import ctypes as C
    from quadrupedal_agility_amd import _capi
    qo = load_oracle()
    assert qo.qo_tsc_goal_step(None, None, None) != 0
    a = np.zeros((4, 19), np.float32)
    assert qo.qo_tsc_set_commands(a.ctypes.data, None, 4, 3, 6, 5, 1, None, None, None, None, None, None, None, None, None, None) != 0
    assert C.sizeof(_capi.QaTscGoalIo) == 8 * len(_capi.TSC_GOAL_IO_FIELDS)


def test_oracle_observations_match_the_reference():
    """_get_heights + compute_observations: scan heights exactly, the 800 / 671 / 49 rows and the pushed history to 1e-5"""
    fx = proto.load_fixture()
    results = proto.run_observations(proto.NumpyBackend(load_oracle()), fx)
    proto.compare_observations(results, fx)
    assert (results[0]["measured_heights"] > 0).any() and (np.abs(results[2]["obs_buf"]) == 100.0).any()       # scan hits boxes; the clip acts
    assert int(fx["obs_t1_update_yaw"]) == 0 and np.array_equal(results[1]["delta_yaw"], results[0]["delta_yaw"])  # carried over
    # layout: the task row and the behaviour row share proprio, privileged values and the pre-push history
    o, b = results[1]["obs_buf"], results[1]["obs_bbc_buf"]
    np.testing.assert_array_equal(o[:, :57], b[:, :57])
    np.testing.assert_array_equal(o[:, 197:230], b[:, 57:90])
    np.testing.assert_array_equal(o[:, 230:800], b[:, 90:660])
    np.testing.assert_array_equal(o[:, 230:800], results[0]["obs_history"].reshape(-1, 570))


@pytest.mark.gpu
def test_hip_observations_match_reference_and_oracle():
    from quadrupedal_agility_amd import _capi
    fx = proto.load_fixture()
    hip = proto.run_observations(proto.TorchBackend(_capi.load_library()), fx)
    proto.compare_observations(hip, fx)
    cpu = proto.run_observations(proto.NumpyBackend(load_oracle()), fx)
    for h, c in zip(hip, cpu):
        np.testing.assert_array_equal(h["measured_heights"], c["measured_heights"])
        for k in h:
            np.testing.assert_allclose(h[k], c[k], rtol=2e-6, atol=1e-6, err_msg=k)           # sinf / cosf / atan2f differ by ulps
        for k in ("obs_buf", "obs_bbc_buf"):                                                   # everything but the yaw errors is bit-exact
            keep = np.ones(h[k].shape[1], bool); keep[57:59] = k != "obs_buf"
            np.testing.assert_array_equal(h[k][:, keep], c[k][:, keep], err_msg=k)


@pytest.mark.gpu
def test_hip_observations_at_full_size_are_the_tiled_small_case():
    """8192 envs (config 4): envs are independent, so tiling the fixture 128x must tile the outputs (ragged last workgroup included)"""
    from quadrupedal_agility_amd import _capi
    fx = proto.load_fixture()
    lib = _capi.load_library()
    reps = 8192 // fx["obs_history0"].shape[0]
    big = proto.run_observations(proto.TorchBackend(lib), fx, reps=reps)
    small = proto.run_observations(proto.TorchBackend(lib), fx)
    for rb, rs in zip(big, small):
        for k in rb:
            np.testing.assert_array_equal(rb[k], np.tile(rs[k], (reps,) + (1,) * (rs[k].ndim - 1)), err_msg=k)
    odd = proto.run_observations(proto.TorchBackend(lib), {k: (fx[k][:61] if k.startswith("obs_") and fx[k].ndim and fx[k].shape[0] == 64 else
                                                               (fx[k][:, :61] if k == "obs_motor_strength" else fx[k])) for k in fx.files})
    for ro, rs in zip(odd, small):
        for k in ro:
            np.testing.assert_array_equal(ro[k], rs[k][:61], err_msg=k)


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["every", "sparse"])
def test_hip_set_commands_matches_reference_and_oracle(case):
    from quadrupedal_agility_amd import _capi
    fx = proto.load_fixture()
    hip = proto.run_set_commands(proto.TorchBackend(_capi.load_library()), fx, case)
    _check_commands(hip, fx, case, 1e-6)
    cpu = proto.run_set_commands(proto.NumpyBackend(load_oracle()), fx, case)
    for k in hip:
        np.testing.assert_array_equal(hip[k], cpu[k], err_msg=k)           # no transcendental in this kernel: bit-exact
    nonoise = proto.run_set_commands(proto.TorchBackend(_capi.load_library()), fx, case, with_noise=False)
    cpu0 = proto.run_set_commands(proto.NumpyBackend(load_oracle()), fx, case, with_noise=False)
    np.testing.assert_array_equal(nonoise["next_commands"], cpu0["next_commands"])


@pytest.mark.gpu
def test_hip_goal_steps_match_reference_and_oracle():
    from quadrupedal_agility_amd import _capi
    fx = proto.load_fixture()
    hip = proto.run_goal_steps(proto.TorchBackend(_capi.load_library()), fx)
    proto.compare_goal_steps(hip, fx)
    cpu = proto.run_goal_steps(proto.NumpyBackend(load_oracle()), fx)
    for h, c in zip(hip, cpu):
        for k in h:
            if h[k].dtype.kind in "ui":
                np.testing.assert_array_equal(h[k], c[k], err_msg=k)
            else:
                np.testing.assert_allclose(h[k], c[k], rtol=2e-6, atol=1e-6, err_msg=k)     # atan2f / asinf / expf differ by ulps


@pytest.mark.gpu
def test_hip_goal_step_at_full_size_properties():
    """8192 envs (config 4): invariants that need no oracle -- flags consistent with each other, reward = clipped sum +
    termination term, goals gathered from the table, a second call advances episode length by one."""
    import ctypes as C
    import torch
    from quadrupedal_agility_amd import _capi
    lib = _capi.load_library()
    fx = proto.load_fixture()
    n, reps = 8192, 8192 // fx["goal_cur_goal_idx0"].shape[0]
    big = {k: (np.tile(fx[k], (reps,) + (1,) * (fx[k].ndim - 1)) if k.startswith("goal_") and fx[k].ndim and fx[k].shape[0] == n // reps and
               k not in ("goal_x_edge_mask",) else fx[k]) for k in fx.files}
    big["goal_x_edge_mask"] = fx["goal_x_edge_mask"]
    results = proto.run_goal_steps(proto.TorchBackend(lib), big)
    small = proto.run_goal_steps(proto.TorchBackend(lib), fx)
    for rb, rs in zip(results, small):
        for k in rb:
            tiled = np.tile(rs[k], (1, reps) if k == "episode_sums" else (reps,) + (1,) * (rs[k].ndim - 1))
            np.testing.assert_array_equal(rb[k], tiled, err_msg=k)         # envs are independent: tiling the inputs tiles the outputs
        assert (rb["reset_buf"] >= rb["time_out_buf"]).all() and (rb["time_out_buf"] >= rb["reach_goal_cutoff"]).all()
        scales = fx["goal_reward_scales"]
        assert (rb["rew_buf"] >= scales[-1] - 1e-6).all()
        term = rb["reset_buf"].astype(bool) & ~rb["time_out_buf"].astype(bool)
        assert (rb["rew_buf"][~term] >= 0).all()
    assert C.sizeof(_capi.QaTscGoalCfg) > 0 and torch.cuda.is_available()


@pytest.mark.gpu
def test_task_level_bookkeeping_mirror_replays_the_reference():
    """The host-side mirror (reference attribute names) over the same kernels: the fixture's 8 steps and a command mapping."""
    import types
    import torch
    from quadrupedal_agility_amd.tsc.legged_gym import TaskLevelBookkeeping
    fx = proto.load_fixture()
    slots, repeat, per, k, rows, cols = (int(v) for v in fx["goal_ints"])
    delay, thr, leave, max_len, vt, border, hs = (float(v) for v in fx["goal_scalars"])
    ns = types.SimpleNamespace
    dt = 0.02
    names = [str(s) for s in fx["goal_reward_names"]]
    cfg = ns(env=ns(mocap_category=["trot", "canter", "jump"], mocap_category_all=["walk", "pace", "trot", "canter", "jump"], num_actions_c=6,
                    reach_goal_delay=delay * dt, next_goal_threshold=thr, leave_goal_threshold=leave, episode_length_s=max_len * dt),
             commands=ns(resampling_time=0.02, ranges=ns(lin_vel_x=fx["cmd_vel_ranges"][0].tolist(), lin_vel_y=fx["cmd_vel_ranges"][1].tolist(),
                                                         ang_vel_yaw=fx["cmd_vel_ranges"][2].tolist(), jump_height=fx["cmd_jump_range"].tolist(),
                                                         locomotion_height=fx["cmd_height_range"].tolist())),
             rewards=ns(target_lin_vel=vt, scales=ns(**{n: float(s) / dt for n, s in zip(names, fx["goal_reward_scales"])})),
             obstacle=ns(num_goals=per, last_goal_repeat=repeat, border_size=border, horizontal_scale=hs), depth=ns(use_camera=False),
             control=ns(decimation=4), sim=ns(dt=0.005))
    T = torch.from_numpy
    env = TaskLevelBookkeeping(cfg, T(fx["goal_env_goals"]), T(fx["goal_obstacle_types"]), T(fx["goal_x_edge_mask"]), fx["goal_feet"],
                               fx["goal_penalised"], fx["goal_termination"], num_bodies=19)
    env.episode_length_buf.copy_(T(fx["cmd_every_episode_length"]))
    env.commands.copy_(T(fx["cmd_every_commands0"])); env.latent_eps.copy_(T(fx["cmd_every_latent_eps0"])); env.latent_c.copy_(T(fx["cmd_every_latent_c0"]))
    nxt = env.set_commands(T(fx["cmd_every_actions"]), T(fx["cmd_every_noise"]))
    np.testing.assert_allclose(nxt.cpu().numpy(), fx["cmd_every_next_commands"], rtol=1e-6, atol=1e-6)
    env.episode_length_buf.copy_(T(fx["goal_episode_length0"])); env.cur_goal_idx.copy_(T(fx["goal_cur_goal_idx0"]))
    env.reach_goal_timer.copy_(T(fx["goal_timer0"])); env.last_contacts.copy_(T(fx["goal_last_contacts0"]))
    env.cur_goals.copy_(T(fx["goal_cur_goals0"])); env.next_goals.copy_(T(fx["goal_next_goals0"]))
    for cam in (0, 1):
        env._cfg.use_camera = cam
        for t in range(proto.STEPS):
            tag = f"goal_c{cam}_t{t}_"
            hist = fx[tag + "action_hl_history"]
            ids = env.post_physics_step(T(fx[tag + "root_states"]), T(fx[tag + "contact_forces"]), T(fx[tag + "rigid_body_states"]),
                                        T(hist) if hist.size else None)
            np.testing.assert_array_equal(env.reset_buf.cpu().numpy(), fx[tag + "reset_buf"])
            np.testing.assert_array_equal(np.sort(ids.cpu().numpy()), np.flatnonzero(fx[tag + "reset_buf"]))
            np.testing.assert_allclose(env.rew_buf.cpu().numpy(), fx[tag + "rew_buf"], rtol=1e-5, atol=2e-6)
            np.testing.assert_allclose(env.target_yaw.cpu().numpy(), fx[tag + "target_yaw"], rtol=1e-5, atol=2e-6)
            env.reset_idx(ids)
            # after the reset bookkeeping the state is the reference's, reset envs included
            np.testing.assert_array_equal(env.cur_goal_idx.cpu().numpy(), fx[tag + "cur_goal_idx"])
            np.testing.assert_array_equal(env.episode_length_buf.cpu().numpy(), fx[tag + "episode_length"])
            np.testing.assert_array_equal(env.cur_goals.cpu().numpy(), fx[tag + "cur_goals"])
            np.testing.assert_array_equal(env.next_goals.cpu().numpy(), fx[tag + "next_goals"])
            np.testing.assert_allclose(env.episode_sums_buf.cpu().numpy(), fx[tag + "episode_sums"], rtol=1e-5, atol=2e-6)
    # observations through the mirror: the fixture's three steps, state set to what the reference had
    border, hsc, vs, lin, ang, dp, dv, lin_d, ang_d, key, foot, clip = (float(v) for v in fx["obs_scalars"])
    cfg.env.root_height_obs = True
    cfg.obstacle.vertical_scale = vs
    cfg.normalization = ns(clip_observations=clip, obs_scales=ns(lin_vel=lin, ang_vel=ang, dof_pos=dp, dof_vel=dv, lin_vel_dist=lin_d,
                                                               ang_vel_dist=ang_d, key_pos=key, foot_contact=foot))
    env.init_observations(T(fx["obs_height_samples"]), T(fx["obs_height_points"]), fx["obs_default_dof_pos"], fx["obs_default_dof_pos_all"])
    env.obs_history_buf.copy_(T(fx["obs_history0"]))
    env.commands.copy_(T(fx["obs_commands"])); env.latent_eps.copy_(T(fx["obs_latent_eps"])); env.latent_c.copy_(T(fx["obs_latent_c"]))
    for t in range(proto.OBS_STEPS):
        tag = f"obs_t{t}_"
        for name, key_ in (("rpy", "rpy"), ("base_lin_vel", "base_lin_vel"), ("base_ang_vel", "base_ang_vel"), ("contact_filt", "contact_filt"),
                           ("cur_obstacle_types", "cur_obstacle_type"), ("target_yaw", "target_yaw"), ("next_target_yaw", "next_target_yaw"),
                           ("episode_length_buf", "episode_length")):
            getattr(env, name).copy_(T(fx[tag + key_]))
        obs = env.compute_observations(T(fx[tag + "root_states"]), T(fx[tag + "dof_pos"]), T(fx[tag + "dof_vel"]), T(fx[tag + "action_history"]),
                                       T(fx[tag + "rigid_body_states"]), T(fx["obs_mass_params"]), T(fx["obs_friction"]),
                                       T(fx["obs_motor_strength"]), update_yaw=bool(fx[tag + "update_yaw"]))
        np.testing.assert_allclose(obs.cpu().numpy(), fx[tag + "obs_buf"], rtol=1e-5, atol=2e-6)
        np.testing.assert_allclose(env.get_observations_bbc().cpu().numpy(), fx[tag + "obs_bbc_buf"], rtol=1e-5, atol=2e-6)
        np.testing.assert_allclose(env.get_observations_disc().cpu().numpy(), fx[tag + "obs_disc_buf"], rtol=1e-5, atol=2e-6)
        np.testing.assert_array_equal(env.measured_heights.cpu().numpy(), fx[tag + "measured_heights"])
        np.testing.assert_allclose(env.obs_history_buf.cpu().numpy(), fx[tag + "obs_history"], rtol=1e-5, atol=2e-6)


def test_task_level_mirror_needs_the_gpu_library():
    import torch
    from quadrupedal_agility_amd.tsc.legged_gym import TaskLevelBookkeeping
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError):
        TaskLevelBookkeeping(None, torch.zeros(2, 4, 3), torch.zeros(2, 1), torch.zeros(2, 2), [0] * 4, [], [], 19)


def _tsc_modules():
    import torch
    from quadrupedal_agility_amd.tsc.rsl_rl import modules as mods
    from quadrupedal_agility_amd.tsc.rsl_rl import algorithms as algs
    from tests import tsc_protocol
    torch.manual_seed(0)
    ac, bbc, est, alg = tsc_protocol.build(mods, algs)
    return bbc, est, alg


def test_frozen_behaviour_policy_of_the_task_level_loop_as_one_chain():
    """PPO.act_bbc (tsc/rsl_rl/algorithms/ppo.py:127-137: estimator overwrite, history encoder, actor mean) on the 671-wide
    obs_bbc row that qa_tsc_observations writes = the same qa_mlp_forward chain as the BBC tree's inference policy; here through
    the oracle's C twin against the torch modules."""
    import torch
    from quadrupedal_agility_amd.rsl_rl.algorithms.fused import PolicyChain
    from tests.test_policy_chain import run_oracle
    bbc, est, alg = _tsc_modules()
    assert alg.train_with_estimated_states
    chain = PolicyChain.describe(bbc, est, True, hist_encoding=True, with_critic=False, estimate_col=alg._priv_slice(False).start)
    assert chain is not None and alg._priv_slice(False).start == 57 + 132       # the reference writes the estimate into the history block
    fx = proto.load_fixture()
    obs = torch.from_numpy(fx["obs_t1_obs_bbc_buf"]).clamp(-3, 3)          # the reference's own behaviour-policy rows
    with torch.no_grad():
        ref = alg.act_bbc(obs)
    mean = run_oracle(chain, obs)[0]
    np.testing.assert_allclose(mean, ref.numpy(), rtol=1e-4, atol=2e-5)


@pytest.mark.gpu
def test_hip_observation_rows_feed_the_behaviour_policy_chain():
    """device-resident path of one inner step: qa_tsc_observations -> obs_bbc_buf -> qa_mlp_forward (no host copy in between)"""
    import torch
    from quadrupedal_agility_amd import _capi
    from quadrupedal_agility_amd.rsl_rl.algorithms.fused import PolicyChain
    bbc, est, alg = _tsc_modules()
    bbc, est = bbc.cuda(), est.cuda()
    alg.actor_critic_bbc, alg.estimator = bbc, est
    chain = PolicyChain.describe(bbc, est, True, hist_encoding=True, with_critic=False, estimate_col=alg._priv_slice(False).start)
    fx = proto.load_fixture()
    be = proto.TorchBackend(_capi.load_library())
    cfg, const, state, outs, tile = proto.run_observations(be, fx, prepare_only=True)
    import ctypes as C
    for t in range(2):
        io, dev = proto.obs_io(be, fx, cfg, const, state, outs, t, tile)
        assert be.lib.qa_tsc_observations(C.byref(cfg), C.byref(io), be.stream) == 0
    rows = outs["obs_bbc_buf"]
    with torch.inference_mode():
        chain.pack()
        mean = chain.forward(rows)[0]
    torch.cuda.synchronize()
    with torch.no_grad():
        ref = alg.act_bbc(rows.clone())
    np.testing.assert_allclose(rows.cpu().numpy(), fx["obs_t1_obs_bbc_buf"], rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(mean.cpu().numpy(), ref.cpu().numpy(), rtol=2e-4, atol=1e-4)


# ---------------------------------------------------------------- reset bookkeeping: the flag OR + extras["episode"] means in one launch
def _reset_stats_case(n, seed, p):
    g = torch.Generator().manual_seed(seed)
    flags = (torch.rand(n, generator=g) < p).to(torch.uint8)
    sums = torch.randn(8, n, generator=g) * 30
    means0 = torch.randn(8, generator=g)
    return flags, sums, means0


def _reference_reset_stats(flags, sums, means0, len_s):
    """tsc/legged_gym/envs/base/legged_robot.py:396-404: mean over the resetting envs of episode_sums / max_episode_length_s (kept when
    nobody resets), and :382-384's `len(env_ids) > 0`"""
    ids = flags.nonzero(as_tuple=False).flatten()
    if len(ids) == 0:
        return means0.clone(), 0
    return torch.stack([torch.mean(sums[k][ids].double()) / len_s for k in range(sums.shape[0])]).float(), 1


@pytest.mark.parametrize("n,p", [(1, 1.0), (64, 0.0), (64, 0.3), (1000, 0.05), (8192, 0.01), (8192, 1.0)])
def test_oracle_reset_stats_match_the_reference_expression(n, p):
    lib = load_oracle()
    flags, sums, means0 = _reset_stats_case(n, n + 3, p)
    means, any_r = means0.clone().numpy(), np.zeros(1, np.uint8)
    f, s = flags.numpy(), np.ascontiguousarray(sums.numpy())
    assert lib.qo_tsc_reset_stats(f.ctypes.data, s.ctypes.data, n, 8, 20.0, means.ctypes.data, any_r.ctypes.data, None) == 0
    want, want_any = _reference_reset_stats(flags, sums, means0, 20.0)
    assert int(any_r[0]) == want_any
    np.testing.assert_allclose(means, want.numpy(), rtol=2e-5, atol=1e-6)
    assert lib.qo_tsc_reset_stats(None, s.ctypes.data, n, 8, 20.0, means.ctypes.data, any_r.ctypes.data, None) != 0
    assert lib.qo_tsc_reset_stats(f.ctypes.data, s.ctypes.data, n, 8, 0.0, means.ctypes.data, any_r.ctypes.data, None) != 0


@pytest.mark.gpu
@pytest.mark.parametrize("terms", [1, 9, 11, 16])
def test_hip_reset_stats_with_other_term_counts(terms):
    """r6: the kernel adds eight terms per pass; fewer and more than eight (a second, partial group) against the twin, bit for bit"""
    hip, lib = _capi.load_library(), load_oracle()
    n = 3000
    g = torch.Generator().manual_seed(terms)
    flags = (torch.rand(n, generator=g) < 0.2).to(torch.uint8)
    sums, means0 = torch.randn(terms, n, generator=g) * 30, torch.randn(terms, generator=g)
    means, any_r = means0.clone().numpy(), np.zeros(1, np.uint8)
    f, s = flags.numpy(), np.ascontiguousarray(sums.numpy())
    assert lib.qo_tsc_reset_stats(f.ctypes.data, s.ctypes.data, n, terms, 20.0, means.ctypes.data, any_r.ctypes.data, None) == 0
    fd, sd, md, ad = flags.cuda(), sums.cuda().contiguous(), means0.clone().cuda(), torch.zeros(1, dtype=torch.uint8, device="cuda")
    assert hip.qa_tsc_reset_stats(fd.data_ptr(), sd.data_ptr(), n, terms, 20.0, md.data_ptr(), ad.data_ptr(), torch.cuda.current_stream().cuda_stream) == 0
    assert int(ad.cpu()[0]) == int(any_r[0]) == 1 and np.array_equal(md.cpu().numpy(), means)


@pytest.mark.gpu
@pytest.mark.parametrize("n,p", [(1, 1.0), (64, 0.0), (64, 0.3), (1000, 0.05), (8192, 0.01), (8192, 1.0)])
def test_hip_reset_stats_equal_the_twin_bit_for_bit(n, p):
    hip, lib = _capi.load_library(), load_oracle()
    flags, sums, means0 = _reset_stats_case(n, n + 3, p)
    means, any_r = means0.clone().numpy(), np.zeros(1, np.uint8)
    f, s = flags.numpy(), np.ascontiguousarray(sums.numpy())
    assert lib.qo_tsc_reset_stats(f.ctypes.data, s.ctypes.data, n, 8, 20.0, means.ctypes.data, any_r.ctypes.data, None) == 0
    fd, sd, md, ad = flags.cuda(), sums.cuda().contiguous(), means0.clone().cuda(), torch.full((1,), 7, dtype=torch.uint8, device="cuda")
    assert hip.qa_tsc_reset_stats(fd.data_ptr(), sd.data_ptr(), n, 8, 20.0, md.data_ptr(), ad.data_ptr(), torch.cuda.current_stream().cuda_stream) == 0
    assert int(ad.cpu()[0]) == int(any_r[0])
    assert np.array_equal(md.cpu().numpy(), means)                 # same summation order, fp-contract off on both sides
