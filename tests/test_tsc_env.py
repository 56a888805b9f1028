"""SURVEY 8a row a18, env side: the task-level command mapping and goal / termination / reward bookkeeping.
CPU: the oracle's qo_tsc_* against the reference's own outputs (tests/golden/tsc_env.npz).
GPU: the HIP qa_tsc_* against the same goldens and against the oracle, through the C ABI."""
import numpy as np
import pytest

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


def test_task_level_mirror_needs_the_gpu_library():
    import torch
    from quadrupedal_agility_amd.tsc.legged_gym import TaskLevelBookkeeping
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError):
        TaskLevelBookkeeping(None, torch.zeros(2, 4, 3), torch.zeros(2, 1), torch.zeros(2, 2), [0] * 4, [], [], 19)
