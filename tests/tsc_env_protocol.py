"""Replays tests/golden/tsc_env.npz (made by tools/gen_golden_tsc_env.py from the reference's own set_commands /
post_physics_step) through qa_tsc_set_commands / qa_tsc_goal_step of either library: the CPU oracle (numpy arrays) or the HIP
library (torch tensors on the GPU).  A backend supplies `put(np) -> handle`, `ptr(handle)`, `get(handle) -> np`."""
import ctypes as C
import os

import numpy as np

from quadrupedal_agility_amd import _capi

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "tsc_env.npz")
STEPS = 4


class NumpyBackend:
    def __init__(self, lib, prefix="qo_"):
        self.lib, self.prefix, self.stream = lib, prefix, None
    put = staticmethod(lambda a: np.ascontiguousarray(a).copy())
    ptr = staticmethod(lambda a: a.ctypes.data)
    get = staticmethod(lambda a: a.copy())


class TorchBackend:
    def __init__(self, lib, prefix="qa_"):
        import torch
        self.torch, self.lib, self.prefix = torch, lib, prefix
        self.stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def put(self, a):
        return self.torch.from_numpy(np.ascontiguousarray(a)).cuda()
    ptr = staticmethod(lambda t: t.data_ptr())

    def get(self, t):
        return t.cpu().numpy()


def load_fixture():
    return np.load(GOLD)


def run_set_commands(be, fx, case, with_noise=True):
    g = lambda k: fx[f"cmd_{case}_{k}"]          # noqa: E731
    n = g("actions").shape[0]
    dim_c = g("latent_c0").shape[1]
    num_d = len(fx["cmd_mocap_index"])
    dev = {k: be.put(v) for k, v in dict(actions=g("actions"), ep=g("episode_length").astype(np.int64), noise=g("noise"),
                                         commands=g("commands0"), eps=g("latent_eps0"), c=g("latent_c0"),
                                         nxt=np.zeros((n, 6 + dim_c), np.float32)).items()}
    mi = np.ascontiguousarray(fx["cmd_mocap_index"], np.int32)
    vr, jr, hr = (np.ascontiguousarray(fx[k], np.float32) for k in ("cmd_vel_ranges", "cmd_jump_range", "cmd_height_range"))
    f = getattr(be.lib, be.prefix + "tsc_set_commands")
    rc = f(be.ptr(dev["actions"]), be.ptr(dev["ep"]), n, num_d, 6, dim_c, int(g("interval")), mi.ctypes.data, vr.ctypes.data, jr.ctypes.data,
           hr.ctypes.data, be.ptr(dev["noise"]) if with_noise else None, be.ptr(dev["commands"]), be.ptr(dev["eps"]), be.ptr(dev["c"]),
           be.ptr(dev["nxt"]), be.stream)
    assert rc == 0
    return {"commands": be.get(dev["commands"]), "latent_eps": be.get(dev["eps"]), "latent_c": be.get(dev["c"]),
            "next_commands": be.get(dev["nxt"])}


def goal_cfg(fx, n, use_camera):
    slots, repeat, per, k, rows, cols = (int(v) for v in fx["goal_ints"])
    delay, thr, leave, max_len, vt, border, hs = (float(v) for v in fx["goal_scalars"])
    c = _capi.QaTscGoalCfg()
    c.num_envs, c.num_bodies, c.num_goal_slots, c.last_goal_repeat, c.goals_per_obstacle, c.num_obstacles = n, 19, slots, repeat, per, k
    c.history_len, c.history_width, c.mask_rows, c.mask_cols, c.use_camera = 5, 19, rows, cols, use_camera
    term, pen, feet = fx["goal_termination"], fx["goal_penalised"], fx["goal_feet"]
    c.num_termination_bodies, c.num_penalised_bodies = len(term), len(pen)
    for i, b in enumerate(term):
        c.termination_bodies[i] = int(b)
    for i, b in enumerate(pen):
        c.penalised_bodies[i] = int(b)
    for i, b in enumerate(feet):
        c.feet_bodies[i] = int(b)
    c.reach_goal_delay_steps, c.next_goal_threshold, c.leave_goal_threshold, c.max_episode_length = delay, thr, leave, max_len
    c.target_lin_vel, c.border_size, c.horizontal_scale = vt, border, hs
    assert [str(s) for s in fx["goal_reward_names"]] == list(_capi.TSC_REWARD_NAMES)
    for i, s in enumerate(fx["goal_reward_scales"]):
        c.reward_scales[i] = float(s)
    return c


OUT_SHAPES = dict(base_lin_vel=(3, np.float32), base_ang_vel=(3, np.float32), projected_gravity=(3, np.float32), rpy=(3, np.float32),
                  contact_filt=(4, np.uint8), target_pos_rel=(2, np.float32), next_target_pos_rel=(2, np.float32), target_yaw=(0, np.float32),
                  next_target_yaw=(0, np.float32), reached_goal=(0, np.uint8), cur_obstacle_type=(0, np.int64), reset_buf=(0, np.uint8),
                  time_out_buf=(0, np.uint8), reach_goal_cutoff=(0, np.uint8), rew_buf=(0, np.float32))
STATE = ("episode_length", "cur_goal_idx", "reach_goal_timer", "last_contacts", "cur_goals", "next_goals", "episode_sums")


def run_goal_steps(be, fx):
    """All 2 x STEPS fixture steps in sequence; after each step the reference's reset bookkeeping is applied on the host
    (goal index / timer / episode sums / episode length zeroed, goals re-gathered), as the kernel's contract asks of the caller.
    Returns one dict of numpy outputs per step (state members as they are BEFORE that host-side reset)."""
    n = fx["goal_cur_goal_idx0"].shape[0]
    env_goals = fx["goal_env_goals"]
    state = dict(episode_length=fx["goal_episode_length0"].astype(np.int64), cur_goal_idx=fx["goal_cur_goal_idx0"].astype(np.int64),
                 reach_goal_timer=fx["goal_timer0"].astype(np.float32), last_contacts=fx["goal_last_contacts0"].astype(np.uint8),
                 cur_goals=fx["goal_cur_goals0"].astype(np.float32), next_goals=fx["goal_next_goals0"].astype(np.float32),
                 episode_sums=np.zeros((len(_capi.TSC_REWARD_NAMES), n), np.float32))
    const = {k: be.put(v) for k, v in dict(env_goals=env_goals, obstacle_types=fx["goal_obstacle_types"].astype(np.int64),
                                           x_edge_mask=fx["goal_x_edge_mask"].astype(np.uint8)).items()}
    f = getattr(be.lib, be.prefix + "tsc_goal_step")
    results = []
    for cam in (0, 1):
        cfg = goal_cfg(fx, n, cam)
        for t in range(STEPS):
            tag = f"goal_c{cam}_t{t}_"
            dev = {k: be.put(v) for k, v in state.items()}
            dev.update(const)
            for k in ("root_states", "contact_forces", "rigid_body_states"):
                dev[k] = be.put(fx[tag + k])
            hist = fx[tag + "action_hl_history"]
            if hist.size:
                dev["action_hl_history"] = be.put(hist)
            for k, (w, dt) in OUT_SHAPES.items():
                dev[k] = be.put(np.zeros((n, w) if w else (n,), dt))
            io = _capi.QaTscGoalIo()
            for name in _capi.TSC_GOAL_IO_FIELDS:
                setattr(io, name, be.ptr(dev[name]) if name in dev else None)
            assert f(C.byref(cfg), C.byref(io), be.stream) == 0
            res = {k: be.get(dev[k]) for k in list(OUT_SHAPES) + list(STATE)}
            results.append(res)
            # host-side reset bookkeeping (tsc legged_robot.py:376, 396-404) and the re-gather of :272-273
            state = {k: res[k].copy() for k in STATE}
            ids = res["reset_buf"].astype(bool)
            state["cur_goal_idx"][ids] = 0
            state["reach_goal_timer"][ids] = 0
            state["episode_sums"][:, ids] = 0
            state["episode_length"][ids] = 0
            state["cur_goals"][ids] = env_goals[ids, 0]
            state["next_goals"][ids] = env_goals[ids, 1]
    return results


def compare_goal_steps(results, fx, rtol=1e-5, atol=2e-6, exact_flags=True):
    """Against the reference's outputs.  Flags and indices exactly; fp32 values to 1e-5.  State members that the
    reference's reset rewrites are compared on the envs that did not reset."""
    i = 0
    for cam in (0, 1):
        for t in range(STEPS):
            tag, res = f"goal_c{cam}_t{t}_", results[i]
            i += 1
            keep = ~fx[tag + "reset_buf"].astype(bool)
            for k in ("reset_buf", "time_out_buf", "reach_goal_cutoff", "reached_goal", "contact_filt", "last_contacts", "cur_obstacle_type"):
                np.testing.assert_array_equal(res[k], fx[tag + k].astype(res[k].dtype), err_msg=tag + k)
            for k in ("base_lin_vel", "base_ang_vel", "projected_gravity", "rpy", "target_pos_rel", "next_target_pos_rel", "target_yaw",
                      "next_target_yaw", "rew_buf"):
                np.testing.assert_allclose(res[k], fx[tag + k], rtol=rtol, atol=atol, err_msg=tag + k)
            np.testing.assert_array_equal(res["cur_goal_idx"][keep], fx[tag + "cur_goal_idx"][keep], err_msg=tag + "cur_goal_idx")
            np.testing.assert_array_equal(res["episode_length"][keep], fx[tag + "episode_length"][keep], err_msg=tag + "episode_length")
            np.testing.assert_array_equal(res["reach_goal_timer"][keep], fx[tag + "timer"][keep], err_msg=tag + "timer")
            np.testing.assert_allclose(res["episode_sums"][:, keep], fx[tag + "episode_sums"][:, keep], rtol=rtol, atol=atol, err_msg=tag + "sums")
            for k in ("cur_goals", "next_goals"):
                np.testing.assert_array_equal(res[k][keep], fx[tag + k][keep], err_msg=tag + k)


# ------------------------------------------------------------------------------------------------- observations
OBS_STEPS = 3


def obs_cfg(fx, n):
    border, hs, vs, lin, ang, dp, dv, lin_d, ang_d, key, foot, clip = (float(v) for v in fx["obs_scalars"])
    c = _capi.QaTscObsCfg()
    c.num_envs, c.num_bodies = n, 19
    for i, b in enumerate(fx["goal_feet"]):
        c.key_bodies[i] = int(b)
    c.map_rows, c.map_cols = fx["obs_height_samples"].shape
    c.root_height_obs, c.action_stride = 1, 8 * 12
    c.points_env_stride, c.point_stride = 132 * 3, 3                  # the reference's (N,132,3) tensor as it is
    c.border_size, c.horizontal_scale, c.vertical_scale = border, hs, vs
    c.lin_vel, c.ang_vel, c.dof_pos, c.dof_vel = lin, ang, dp, dv
    c.lin_vel_dist, c.ang_vel_dist, c.key_pos, c.foot_contact, c.clip_observations = lin_d, ang_d, key, foot, clip
    for i in range(12):
        c.default_dof_pos[i] = float(fx["obs_default_dof_pos"][i])
        c.default_dof_pos_all[i] = float(fx["obs_default_dof_pos_all"][i])
    return c


def run_observations(be, fx, reps=1, prepare_only=False):
    """The fixture's 3 observation steps in sequence (history and yaw errors carried from step to step).  reps > 1 tiles the envs."""
    tile = (lambda a: np.tile(a, (reps,) + (1,) * (a.ndim - 1))) if reps > 1 else (lambda a: a)
    n = fx["obs_history0"].shape[0] * reps
    cfg = obs_cfg(fx, n)
    f = getattr(be.lib, be.prefix + "tsc_observations")
    const = {k: be.put(v) for k, v in dict(
        height_samples=fx["obs_height_samples"].astype(np.int16), height_points=tile(fx["obs_height_points"].astype(np.float32)),
        mass_params=tile(fx["obs_mass_params"]), friction=tile(fx["obs_friction"]),
        motor_strength=np.ascontiguousarray(np.tile(fx["obs_motor_strength"], (1, reps, 1))), commands=tile(fx["obs_commands"]),
        latent_eps=tile(fx["obs_latent_eps"]), latent_c=tile(fx["obs_latent_c"])).items()}
    state = {k: be.put(v) for k, v in dict(obs_history=tile(fx["obs_history0"]), delta_yaw=np.zeros(n, np.float32),
                                           delta_next_yaw=np.zeros(n, np.float32)).items()}
    outs = {k: be.put(np.zeros((n, w), np.float32)) for k, w in dict(measured_heights=132, obs_buf=800, obs_bbc_buf=671, obs_disc_buf=49).items()}
    if prepare_only:
        return cfg, const, state, outs, tile
    results = []
    for t in range(OBS_STEPS):
        io, dev = obs_io(be, fx, cfg, const, state, outs, t, tile)
        assert f(C.byref(cfg), C.byref(io), be.stream) == 0
        results.append({k: be.get(dev[k]) for k in ("measured_heights", "obs_buf", "obs_bbc_buf", "obs_disc_buf", "obs_history", "delta_yaw",
                                                     "delta_next_yaw")})
    return results


def obs_io(be, fx, cfg, const, state, outs, t, tile):
    """qa_tsc_obs_io for fixture step t (sets cfg.update_yaw); returns it with the dict that keeps the buffers alive"""
    tag = f"obs_t{t}_"
    dev = dict(const); dev.update(state); dev.update(outs)
    for k in ("root_states", "rpy", "base_lin_vel", "base_ang_vel", "contact_filt", "dof_pos", "dof_vel", "rigid_body_states", "target_yaw",
              "next_target_yaw"):
        dev[k] = be.put(tile(fx[tag + k]))
    dev["cur_obstacle_type"] = be.put(tile(fx[tag + "cur_obstacle_type"].astype(np.int64)))
    dev["episode_length"] = be.put(tile(fx[tag + "episode_length"].astype(np.int64)))
    dev["_action_history"] = be.put(tile(fx[tag + "action_history"]))     # (N,8,12); the kernel reads the last slot through the stride
    cfg.update_yaw = int(fx[tag + "update_yaw"])
    io = _capi.QaTscObsIo()
    for name in _capi.TSC_OBS_IO_FIELDS:
        if name != "last_action":
            setattr(io, name, be.ptr(dev[name]))
    io.last_action = be.ptr(dev["_action_history"]) + 7 * 12 * 4
    return io, dev


def compare_observations(results, fx, rtol=1e-5, atol=2e-6):
    for t, res in enumerate(results):
        tag = f"obs_t{t}_"
        np.testing.assert_array_equal(res["measured_heights"], fx[tag + "measured_heights"], err_msg=tag + "heights")   # integer cells x scale
        for k in ("delta_yaw", "delta_next_yaw", "obs_buf", "obs_bbc_buf", "obs_disc_buf"):
            np.testing.assert_allclose(res[k], fx[tag + k], rtol=rtol, atol=atol, err_msg=tag + k)
        np.testing.assert_allclose(res["obs_history"], fx[tag + "obs_history"].reshape(res["obs_history"].shape), rtol=rtol, atol=atol,
                                   err_msg=tag + "history")
