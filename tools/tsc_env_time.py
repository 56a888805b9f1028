#!/usr/bin/env python3
"""Time qa_tsc_set_commands / qa_tsc_goal_step at 8192 envs (config 4's env count) with HIP events; prints one JSON line."""
import ctypes as C
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quadrupedal_agility_amd import _capi          # noqa: E402
from tests import tsc_env_protocol as proto        # noqa: E402

lib = _capi.load_library()
fx = proto.load_fixture()
n0 = fx["goal_cur_goal_idx0"].shape[0]
N = 8192
reps = N // n0
be = proto.TorchBackend(lib)
tile = lambda a: np.tile(a, (reps,) + (1,) * (a.ndim - 1))         # noqa: E731
tag = "goal_c0_t0_"
dev = {k: be.put(tile(v)) for k, v in dict(
    root_states=fx[tag + "root_states"], contact_forces=fx[tag + "contact_forces"], rigid_body_states=fx[tag + "rigid_body_states"],
    env_goals=fx["goal_env_goals"], obstacle_types=fx["goal_obstacle_types"].astype(np.int64), action_hl_history=fx[tag + "action_hl_history"],
    episode_length=fx["goal_episode_length0"].astype(np.int64), cur_goal_idx=fx["goal_cur_goal_idx0"].astype(np.int64),
    reach_goal_timer=fx["goal_timer0"], last_contacts=fx["goal_last_contacts0"], cur_goals=fx["goal_cur_goals0"],
    next_goals=fx["goal_next_goals0"]).items()}
dev["x_edge_mask"] = be.put(fx["goal_x_edge_mask"])
dev["episode_sums"] = torch.zeros(8, N, device="cuda")
for k, (w, dt) in proto.OUT_SHAPES.items():
    dev[k] = be.put(np.zeros((N, w) if w else (N,), dt))
cfg = proto.goal_cfg(fx, N, 0)
io = _capi.QaTscGoalIo()
for name in _capi.TSC_GOAL_IO_FIELDS:
    setattr(io, name, dev[name].data_ptr())
bytes_goal = sum(dev[k].numel() * dev[k].element_size() for k in _capi.TSC_GOAL_IO_FIELDS if k not in ("x_edge_mask", "env_goals", "obstacle_types"))


def timeit(fn, iters=200):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
t_goal = timeit(lambda: lib.qa_tsc_goal_step(C.byref(cfg), C.byref(io), st))
acts = be.put(tile(fx["cmd_every_actions"])); ep = be.put(tile(fx["cmd_every_episode_length"].astype(np.int64))); noise = be.put(tile(fx["cmd_every_noise"]))
cmd, eps, lc = be.put(tile(fx["cmd_every_commands0"])), be.put(tile(fx["cmd_every_latent_eps0"])), be.put(tile(fx["cmd_every_latent_c0"]))
nxt = torch.zeros(N, 11, device="cuda")
mi = np.ascontiguousarray(fx["cmd_mocap_index"], np.int32)
vr, jr, hr = (np.ascontiguousarray(fx[k], np.float32) for k in ("cmd_vel_ranges", "cmd_jump_range", "cmd_height_range"))
t_cmd = timeit(lambda: lib.qa_tsc_set_commands(acts.data_ptr(), ep.data_ptr(), N, 3, 6, 5, 1, mi.ctypes.data, vr.ctypes.data, jr.ctypes.data,
                                               hr.ctypes.data, noise.data_ptr(), cmd.data_ptr(), eps.data_ptr(), lc.data_ptr(), nxt.data_ptr(), st))
ocfg, const, state, outs, otile = proto.run_observations(be, fx, reps=reps, prepare_only=True)
oio, okeep = proto.obs_io(be, fx, ocfg, const, state, outs, 2, otile)
t_obs = timeit(lambda: lib.qa_tsc_observations(C.byref(ocfg), C.byref(oio), st))
obs_bytes = N * 4 * ((800 + 671 + 49 + 570 + 132) + (570 + 13 + 9 + 1 + 36 + 4 * 3 + 29 + 11 + 4 + 2))      # written + read per env (scan gathers apart)
print(json.dumps({"num_envs": N, "qa_tsc_observations_us": round(t_obs, 2), "observations_algorithmic_bytes_per_launch": obs_bytes,
                  "observations_GBps": round(obs_bytes / t_obs / 1e3, 1), "observations_frac_of_8TBps": round(obs_bytes / t_obs / 1e3 / 8000, 3), "qa_tsc_goal_step_us": round(t_goal, 2), "goal_step_bytes_per_launch": int(bytes_goal),
                  "goal_step_GBps": round(bytes_goal / t_goal / 1e3, 1), "qa_tsc_set_commands_us": round(t_cmd, 2),
                  "note": "launch-to-launch time of back-to-back launches from the host (includes launch overhead)"}))
