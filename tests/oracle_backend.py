"""Test-only engine with QaSim's interface, backed by the CPU oracle, so that the host-side mirror
(LeggedRobot / OnPolicyRunner / SSInfoGAIL) can be exercised end to end without a GPU -- BASELINE
config 1 ("64 envs, CPU physics + CPU torch, plumbing").  Never imported by the product."""
import ctypes as C

import numpy as np
import torch

from tests.oracle_lib import OracleSim


class OracleBackend:
    def __init__(self, qcfg):
        self.o = OracleSim(qcfg)
        self.cfg = qcfg
        self.device = torch.device("cpu")
        self.t = {k: torch.from_numpy(v) for k, v in self.o.t.items()}     # zero-copy views of the host arena
        self.global_step = 0

    def reset_all(self):
        self.o.global_step = self.global_step
        self.o.reset_all()

    def step(self, actions, delay=0):
        self.o.global_step = self.global_step
        self.o.step(actions.detach().cpu().numpy(), delay)
        self.global_step += 1

    def physics_step(self, actions, delay=0):
        a = np.ascontiguousarray(actions.detach().cpu().numpy(), dtype=np.float32)
        assert self.o.lib.qo_env_physics_step(self.o.h, a.ctypes.data, int(delay), None) == 0

    def tsc_reset(self, flags, start_xy, start_yaw, yaw_range, x_range, y_range, pitch_range, step):
        f, xy, yw = flags.contiguous(), start_xy.contiguous(), start_yaw.contiguous()
        assert self.o.lib.qo_tsc_reset(self.o.h, f.data_ptr(), xy.data_ptr(), yw.data_ptr(), C.c_float(yaw_range), C.c_float(x_range),
                                       C.c_float(y_range), C.c_float(pitch_range), C.c_int64(int(step)), None) == 0     # int(tensor) reads the 1-element step tensor

    def simulate_if(self, torques, cond):
        assert self.o.lib.qo_simulate_if(self.o.h, torques.data_ptr() if torques is not None else None, cond.data_ptr(), None) == 0

    def set_mocap(self, frames, clips, first_clip):
        f = np.ascontiguousarray(frames, dtype=np.float32)
        ct = np.ascontiguousarray(clips, dtype=np.float64)
        first = (C.c_int32 * 6)(*[int(x) for x in first_clip])
        assert self.o.lib.qo_set_mocap(self.o.h, f.ctypes.data, f.shape[0], ct.ctypes.data, ct.shape[0], first, None) == 0

    def gae(self, rewards, values, dones, last_values, returns, advantages, gamma, lam, normalize=True):
        T, N = rewards.shape
        p = lambda x: x.numpy().ctypes.data
        rc = self.o.lib.qo_gae(p(rewards), p(values), p(dones), p(last_values), p(returns), p(advantages), T, N,
                               float(gamma), float(lam), int(bool(normalize)), None, None)
        assert rc == 0
