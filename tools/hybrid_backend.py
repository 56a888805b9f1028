"""Checker-side engine (r5) with QaSim's interface: the ORACLE's physics on the host cores, the arena MIRRORED on the GPU.

What it is for: the north-star parity criterion compares return curves of the HIP trainer with those of the CPU physics path.  The all-CPU arm
(oracle physics + torch-CPU learner, tools/return_curve_parity.py --side cpu) costs ~2 h per seed, 90 % of it the torch-CPU learner, which is
not what the arm is meant to test.  This backend keeps the oracle's physics (double precision, dense solve, OpenMP over envs) and lets the
PRODUCT's learner run on the GPU: the env's tensors are device views of a mirror of the host arena; a step copies the mirror back (the
learner writes PRIOR_PARAMETERS / EPISODE_LENGTH there), steps the oracle on the actions, and copies the arena forward.  Two PCIe crossings
of a few MB per env step.  Never imported by the product (it loads the oracle)."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quadrupedal_agility_amd import _capi                      # noqa: E402
from tests.oracle_lib import OracleSim                         # noqa: E402

_TORCH_DT = {_capi.DTYPE_F32: torch.float32, _capi.DTYPE_I64: torch.int64, _capi.DTYPE_U8: torch.uint8,
             _capi.DTYPE_I32: torch.int32, _capi.DTYPE_I16: torch.int16, _capi.DTYPE_F64: torch.float64}
_ITEM = {_capi.DTYPE_F32: 4, _capi.DTYPE_I64: 8, _capi.DTYPE_U8: 1, _capi.DTYPE_I32: 4, _capi.DTYPE_I16: 2, _capi.DTYPE_F64: 8}


class HybridBackend:
    def __init__(self, qcfg, device="cuda:0"):
        self.o = OracleSim(qcfg)
        self.cfg = qcfg
        self.device = torch.device(device)
        self.lib = _capi.load_library()                      # qa_gae of the product (the learner's side of the seam)
        n = self.o.arena.nbytes
        self._host = torch.from_numpy(self.o.arena)          # zero-copy view of the oracle's arena
        self._pin = torch.empty(n, dtype=torch.uint8).pin_memory()
        slab = torch.zeros(n + 256, dtype=torch.uint8, device=self.device)
        shift = (-slab.data_ptr()) % 256
        self._slab, self.arena = slab, slab[shift:shift + n]
        self.t = {}
        for name, idx in _capi.T.items():
            off, shape, dt = _capi.tensor_info(self.o.lib, "qo_", qcfg, idx)
            k = 1
            for s in shape:
                k *= s
            self.t[name] = self.arena[off:off + k * _ITEM[dt]].view(_TORCH_DT[dt]).view(*shape)
        self._gae_scratch = torch.zeros(4096, dtype=torch.uint8, device=self.device)
        self.global_step = 0
        self.lean_exports = 0
        self._forward()

    def _forward(self):          # host arena -> device mirror
        self._pin.copy_(self._host)
        self.arena.copy_(self._pin, non_blocking=True)
        torch.cuda.current_stream(self.device).synchronize()

    def _back(self):             # device mirror -> host arena (whatever the learner wrote into the env's tensors)
        self._pin.copy_(self.arena)                          # synchronises with the stream
        self._host.copy_(self._pin)

    def reset_all(self):
        self._back()
        self.o.global_step = self.global_step
        self.o.reset_all()
        self._forward()

    def step(self, actions, delay=0):
        a = actions.detach().to("cpu", torch.float32).numpy()
        self._back()
        self.o.global_step = self.global_step
        self.o.step(a, delay)
        self.global_step += 1
        self._forward()

    def step_dev(self, actions, delay, step_counter):
        raise RuntimeError("the hybrid arm steps from the host: run it with QA_ROLLOUT_GRAPH=0")

    def set_lean_exports(self, mask):      # the oracle always writes every tensor
        self.lean_exports = 0

    def set_mocap(self, frames, clips, first_clip):
        f = np.ascontiguousarray(frames, dtype=np.float32)
        ct = np.ascontiguousarray(clips, dtype=np.float64)
        first = (C.c_int32 * (_capi.NUM_GAITS + 1))(*[int(x) for x in first_clip])
        self._back()
        assert self.o.lib.qo_set_mocap(self.o.h, f.ctypes.data, f.shape[0], ct.ctypes.data, ct.shape[0], first, None) == 0
        self._forward()          # the clip table lives in the arena: the mirror must hold it too, or the next _back() would erase it

    def gae(self, rewards, values, dones, last_values, returns, advantages, gamma, lam, normalize=True):
        T, N = rewards.shape[0], rewards.shape[1]
        rc = self.lib.qa_gae(rewards.data_ptr(), values.data_ptr(), dones.data_ptr(), last_values.data_ptr(), returns.data_ptr(), advantages.data_ptr(),
                             T, N, float(gamma), float(lam), int(bool(normalize)), self._gae_scratch.data_ptr(),
                             C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream))
        assert rc == 0, self.lib.qa_last_error()
