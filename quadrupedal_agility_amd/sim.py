"""Thin torch-side owner of the device arena behind include/qa_sim.h.

torch is plumbing here: it allocates the slab (so the caching allocator accounts for it),
provides the stream, and exposes zero-copy views of the engine's tensors -- the same role
`gymtorch.wrap_tensor` plays in the reference (legged_robot.py:757-770).
"""
import ctypes as C

import torch

from . import _capi

_TORCH_DT = {_capi.DTYPE_F32: torch.float32, _capi.DTYPE_I64: torch.int64, _capi.DTYPE_U8: torch.uint8,
             _capi.DTYPE_I32: torch.int32, _capi.DTYPE_I16: torch.int16, _capi.DTYPE_F64: torch.float64}
_ITEM = {_capi.DTYPE_F32: 4, _capi.DTYPE_I64: 8, _capi.DTYPE_U8: 1, _capi.DTYPE_I32: 4, _capi.DTYPE_I16: 2, _capi.DTYPE_F64: 8}


def _check(rc, lib, what):
    if rc != 0:
        raise RuntimeError(f"{what} failed with code {rc}: {lib.qa_last_error().decode()}")


class QaSim:
    """One handle = one set of envs on one GPU."""

    def __init__(self, qcfg, device="cuda:0"):
        self.lib = _capi.load_library()          # raises if the HIP library is missing
        if not torch.cuda.is_available():
            raise RuntimeError("quadrupedal_agility_amd needs a ROCm GPU; there is no CPU path")
        self.cfg = qcfg
        self.device = torch.device(device)
        nbytes = self.lib.qa_arena_bytes(C.byref(qcfg))
        if nbytes <= 0:
            raise RuntimeError(f"qa_arena_bytes -> {nbytes}")
        with torch.cuda.device(self.device):
            # over-allocate so that the base can be rounded up to 256 B whatever the allocator returns
            self._slab = torch.zeros(nbytes + 256, dtype=torch.uint8, device=self.device)
            shift = (-self._slab.data_ptr()) % 256
            self.arena = self._slab[shift:shift + nbytes]
            h = C.c_void_p()
            _check(self.lib.qa_create(C.byref(qcfg), self.arena.data_ptr(), nbytes, self._stream(), C.byref(h)),
                   self.lib, "qa_create")
        self.h = h
        self.t = {}
        for name, idx in _capi.T.items():
            off, shape, dt = _capi.tensor_info(self.lib, "qa_", qcfg, idx)
            n = 1
            for s in shape:
                n *= s
            self.t[name] = self.arena[off:off + n * _ITEM[dt]].view(_TORCH_DT[dt]).view(*shape)
        self._gae_scratch = torch.zeros(4096, dtype=torch.uint8, device=self.device)
        self.global_step = 0

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def reset_all(self):
        _check(self.lib.qa_reset_all(self.h, self.global_step, self._stream()), self.lib, "qa_reset_all")

    def step(self, actions, delay=0):
        assert actions.is_cuda and actions.dtype == torch.float32 and actions.is_contiguous()
        assert actions.shape == (self.cfg.num_envs, 12)
        _check(self.lib.qa_env_step(self.h, actions.data_ptr(), int(delay), self.global_step, self._stream()),
               self.lib, "qa_env_step")
        self.global_step += 1

    def set_lean_exports(self, mask):
        """qa_set_lean_exports: 0 = every tensor refreshed each step (default), 1 = seam-1 / logging exports skipped + two-slot action ring,
        3 = discriminator observations skipped too (training without AMP)"""
        _check(self.lib.qa_set_lean_exports(self.h, int(mask)), self.lib, "qa_set_lean_exports")
        self.lean_exports = int(mask)

    def step_dev(self, actions, delay, step_counter):
        """qa_env_step_dev: the step counter lives in device memory (graph-replayable)."""
        assert actions.is_cuda and actions.dtype == torch.float32 and actions.is_contiguous()
        assert step_counter.is_cuda and step_counter.dtype == torch.int64
        _check(self.lib.qa_env_step_dev(self.h, actions.data_ptr(), int(delay), step_counter.data_ptr(), self._stream()),
               self.lib, "qa_env_step_dev")

    def physics_step(self, actions, delay=0):
        """qa_env_physics_step: the step without post_physics_step (the task-level env brings its own)"""
        assert actions.is_cuda and actions.dtype == torch.float32 and actions.is_contiguous() and actions.shape == (self.cfg.num_envs, 12)
        _check(self.lib.qa_env_physics_step(self.h, actions.data_ptr(), int(delay), self._stream()), self.lib, "qa_env_physics_step")

    def tsc_reset(self, flags, start_xy, start_yaw, yaw_range, x_range, y_range, pitch_range, step):
        """`step`: the draws' key -- an int, or a 1-element int64 device tensor read when the launch executes (recorded rollouts)"""
        assert flags.is_cuda and flags.dtype == torch.uint8 and start_xy.is_contiguous() and start_yaw.is_contiguous()
        if torch.is_tensor(step):
            assert step.is_cuda and step.dtype == torch.int64
            return _check(self.lib.qa_tsc_reset_dev(self.h, flags.data_ptr(), start_xy.data_ptr(), start_yaw.data_ptr(), float(yaw_range), float(x_range),
                                                    float(y_range), float(pitch_range), step.data_ptr(), self._stream()), self.lib, "qa_tsc_reset_dev")
        _check(self.lib.qa_tsc_reset(self.h, flags.data_ptr(), start_xy.data_ptr(), start_yaw.data_ptr(), float(yaw_range), float(x_range),
                                     float(y_range), float(pitch_range), int(step), self._stream()), self.lib, "qa_tsc_reset")

    def simulate_if(self, torques, cond):
        """one more substep for every env iff the device byte `cond` is non-zero; torques None = the ones applied last"""
        assert cond.is_cuda and cond.dtype == torch.uint8
        _check(self.lib.qa_simulate_if(self.h, torques.data_ptr() if torques is not None else None, cond.data_ptr(), self._stream()),
               self.lib, "qa_simulate_if")

    def simulate(self, torques):
        assert torques.is_cuda and torques.dtype == torch.float32 and torques.is_contiguous()
        _check(self.lib.qa_simulate(self.h, torques.data_ptr(), self._stream()), self.lib, "qa_simulate")

    def set_mocap(self, frames, clips, first_clip):
        """qa_set_mocap: frames (F,37) fp32, clip table (C,8) float64, clip ranges per gait (MotionLoader.reset_clip_table)"""
        import numpy as np
        f = np.ascontiguousarray(frames, dtype=np.float32)
        ct = np.ascontiguousarray(clips, dtype=np.float64)
        assert f.ndim == 2 and f.shape[1] == _capi.MOCAP_FRAME and ct.ndim == 2 and ct.shape[1] == _capi.MOCAP_CLIP
        first = (C.c_int32 * (_capi.NUM_GAITS + 1))(*[int(x) for x in first_clip])
        _check(self.lib.qa_set_mocap(self.h, f.ctypes.data, f.shape[0], ct.ctypes.data, ct.shape[0], first, self._stream()), self.lib, "qa_set_mocap")
        torch.cuda.current_stream(self.device).synchronize()   # host buffers must outlive the async copies

    def debug_post_physics(self, step):
        """qa_debug_post_physics: post_physics_step alone on the arena's state (verification entry)"""
        _check(self.lib.qa_debug_post_physics(self.h, int(step), self._stream()), self.lib, "qa_debug_post_physics")

    def gae(self, rewards, values, dones, last_values, returns, advantages, gamma, lam, normalize=True):
        T, N = rewards.shape[0], rewards.shape[1]
        for x in (rewards, values, dones, last_values, returns, advantages):
            assert x.is_cuda and x.is_contiguous()
        assert dones.dtype == torch.uint8
        _check(self.lib.qa_gae(rewards.data_ptr(), values.data_ptr(), dones.data_ptr(), last_values.data_ptr(),
                               returns.data_ptr(), advantages.data_ptr(), T, N, float(gamma), float(lam),
                               int(bool(normalize)), self._gae_scratch.data_ptr(), self._stream()),
               self.lib, "qa_gae")

    def close(self):
        if getattr(self, "h", None) is not None:
            self.lib.qa_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
