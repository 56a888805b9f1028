"""The mocap reset path (reset_mode 1; BASELINE config 3) on the REAL clips.

CPU: the baked dataset loads into the mirror MotionLoader; the oracle's frame sampling -- clip choice, sample time, float64 frame
indices, fp32 blend, the reference's slerp -- reproduces tests/golden/mocap_reset.npz, i.e. what the reference's OWN
MotionLoader.get_full_frame_at_time_batch and _reset_root_states_mocap / _reset_dofs_mocap return for the same draws
(tools/bake_mocap.py).  GPU (-m gpu): the fused kernel's reset_mode-1 branch against the oracle, single-step from identical
arenas with many resets per step, same tolerances as the default-pose parity test.
"""
import ctypes as C
import os

import numpy as np
import pytest

from quadrupedal_agility_amd import _capi
from tests.oracle_lib import OracleSim, go2_cfg

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "mocap_reset.npz")
CATS = ["walk", "pace", "trot", "canter", "jump"]


def real_clip_table():
    from quadrupedal_agility_amd.rsl_rl.datasets.motion_loader import MotionLoader
    ml = MotionLoader(device="cpu", time_between_frames=0.02, mocap_state_init=True, motion_files_lb=[], motion_files_ulb=[], mocap_category=CATS)
    assert ml.source == "baked", "quadrupedal_agility_amd/resources/go2_mocap.npz is missing"
    return ml, ml.reset_clip_table()


def mocap_oracle(n, seed=1, **over):
    ml, (frames, clips, first) = real_clip_table()
    q = go2_cfg(n, seed=seed, reset_mode=1, num_mocap_frames=int(frames.shape[0]), **over)
    o = OracleSim(q)
    fr = np.ascontiguousarray(frames); ct = np.ascontiguousarray(clips)
    assert o.lib.qo_set_mocap(o.h, fr.ctypes.data, fr.shape[0], ct.ctypes.data, ct.shape[0], (C.c_int32 * 6)(*first), None) == 0
    return q, o, (frames, clips, first)


def test_baked_dataset_is_the_reference_dataset():
    ml, (frames, clips, first) = real_clip_table()
    assert ml.lb.n == 17 and ml.ulb.frames.shape[0] == 39196 and frames.shape == (1196, _capi.MOCAP_FRAME)
    assert first == [0, 3, 6, 9, 13, 17]                                   # walk 3, pace 3, trot 3, canter 4, jump 4
    assert np.allclose(np.linalg.norm(frames[:, 3:7], axis=1), 1.0, atol=1e-6) and (frames[:, 6] >= 0).all()
    assert (clips[:, 3] < clips[:, 2]).all() and np.allclose(clips[[f - 1 for f in first[1:]], 4], 1.0)


def test_clip_table_rejects_bad_input():
    q, o, (frames, clips, first) = mocap_oracle(4)
    fr = np.ascontiguousarray(frames); ct = np.ascontiguousarray(clips)
    bad = list(first); bad[2] = bad[1]                                       # a gait without clips
    assert o.lib.qo_set_mocap(o.h, fr.ctypes.data, fr.shape[0], ct.ctypes.data, ct.shape[0], (C.c_int32 * 6)(*bad), None) != 0
    assert o.lib.qo_set_mocap(o.h, fr.ctypes.data, fr.shape[0] + 1, ct.ctypes.data, ct.shape[0], (C.c_int32 * 6)(*first), None) != 0


def test_oracle_frame_sampling_matches_the_reference():
    g = np.load(GOLD)
    q, o, (frames, clips, first) = mocap_oracle(4)
    o.lib.qo_debug_mocap_reset.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_void_p]
    n = len(g["gait"])
    out = np.zeros((n, 37), np.float32)
    for i in range(n):
        assert o.lib.qo_debug_mocap_reset(o.h, int(g["gait"][i]), float(g["u0"][i]), float(g["u1"][i]), out[i].ctypes.data) == 0
    # the clip the reference's weighted choice picked: compare through the frames (a wrong clip is off by far more than 1e-3)
    assert np.allclose(out[:, 0:13], g["root_state"], atol=2e-6, rtol=1e-5)
    assert np.allclose(out[:, 13:25], g["dof_pos"], atol=2e-6, rtol=1e-5)
    assert np.allclose(out[:, 25:37], g["dof_vel"], atol=2e-5, rtol=1e-5)
    assert len(set(g["traj_names"])) == 17                                     # every labelled clip is hit
    # the reference's slerp divides by the angle, not its sine: quaternions come out (slightly) non-unit, as in the reference
    nrm = np.linalg.norm(out[:, 3:7], axis=1)
    assert np.allclose(nrm, np.linalg.norm(g["root_state"][:, 3:7], axis=1), atol=1e-6) and nrm.min() > 0.99


def test_mirror_loader_agrees_too():
    """the host-side MotionLoader (expert pairs for the discriminator) blends frames like the reference as well"""
    import torch
    g = np.load(GOLD)
    ml, _ = real_clip_table()
    names = [os.path.basename(n) for n in g["traj_names"]]
    mine = {c: i for i, c in enumerate(ml.lb_names)}
    traj = np.array([mine[n] for n in names])
    fr = ml.get_full_frame_at_time_batch(traj, g["times"], labeled=True)
    assert torch.allclose(fr[:, :7], torch.from_numpy(g["root_state"][:, :7]), atol=2e-6)
    assert torch.allclose(fr[:, 7:19], torch.from_numpy(g["dof_pos"]), atol=2e-6)


def test_env_resets_from_real_clips_on_cpu():
    """whole env step with reset_mode 1: resetting envs land on frames of their gait's clips"""
    q, o, (frames, clips, first) = mocap_oracle(256, seed=3)
    o.reset_all()
    root, dof = o.t["ROOT_STATES"], o.t["DOF_STATE"]
    gait = o.t["LATENT_C"].argmax(1)
    z = root[:, 2] - o.t["ENV_ORIGINS"][:, 2]
    assert 0.15 < z.min() and z.max() < 0.7
    for e in range(256):
        lo, hi = int(clips[first[gait[e]], 0]), int(clips[first[gait[e] + 1] - 1, 0] + clips[first[gait[e] + 1] - 1, 1])
        d = np.abs(frames[lo:hi, 7:19] - dof[e, :, 0]).max(axis=1)
        step = np.abs(np.diff(frames[lo:hi, 7:19], axis=0)).max()
        assert d.min() <= step, (e, d.min(), step)        # between two neighbouring frames of one of the gait's clips


# ------------------------------------------------------------------------------------------------------------------ GPU
@pytest.mark.gpu
@pytest.mark.parametrize("n_envs,seed", [(64, 1), (1000, 7)])
def test_single_step_parity_with_mocap_reset(n_envs, seed):
    import torch
    from quadrupedal_agility_amd.sim import QaSim
    from tests.test_hip_parity import TOL, env_mismatch, push_arena
    q, o, (frames, clips, first) = mocap_oracle(n_envs, seed=seed)
    h = QaSim(q)
    h.set_mocap(frames, clips, first)
    rng = np.random.default_rng(seed)
    o.reset_all()
    o.t["EPISODE_LENGTH"][:] = rng.integers(0, 1000, n_envs)
    o.global_step = 380
    flips = resets = 0
    steps = 40
    worst = {}
    for k in range(steps):
        # force extra time-outs so that every step resets ~5 % of the envs through the mocap branch
        o.t["EPISODE_LENGTH"][rng.random(n_envs) < 0.05] = 1000
        push_arena(o, h)
        act = rng.normal(0, 1.0, (n_envs, 12)).astype(np.float32)
        o.step(act); h.step(torch.from_numpy(act).cuda()); torch.cuda.synchronize()
        bad_env = np.zeros(n_envs, bool)
        for name in TOL:
            got = h.t[name].cpu().numpy(); exp = o.t[name]
            worst[name] = max(worst.get(name, 0.0), float(np.abs(got.astype(np.float64) - exp.astype(np.float64)).max()))
            bad_env |= env_mismatch(name, got, exp, n_envs)
        rs = o.t["RESET"] != 0
        resets += int(rs.sum())
        # the reset state itself (mocap frame blend, slerp) must agree tightly wherever both sides reset
        both = rs & (h.t["RESET"].cpu().numpy() != 0)
        assert np.allclose(h.t["ROOT_STATES"].cpu().numpy()[both], o.t["ROOT_STATES"][both], atol=3e-6, rtol=1e-5)
        assert np.allclose(h.t["DOF_STATE"].cpu().numpy()[both], o.t["DOF_STATE"][both], atol=3e-5, rtol=1e-5)
        flips += int(bad_env.sum())
    print(f"mocap-reset parity: {resets} resets, env-steps outside tolerance: {flips} of {steps * n_envs};", {k: f"{v:.1e}" for k, v in worst.items() if v > 0})
    assert resets > steps * n_envs * 0.04
    from tests.test_hip_parity import BUDGET, check_flips
    check_flips(f"mocap_{n_envs}", flips, steps * n_envs, BUDGET["mocap"])
