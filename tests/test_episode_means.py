"""qa_episode_means (ABI 13): extras["episode"] of reset_idx (bbc/legged_gym/envs/base/legged_robot.py:229-240) from the step's EPISODE_STATS bin in
one launch.  The C twin against the expression the env used to evaluate with torch ops; the HIP kernel against the twin; the env's extras
through a recorded-style device step counter."""
import ctypes as C

import numpy as np
import pytest
import torch

from tests.oracle_lib import load_oracle


def torch_expression(stats, step, means, max_len_s):
    st = stats[(step - 1) & 1]
    cnt = st[14]
    mean = st[:14] / torch.clamp(cnt, min=1.0) / max_len_s
    return torch.where(cnt > 0, mean, means)


@pytest.mark.parametrize("step,cnt", [(7, 3.0), (8, 0.0), (1, 1.0)])
def test_twin_matches_the_torch_expression(step, cnt):
    lib = load_oracle()
    g = torch.Generator().manual_seed(step)
    stats = torch.randn(2, 16, generator=g) * 30
    stats[(step - 1) & 1, 14] = cnt
    stats[step & 1, 14] = 5.0                      # the OTHER bin must not be read
    means = torch.randn(14, generator=g)
    want = torch_expression(stats, step, means.clone(), 20.0)
    for dev_step in (False, True):
        m, snap = means.clone(), torch.zeros(14)
        ctr = torch.tensor([step], dtype=torch.int64)
        rc = lib.qo_episode_means(stats.data_ptr(), ctr.data_ptr() if dev_step else None, 0 if dev_step else step, 14, 20.0, m.data_ptr(), snap.data_ptr(), None)
        assert rc == 0
        assert torch.allclose(m, want, rtol=1e-6, atol=0) and torch.equal(m, snap)
    assert lib.qo_episode_means(stats.data_ptr(), None, step, 15, 20.0, means.data_ptr(), means.data_ptr(), None) != 0


@pytest.mark.gpu
def test_hip_matches_twin_and_env_extras_follow_it():
    from quadrupedal_agility_amd import _capi
    lib, olib = _capi.load_library(), load_oracle()
    for step, cnt in ((11, 4.0), (12, 0.0)):
        g = torch.Generator().manual_seed(step)
        stats = torch.randn(2, 16, generator=g) * 30
        stats[(step - 1) & 1, 14] = cnt
        means = torch.randn(14, generator=g)
        m_ref, s_ref = means.clone(), torch.zeros(14)
        assert olib.qo_episode_means(stats.data_ptr(), None, step, 14, 20.0, m_ref.data_ptr(), s_ref.data_ptr(), None) == 0
        sd, md, snd = stats.cuda(), means.cuda(), torch.zeros(14, device="cuda")
        ctr = torch.tensor([step], dtype=torch.int64, device="cuda")
        assert lib.qa_episode_means(sd.data_ptr(), ctr.data_ptr(), 0, 14, 20.0, md.data_ptr(), snd.data_ptr(), None) == 0
        torch.cuda.synchronize()
        assert torch.equal(md.cpu(), m_ref) and torch.equal(snd.cpu(), s_ref)
    # the env: extras["episode"] after real steps equal the torch expression on the arena's own statistics
    from quadrupedal_agility_amd.legged_gym.envs import task_registry
    from quadrupedal_agility_amd.legged_gym.envs.go2.go2_locomotion_config import Go2LocomotionCfg
    from quadrupedal_agility_amd.legged_gym.utils import get_args
    cfg = Go2LocomotionCfg(); cfg.env.num_envs = 256; cfg.terrain.mesh_type = "plane"; cfg.env.mocap_state_init = False; cfg.seed = 3
    env, _ = task_registry.make_env("go2_locomotion", args=get_args(["--device", "gpu"]), env_cfg=cfg)
    env.episode_length_buf[:] = torch.randint(900, 1001, (256,), device=env.device)       # time-outs within the next steps
    prev = env._episode_means.clone()
    for k in range(6):
        env.step(torch.randn(256, 12, device=env.device) * 0.3)
        want = torch_expression(env.sim.t["EPISODE_STATS"], env.common_step_counter, prev, env.max_episode_length_s)
        got = torch.stack([env.extras["episode"]["rew_" + n] for n in env.reward_names])
        assert torch.allclose(got, want[[_capi.REWARD_NAMES.index(n) for n in env.reward_names]], rtol=1e-6, atol=0), k
        prev = env._episode_means.clone()
    assert float(env.sim.t["EPISODE_STATS"][:, 14].sum()) > 0
