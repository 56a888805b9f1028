"""Learner-side mirror (quadrupedal_agility_amd.rsl_rl) against golden vectors produced by the reference's own
rsl_rl code (tools/gen_golden.py): network forwards from a fixed state_dict, discriminator reward, GAE, one PPO
step and one discriminator step (losses + post-step weights), mocap re-ordering and frame blending."""
import os
import types

import numpy as np
import pytest
import torch

from quadrupedal_agility_amd.legged_gym.envs.go2.go2_locomotion_config import Go2LocomotionCfg
from quadrupedal_agility_amd.rsl_rl.algorithms import SSInfoGAIL
from quadrupedal_agility_amd.rsl_rl.algorithms.discriminator import Discriminator
from quadrupedal_agility_amd.rsl_rl.datasets.motion_loader import MotionLoader, load_clip
from quadrupedal_agility_amd.rsl_rl.modules import ActorCritic, Estimator
from quadrupedal_agility_amd.rsl_rl.storage import RolloutStorage
from quadrupedal_agility_amd.rsl_rl.utils.utils import Normalizer, TorchNormalizer

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TOL = dict(atol=2e-5, rtol=1e-4)


@pytest.fixture(scope="module")
def fx():
    return torch.load(os.path.join(GOLD, "learner.pt"), weights_only=False)


def _env(task_obs_weight):
    return types.SimpleNamespace(cfg=Go2LocomotionCfg(), dim_c=5, num_obs_disc=49, num_envs=32, task_obs_weight_decay=True,
                                 task_obs_weight=task_obs_weight, latent_eps=torch.zeros(32, 1), latent_c=torch.zeros(32, 5),
                                 prior_parameters=torch.ones(5) / 5, dt=0.02)


def _nets(fx):
    ac = ActorCritic(101, 671, 12, 57, 10, 4, 29, 11, **fx["policy_cfg"])
    est = Estimator(input_dim=57, output_dim=4, hidden_dims=fx["estimator_cfg"]["hidden_dims"])
    env = _env(fx["task_obs_weight"])
    disc = Discriminator(env, 98, 49, 5, 0.02, "MSELoss", None, 1.0, 0.01, 0.2, 0.2, 2, 2, 0.0, [512, 256], "cpu")
    ac.load_state_dict(fx["init"]["actor_critic"]); est.load_state_dict(fx["init"]["estimator"]); disc.load_state_dict(fx["init"]["disc"])
    norm = Normalizer(98)
    norm.mean, norm.var, norm.count = fx["normalizer"]["mean"].copy(), fx["normalizer"]["var"].copy(), fx["normalizer"]["count"]
    return env, ac, est, disc, norm


def test_state_dict_names_and_shapes_are_the_references(fx):
    _, ac, est, disc, _ = _nets(fx)       # load_state_dict(strict) already proves names; check counts too
    assert sum(p.numel() for p in ac.parameters()) == 735699
    assert sum(p.numel() for p in est.parameters()) == 15940
    assert sum(p.numel() for p in disc.parameters()) == 183815


def test_network_forwards(fx):
    _, ac, est, disc, _ = _nets(fx)
    obs, actions = fx["obs"], fx["actions"]
    with torch.no_grad():
        assert torch.allclose(ac.act_inference(obs, hist_encoding=False), fx["act_inference_priv"], **TOL)
        assert torch.allclose(ac.act_inference(obs, hist_encoding=True), fx["act_inference_hist"], **TOL)   # conv-as-GEMM path
        assert torch.allclose(ac.infer_hist_latent(obs[:, 90:660]), fx["hist_latent"], **TOL)
        assert torch.allclose(ac.evaluate(obs), fx["value"], **TOL)
        ac.update_distribution(obs, False)
        assert torch.allclose(ac.get_actions_log_prob(actions), fx["log_prob"], **TOL)
        assert torch.allclose(ac.entropy, fx["entropy"], **TOL)
        assert torch.allclose(est(obs[:, :57]), fx["estimator_out"], **TOL)
        d, eps, c = disc(fx["disc_in"])
        assert torch.allclose(d, fx["disc_d"], **TOL) and torch.allclose(eps, fx["disc_eps"], **TOL) and torch.allclose(c, fx["disc_c"], **TOL)


@pytest.mark.parametrize("device_norm", [False, True])
def test_predict_disc_reward(fx, device_norm):
    _, _, _, disc, norm = _nets(fx)
    n = TorchNormalizer.from_reference(norm, "cpu") if device_norm else norm
    out = disc.predict_disc_reward(fx["pdr_reward_t"], fx["obs"], fx["pdr_obs_disc"], normalizer=n)
    for got, exp in zip(out, fx["pdr_out"]):
        assert torch.allclose(got, exp.to(got.dtype), atol=2e-6, rtol=1e-4)


def test_gae_eager_storage_path():
    g = np.load(os.path.join(GOLD, "gae.npz"))
    for i in range(int(g["num_cases"])):
        T, N = g[f"c{i}_rewards"].shape[:2]
        st = RolloutStorage(N, T, [4], [4], [2], "cpu")
        st.rewards[:] = torch.tensor(g[f"c{i}_rewards"]); st.values[:] = torch.tensor(g[f"c{i}_values"]); st.dones[:] = torch.tensor(g[f"c{i}_dones"])
        st.compute_returns(torch.tensor(g[f"c{i}_last"]), 0.99, 0.95)
        assert torch.allclose(st.returns, torch.tensor(g[f"c{i}_returns"]), atol=1e-6)
        assert torch.allclose(st.advantages, torch.tensor(g[f"c{i}_advantages"]), atol=1e-5)


def test_gae_oracle_kernel_twin(oracle_lib):
    """qo_gae (whose HIP twin qa_gae is what the GPU path runs) against the reference's compute_returns."""
    g = np.load(os.path.join(GOLD, "gae.npz"))
    for i in range(int(g["num_cases"])):
        rew = np.ascontiguousarray(g[f"c{i}_rewards"][..., 0]); val = np.ascontiguousarray(g[f"c{i}_values"][..., 0])
        done = np.ascontiguousarray(g[f"c{i}_dones"][..., 0]); last = np.ascontiguousarray(g[f"c{i}_last"][:, 0])
        T, N = rew.shape
        ret = np.zeros_like(rew); adv = np.zeros_like(rew)
        assert oracle_lib.qo_gae(rew.ctypes.data, val.ctypes.data, done.ctypes.data, last.ctypes.data, ret.ctypes.data, adv.ctypes.data,
                                 T, N, 0.99, 0.95, 1, None, None) == 0
        assert np.allclose(ret, g[f"c{i}_returns"][..., 0], atol=1e-6)
        assert np.allclose(adv, g[f"c{i}_advantages"][..., 0], atol=1e-5)


def _check_compact(sd, compact, atol):
    for k, c in compact.items():
        v = sd[k]
        assert torch.allclose(v.flatten()[::97], c["sample"], atol=atol, rtol=1e-4), k
        assert v.double().sum().item() == pytest.approx(c["sum"], abs=atol * v.numel() ** 0.5 * 10 + 1e-6), k


def _alg(fx):
    env, ac, est, disc, norm = _nets(fx)
    tn = TorchNormalizer.from_reference(norm, "cpu")
    alg = SSInfoGAIL(env, ac, disc, est, fx["estimator_cfg"], types.SimpleNamespace(), tn, 2, 2, 49, 0.0, device="cpu",
                     min_std=torch.full((12,), 0.05), **fx["alg_cfg"])
    return env, alg


def test_one_ppo_step_losses_and_weights(fx):
    env, alg = _alg(fx)
    alg.priv_reg_counter = 1500
    sample = tuple(fx["ppo_sample"]) + ((None, None), None)
    losses = alg.update_actor_critic(sample)
    for got, exp in zip(losses, fx["ppo_losses"]):
        assert float(got) == pytest.approx(float(exp), rel=2e-4, abs=2e-6)
    assert alg.lr_ac == pytest.approx(fx["ppo_lr_after"])                    # adaptive-KL schedule took the same branch
    _check_compact(alg.actor_critic.state_dict(), fx["ppo_after"]["actor_critic"], 3e-6)
    _check_compact(alg.estimator.state_dict(), fx["ppo_after"]["estimator"], 3e-6)


def test_one_discriminator_step_losses_and_weights(fx):
    env, alg = _alg(fx)
    alg.info_max_coef_on = 0.3
    alg.actor_critic.std.data[:3] = 0.01                        # below min_std: the step must clamp it (gail.py:522-523)
    s = fx["disc_samples"]
    out = alg.update_ss_info_gail(tuple(s["policy"]), tuple(s["lb"]), s["ulb"])
    for got, exp in zip(out, fx["disc_losses"]):
        assert float(got) == pytest.approx(float(exp), rel=3e-4, abs=3e-6)
    _check_compact(alg.disc.state_dict(), fx["disc_after"], 5e-6)
    assert torch.allclose(env.prior_parameters, fx["prior_after"], atol=1e-7)
    assert torch.allclose(alg.actor_critic.std.detach()[:3], torch.full((3,), 0.05)) and (alg.actor_critic.std.detach()[3:] == 1.0).all()
    ref = alg.disc_normalizer.to_reference()
    assert np.allclose(ref.mean, fx["normalizer_after"]["mean"], atol=1e-6) and np.allclose(ref.var, fx["normalizer_after"]["var"], rtol=1e-5)
    assert ref.count == pytest.approx(fx["normalizer_after"]["count"])


def test_mocap_reorder_and_frame_blending():
    g = np.load(os.path.join(GOLD, "mocap.npz"))
    clip = load_clip(os.path.join(GOLD, "trot_clip40.json"))
    assert np.allclose(clip["frames"][:, :49], g["reordered"], atol=1e-6)
    q0 = torch.tensor([[0, 0.9, -1.8] * 4])
    cfg = Go2LocomotionCfg()
    ml = MotionLoader("cpu", 0.02, mocap_state_init=True, motion_files_lb=[os.path.join(GOLD, "trot_clip40.json")], motion_files_ulb=[],
                      mocap_category=cfg.env.mocap_category, default_dof_pos=q0, obs_scales=cfg.normalization.obs_scales)
    fr = ml.get_full_frame_at_time_batch(np.zeros(len(g["times"]), dtype=np.int64), g["times"], labeled=True)
    assert np.allclose(fr.numpy(), g["frames"], atol=2e-6)


def test_analytic_input_gradient_equals_autograd_double_backward():
    """Discriminator.forward_with_input_gradient vs autograd.grad(create_graph=True): value of d logit / d x AND the
    gradient of its squared norm w.r.t. every weight (what the gradient penalty contributes to the step)"""
    from quadrupedal_agility_amd.rsl_rl.algorithms.discriminator import Discriminator

    class _Env:
        task_obs_weight_decay = False
    torch.manual_seed(4)
    disc = Discriminator(_Env(), 98, 49, 5, 0.02, "MSELoss", None, 1.0, 0.01, 0.2, 0.2, 2, 2, 0.0, [512, 256], "cpu")
    x = torch.randn(90, 98)
    rows = slice(60, None)
    (d, e, c), g = disc.forward_with_input_gradient(x, rows)
    pen = g.square().sum(-1).mean()
    grads = torch.autograd.grad(pen, [p for p in disc.parameters()], allow_unused=True)
    xr = x[rows].clone().requires_grad_(True)
    d2, e2, c2 = disc(torch.cat([x[:60], xr]))
    g2 = torch.autograd.grad(d2[60:], xr, grad_outputs=torch.ones_like(d2[60:]), create_graph=True)[0]
    pen2 = g2.square().sum(-1).mean()
    grads2 = torch.autograd.grad(pen2, [p for p in disc.parameters()], allow_unused=True)
    assert torch.allclose(d, d2, atol=1e-6) and torch.allclose(c, c2, atol=1e-6) and torch.allclose(e, e2, atol=1e-6)
    assert torch.allclose(g, g2, atol=1e-6) and float(pen) == pytest.approx(float(pen2), rel=1e-6)
    for a, b in zip(grads, grads2):
        if a is None:            # the penalty does not depend on the biases: no gradient here, exact zeros from the double backward
            assert b is None or float(b.abs().max()) == 0.0
        else:
            assert torch.allclose(a, b, atol=1e-6, rtol=1e-5)


def test_only_networks_whose_backward_we_reduce_ourselves_are_recorded(fx):
    """the recorded (hipGraph) learner steps are offered only for Linear+ELU networks / a ReLU MSE discriminator: with any other
    activation the backward contains torch's batch `sum(0)`, which goes stale under replay (profiles/r2_hipgraph_stale_reductions.md)"""
    env, alg = _alg(fx)
    assert alg.use_update_graph is True and alg._recordable_networks()
    alg.actor_critic.actor_trunk[1] = torch.nn.Tanh()
    assert not alg._recordable_networks()
    alg.actor_critic.actor_trunk[1] = torch.nn.ELU()
    alg.disc_loss_function = "WassersteinLoss"
    assert not alg._recordable_networks()
