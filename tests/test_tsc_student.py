"""The vision student's learner (SURVEY 8f row 3 / BASELINE configs[4]) against the reference's OWN classes: tests/golden/tsc_student.npz is
tests/tsc_student_protocol.py run on tsc/rsl_rl's DepthOnlyFCBackbone58x87 / RecurrentDepthBackbone / BYOL / PPO.update_depth_actor
(tools/gen_golden_tsc_student.py); here the same protocol runs on the mirror.  CPU: forwards, GRU state, BYOL loss, the four DAgger
losses, and the weights after one update_depth_actor (one Adam step over actor + encoder, 6 BYOL minibatches with EMA target updates).
GPU (-m gpu): the same protocol on the device against the same fixture."""
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from tests import tsc_student_protocol as SP

GOLD = os.path.join(os.path.dirname(__file__), "golden", "tsc_student.npz")


def _mine():
    import quadrupedal_agility_amd.tsc.rsl_rl.algorithms as algs
    import quadrupedal_agility_amd.tsc.rsl_rl.modules as mods
    from quadrupedal_agility_amd.tsc.rsl_rl.modules import depth_backbone as db
    return SimpleNamespace(DepthOnlyFCBackbone58x87=db.DepthOnlyFCBackbone58x87, RecurrentDepthBackbone=db.RecurrentDepthBackbone,
                           ActorCriticTSC=mods.ActorCriticTSC, ActorCriticBBC=mods.ActorCriticBBC, Estimator=mods.Estimator, PPO=algs.PPO)


def _compare(out, g, tol_fwd, tol_upd, enc_names):
    assert set(out) == set(g.files)
    for k in g.files:
        a, b = np.asarray(out[k], dtype=np.float64), np.asarray(g[k], dtype=np.float64)
        assert a.shape == b.shape, k
        tol = tol_upd if k.startswith(("probe", "update", "byol_loss")) else tol_fwd
        if k == "probe_encoder_after":
            # rows of the tensors the DAgger step alone trains: tight.  Rows the six BYOL minibatches train after it: Adam on the zero-gradient
            # pre-BatchNorm biases turns fp32 rounding noise into +-lr steps (tests/tsc_student_protocol.py), so their sums agree to a few 1e-2
            # of their size only; `probe_byol_after_2_steps` pins that arithmetic tightly from identical inputs.
            byol = np.array([n.startswith(("base_backbone.", "byol_learner.")) for n in enc_names])
            np.testing.assert_allclose(a[~byol], b[~byol], rtol=tol, atol=tol, err_msg=k)
            np.testing.assert_allclose(a[byol][:, 3:5], b[byol][:, 3:5], rtol=5e-2, atol=5e-2, err_msg=k + " (BYOL-trained rows)")
            continue
        np.testing.assert_allclose(a, b, rtol=tol, atol=tol, err_msg=k)


def _encoder_names(ns):
    enc, _, _ = SP.build(ns)
    return [n for n, _ in sorted(enc.named_parameters())]


def test_student_protocol_matches_reference_golden():
    g = np.load(GOLD)
    _compare(SP.run(_mine()), g, 2e-5, 2e-4, _encoder_names(_mine()))
    assert np.all(np.abs(g["encoder_step0"][:, 34:].sum(1) - 1.0) < 1e-5)           # the obstacle-class block is a softmax
    assert not np.allclose(g["encoder_step0"], g["encoder_step1"])                  # and the second step sees the GRU state of the first


def test_parameter_names_are_the_references():
    """a reference `depth_encoder_state_dict` / `depth_actor_state_dict` loads into the mirror (on_policy_runner.py save/load)"""
    ns = _mine()
    enc, depth_actor, _ = SP.build(ns)
    keys = set(enc.state_dict())
    for k in ("base_backbone.image_compression.0.weight", "base_backbone.image_compression.3.weight", "base_backbone.image_compression.6.weight",
              "base_backbone.image_compression.8.bias", "byol_learner.online_encoder.projector.0.weight", "byol_learner.online_encoder.projector.1.running_mean",
              "byol_learner.online_predictor.3.weight", "byol_learner.target_encoder.net.image_compression.0.weight", "combination_mlp.0.weight",
              "rnn.weight_ih_l0", "rnn.weight_hh_l0", "output_mlp.0.weight"):
        assert k in keys, k
    assert enc.state_dict()["output_mlp.0.weight"].shape == (40, 512) and enc.state_dict()["base_backbone.image_compression.6.weight"].shape == (128, 64 * 25 * 39)


def _probe_after_one_adam_step(a, b, k):
    """parameter probes (entries | sum | abs-sum | numel) of tensors trained by ONE Adam step at lr 1e-3, device vs the CPU fixture"""
    np.testing.assert_allclose(a[:, :3], b[:, :3], rtol=5e-3, atol=5e-3, err_msg=k)
    allow = 5e-3 * np.abs(b[:, 3:5]) + 5e-3 + 2e-6 * b[:, 5:6]
    assert np.all(np.abs(a[:, 3:5] - b[:, 3:5]) <= allow), k


@pytest.mark.gpu
def test_student_protocol_on_the_gpu_matches_reference_golden():
    """the same protocol with every module and input on the device (the hand-written image stem of csrc/qa_conv.hip / qa_gemm.hip, rocBLAS
    linears and GRU): forwards 1e-4, the DAgger-trained tensors 5e-3 (the stem's fp32 sums run in another order than the CPU's; the
    first Adam step is +-lr per element whatever the gradient's size, so an element whose gradient is ~0 can land 2e-3 away and the probe's
    abs-sums move by that much per such element -- r3 measured 1 of 60 probe entries at 2.9e-3, MIOpen's convolutions gave < 2e-3); the
    BYOL-trained probes are held to a few 1e-2 of their size (Adam on the zero-gradient biases, see the protocol)"""
    g = np.load(GOLD)
    out = SP.run(_mine(), device="cuda")
    names = _encoder_names(_mine())
    for k in g.files:
        a, b = np.asarray(out[k], dtype=np.float64), np.asarray(g[k], dtype=np.float64)
        assert a.shape == b.shape, k
        if k == "probe_byol_after_2_steps":
            np.testing.assert_allclose(a[:, 3:5], b[:, 3:5], rtol=5e-2, atol=5e-2, err_msg=k)
        elif k == "probe_encoder_after":
            byol = np.array([n.startswith(("base_backbone.", "byol_learner.")) for n in names])
            # single entries at 5e-3; the sums get 2 lr x 0.1 % of the tensor's elements on top: the first Adam step moves EVERY element by
            # +-lr = 1e-3 whatever its gradient's size, so the few elements whose gradient is ~0 may step the other way under another
            # summation order (r3, split 62,400 -> 128 layer: one signed sum of a 786 k-element tensor off by 0.099 = ~50 elements; a wrong
            # gradient would flip about half of them, 400 x this allowance)
            _probe_after_one_adam_step(a[~byol], b[~byol], k)
            # |sum| within 5 %; the SIGNED sum on the scale of |sum| (it cancels: N elements moved by +-lr each change it by ~lr sqrt(N), which is
            # not small against a sum that happens to be near zero -- r3's stem moved one of 68 such sums by 0.13 at |sum| = 80)
            np.testing.assert_allclose(a[byol][:, 4], b[byol][:, 4], rtol=5e-2, atol=5e-2, err_msg=k)
            assert np.all(np.abs(a[byol][:, 3] - b[byol][:, 3]) <= 5e-2 * np.abs(b[byol][:, 3]) + 5e-2 + 5e-3 * b[byol][:, 4]), k
        elif k == "probe_actor_after":
            _probe_after_one_adam_step(a, b, k)
        else:
            tol = 2e-3 if k.startswith(("probe", "update", "byol_loss")) else 1e-4
            np.testing.assert_allclose(a, b, rtol=tol, atol=tol, err_msg=k)
