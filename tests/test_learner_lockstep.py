"""The learner lockstep (tools/learner_lockstep.py): the product's GPU learner against the torch-CPU learner on the same
state, rollout and sample tables (bbc/rsl_rl/algorithms/gail.py:231-326 is what both implement).

CPU: the tool's own plumbing -- with a torch-CPU learner as the driver every difference must be EXACTLY zero (state copy, rollout copy, ring
mirroring and table injection cover everything an iteration depends on).  -m gpu: the first 3 iterations at 256 envs, config 3."""
import argparse
import os
import sys

import pytest

torch = pytest.importorskip("torch")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _args(**kw):
    d = dict(num_envs=64, iters=2, seed=1, amp=True, physics="oracle", driver="cpu", free=1, control=0.0, ring=20000, bound=1e-5, keep_rows=1, verbose=0, out=None)
    d.update(kw)
    return argparse.Namespace(**d)


def test_lockstep_of_a_cpu_learner_with_itself_is_exact(monkeypatch):
    monkeypatch.setenv("QA_CPU_THREADS", "4")
    from tools import learner_lockstep as ll
    res = ll.run(_args())
    for arm in ("forced", "free"):
        worst = res[arm]["worst_over_iterations"]
        assert worst and all(v["rel_l2"] == 0.0 for v in worst.values()), {k: v for k, v in worst.items() if v["rel_l2"] != 0.0}
        assert res[arm]["logged_scalars_max_abs_diff"] == 0.0
    # both arms really stepped: the optimisers have moments, the ring filled, the DAgger iteration (it 0) ran
    assert "adam_ac_m" in res["forced"]["worst_over_iterations"] and "adam_hist_encoder_m" in res["forced"]["worst_over_iterations"]
    assert res["rows"][0]["hist_encoding"] and res["rows"][0]["forced"]["hist_latent_loss"] is not None


def test_tables_hook_makes_two_cpu_learners_step_on_the_same_samples():
    """SSInfoGAIL.update(tables=...) / update_dagger(perm=...): a given permutation is the one the minibatches follow"""
    from quadrupedal_agility_amd.rsl_rl.storage import RolloutStorage
    st = RolloutStorage(8, 6, [11], [11], [3])
    st.observations.copy_(torch.arange(48.0).view(6, 8, 1).expand(6, 8, 11))
    perm = torch.randperm(48, generator=torch.Generator().manual_seed(0))
    seen = [s[0][:, 0].long() for s in st.mini_batch_generator(4, 2, perm=perm)]
    assert len(seen) == 8 and all(torch.equal(seen[i], perm[(i % 4) * 12:(i % 4 + 1) * 12]) for i in range(8))


def test_control_arm_measures_what_one_ulp_does_to_an_iteration(monkeypatch):
    """the yardstick: two torch-CPU learners from the same state, one with every parameter moved by ~1 fp32 ulp.  The PPO side answers at rounding
    level; the discriminator's 80 Adam steps (ReLU masks, gradient penalty, Adam's normalisation of near-zero gradients) amplify the ulp by orders
    of magnitude within ONE iteration -- which is why "GPU learner == CPU learner to 1e-5" cannot be asked of the discriminator's weights, of any
    two arithmetics, and is asked relative to this arm instead."""
    monkeypatch.setenv("QA_CPU_THREADS", "4")
    from tools import learner_lockstep as ll
    res = ll.run(_args(iters=2, free=0, control=1e-7))
    c = res["control"]["worst_over_iterations"]
    assert all(v["rel_l2"] == 0.0 for v in res["forced"]["worst_over_iterations"].values())      # the driver is a torch-CPU learner here
    assert 1e-8 < c["critic_trunk"]["rel_l2"] < 1e-5 and 1e-8 < c["estimator"]["rel_l2"] < 1e-5
    assert c["disc_trunk"]["rel_l2"] > 10 * c["critic_trunk"]["rel_l2"]          # the discriminator is the ill-conditioned half
    assert c["normaliser_count"]["rel_l2"] == 0.0 and c["adam_ac_step"]["rel_l2"] == 0.0


# The GPU test.  `forced` = what ONE iteration's arithmetic differs by between the product's GPU learner and the torch-CPU learner (fp32 GEMMs through
# MFMA add in another order than the CPU's blocked sums, fused objectives, Adam in one kernel); `control` = what one iteration does to a one-ulp
# difference in its input (CPU vs CPU).  Measured at 1024 envs x 200 iterations (profiles/r6_learner_lockstep_cfg3.json): PPO-side groups 1e-7 .. 3e-6
# in both arms; discriminator groups 1e-5 .. 1e-1 in BOTH arms, growing with training -- the GPU learner sits at the control's level.
PPO_GROUPS = ("actor_trunk", "actor_head", "critic_trunk", "critic_head", "priv_encoder", "estimator", "std")
DISC_GROUPS = ("disc_trunk", "disc_head", "disc_encoder_eps", "disc_classifier")


@pytest.mark.gpu
def test_gpu_learner_steps_like_the_torch_cpu_learner_for_three_iterations():
    from tools import learner_lockstep as ll
    res = ll.run(_args(num_envs=256, iters=3, driver="gpu", ring=50000, free=0, control=1e-7))
    f, c = res["forced"], res["control"]
    w, wc = f["worst_over_iterations"], c["worst_over_iterations"]
    print({g: (f"{w[g]['rel_l2']:.1e}", f"{wc[g]['rel_l2']:.1e}") for g in PPO_GROUPS + DISC_GROUPS})
    for g in PPO_GROUPS:                       # rounding level, absolutely
        assert w[g]["rel_l2"] <= 2e-5, (g, w[g])
    for g in DISC_GROUPS:                      # at the level a one-ulp input change produces (x 30: the GPU's arithmetic differs in every operation, not once)
        assert w[g]["rel_l2"] <= 30 * wc[g]["rel_l2"] + 1e-6, (g, w[g], wc[g])
    assert w["advantages"]["rel_l2"] < 1e-4 and w["returns"]["rel_l2"] < 1e-5            # GAE kernel vs the CPU scan
    assert w["lr_ac"]["rel_l2"] < 1e-6                                                   # the KL rule took the same branch in every minibatch step
    assert all(w[k]["rel_l2"] == 0.0 for k in w if k.endswith("_step"))
    assert w["normaliser_mean"]["rel_l2"] < 1e-6 and w["normaliser_var"]["rel_l2"] < 1e-6 and w["normaliser_count"]["rel_l2"] == 0.0
    assert res["gpu_path_last"]["ppo_steps_recorded"] and res["gpu_path_last"]["disc_steps_recorded"]      # the product's recorded steps were the ones compared
