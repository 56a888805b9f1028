"""-m gpu: the checker-side hybrid arm (tools/hybrid_backend.py: the ORACLE's physics on the host under the product's GPU learner, r5).

It is test infrastructure for the return-curve parity question (tools/three_arm_parity.py), so what is pinned here is only that it is what it says:
its env steps ARE the oracle's env steps (bit for bit, through the device mirror), what the learner writes into env tensors on the device reaches the
oracle, and a short training run on it works."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


def test_hybrid_env_steps_are_the_oracles_env_steps():
    from tests.oracle_lib import OracleSim, go2_cfg
    from tools.hybrid_backend import HybridBackend
    n = 96
    q = go2_cfg(n, seed=3)
    h, o = HybridBackend(q), OracleSim(go2_cfg(n, seed=3))
    h.reset_all(); o.reset_all()
    g = torch.Generator().manual_seed(0)
    fired = False
    for k in range(5):
        a = torch.randn(n, 12, generator=g)
        if k == 2:          # the learner's side of the seam writes on the DEVICE: it must reach the oracle before its next step
            h.t["EPISODE_LENGTH"].fill_(10 ** 6); o.t["EPISODE_LENGTH"][:] = 10 ** 6
        h.step(a.cuda()); o.global_step = k; o.step(a.numpy())
        for name in ("OBS", "REW", "RESET", "ROOT_STATES", "DOF_STATE", "EPISODE_LENGTH", "OBS_DISC"):
            assert np.array_equal(h.t[name].cpu().numpy(), o.t[name]), (k, name)
        fired = fired or bool((o.t["RESET"] != 0).any())
    assert fired          # the time-outs planted at k = 2 fired (in the step that followed)


@pytest.mark.parametrize("amp", [False, True])
def test_short_training_run_on_the_hybrid_arm(amp, monkeypatch):
    monkeypatch.setenv("QA_PARITY_NO_LOG", "0")
    from tools.return_curve_parity import run
    curves, wall, fps = run("hybrid", 128, 3, seed=2, amp=amp)
    losses = [v for k, vs in curves.items() if k.startswith("Loss/") for v in vs]
    assert losses and all(np.isfinite(losses)), "the learner's losses on the hybrid arm must be finite"
