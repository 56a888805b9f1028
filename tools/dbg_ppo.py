import sys; sys.path.insert(0, ".")
import numpy as np, torch
from tests.test_fused_learner import batch, oracle, KW, reference
from quadrupedal_agility_amd.rsl_rl.algorithms.fused import ppo_loss
B = 100001
t = batch(B, seed=B + 1)
g = {k: v.cuda() for k, v in t.items()}
mu = g["mu"].clone().requires_grad_(True); std = g["std"].clone().requires_grad_(True); value = g["value"].clone().requires_grad_(True)
loss, stats = ppo_loss(mu, std, value, g["actions"], g["old_logp"], g["old_mu"], g["old_sigma"], g["advantages"], g["returns"], g["target_values"], clipped_value=True, **KW)
loss.backward()
out, dmu, dstd, dval = oracle(t, True)
got = mu.grad.cpu().numpy()
bad = ~np.isclose(got, dmu, rtol=3e-4, atol=1e-7 + 2e-6 / B)
rows = np.unique(np.nonzero(bad)[0])
print("bad rows", len(rows), rows[:10])
rs, gmu, gstd, gval = reference(g, True)
gm = gmu.cpu().numpy()
for r in rows[:5]:
    print(r, got[r, :4], dmu[r, :4], gm[r, :4])
    a = {k: v[r].double() if v.dim() > 1 else v.double() for k, v in t.items()}
    logp = (-(a["actions"] - a["mu"]) ** 2 / (2 * a["std"] ** 2) - a["std"].log() - 0.9189385332046727).sum()
    print("  ratio", float(torch.exp(logp - a["old_logp"][0])), "adv", float(a["advantages"][0]))
