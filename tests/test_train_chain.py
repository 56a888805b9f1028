"""The PPO minibatch step's networks as two chain launches (quadrupedal_agility_amd/rsl_rl/algorithms/train_chain.py, include/qa_sim.h ABI 17):
forward with saved activations, input-gradient chain on transposed weights, weight gradients as GEMMs of tape columns.

What they must equal: PyTorch autograd through the reference's modules as SSInfoGAIL.update_actor_critic runs them
(bbc/rsl_rl/algorithms/gail.py:328-413; actor_critic.py:171-225; estimator.py:35-36) -- outputs, and the gradient of EVERY parameter for given
gradients at the four outputs (action mean, value, estimate, privileged latent; the latent receives the regulariser's gradient AND the actor's).
CPU: the op programs through the oracle's twins (double accumulation).  -m gpu: the HIP launches, finished and in-parts gradients, and the whole
minibatch step against the three-stream autograd step."""
import numpy as np
import pytest

from tests.oracle_lib import load_oracle

torch = pytest.importorskip("torch")


def _modules(seed=0):
    from quadrupedal_agility_amd.legged_gym.envs.go2.go2_locomotion_config import Go2LocomotionCfg, Go2LocomotionCfgAlgo
    from quadrupedal_agility_amd.legged_gym.utils.helpers import class_to_dict
    from quadrupedal_agility_amd.rsl_rl.modules import ActorCritic, Estimator
    torch.manual_seed(seed)
    e = Go2LocomotionCfg.env
    pol = class_to_dict(Go2LocomotionCfgAlgo.policy)
    n_obs = e.num_obs + e.history_len * e.num_prop
    ac = ActorCritic(e.num_obs, n_obs, 12, e.num_prop, e.history_len, e.num_explicit, e.num_latent, e.num_command, **pol)
    est = Estimator(input_dim=e.num_prop, output_dim=e.num_explicit, hidden_dims=class_to_dict(Go2LocomotionCfgAlgo.estimator)["hidden_dims"])
    with torch.no_grad():          # biases away from zero, so that a dropped bias shows
        for p in list(ac.parameters()) + list(est.parameters()):
            if p.dim() == 1:
                p.add_(torch.randn_like(p) * 0.05)
    return ac, est, n_obs


def _reference(ac, est, obs, g_est, dmu, dvalue, g_priv):
    """autograd through the modules exactly as _ac_forward_backward_direct calls them"""
    for p in list(ac.parameters()) + list(est.parameters()):
        p.grad = None
    a = ac.num_prop; b = a + ac.num_explicit; c = b + ac.num_latent
    value = ac.evaluate(obs)
    priv = ac.infer_priv_latent(obs[:, b:c])
    e = est(obs[:, :a])
    mu = ac._actor_mean(obs, False)
    torch.autograd.backward([e, mu, value, priv], [g_est, dmu, dvalue.view_as(value), g_priv])
    return e.detach(), mu.detach(), value.detach(), priv.detach()


def _inputs(rows, n_obs, seed=1):
    g = torch.Generator().manual_seed(seed)
    obs = torch.randn(rows, n_obs, generator=g) * 0.7
    return obs, torch.randn(rows, 4, generator=g) / rows, torch.randn(rows, 12, generator=g) / rows, torch.randn(rows, generator=g) / rows, torch.randn(rows, 29, generator=g) * 0.1 / rows


def _check_grads(params_ref, params_got, rtol, tag):
    for (name, p), q in zip(params_ref, params_got):
        if p.grad is None:
            assert q.grad is None or float(q.grad.abs().max()) == 0.0, name
            continue
        scale = float(p.grad.abs().max()) + 1e-30
        err = float((p.grad - q.grad.to(p.grad.device)).abs().max())
        assert err <= rtol * scale, (tag, name, err, scale)


def test_chain_programs_equal_autograd_through_the_oracle_twins():
    from quadrupedal_agility_amd.rsl_rl.algorithms import train_chain
    lib = load_oracle()
    ac, est, n_obs = _modules()
    rows = 37                                  # ragged: not a multiple of the 16-row tile
    chain = train_chain.PpoTrainChain.describe(ac, est, rows, lib=lib, prefix="qo_")
    assert chain is not None and chain.fwd.n_ops <= 24 and chain.bwd.n_ops <= 24
    obs, g_est, dmu, dvalue, g_priv = _inputs(rows, n_obs)
    e_ref, mu_ref, v_ref, p_ref = _reference(ac, est, obs, g_est, dmu, dvalue, g_priv)
    ref = [(n, p) for n, p in list(ac.named_parameters()) + list(est.named_parameters())]
    grads_ref = {n: (p.grad.clone() if p.grad is not None else None) for n, p in ref}
    chain.pack()
    e, mu, v, priv = chain.forward(obs)
    for got, exp, tag in ((e, e_ref, "est"), (mu, mu_ref, "mu"), (v, v_ref, "value"), (priv, p_ref, "priv")):
        assert torch.allclose(got, exp, rtol=1e-5, atol=2e-6), (tag, float((got - exp).abs().max()))
    # the saved activations are the modules' hidden layers
    h = obs
    for i, name in ((0, "c1"), (2, "c2"), (4, "c3")):
        h = torch.nn.functional.elu(ac.critic_trunk[i](h))
        c0, w = chain.t[name]
        assert torch.allclose(chain.tape[:, c0:c0 + w], h.detach(), rtol=1e-5, atol=2e-6), name
    for p in list(ac.parameters()) + list(est.parameters()):
        p.grad = None
    chain.backward(g_est, dmu, dvalue, g_priv, defer=False)
    for n, p in ref:
        gr = grads_ref[n]
        if gr is None:
            continue               # history encoder, std: not the chain's
        scale = float(gr.abs().max()) + 1e-30
        assert p.grad is not None, n
        assert float((p.grad - gr).abs().max()) <= 2e-5 * scale, (n, float((p.grad - gr).abs().max()), scale)


def test_describe_declines_what_the_programs_do_not_cover():
    from quadrupedal_agility_amd.rsl_rl.algorithms import train_chain
    lib = load_oracle()
    ac, est, _ = _modules()
    assert train_chain.PpoTrainChain.describe(ac, est, train_chain.MAX_ROWS + 1, lib=lib, prefix="qo_") is None          # many rows: the library GEMMs' regime
    ac.train_with_estimated_latent = False
    assert train_chain.PpoTrainChain.describe(ac, est, 64, lib=lib, prefix="qo_") is None
    ac.train_with_estimated_latent = True
    ac.actor_trunk[1] = torch.nn.Tanh()
    assert train_chain.PpoTrainChain.describe(ac, est, 64, lib=lib, prefix="qo_") is None


@pytest.mark.gpu
@pytest.mark.parametrize("rows", [48, 3072, 6144])
def test_hip_chain_equals_autograd(rows):
    from quadrupedal_agility_amd.rsl_rl.algorithms import fused, train_chain
    ac, est, n_obs = _modules(3)
    ac_r, est_r, _ = _modules(3)
    ac, est = ac.cuda(), est.cuda()
    obs, g_est, dmu, dvalue, g_priv = _inputs(rows, n_obs, seed=rows)
    e_ref, mu_ref, v_ref, p_ref = _reference(ac_r, est_r, obs, g_est, dmu, dvalue, g_priv)
    chain = train_chain.PpoTrainChain.describe(ac, est, rows)
    assert chain is not None
    chain.pack()
    dv = [x.cuda() for x in (obs, g_est, dmu, dvalue, g_priv)]
    e, mu, v, priv = chain.forward(dv[0])
    torch.cuda.synchronize()
    for got, exp, tag in ((e, e_ref, "est"), (mu, mu_ref, "mu"), (v, v_ref, "value"), (priv, p_ref, "priv")):
        assert torch.allclose(got.cpu(), exp, rtol=2e-4, atol=2e-5), (tag, float((got.cpu() - exp).abs().max()))
    ref = list(ac_r.named_parameters()) + list(est_r.named_parameters())
    got = list(ac.parameters()) + list(est.parameters())
    chain.backward(*dv[1:], defer=False)
    torch.cuda.synchronize()
    _check_grads(ref, got, 3e-4, "finished")
    first = [p.grad.clone() if p.grad is not None else None for p in got]
    # in parts: nothing finished until somebody asks
    for p in got:
        p.grad = None
    chain.forward(dv[0])
    chain.backward(*dv[1:], defer=True)
    assert fused.pending_grads() == 26
    fused.flush_pending_grads()
    assert fused.pending_grads() == 0
    torch.cuda.synchronize()
    _check_grads(ref, got, 3e-4, "parts")
    for a, b in zip(first, got):          # the same partial products, added by either finish (qa_slab_reduce / qa_grad_reduce: fixed orders, not the same one)
        if a is not None:
            assert torch.allclose(a, b.grad, rtol=1e-5, atol=1e-6 * float(a.abs().max() + 1e-30))


@pytest.mark.gpu
@pytest.mark.parametrize("graph", [False, True])
def test_training_with_chain_steps_equals_training_with_autograd_steps(graph):
    """3 iterations at 256 envs (1,536-row minibatches, 20 steps each; iterations 2-3 replay recorded steps when `graph`): the same rollouts,
    the same permutations -- the weights after 60 optimiser steps agree to what 60 steps of fp32 rounding allow"""
    from tests.test_gpu_train import _make
    from quadrupedal_agility_amd.legged_gym.envs import task_registry
    from quadrupedal_agility_amd.rsl_rl.algorithms import train_chain
    res = []
    try:
        for chain in (True, False):
            train_chain.ENABLED = chain
            torch.manual_seed(0)
            env, args, tcfg = _make(256, False)
            runner, _ = task_registry.make_alg_runner(env, name="go2_locomotion", args=args, train_cfg=tcfg, log_root=None)
            runner.alg.use_update_graph = graph
            runner.learn(3, init_at_random_ep_len=True)
            used = bool(getattr(runner.alg, "_train_chains", {}))
            assert used == chain
            res.append(({k: v.clone() for k, v in runner.alg.actor_critic.state_dict().items()},
                        {k: v.clone() for k, v in runner.alg.estimator.state_dict().items()}, float(runner.alg.lr_ac)))
    finally:
        train_chain.ENABLED = True
    (wa, ea, lra), (wb, eb, lrb) = res
    assert lra == lrb
    # 60 Adam steps apart in arithmetic (the chain's ELU is __expf, its sums run in another order): Adam turns a rounding-sized difference of a
    # near-zero gradient into a full step of one element, so single elements differ by up to a few steps' worth (lr 1e-3) while the tensors agree
    for k in list(wa) + ["estimator." + k for k in ea]:
        a, b = (wa[k], wb[k]) if k in wa else (ea[k[10:]], eb[k[10:]])
        if a.dtype.is_floating_point and a.numel() > 1:
            rel = float((a - b).norm() / (b.norm() + 1e-12))
            assert float((a - b).abs().max()) < 3e-3 and (rel < 2e-3 or float(b.norm()) < 1.0), (k, rel, float((a - b).abs().max()))      # (zero-initialised biases: no norm to be relative to)


# ------------------------------------------------------------------ the discriminator step's three chain launches
def _disc(seed=0):
    from quadrupedal_agility_amd.rsl_rl.algorithms.discriminator import Discriminator

    class _Env:
        task_obs_weight_decay = False
    torch.manual_seed(seed)
    d = Discriminator(_Env(), 98, 49, 5, 0.02, "MSELoss", None, 1.0, 1.0, 1.0, 1.0, 2, 2, 0.0, [512, 256], "cpu")
    with torch.no_grad():
        for p in d.parameters():
            if p.dim() == 1:
                p.add_(torch.randn_like(p) * 0.05)
    return d


def _disc_reference(disc, x, n_u, g_d, g_eps, g_l, c_gp):
    """autograd, with the penalty's input gradient by DOUBLE BACKWARD as the reference takes it (gail.py:487-492)"""
    for p in disc.parameters():
        p.grad = None
    xu = x[-n_u:].clone().requires_grad_(True)
    xa = torch.cat([x[:-n_u], xu], 0)
    h = disc.trunk(xa)
    d, eps, logits = disc.linear(h), disc.encoder_eps(h), disc.classifier(h)
    g = torch.autograd.grad(d[-n_u:], xu, grad_outputs=torch.ones_like(d[-n_u:]), create_graph=True, retain_graph=True)[0]
    loss = (d * g_d).sum() + (eps * g_eps).sum() + (logits * g_l).sum() + c_gp * g.square().sum(-1).mean()
    loss.backward()
    return d.detach(), eps.detach(), logits.detach(), g.detach()


def _disc_inputs(rows, seed):
    gen = torch.Generator().manual_seed(seed)
    x = torch.randn(rows, 98, generator=gen)
    return x, torch.randn(rows, 1, generator=gen) / rows, torch.randn(rows, 1, generator=gen) / rows, torch.randn(rows, 5, generator=gen) / rows


def test_discriminator_chain_programs_equal_double_backward_through_the_oracle_twins():
    from quadrupedal_agility_amd.rsl_rl.algorithms import train_chain
    lib = load_oracle()
    disc = _disc()
    rows, n_u, c_gp = 3 * 23, 23, 0.2
    x, g_d, g_eps, g_l = _disc_inputs(rows, 4)
    d_ref, e_ref, l_ref, g_ref = _disc_reference(disc, x, n_u, g_d, g_eps, g_l, c_gp)
    ref = {n: p.grad.clone() for n, p in disc.named_parameters()}
    chain = train_chain.DiscTrainChain.describe(disc, rows, n_u, lib=lib, prefix="qo_")
    assert chain is not None
    chain.pack()
    d, e, l = chain.forward(x)
    g = chain.penalty_gradient()
    for got, exp, tag in ((d, d_ref, "logit"), (e, e_ref, "eps"), (l, l_ref, "class logits"), (g, g_ref, "d logit / d x")):
        assert torch.allclose(got, exp, rtol=1e-5, atol=2e-6), (tag, float((got - exp).abs().max()))
    for p in disc.parameters():
        p.grad = None
    chain.backward(g_d, g_eps, g_l, c_gp)
    for n, p in disc.named_parameters():
        scale = float(ref[n].abs().max()) + 1e-30
        assert p.grad is not None and p.grad.shape == p.shape and float((p.grad - ref[n]).abs().max()) <= 2e-5 * scale, (n, float((p.grad - ref[n]).abs().max()), scale)


@pytest.mark.gpu
@pytest.mark.parametrize("mb", [16, 307, 1228])
def test_hip_discriminator_chain_equals_double_backward(mb):
    from quadrupedal_agility_amd.rsl_rl.algorithms import train_chain
    disc_r, disc = _disc(5), _disc(5).cuda()
    rows, n_u, c_gp = 3 * mb, mb, 0.2
    x, g_d, g_eps, g_l = _disc_inputs(rows, mb)
    d_ref, e_ref, l_ref, g_ref = _disc_reference(disc_r, x, n_u, g_d, g_eps, g_l, c_gp)
    chain = train_chain.DiscTrainChain.describe(disc, rows, n_u)
    assert chain is not None
    chain.pack()
    d, e, l = chain.forward(x.cuda())
    g = chain.penalty_gradient()
    torch.cuda.synchronize()
    for got, exp, tag in ((d, d_ref, "logit"), (e, e_ref, "eps"), (l, l_ref, "class logits"), (g, g_ref, "d logit / d x")):
        assert torch.allclose(got.cpu(), exp, rtol=2e-4, atol=2e-5), (tag, float((got.cpu() - exp).abs().max()))
    chain.backward(g_d.cuda(), g_eps.cuda(), g_l.cuda(), c_gp)
    torch.cuda.synchronize()
    _check_grads(list(disc_r.named_parameters()), list(disc.parameters()), 3e-4, "disc")


@pytest.mark.gpu
def test_amp_training_with_the_discriminator_chain_equals_training_without_it():
    """2 iterations of config 3 at 256 envs (iteration 1: 80 eager discriminator steps; iteration 2: recorded ones): same rollouts, same sample
    tables (same generator calls).  The discriminator's iteration amplifies rounding by orders of magnitude (tests/test_learner_lockstep.py's
    control arm), so its weights are held to a relative L2, the PPO side -- which does not see the discriminator's weights until the next
    rollout -- elementwise."""
    from tests.test_gpu_train import _make
    from quadrupedal_agility_amd.legged_gym.envs import task_registry
    from quadrupedal_agility_amd.rsl_rl.algorithms import train_chain
    res = []
    try:
        for chain in (True, False):
            train_chain.DISC_ENABLED = chain
            torch.manual_seed(0)
            env, args, tcfg = _make(256, True)
            runner, _ = task_registry.make_alg_runner(env, name="go2_locomotion", args=args, train_cfg=tcfg, log_root=None)
            runner.alg.eager_from_tables = True            # the eager warm-up update draws its samples the way the recorded ones do
            runner.learn(2, init_at_random_ep_len=True)
            assert bool(getattr(runner.alg, "_disc_train_chains", {})) == chain
            a = runner.alg
            res.append((torch.cat([p.detach().flatten() for p in a.disc.parameters()]), {k: v.clone() for k, v in a.actor_critic.state_dict().items()},
                        a.disc_normalizer.mean.clone(), float(a.lr_ac)))
    finally:
        train_chain.DISC_ENABLED = True
    (da, wa, na, lra), (db, wb, nb, lrb) = res
    rel = float((da - db).norm() / db.norm())
    print("discriminator weights, relative L2 after 160 steps:", rel)
    assert torch.isfinite(da).all() and rel < 2e-3
    assert torch.allclose(na, nb, rtol=1e-9, atol=1e-12) and lra == lrb


def _stacked_case(dev, lib=None, prefix="qa_", steps=3):
    """the discriminator step's optimiser half both ways: chain backward finished + regulariser adds + three torch Adam steps, one after the other
    (gail.py:503-520), against chain backward in parts + ONE stacked launch"""
    import copy
    from quadrupedal_agility_amd.rsl_rl.algorithms import fused, train_chain
    c_gp, c_wd, c_lr = 0.2, 1e-4, 0.05
    out = []
    for stacked in (False, True):
        disc = _disc(7).to(dev)
        adam = dict(fused=True, capturable=True) if dev == "cuda" else {}
        groups = lambda head: [{"params": disc.trunk.parameters(), "weight_decay": 1e-3}, {"params": head.parameters(), "weight_decay": 1e-3}]
        opts = [torch.optim.Adam(groups(disc.linear), lr=1e-3, **adam), torch.optim.Adam(groups(disc.encoder_eps), lr=3e-4, **adam),
                torch.optim.Adam(groups(disc.classifier), lr=3e-4, **adam)]
        rows, n_u = 3 * 41, 41
        chain = train_chain.DiscTrainChain.describe(disc, rows, n_u, lib=lib, prefix=prefix)
        stack = fused.StackedAdam(opts, lib=lib, prefix=prefix)
        reg_w = [m.weight for m in disc.trunk.modules() if isinstance(m, torch.nn.Linear)] + [disc.linear.weight]
        for it in range(steps):
            x, g_d, g_eps, g_l = (t.to(dev) for t in _disc_inputs(rows, 20 + it))
            chain.pack()
            chain.forward(x)
            chain.penalty_gradient()
            for o in opts:
                o.zero_grad()
            use = stacked and it > 0                     # (the first step creates the optimisers' state)
            assert stack.ready() == (it > 0)
            src = chain.backward(g_d, g_eps, g_l, c_gp, finish=not use)
            if use:
                reg = {w: 2.0 * c_wd for w in reg_w[:-1]}
                reg[reg_w[-1]] = 2.0 * (c_wd + c_lr)
                stack.step(src, reg)
            else:
                assert src is None
                with torch.no_grad():
                    torch._foreach_add_([w.grad for w in reg_w[:-1]], reg_w[:-1], alpha=2.0 * c_wd)
                    reg_w[-1].grad.add_(reg_w[-1], alpha=2.0 * (c_wd + c_lr))
                for o in opts:
                    o.step()
        if dev == "cuda":
            torch.cuda.synchronize()
        state = {}
        for n, p in disc.named_parameters():
            state["p." + n], state["g." + n] = p.detach().cpu().clone(), p.grad.detach().cpu().clone()
            for k, o in enumerate(opts):
                if p in o.state:
                    st = o.state[p]
                    state[f"m{k}." + n], state[f"v{k}." + n], state[f"s{k}." + n] = st["exp_avg"].cpu().clone(), st["exp_avg_sq"].cpu().clone(), st["step"].cpu().clone().float()
        out.append(state)
    a, b = out
    assert a.keys() == b.keys() and sum(k.startswith("m") for k in a) == 3 * 4 + 6
    for k in a:
        scale = float(a[k].abs().max()) + 1e-30
        tol = 0.0 if k[0] == "s" else (2e-5 if k[0] in "gmv" else 2e-6)        # (parameters: lr x O(1) x the moments' relative difference)
        assert float((a[k] - b[k]).abs().max()) <= tol * scale, (k, float((a[k] - b[k]).abs().max()), scale)
        if k[0] == "s":
            assert float(a[k]) == steps


def test_stacked_adam_twin_equals_three_optimisers_one_after_the_other():
    _stacked_case("cpu", lib=load_oracle(), prefix="qo_")


@pytest.mark.gpu
def test_hip_stacked_adam_equals_three_optimisers_one_after_the_other():
    _stacked_case("cuda")


# ------------------------------------------------------------------ several weight-gradient products in one call
WG_SHAPES = [(3072, 671, 512), (3072, 512, 256), (3072, 101, 512), (3072, 128, 12), (3072, 128, 1), (3072, 29, 64), (3072, 64, 4), (1228, 98, 512), (1228, 1, 256), (77, 57, 128)]


def _x_width(i, k):
    """row width of product i's x: the first three are rows padded to the next multiple of 4 (the 671 observation columns of 672-wide rows, the actor's 101
    input columns of a 104-wide tape region: read in 16-byte pieces INCLUDING the padding), the others k + 3 (rows that are not 16-byte aligned)"""
    return (k + 3) // 4 * 4 if i < 3 else k + 3


def _wg_case(seed=0):
    gen = torch.Generator().manual_seed(seed)
    xs = []
    for i, (r, k, n) in enumerate(WG_SHAPES):
        full = torch.randn(r, _x_width(i, k), generator=gen)
        full[:, k:] = float("nan")                    # whatever lies behind in_features must reach no output
        xs.append(full[:, :k])
    gs = [torch.randn(r, n + 5, generator=gen)[:, 1:1 + n] / r for r, k, n in WG_SHAPES]     # ... and a start that is not 16-byte aligned
    return xs, gs


def _run_batch(lib, prefix, xs, gs, dev, finished):
    import ctypes as C
    from quadrupedal_agility_amd import _capi
    fn = lambda name: getattr(lib, prefix + name)
    descs = (_capi.QaWgradDesc * len(xs))()
    keep, out = [], []
    for i, ((r, k, n), x, g) in enumerate(zip(WG_SHAPES, xs, gs)):
        nb = int(fn("linear_backward_weight_batch_scratch_bytes")(r, k, n))
        lay = (C.c_int64 * 5)()
        assert fn("linear_backward_weight_batch_layout")(r, k, n, lay) == 0
        sc = torch.zeros(nb // 4 + 4, device=dev); gw = torch.full((n, k), float("nan"), device=dev); gb = torch.full((n,), float("nan"), device=dev)
        descs[i] = _capi.QaWgradDesc(g.data_ptr(), g.stride(0), x.data_ptr(), x.stride(0), gw.data_ptr() if finished else None, gb.data_ptr() if finished else None, r, k, n, sc.data_ptr(), nb)
        keep.append((sc, list(lay))); out.append((gw, gb))
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream) if dev != "cpu" else None
    assert fn("linear_backward_weight_batch")(descs, len(xs), stream) == 0
    if not finished:          # add the parts up here: what qa_grad_reduce / the optimiser's first pass do
        out = []
        for (sc, lay), (r, k, n) in zip(keep, WG_SHAPES):
            w = torch.stack([sc[j * lay[1]: j * lay[1] + n * k] for j in range(lay[0])]).sum(0).view(n, k)
            b = torch.stack([sc[lay[4] + j * lay[3]: lay[4] + j * lay[3] + n] for j in range(lay[2])]).sum(0)
            out.append((w, b))
    return out


def test_wgrad_batch_twin_equals_the_products():
    xs, gs = _wg_case()
    for finished in (True, False):
        for (gw, gb), x, g in zip(_run_batch(load_oracle(), "qo_", xs, gs, "cpu", finished), xs, gs):
            assert torch.allclose(gw, g.t().double().mm(x.double()).float(), rtol=1e-5, atol=1e-7) and torch.allclose(gb, g.double().sum(0).float(), rtol=1e-5, atol=1e-7)


@pytest.mark.gpu
@pytest.mark.parametrize("finished", [True, False])
def test_hip_wgrad_batch_equals_the_products_one_by_one(finished):
    """qa_linear_backward_weight_batch (<= 4 launches for the ten products: 16-byte / 4-byte readable operands in every combination, a 1-column input,
    a 1-column gradient, 77 rows) against float64 products, finished by its one reduction or left in parts"""
    from quadrupedal_agility_amd import _capi
    xs, gs = _wg_case(1)
    dx, dg = [], []          # device copies that keep the strides and the misalignment
    for i, (x, g) in enumerate(zip(xs, gs)):
        bx = torch.full((x.shape[0], _x_width(i, x.shape[1])), float("nan"), device="cuda"); bx[:, :x.shape[1]] = x.cuda(); dx.append(bx[:, :x.shape[1]])
        bg = torch.zeros(g.shape[0], g.shape[1] + 5, device="cuda"); bg[:, 1:1 + g.shape[1]] = g.cuda(); dg.append(bg[:, 1:1 + g.shape[1]])
    res = _run_batch(_capi.load_library(), "qa_", dx, dg, "cuda", finished)
    torch.cuda.synchronize()
    for (gw, gb), x, g, shape in zip(res, xs, gs, WG_SHAPES):
        rw, rb = g.t().double().mm(x.double()), g.double().sum(0)
        assert torch.allclose(gw.cpu().double(), rw, rtol=2e-4, atol=2e-6 * float(rw.abs().max())), shape
        assert torch.allclose(gb.cpu().double(), rb, rtol=2e-4, atol=2e-6 * float(rb.abs().max()) + 1e-9), shape


# ------------------------------------------------------------------ the task-level learner's step
def _tsc_reference(ac, est, obs, g_est, dlogits, dmean, dvalue, g_priv):
    """autograd through the modules as tsc/rsl_rl/algorithms/ppo.py:_minibatch_forward_backward calls them"""
    from quadrupedal_agility_amd.rsl_rl.modules.actor_critic import _head
    for p in list(ac.parameters()) + list(est.parameters()):
        p.grad = None
    emb = ac.actor(obs, False)
    logits, mean = _head(ac.actor.actor_d, emb), _head(ac.actor.actor_c, emb)
    value = ac.evaluate(obs)
    priv = ac.actor.infer_priv_latent(obs)
    e = est(obs[:, :57])
    torch.autograd.backward([e, logits, mean, value, priv], [g_est, dlogits, dmean, dvalue.view_as(value), g_priv])
    return e.detach(), logits.detach(), mean.detach(), value.detach(), priv.detach()


def _tsc_inputs(rows, n_obs, nd, nc, seed=1):
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.randn(*s, generator=g)
    return r(rows, n_obs) * 0.7, r(rows, 4) / rows, r(rows, nd) / rows, r(rows, nc) / rows, r(rows) / rows, r(rows, 29) * 0.1 / rows


def test_task_level_chain_programs_equal_autograd_through_the_oracle_twins():
    from tests.test_policy_chain import tsc_modules
    from quadrupedal_agility_amd.rsl_rl.algorithms import train_chain
    lib = load_oracle()
    ac, est, n_obs = tsc_modules(4)
    rows = 21
    chain = train_chain.TscTrainChain.describe(ac, est, rows, 57, lib=lib, prefix="qo_")
    assert chain is not None and chain.fwd.n_ops <= 24 and chain.bwd.n_ops <= 24
    nd, nc = chain.dims["nd"], chain.dims["nc"]
    obs, g_est, dlogits, dmean, dvalue, g_priv = _tsc_inputs(rows, n_obs, nd, nc)
    refs = _tsc_reference(ac, est, obs, g_est, dlogits, dmean, dvalue, g_priv)
    named = list(ac.named_parameters()) + list(est.named_parameters())
    grads_ref = {n: (p.grad.clone() if p.grad is not None else None) for n, p in named}
    chain.pack()
    outs = chain.forward(obs)
    for got, exp, tag in zip(outs, refs, ("est", "logits", "mean", "value", "priv")):
        assert torch.allclose(got, exp, rtol=1e-5, atol=2e-6), (tag, float((got - exp).abs().max()))
    for p in list(ac.parameters()) + list(est.parameters()):
        p.grad = None
    chain.backward(g_est, dlogits, dmean, dvalue, g_priv, defer=False)
    checked = 0
    for n, p in named:
        gr = grads_ref[n]
        if gr is None:
            continue               # history encoder, std: not the chain's
        scale = float(gr.abs().max()) + 1e-30
        assert p.grad is not None and p.grad.shape == p.shape, n
        assert float((p.grad - gr).abs().max()) <= 2e-5 * scale, (n, float((p.grad - gr).abs().max()), scale)
        checked += 1
    assert checked == 34


@pytest.mark.gpu
@pytest.mark.parametrize("rows", [40, 6144])
def test_hip_task_level_chain_equals_autograd(rows):
    from tests.test_policy_chain import tsc_modules
    from quadrupedal_agility_amd.rsl_rl.algorithms import fused, train_chain
    ac_r, est_r, n_obs = tsc_modules(6)
    ac, est, _ = tsc_modules(6)
    ac, est = ac.cuda(), est.cuda()
    chain = train_chain.TscTrainChain.describe(ac, est, rows, 57)
    assert chain is not None
    ins = _tsc_inputs(rows, n_obs, chain.dims["nd"], chain.dims["nc"], seed=rows)
    refs = _tsc_reference(ac_r, est_r, *ins)
    dv = [x.cuda() for x in ins]
    chain.pack()
    outs = chain.forward(dv[0])
    torch.cuda.synchronize()
    for got, exp, tag in zip(outs, refs, ("est", "logits", "mean", "value", "priv")):
        assert torch.allclose(got.cpu(), exp, rtol=2e-4, atol=2e-5), (tag, float((got.cpu() - exp).abs().max()))
    ref = list(ac_r.named_parameters()) + list(est_r.named_parameters())
    got = list(ac.parameters()) + list(est.parameters())
    chain.backward(*dv[1:], defer=False)
    torch.cuda.synchronize()
    _check_grads(ref, got, 3e-4, "finished")
    for p in got:
        p.grad = None
    chain.forward(dv[0])
    chain.backward(*dv[1:], defer=True)
    assert fused.pending_grads() == 34
    fused.flush_pending_grads()
    torch.cuda.synchronize()
    _check_grads(ref, got, 3e-4, "parts")
