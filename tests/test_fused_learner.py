"""Fused PPO objective (qa_ppo_loss): the C restatement against the eager PyTorch expression (CPU), the HIP kernel
against both (GPU), and the update step with the kernel against the update step without it."""
import ctypes as C

import numpy as np
import pytest
import torch

from quadrupedal_agility_amd.rsl_rl.algorithms.fused import ppo_loss_reference
from tests.oracle_lib import load_oracle

KW = dict(clip=0.2, c_surr=2.0, c_value=5.0, c_bound=0.3, c_entropy=0.01)


def batch(B, seed, spread=1.0, dev="cpu"):
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.randn(*s, generator=g)
    old_mu = r(B, 12) * 0.8
    old_sigma = (0.3 + torch.rand(B, 12, generator=g)) * torch.ones(B, 12)
    actions = old_mu + old_sigma * r(B, 12)
    mu = old_mu + 0.15 * spread * r(B, 12)
    mu[::7] *= 3.0                                   # some means beyond +-1: the bound loss is live
    std = 0.4 + torch.rand(12, generator=g)
    old_logp = (-(actions - old_mu) ** 2 / (2 * old_sigma ** 2) - old_sigma.log() - 0.9189385332).sum(-1, keepdim=True)
    target_values = r(B, 1)
    value = target_values + 0.3 * spread * r(B, 1)     # both inside and outside the +-0.2 clip
    returns = target_values + 0.5 * r(B, 1)
    adv = r(B, 1)
    adv[::11] = 0.0
    t = dict(mu=mu, std=std, value=value, actions=actions, old_logp=old_logp, old_mu=old_mu, old_sigma=old_sigma,
             advantages=adv, returns=returns, target_values=target_values)
    return {k: v.to(dev) for k, v in t.items()}


def reference(t, clipped=True):
    mu = t["mu"].clone().requires_grad_(True); std = t["std"].clone().requires_grad_(True); value = t["value"].clone().requires_grad_(True)
    loss, stats = ppo_loss_reference(mu, std, value, t["actions"], t["old_logp"], t["old_mu"], t["old_sigma"], t["advantages"],
                                     t["returns"], t["target_values"], clipped_value=clipped, **KW)
    loss.backward()
    return stats, mu.grad, std.grad, value.grad


def oracle(t, clipped=True):
    lib = load_oracle()
    lib.qo_ppo_loss.argtypes = [C.c_void_p] * 10 + [C.c_int64, C.c_int32] + [C.c_float] * 5 + [C.c_int32] + [C.c_void_p] * 5 + [C.c_int64, C.c_void_p]
    a = {k: np.ascontiguousarray(v.detach().cpu().numpy(), dtype=np.float32) for k, v in t.items()}
    B = a["mu"].shape[0]
    dmu = np.zeros((B, 12), np.float32); dstd = np.zeros(12, np.float32); dval = np.zeros(B, np.float32); out = np.zeros(8, np.float32)
    p = lambda x: x.ctypes.data
    rc = lib.qo_ppo_loss(p(a["mu"]), p(a["std"]), p(a["value"]), p(a["actions"]), p(a["old_logp"]), p(a["old_mu"]), p(a["old_sigma"]),
                         p(a["advantages"]), p(a["returns"]), p(a["target_values"]), B, 12, KW["clip"], KW["c_surr"], KW["c_value"],
                         KW["c_bound"], KW["c_entropy"], int(clipped), p(dmu), p(dstd), p(dval), p(out), None, 0, None)
    assert rc == 0
    return out, dmu, dstd, dval


@pytest.mark.parametrize("B,clipped", [(1, True), (37, True), (4096, True), (4096, False)])
def test_oracle_matches_eager_pytorch(B, clipped):
    t = batch(B, seed=B)
    stats, gmu, gstd, gval = reference(t, clipped)
    out, dmu, dstd, dval = oracle(t, clipped)
    assert np.allclose(out[:6], stats.numpy(), rtol=2e-5, atol=2e-6)
    assert np.allclose(dmu, gmu.numpy(), rtol=2e-4, atol=1e-7 + 2e-6 / B)
    assert np.allclose(dstd, gstd.numpy(), rtol=2e-4, atol=2e-6)
    assert np.allclose(dval, gval.numpy().ravel(), rtol=2e-5, atol=1e-9)


def test_oracle_rejects_bad_arguments():
    lib = load_oracle()
    lib.qo_ppo_loss.argtypes = [C.c_void_p] * 10 + [C.c_int64, C.c_int32] + [C.c_float] * 5 + [C.c_int32] + [C.c_void_p] * 5 + [C.c_int64, C.c_void_p]
    assert lib.qo_ppo_loss(*([None] * 10), 4, 12, 0.2, 1, 1, 1, 1, 1, *([None] * 5), 0, None) == -1


@pytest.mark.gpu
@pytest.mark.parametrize("B,clipped", [(1, True), (37, True), (24576, True), (24576, False), (100001, True)])
def test_hip_kernel_matches_oracle_and_pytorch(B, clipped):
    from quadrupedal_agility_amd.rsl_rl.algorithms.fused import ppo_loss
    t = batch(B, seed=B + 1)
    g = {k: v.cuda() for k, v in t.items()}
    mu = g["mu"].clone().requires_grad_(True); std = g["std"].clone().requires_grad_(True); value = g["value"].clone().requires_grad_(True)
    loss, stats = ppo_loss(mu, std, value, g["actions"], g["old_logp"], g["old_mu"], g["old_sigma"], g["advantages"], g["returns"],
                           g["target_values"], clipped_value=clipped, **KW)
    (3.0 * loss).backward()                                     # upstream gradient is applied
    out, dmu, dstd, dval = oracle(t, clipped)
    assert np.allclose(stats.cpu().numpy()[:6], out[:6], rtol=2e-5, atol=2e-6)
    assert float(loss.detach()) == pytest.approx(float(out[0]), rel=2e-5, abs=2e-6)
    # a sample whose ratio sits within fp32 rounding of 1 +- clip may take the other branch of the max (its gradient is
    # then 0 instead of -A ratio, or vice versa): count such rows, do not hide them
    bad = ~np.isclose(mu.grad.cpu().numpy(), 3.0 * dmu, rtol=3e-4, atol=1e-7 + 6e-6 / B)
    flipped = np.unique(np.nonzero(bad)[0])
    assert len(flipped) <= B // 30000, flipped
    if len(flipped):
        a = {k: v.double() for k, v in t.items()}
        logp = (-(a["actions"] - a["mu"]) ** 2 / (2 * a["std"] ** 2) - a["std"].log() - 0.9189385332046727).sum(-1)
        ratio = torch.exp(logp - a["old_logp"][:, 0])[flipped]
        assert (torch.minimum((ratio - 0.8).abs(), (ratio - 1.2).abs()) < 5e-6).all()
    assert np.allclose(std.grad.cpu().numpy(), 3.0 * dstd, rtol=3e-4, atol=6e-6 + 3.0 * len(flipped) / B)
    assert np.allclose(value.grad.cpu().numpy().ravel(), 3.0 * dval, rtol=2e-5, atol=1e-9)
    rs, gmu, gstd, gval = reference(g, clipped)                 # eager PyTorch on the GPU
    assert np.allclose(stats.cpu().numpy()[:6], rs.cpu().numpy(), rtol=2e-5, atol=2e-6)
    bad = ~np.isclose(mu.grad.cpu().numpy(), 3.0 * gmu.cpu().numpy(), rtol=3e-4, atol=1e-7 + 6e-6 / B)
    assert len(np.unique(np.nonzero(bad)[0])) <= B // 30000


@pytest.mark.gpu
def test_update_step_with_and_without_the_fused_kernel(tmp_path):
    """same seed, same rollout: one PPO update through qa_ppo_loss and one through eager ops end in the same weights"""
    from tests.test_gpu_train import _make
    from quadrupedal_agility_amd.legged_gym.envs import task_registry
    res = []
    for fused in (True, False):
        torch.manual_seed(0)
        env, args, tcfg = _make(256, False)
        runner, _ = task_registry.make_alg_runner(env, name="go2_locomotion", args=args, train_cfg=tcfg, log_root=None)
        runner.alg.use_fused_loss = fused
        runner.learn(1, init_at_random_ep_len=True)
        w = {k: v.clone() for k, v in runner.alg.actor_critic.state_dict().items()}
        w.update({"estimator." + k: v.clone() for k, v in runner.alg.estimator.state_dict().items()})
        res.append(w)
        lr = float(runner.alg.lr_ac)
        res.append(lr)
    (wa, lra, wb, lrb) = res
    assert lra == lrb
    for k in wa:
        assert torch.allclose(wa[k], wb[k], atol=2e-4, rtol=1e-3), k


# ------------------------------------------------------------------ Linear+ELU backward
def elu_case(rows, cols, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(rows, cols, generator=g) * 1.5
    y = torch.nn.functional.elu(x)
    gy = torch.randn(rows, cols, generator=g)
    return y, gy


@pytest.mark.parametrize("rows,cols", [(1, 1), (33, 29), (1000, 64), (4096, 512)])
def test_elu_backward_bias_oracle_matches_pytorch(rows, cols):
    y, gy = elu_case(rows, cols, rows + cols)
    lib = load_oracle()
    lib.qo_elu_backward_bias.argtypes = [C.c_void_p] * 4 + [C.c_int64, C.c_int32, C.c_float, C.c_void_p, C.c_int64, C.c_void_p]
    yn, gn = y.numpy().copy(), gy.numpy().copy()
    gin = np.zeros_like(yn); gb = np.zeros(cols, np.float32)
    assert lib.qo_elu_backward_bias(gn.ctypes.data, yn.ctypes.data, gin.ctypes.data, gb.ctypes.data, rows, cols, 1.0, None, 0, None) == 0
    want = torch.ops.aten.elu_backward(gy, 1.0, 1.0, 1.0, True, y)
    assert np.array_equal(gin, want.numpy())
    assert np.allclose(gb, want.sum(0).numpy(), rtol=1e-5, atol=1e-5 * np.sqrt(rows))


@pytest.mark.parametrize("rows,cols", [(33, 29), (921, 512), (307, 256)])
def test_elu_backward_bias_oracle_with_alpha_zero_is_the_relu_backward(rows, cols):
    """alpha = 0 (fused.linear_relu, fused.mask_times_row: the discriminator's recorded step): g * [y > 0] and its column sums"""
    g = torch.Generator().manual_seed(rows)
    y = torch.relu(torch.randn(rows, cols, generator=g)); gy = torch.randn(rows, cols, generator=g)
    lib = load_oracle()
    lib.qo_elu_backward_bias.argtypes = [C.c_void_p] * 4 + [C.c_int64, C.c_int32, C.c_float, C.c_void_p, C.c_int64, C.c_void_p]
    yn, gn = y.numpy().copy(), gy.numpy().copy()
    gin = np.zeros_like(yn); gb = np.zeros(cols, np.float32)
    assert lib.qo_elu_backward_bias(gn.ctypes.data, yn.ctypes.data, gin.ctypes.data, gb.ctypes.data, rows, cols, 0.0, None, 0, None) == 0
    want = torch.ops.aten.threshold_backward(gy, y, 0.0)
    assert np.array_equal(gin, want.numpy())
    assert np.allclose(gb, want.double().sum(0).numpy(), rtol=1e-5, atol=1e-5 * np.sqrt(rows))


@pytest.mark.gpu
@pytest.mark.parametrize("rows,cols", [(1, 1), (33, 29), (1000, 64), (24576, 128), (24576, 512), (5000, 700)])
def test_elu_backward_bias_hip(rows, cols):
    from quadrupedal_agility_amd import _capi
    lib = _capi.load_library()
    y, gy = elu_case(rows, cols, rows + cols)
    yd, gd = y.cuda(), gy.cuda()
    gin = torch.empty_like(yd); gb = torch.empty(cols, device="cuda")
    n = int(lib.qa_elu_backward_bias_scratch_bytes(rows, cols))
    scratch = torch.empty(n, dtype=torch.uint8, device="cuda")
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    P = lambda t: C.c_void_p(t.data_ptr())
    assert lib.qa_elu_backward_bias(P(gd), P(yd), P(gin), P(gb), rows, cols, 1.0, P(scratch), n, st) == 0
    assert lib.qa_elu_backward_bias(P(gd), P(yd), P(gin), P(gb), rows, cols, 1.0, P(scratch), n - 1, st) == -1      # scratch too small
    torch.cuda.synchronize()
    want = torch.ops.aten.elu_backward(gy, 1.0, 1.0, 1.0, True, y)
    assert torch.equal(gin.cpu(), want)                                       # elementwise part is exact
    assert np.allclose(gb.cpu().numpy(), want.double().sum(0).numpy(), rtol=2e-5, atol=2e-5 * np.sqrt(rows))
    gb2 = torch.empty_like(gb)
    assert lib.qa_elu_backward_bias(P(gd), P(yd), P(gin), P(gb2), rows, cols, 1.0, P(scratch), n, st) == 0
    assert torch.equal(gb, gb2)                                               # fixed summation order: bit-reproducible


@pytest.mark.gpu
def test_fused_mlp_matches_modules():
    from quadrupedal_agility_amd.rsl_rl.algorithms import fused
    torch.manual_seed(0)
    seq = torch.nn.Sequential(torch.nn.Linear(101, 512), torch.nn.ELU(), torch.nn.Linear(512, 256), torch.nn.ELU(),
                              torch.nn.Linear(256, 128), torch.nn.ELU(), torch.nn.Linear(128, 12)).cuda()
    x = torch.randn(3000, 101, device="cuda", requires_grad=True)
    w = torch.randn(3000, 12, device="cuda")
    (fused.mlp_forward(seq, x) * w).sum().backward()
    got = [p.grad.clone() for p in seq.parameters()] + [x.grad.clone()]
    for p in seq.parameters():
        p.grad = None
    x.grad = None
    (seq(x) * w).sum().backward()
    want = [p.grad for p in seq.parameters()] + [x.grad]
    for a, b in zip(got, want):
        assert torch.allclose(a, b, rtol=2e-4, atol=2e-4 * float(b.abs().max()))


# ------------------------------------------------------------------ running-moment normaliser
def norm_batches(seed, d=98):
    g = torch.Generator().manual_seed(seed)
    return [torch.randn(n, d, generator=g) * (1.0 + 3.0 * torch.rand(d, generator=g)) + torch.randn(d, generator=g) for n in (1228, 1228, 77)]


def test_normalizer_oracle_matches_reference_class():
    """oracle C vs the numpy Normalizer the reference pickles into model.pt (mirror of bbc/rsl_rl/utils/utils.py:62-103)"""
    from quadrupedal_agility_amd.rsl_rl.utils.utils import Normalizer
    lib = load_oracle()
    lib.qo_normalizer_update.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.qo_normalizer_apply.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_void_p]
    ref = Normalizer(98); ref32 = Normalizer(98)
    mean = ref.mean.astype(np.float64).copy(); var = ref.var.astype(np.float64).copy(); count = np.array([ref.count], np.float64)
    for rnd in range(3):
        bs = [b.numpy().copy() for b in norm_batches(rnd)]
        for b in bs:
            ref.update(b.astype(np.float64))         # batch moments in double, as the device path computes them
            ref32.update(b)                          # what the reference does (float32 numpy batches, gail.py:526-529)
        ptrs = (C.c_void_p * 3)(*[b.ctypes.data for b in bs]); rows = (C.c_int64 * 3)(*[b.shape[0] for b in bs])
        assert lib.qo_normalizer_update(ptrs, rows, 3, 98, mean.ctypes.data, var.ctypes.data, count.ctypes.data, None) == 0
    assert np.allclose(mean, ref.mean, rtol=1e-12, atol=1e-12) and np.allclose(var, ref.var, rtol=1e-11) and count[0] == pytest.approx(ref.count)
    assert np.allclose(mean, ref32.mean, rtol=1e-5, atol=2e-6) and np.allclose(var, ref32.var, rtol=1e-5)      # fp32 batch-moment rounding of the reference
    x = norm_batches(9)[0].numpy().copy(); y = np.zeros_like(x)
    assert lib.qo_normalizer_apply(x.ctypes.data, y.ctypes.data, x.shape[0], 98, mean.ctypes.data, var.ctypes.data, 1e-4, 10.0, None) == 0
    assert np.allclose(y, ref.normalize_torch(torch.from_numpy(x), "cpu").numpy(), rtol=1e-6, atol=1e-6)


@pytest.mark.gpu
def test_normalizer_hip_matches_eager_and_oracle():
    from quadrupedal_agility_amd.rsl_rl.algorithms import fused
    from quadrupedal_agility_amd.rsl_rl.utils.utils import TorchNormalizer
    a = TorchNormalizer(98, "cuda"); b = TorchNormalizer(98, "cuda")
    for rnd in range(3):
        bs = [x.cuda() for x in norm_batches(rnd)]
        a.update_torch(bs)                       # fused: one launch
        fused.ENABLED = False
        try:
            b.update_torch(bs)                   # eager double-precision ops
        finally:
            fused.ENABLED = True
    assert torch.allclose(a.mean, b.mean, rtol=1e-12, atol=1e-12) and torch.allclose(a.var, b.var, rtol=1e-11) and float(a.count) == float(b.count)
    x = norm_batches(9)[0].cuda()
    y = a.normalize_torch(x)
    fused.ENABLED = False
    try:
        y2 = b.normalize_torch(x)
    finally:
        fused.ENABLED = True
    assert torch.allclose(y, y2, rtol=1e-6, atol=1e-6) and float(y.abs().max()) <= 10.0
    ref = a.to_reference()
    assert type(ref).__name__ == "Normalizer" and np.allclose(ref.mean, a.mean.cpu().numpy())


@pytest.mark.gpu
def test_recorded_ppo_update_matches_eager_launches():
    """3 iterations with the PPO minibatch step replayed from a hipGraph (iterations 2-3) vs launched eagerly: same
    weights, same learning rate, same loss means"""
    from tests.test_gpu_train import _make
    from quadrupedal_agility_amd.legged_gym.envs import task_registry
    res = []
    for graph in (True, False):
        torch.manual_seed(0)
        env, args, tcfg = _make(256, False)
        runner, _ = task_registry.make_alg_runner(env, name="go2_locomotion", args=args, train_cfg=tcfg, log_root=None)
        runner.alg.use_update_graph = graph
        runner.learn(3, init_at_random_ep_len=True)
        assert (runner.alg._ac_graph is not None and runner.alg._ac_graph is not False) == graph
        res.append(({k: v.clone() for k, v in runner.alg.actor_critic.state_dict().items()},
                    {k: v.clone() for k, v in runner.alg.estimator.state_dict().items()}, float(runner.alg.lr_ac)))
    (wa, ea, lra), (wb, eb, lrb) = res
    assert lra == lrb
    for k in wa:
        assert torch.allclose(wa[k], wb[k], atol=3e-4, rtol=3e-3), k
    for k in ea:
        assert torch.allclose(ea[k], eb[k], atol=3e-4, rtol=3e-3), k


@pytest.mark.gpu
def test_data_parallel_update_as_two_graphs_around_the_collective(monkeypatch):
    """a 1-rank RCCL group on one GPU takes the data-parallel path: the PPO step is recorded as two graphs with the
    bucket all-reduce between them; with one rank the result must equal the single-graph run"""
    import torch.distributed as dist
    from tests.test_gpu_train import _make
    from tests.test_distributed_cpu import _free_port
    from quadrupedal_agility_amd.legged_gym.envs import task_registry
    res = []
    for dp in (False, True):
        if dp:
            monkeypatch.setenv("QA_FORCE_DATA_PARALLEL", "1")
            monkeypatch.setenv("MASTER_ADDR", "127.0.0.1"); monkeypatch.setenv("MASTER_PORT", str(_free_port()))
            dist.init_process_group("nccl", rank=0, world_size=1)
        try:
            torch.manual_seed(0)
            env, args, tcfg = _make(256, False)
            runner, _ = task_registry.make_alg_runner(env, name="go2_locomotion", args=args, train_cfg=tcfg, log_root=None)
            assert (runner.alg.grad_sync is not None) == dp
            runner.learn(3, init_at_random_ep_len=True)
            # data-parallel: one recorded step per minibatch slot (the collective sits between its two graphs); one GPU, chain steps (r6): the slots of an
            # epoch in ONE recording
            assert len(runner.alg._ac_graph) == (runner.alg.num_mini_batches if dp else 1)
            assert all((gb is not None) == dp for _, gb, _ in runner.alg._ac_graph)
            res.append({k: v.clone() for k, v in runner.alg.actor_critic.state_dict().items()})
            res.append(float(runner.alg.lr_ac))
        finally:
            if dp:
                dist.destroy_process_group()
    wa, lra, wb, lrb = res
    assert lra == lrb
    # the data-parallel path normalises advantages from all-reduced moments instead of inside the fused GAE kernel; the
    # 1e-7 differences are amplified by 60 Adam steps (sign-like updates early in training), so: bulk tight, tails bounded
    for k in wa:
        d = (wa[k] - wb[k]).abs()
        assert float((d > 3e-4 + 3e-3 * wb[k].abs()).float().mean()) < 0.005 and float(d.max()) < 5e-3, k


# ------------------------------------------------------------------ clip + Adam
def _adam_twins(dev, wd=0.0, lr=1e-3, lr_tensor=False):
    torch.manual_seed(3)
    mk = lambda: torch.nn.Sequential(torch.nn.Linear(37, 129), torch.nn.ELU(), torch.nn.Linear(129, 5000), torch.nn.ELU(), torch.nn.Linear(5000, 3)).to(dev)
    a, b = mk(), mk()
    b.load_state_dict(a.state_dict())
    kw = dict(fused=True, capturable=True) if dev == "cuda" else {}
    mklr = lambda: torch.tensor(lr, device=dev) if lr_tensor else lr
    oa = torch.optim.Adam([{"params": a.parameters(), "weight_decay": wd}], lr=mklr(), **kw)
    ob = torch.optim.Adam([{"params": b.parameters(), "weight_decay": wd}], lr=mklr(), **kw)
    return a, b, oa, ob


def _close_params(x, y, lr=1e-3, steps=6):
    """Adam's update lr * m / (sqrt(v) + eps) is ill-conditioned where the effective gradient (wd p + coef grad) cancels
    to ~eps: there a 1e-6 relative difference in the clip coefficient moves the update by a fraction of lr.  So: all
    but a sliver of the elements agree to fp32 rounding, and nothing is off by more than the steps taken."""
    d = (x.detach().double().cpu() - y.detach().double().cpu()).abs().reshape(-1)
    tol = 2e-7 + 2e-5 * y.detach().double().cpu().abs().reshape(-1)
    assert float((d > tol).double().mean()) < 2e-4 and float(d.max()) < 2 * lr * steps


def _fake_grads(a, b, k, scale):
    g = torch.Generator().manual_seed(100 + k)
    for pa, pb in zip(a.parameters(), b.parameters()):
        gr = (torch.randn(pa.shape, generator=g) * scale).to(pa.device)
        pa.grad = gr.clone(); pb.grad = gr.clone()


def test_clip_adam_oracle_matches_pytorch():
    a, b, oa, ob = _adam_twins("cpu", wd=1e-3)
    lib = load_oracle()
    lib.qo_clip_adam_step.argtypes = [C.c_void_p] * 5 + [C.c_int32] + [C.c_void_p] * 3 + [C.c_int32, C.c_void_p, C.c_void_p] + [C.c_float] * 4 + [C.c_void_p, C.c_int64, C.c_void_p]
    ps = list(b.parameters())
    m = [np.zeros(p.numel(), np.float32) for p in ps]; v = [np.zeros(p.numel(), np.float32) for p in ps]; st = [np.zeros(1, np.float32) for _ in ps]
    pn = [p.detach().numpy().reshape(-1).copy() for p in ps]
    ct, cs, cl = [], [], []
    for t, p in enumerate(ps):
        for s0 in range(0, p.numel(), 2048):
            ct.append(t); cs.append(s0); cl.append(min(2048, p.numel() - s0))
    ct, cs, cl = (np.array(x, np.int32) for x in (ct, cs, cl))
    wd = np.full(len(ps), 1e-3, np.float32); lr = np.array([1e-3], np.float32); scratch = np.zeros(4 + len(ct), np.float32)
    tab = lambda arrs: (C.c_void_p * len(arrs))(*[x.ctypes.data for x in arrs])
    for k in range(4):
        _fake_grads(a, b, k, 3.0 if k % 2 else 0.01)            # clipped and unclipped steps
        gn = [p.grad.numpy().reshape(-1).copy() for p in ps]
        torch.nn.utils.clip_grad_norm_(a.parameters(), 1.0)
        oa.step()
        rc = lib.qo_clip_adam_step(tab(pn), tab(gn), tab(m), tab(v), tab(st), len(ps), ct.ctypes.data, cs.ctypes.data, cl.ctypes.data, len(ct),
                                   wd.ctypes.data, lr.ctypes.data, 0.9, 0.999, 1e-8, 1.0, scratch.ctypes.data, scratch.size, None)
        assert rc == 0 and st[0][0] == k + 1
    for p, q in zip(a.parameters(), pn):
        _close_params(p.reshape(-1), torch.from_numpy(q), steps=4)


@pytest.mark.gpu
@pytest.mark.parametrize("wd,lr_tensor,max_norm", [(0.0, True, 1.0), (1e-3, False, None)])
def test_clip_adam_hip_matches_pytorch(wd, lr_tensor, max_norm):
    from quadrupedal_agility_amd.rsl_rl.algorithms.fused import ClipAdam
    a, b, oa, ob = _adam_twins("cuda", wd=wd, lr_tensor=lr_tensor)
    stepper = ClipAdam(ob, max_norm)
    for k in range(6):
        _fake_grads(a, b, k, 3.0 if k % 2 else 0.01)
        if max_norm:
            torch.nn.utils.clip_grad_norm_(a.parameters(), max_norm)
        oa.step()
        stepper.step()                                          # step 0 goes through PyTorch (no state yet), then the kernel
        if k == 2 and lr_tensor:
            for o in (oa, ob):
                o.param_groups[0]["lr"].mul_(0.5)               # the KL rule rewrites the device LR in place
    assert stepper._tab is not None
    for p, q in zip(a.parameters(), b.parameters()):
        _close_params(p, q)
    sa, sb = oa.state_dict()["state"], ob.state_dict()["state"]
    for i in sa:
        assert float(sa[i]["step"]) == float(sb[i]["step"]) == 6.0
        # with weight decay the moments inherit the (ill-conditioned, see _close_params) parameter differences x wd
        assert torch.allclose(sa[i]["exp_avg"], sb[i]["exp_avg"], rtol=2e-5, atol=1e-6 if wd else 1e-8)
        assert torch.allclose(sa[i]["exp_avg_sq"], sb[i]["exp_avg_sq"], rtol=1e-4, atol=1e-9)


def _two_nets(dev, seed=11):
    torch.manual_seed(seed)
    est = torch.nn.Sequential(torch.nn.Linear(53, 128), torch.nn.ELU(), torch.nn.Linear(128, 9)).to(dev)
    ac = torch.nn.Sequential(torch.nn.Linear(101, 512), torch.nn.ELU(), torch.nn.Linear(512, 2300), torch.nn.ELU(), torch.nn.Linear(2300, 12)).to(dev)
    return est, ac


def _grads_for(mods, k, scale):
    g = torch.Generator().manual_seed(200 + k)
    for m in mods:
        for p in m.parameters():
            p.grad = (torch.randn(p.shape, generator=g) * scale).to(p.device)


@pytest.mark.gpu
def test_clip_adam_pair_equals_the_two_steps_and_the_rule_one_after_the_other():
    """qa_clip_adam_pair_step (ABI 18): estimator step, KL rule on the actor-critic's learning rate, actor-critic step in three launches -- the same
    kernels with the same sums as the seven launches one after the other: bit-identical parameters, moments, step counts and learning rate"""
    from quadrupedal_agility_amd.rsl_rl.algorithms import fused
    res = []
    for paired in (False, True):
        est, ac = _two_nets("cuda")
        kw = dict(fused=True, capturable=True)
        lr_ac = torch.tensor(1e-3, device="cuda")
        o_est, o_ac = torch.optim.Adam(est.parameters(), lr=2e-4, **kw), torch.optim.Adam([{"params": ac.parameters(), "weight_decay": 1e-4}], lr=lr_ac, **kw)
        s_est, s_ac = fused.ClipAdam(o_est, 1.0), fused.ClipAdam(o_ac, 0.7)
        pair = fused.ClipAdamPair(s_est, s_ac)
        took = []
        for k in range(7):
            _grads_for((est, ac), k, 3.0 if k % 2 else 0.01)          # clipped and unclipped steps
            kl = torch.tensor([0.05, 0.001, 0.01, 0.0, 0.03, 0.002, 0.01][k], device="cuda")       # down, up, keep, keep (0), down, up, keep
            if paired and pair.step(kl, 0.01):
                took.append(k)
                continue
            s_est.step()
            fused.kl_lr_rule(kl, 0.01, lr_ac)
            s_ac.step()
        assert took == (list(range(1, 7)) if paired else [])          # step 0 creates the optimisers' state through PyTorch
        torch.cuda.synchronize()
        st = [p.detach().clone() for m in (est, ac) for p in m.parameters()]
        for o in (o_est, o_ac):
            for p in o.param_groups[0]["params"]:
                st += [o.state[p]["exp_avg"].clone(), o.state[p]["exp_avg_sq"].clone(), o.state[p]["step"].clone().float()]
        res.append((st, float(lr_ac)))
    (a, la), (b, lb) = res
    assert la == lb and la != 1e-3
    for i, (x, y) in enumerate(zip(a, b)):
        assert torch.equal(x, y), (i, float((x - y).abs().max()))


def test_clip_adam_pair_twin_equals_its_halves():
    """the C twin of the pair entry against the single-optimiser twin called twice with the rule between (host tables)"""
    lib = load_oracle()
    single = lib.qo_clip_adam_step_reduce
    single.argtypes = [C.c_void_p] * 5 + [C.c_int32] + [C.c_void_p] * 3 + [C.c_int32, C.c_void_p, C.c_void_p] + [C.c_float] * 4 + [C.c_void_p, C.c_int64] + [C.c_void_p] * 4
    pairf = lib.qo_clip_adam_pair_step
    pairf.argtypes = [C.c_void_p] * 5 + [C.c_int32] + [C.c_void_p] * 3 + [C.c_int32, C.c_void_p, C.c_void_p] + [C.c_float] * 4 + [C.c_void_p, C.c_int64] + [C.c_void_p] * 5
    from quadrupedal_agility_amd._capi import QaAdamPair
    rng = np.random.default_rng(4)
    sizes = [53 * 16, 16, 16 * 3, 101 * 40, 40, 5000]
    nt = 3
    def state():
        r = np.random.default_rng(9)
        return ([r.standard_normal(n).astype(np.float32) for n in sizes], [np.zeros(n, np.float32) for n in sizes], [np.zeros(n, np.float32) for n in sizes],
                [np.zeros(1, np.float32) for _ in sizes])
    ct, cs, cl = [], [], []
    for t, n in enumerate(sizes):
        for s0 in range(0, n, 2048):
            ct.append(t); cs.append(s0); cl.append(min(2048, n - s0))
    nc = sum(1 for t in ct if t < nt)
    ct, cs, cl = (np.array(x, np.int32) for x in (ct, cs, cl))
    wd = np.array([0, 0, 0, 1e-4, 1e-4, 1e-4], np.float32)
    tab = lambda arrs: (C.c_void_p * len(arrs))(*[x.ctypes.data for x in arrs])
    zero = lambda ty, n: (ty * n)()
    out = []
    for paired in (True, False):
        p, m, v, st = state()
        lr1, lr2 = np.array([2e-4], np.float32), np.array([1e-3], np.float32)
        scratch = np.zeros(len(ct) + 9, np.float32)
        for k in range(4):
            r = np.random.default_rng(50 + k)
            g = [(r.standard_normal(n) * (3.0 if k % 2 else 0.01)).astype(np.float32) for n in sizes]
            kl = np.array([[0.05, 0.001, 0.01, 0.03][k]], np.float32)
            if paired:
                pr = QaAdamPair(nt, nc, lr2.ctypes.data, 0.7, kl.ctypes.data, 0.01, 1.5, 1e-5, 1e-2)
                rc = pairf(tab(p), tab(g), tab(m), tab(v), tab(st), len(sizes), ct.ctypes.data, cs.ctypes.data, cl.ctypes.data, len(ct), wd.ctypes.data, lr1.ctypes.data,
                           0.9, 0.999, 1e-8, 1.0, scratch.ctypes.data, scratch.size, None, None, None, C.byref(pr), None)
                assert rc == 0
            else:
                n2 = len(sizes) - nt
                rc = single(tab(p[:nt]), tab(g[:nt]), tab(m[:nt]), tab(v[:nt]), tab(st[:nt]), nt, ct.ctypes.data, cs.ctypes.data, cl.ctypes.data, nc, wd.ctypes.data,
                            lr1.ctypes.data, 0.9, 0.999, 1e-8, 1.0, scratch.ctypes.data, scratch.size, zero(C.c_void_p, nt), zero(C.c_int64, nt), zero(C.c_int32, nt), None)
                assert rc == 0
                if kl[0] > 0.02: lr2[0] = max(np.float32(1e-5), lr2[0] / np.float32(1.5))
                elif 0 < kl[0] < 0.005: lr2[0] = min(np.float32(1e-2), lr2[0] * np.float32(1.5))
                ct2 = (ct[nc:] - nt).astype(np.int32); cs2 = cs[nc:].copy(); cl2 = cl[nc:].copy(); wd2 = wd[nt:].copy()
                rc = single(tab(p[nt:]), tab(g[nt:]), tab(m[nt:]), tab(v[nt:]), tab(st[nt:]), n2, ct2.ctypes.data, cs2.ctypes.data, cl2.ctypes.data, len(ct2), wd2.ctypes.data,
                            lr2.ctypes.data, 0.9, 0.999, 1e-8, 0.7, scratch.ctypes.data, scratch.size, zero(C.c_void_p, n2), zero(C.c_int64, n2), zero(C.c_int32, n2), None)
                assert rc == 0
        out.append((p, m, v, st, float(lr2[0])))
    a, b = out
    assert a[4] == b[4] != 1e-3
    for xs, ys in zip(a[:4], b[:4]):
        for x, y in zip(xs, ys):
            assert np.array_equal(x, y)
    assert a[3][0][0] == 4 and a[3][-1][0] == 4


# ------------------------------------------------------------------ rollout bookkeeping kernels
def _rollout_case(N, seed):
    g = torch.Generator().manual_seed(seed)
    return dict(mean=torch.randn(N, 12, generator=g), std=0.3 + torch.rand(12, generator=g), value=torch.randn(N, 1, generator=g),
                noise=torch.randn(N, 12, generator=g), rew=torch.rand(N, generator=g), reset=(torch.rand(N, generator=g) < 0.2).long(),
                time_out=(torch.rand(N, generator=g) < 0.1).to(torch.uint8), cur=torch.rand(6, N, generator=g))


def _bind_rollout(lib, prefix):
    getattr(lib, prefix + "rollout_act").argtypes = [C.c_void_p] * 4 + [C.c_uint64, C.c_void_p, C.c_int64, C.c_int32, C.c_int32] + [C.c_void_p] * 7
    getattr(lib, prefix + "rollout_post").argtypes = [C.c_void_p] * 4 + [C.c_float, C.c_float, C.c_int32] + [C.c_void_p] * 6


def test_rollout_kernels_oracle_matches_pytorch_expressions():
    from torch.distributions import Normal
    N = 777
    c = _rollout_case(N, 1)
    lib = load_oracle(); _bind_rollout(lib, "qo_")
    n = {k: np.ascontiguousarray(v.numpy()) for k, v in c.items()}
    out = {k: np.zeros((N, 12), np.float32) for k in ("actions", "sa", "smu", "ssig")}
    logp = np.zeros(N, np.float32); sval = np.zeros(N, np.float32)
    p = lambda x: x.ctypes.data
    assert lib.qo_rollout_act(p(n["mean"]), p(n["std"]), p(n["value"]), p(n["noise"]), 1, None, 5, N, 0, p(out["actions"]), p(out["sa"]), p(out["smu"]),
                              p(out["ssig"]), p(logp), p(sval), None) == 0
    act = c["mean"] + c["std"] * c["noise"]
    want = Normal(c["mean"], c["mean"] * 0 + c["std"]).log_prob(act).sum(-1)
    assert np.allclose(out["actions"], act.numpy(), atol=1e-6) and np.array_equal(out["actions"], out["sa"])
    assert np.allclose(logp, want.numpy(), rtol=1e-5, atol=1e-5) and np.array_equal(out["smu"], n["mean"]) and np.allclose(out["ssig"], n["std"][None])
    assert np.array_equal(sval, n["value"][:, 0])
    # Philox path: N(0,1) noise, reproducible, different per step
    a1 = np.zeros((N, 12), np.float32); a2 = np.zeros((N, 12), np.float32); a3 = np.zeros((N, 12), np.float32)
    zero = np.zeros((N, 12), np.float32); one = np.ones(12, np.float32)
    for buf, step in ((a1, 7), (a2, 7), (a3, 8)):
        lib.qo_rollout_act(p(zero), p(one), p(n["value"]), None, 42, None, step, N, 0, p(buf), p(out["sa"]), p(out["smu"]), p(out["ssig"]), p(logp), p(sval), None)
    assert np.array_equal(a1, a2) and not np.array_equal(a1, a3)
    assert abs(a1.mean()) < 0.03 and abs(a1.std() - 1.0) < 0.03
    # post
    st_r = np.zeros(N, np.float32); st_d = np.zeros(N, np.uint8); cur = n["cur"].copy(); fin = np.zeros((6, N), np.float32); mask = np.zeros(N, np.uint8)
    assert lib.qo_rollout_post(p(n["rew"]), p(n["reset"]), p(n["time_out"]), p(n["value"]), 0.2, 0.99, N, p(st_r), p(st_d), p(cur), p(fin), p(mask), None) == 0
    r = 0.2 * c["rew"]
    assert np.allclose(st_r, (r + 0.99 * c["value"][:, 0] * c["time_out"].float()).numpy(), atol=1e-6)
    assert np.array_equal(st_d, (c["reset"] > 0).numpy().astype(np.uint8)) and np.array_equal(mask, st_d)
    want_cur = c["cur"] + torch.stack([r, torch.zeros(N), torch.zeros(N), torch.zeros(N), c["rew"], torch.ones(N)])
    assert np.allclose(fin, want_cur.numpy(), atol=1e-6) and np.allclose(cur, (want_cur * (c["reset"] == 0)).numpy(), atol=1e-6)


@pytest.mark.gpu
def test_rollout_kernels_hip_match_oracle():
    from quadrupedal_agility_amd import _capi
    N = 4099
    c = _rollout_case(N, 2)
    lo = load_oracle(); _bind_rollout(lo, "qo_")
    lib = _capi.load_library()
    n = {k: np.ascontiguousarray(v.numpy()) for k, v in c.items()}
    d = {k: v.cuda() for k, v in c.items()}
    p = lambda x: x.ctypes.data
    P = lambda t: C.c_void_p(t.data_ptr())
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    step_dev = torch.tensor([123], dtype=torch.int64, device="cuda")
    for noise in (True, False):
        o = {k: np.zeros((N, 12), np.float32) for k in ("a", "sa", "smu", "ssig")}; ol = np.zeros(N, np.float32); ov = np.zeros(N, np.float32)
        g = {k: torch.zeros(N, 12, device="cuda") for k in ("a", "sa", "smu", "ssig")}; gl = torch.zeros(N, device="cuda"); gv = torch.zeros(N, device="cuda")
        assert lo.qo_rollout_act(p(n["mean"]), p(n["std"]), p(n["value"]), p(n["noise"]) if noise else None, 9, None, 123, N, 0,
                                 p(o["a"]), p(o["sa"]), p(o["smu"]), p(o["ssig"]), p(ol), p(ov), None) == 0
        assert lib.qa_rollout_act(P(d["mean"]), P(d["std"]), P(d["value"]), P(d["noise"]) if noise else None, 9, P(step_dev), 0, N, 0,
                                  P(g["a"]), P(g["sa"]), P(g["smu"]), P(g["ssig"]), P(gl), P(gv), st) == 0
        torch.cuda.synchronize()
        tol = dict(atol=1e-6) if noise else dict(atol=2e-5, rtol=1e-5)          # Philox path: libm vs device log/sincos
        assert np.allclose(g["a"].cpu().numpy(), o["a"], **tol) and np.allclose(gl.cpu().numpy(), ol, rtol=1e-4, atol=1e-4)
        assert torch.equal(g["a"], g["sa"]) and np.array_equal(g["smu"].cpu().numpy(), n["mean"]) and np.array_equal(gv.cpu().numpy(), n["value"][:, 0])
    o_r = np.zeros(N, np.float32); o_d = np.zeros(N, np.uint8); o_cur = n["cur"].copy(); o_fin = np.zeros((6, N), np.float32); o_m = np.zeros(N, np.uint8)
    lo.qo_rollout_post(p(n["rew"]), p(n["reset"]), p(n["time_out"]), p(n["value"]), 0.2, 0.99, N, p(o_r), p(o_d), p(o_cur), p(o_fin), p(o_m), None)
    g_r = torch.zeros(N, device="cuda"); g_d = torch.zeros(N, dtype=torch.uint8, device="cuda"); g_cur = d["cur"].clone()
    g_fin = torch.zeros(6, N, device="cuda"); g_m = torch.zeros(N, dtype=torch.bool, device="cuda")
    assert lib.qa_rollout_post(P(d["rew"]), P(d["reset"]), P(d["time_out"]), P(d["value"]), 0.2, 0.99, N, P(g_r), P(g_d), P(g_cur), P(g_fin), P(g_m), st) == 0
    torch.cuda.synchronize()
    assert np.allclose(g_r.cpu().numpy(), o_r, atol=1e-6) and np.array_equal(g_d.cpu().numpy(), o_d) and np.array_equal(g_m.cpu().numpy().astype(np.uint8), o_m)
    assert np.allclose(g_cur.cpu().numpy(), o_cur, atol=1e-6) and np.allclose(g_fin.cpu().numpy(), o_fin, atol=1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("N", [1, 7, 512, 4099])
def test_rollout_act_store_is_rollout_act_plus_the_row_copy(N):
    """qa_rollout_act_store (ABI 18): the sampling launch with the observation rows' copy into the (padded) storage folded in -- every output bit-identical to
    qa_rollout_act, the rows exact, the storage's padding column untouched"""
    from quadrupedal_agility_amd import _capi
    lib = _capi.load_library()
    c = {k: v.cuda() for k, v in _rollout_case(N, 3).items()}
    P = lambda t: C.c_void_p(t.data_ptr())
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    step_dev = torch.tensor([77], dtype=torch.int64, device="cuda")
    obs = torch.randn(N, 672, device="cuda")[:, :671]                      # rows of a wider arena tensor
    res = []
    for store in (False, True):
        g = {k: torch.zeros(N, 12, device="cuda") for k in ("a", "sa", "smu", "ssig")}; gl = torch.zeros(N, device="cuda"); gv = torch.zeros(N, device="cuda")
        so_full = torch.full((N, 688), -7.0, device="cuda"); so = so_full[:, :671]
        if store:
            assert lib.qa_rollout_act_store(P(c["mean"]), P(c["std"]), P(c["value"]), None, 9, P(step_dev), 0, N, 5, P(g["a"]), P(g["sa"]), P(g["smu"]), P(g["ssig"]), P(gl), P(gv),
                                            P(obs), obs.stride(0), 671, P(so), so.stride(0), st) == 0
        else:
            assert lib.qa_rollout_act(P(c["mean"]), P(c["std"]), P(c["value"]), None, 9, P(step_dev), 0, N, 5, P(g["a"]), P(g["sa"]), P(g["smu"]), P(g["ssig"]), P(gl), P(gv), st) == 0
            so.copy_(obs)
        torch.cuda.synchronize()
        res.append([g[k] for k in ("a", "sa", "smu", "ssig")] + [gl, gv, so_full])
    for x, y in zip(*res):
        assert torch.equal(x, y)
    assert torch.equal(res[1][-1][:, :671], obs) and bool((res[1][-1][:, 671:] == -7.0).all())


def test_rollout_act_store_twin():
    lo = load_oracle()
    lo.qo_rollout_act_store.argtypes = [C.c_void_p] * 4 + [C.c_uint64, C.c_void_p, C.c_int64, C.c_int32, C.c_int32] + [C.c_void_p] * 6 + [C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p]
    N = 37
    n = {k: np.ascontiguousarray(v.numpy()) for k, v in _rollout_case(N, 4).items()}
    p = lambda x: x.ctypes.data
    o = {k: np.zeros((N, 12), np.float32) for k in ("a", "sa", "smu", "ssig")}; ol = np.zeros(N, np.float32); ov = np.zeros(N, np.float32)
    obs = np.random.default_rng(0).standard_normal((N, 16)).astype(np.float32); so = np.full((N, 20), -1.0, np.float32)
    assert lo.qo_rollout_act_store(p(n["mean"]), p(n["std"]), p(n["value"]), p(n["noise"]), 9, None, 1, N, 0, p(o["a"]), p(o["sa"]), p(o["smu"]), p(o["ssig"]), p(ol), p(ov),
                                   p(obs), 16, 13, p(so), 20, None) == 0
    assert np.array_equal(so[:, :13], obs[:, :13]) and (so[:, 13:] == -1.0).all() and np.allclose(o["a"], n["mean"] + n["std"] * n["noise"], atol=1e-6)


@pytest.mark.gpu
def test_fused_rollout_fills_the_storage_consistently():
    from torch.distributions import Normal
    from tests.test_gpu_train import _make
    from quadrupedal_agility_amd.legged_gym.envs import task_registry
    torch.manual_seed(0)
    env, args, tcfg = _make(512, False)
    runner, _ = task_registry.make_alg_runner(env, name="go2_locomotion", args=args, train_cfg=tcfg, log_root=None)
    runner.learn(1, init_at_random_ep_len=True)
    runner._collect(False, True)                           # one more (recorded) rollout; storage is left full
    st = runner.alg.storage
    assert st.step == 24
    logp = Normal(st.mu, st.sigma).log_prob(st.actions).sum(-1, keepdim=True)
    assert torch.allclose(logp, st.actions_log_prob, rtol=1e-4, atol=1e-4)
    z = (st.actions - st.mu) / st.sigma
    assert abs(float(z.mean())) < 0.02 and abs(float(z.std()) - 1.0) < 0.02 and torch.equal(st.sigma[0, 0], runner.alg.actor_critic.std.detach())
    assert torch.isfinite(st.values).all() and torch.isfinite(st.rewards).all() and (st.rewards >= 0).all()
    assert 0 < int(st.dones.sum()) < 24 * 512 // 4
    z0 = z.clone(); runner.alg.storage.clear(); runner._collect(False, True)
    assert not torch.equal(z0, (st.actions - st.mu) / st.sigma)                  # the device step counter moved the Philox key


# ------------------------------------------------------------------ discriminator head losses
def _disc_case(b_lb, b_pi, b_ulb, seed):
    g = torch.Generator().manual_seed(seed)
    B = b_lb + b_pi + b_ulb
    c = torch.clamp(torch.softmax(2.0 * torch.randn(B, 5, generator=g), -1), 1e-20)
    return dict(d=torch.randn(B, 1, generator=g), eps=torch.randn(B, 1, generator=g), c=c, label=torch.randint(0, 5, (b_lb,), generator=g),
                pol_eps=torch.rand(b_pi, 1, generator=g) * 2 - 1, pol_c=torch.nn.functional.one_hot(torch.randint(0, 5, (b_pi,), generator=g), 5).float())


DKW = dict(c_ss=1.0, c_disc=1.0, c_us=1.0)


def _disc_reference(t, sizes, info_coef, dev="cpu"):
    from quadrupedal_agility_amd.rsl_rl.algorithms.fused import disc_loss_reference
    d = t["d"].to(dev).clone().requires_grad_(True); e = t["eps"].to(dev).clone().requires_grad_(True); c = t["c"].to(dev).clone().requires_grad_(True)
    loss, stats = disc_loss_reference(d, e, c, t["label"].to(dev), t["pol_eps"].to(dev), t["pol_c"].to(dev), *sizes, info_coef=info_coef, **DKW)
    loss.backward()
    return stats, d.grad, e.grad, c.grad


def _disc_oracle(t, sizes, info_coef):
    lib = load_oracle()
    lib.qo_disc_loss.argtypes = [C.c_void_p] * 6 + [C.c_int32] * 3 + [C.c_float, C.c_void_p, C.c_float, C.c_float] + [C.c_void_p] * 5 + [C.c_int64, C.c_void_p]
    n = {k: np.ascontiguousarray(v.numpy()) for k, v in t.items()}
    B = sum(sizes)
    gd = np.zeros(B, np.float32); ge = np.zeros(B, np.float32); gc = np.zeros((B, 5), np.float32); out = np.zeros(16, np.float32)
    ic = np.array([info_coef], np.float32)
    p = lambda x: x.ctypes.data
    assert lib.qo_disc_loss(p(n["d"]), p(n["eps"]), p(n["c"]), p(n["label"]), p(n["pol_eps"]), p(n["pol_c"]), *sizes, DKW["c_ss"], p(ic), DKW["c_disc"],
                            DKW["c_us"], p(gd), p(ge), p(gc), p(out), None, 0, None) == 0
    return out, gd, ge, gc


@pytest.mark.parametrize("sizes", [(1, 1, 1), (37, 41, 29), (1228, 1228, 1228)])
def test_disc_loss_oracle_matches_pytorch(sizes):
    t = _disc_case(*sizes, seed=sum(sizes))
    stats, gd, ge, gc = _disc_reference(t, sizes, 0.37)
    out, od, oe, oc = _disc_oracle(t, sizes, 0.37)
    assert np.allclose(out[:14], stats.numpy()[:14], rtol=2e-5, atol=2e-6)
    assert np.allclose(od, gd.numpy().ravel(), rtol=1e-5, atol=1e-9) and np.allclose(oe, ge.numpy().ravel(), rtol=1e-5, atol=1e-9)
    assert np.allclose(oc, gc.numpy(), rtol=2e-4, atol=1e-8)


@pytest.mark.gpu
@pytest.mark.parametrize("sizes", [(1, 1, 1), (37, 41, 29), (1228, 1228, 1228)])
def test_disc_loss_hip_matches_oracle_and_pytorch(sizes):
    from quadrupedal_agility_amd.rsl_rl.algorithms.fused import disc_loss
    t = _disc_case(*sizes, seed=sum(sizes) + 1)
    g = {k: v.cuda() for k, v in t.items()}
    d = g["d"].clone().requires_grad_(True); e = g["eps"].clone().requires_grad_(True); c = g["c"].clone().requires_grad_(True)
    ic = torch.tensor(0.37, device="cuda")
    loss, stats = disc_loss(d, e, c, g["label"], g["pol_eps"], g["pol_c"], *sizes, info_coef_dev=ic, **DKW)
    (2.0 * loss).backward()
    out, od, oe, oc = _disc_oracle(t, sizes, 0.37)
    assert np.allclose(stats.cpu().numpy()[:14], out[:14], rtol=2e-5, atol=2e-6)
    assert np.allclose(d.grad.cpu().numpy().ravel(), 2 * od, rtol=1e-5, atol=1e-9) and np.allclose(e.grad.cpu().numpy().ravel(), 2 * oe, rtol=1e-5, atol=1e-9)
    assert np.allclose(c.grad.cpu().numpy(), 2 * oc, rtol=3e-4, atol=1e-8)
    rs, rd, re_, rc = _disc_reference(t, sizes, 0.37, dev="cuda")
    assert np.allclose(stats.cpu().numpy()[:14], rs.cpu().numpy()[:14], rtol=2e-5, atol=2e-6)
    assert np.allclose(c.grad.cpu().numpy(), 2 * rc.cpu().numpy(), rtol=3e-4, atol=1e-8)


@pytest.mark.gpu
@pytest.mark.parametrize("sizes", [(1, 1, 1), (37, 41, 29), (1228, 1228, 1228)])
def test_disc_loss_from_logits_is_softmax_loss_softmax_backward(sizes):
    """qa_disc_loss_logits (ABI 18): the class head's softmax and its backward inside the objective's launch, against torch.softmax -> qa_disc_loss ->
    torch's softmax backward (three launches more per discriminator step)"""
    from quadrupedal_agility_amd.rsl_rl.algorithms import fused
    b_lb, b_pi, b_ulb = sizes
    B = sum(sizes)
    g = torch.Generator().manual_seed(B)
    d, eps = torch.randn(B, 1, generator=g).cuda(), torch.randn(B, 1, generator=g).cuda()
    logits = (torch.randn(B, 5, generator=g) * 3).cuda()
    logits[0, 0] = 200.0                                              # a saturated row: the other classes' probabilities underflow below the clamp
    label = torch.randint(0, 5, (b_lb,), generator=g).cuda()
    pe, pc = torch.randn(b_pi, 1, generator=g).cuda(), torch.softmax(torch.randn(b_pi, 5, generator=g), -1).cuda()
    kw = dict(c_ss=1.3, info_coef_dev=torch.tensor(0.7, device="cuda"), c_disc=0.9, c_us=0.4)
    hs1, gd1, ge1, gl1 = fused.disc_loss_raw(d, eps, logits, label, pe, pc, b_lb, b_pi, b_ulb, from_logits=True, **kw)
    c = torch.softmax(logits, -1)
    hs0, gd0, ge0, gc0 = fused.disc_loss_raw(d, eps, c, label, pe, pc, b_lb, b_pi, b_ulb, **kw)
    gl0 = torch._softmax_backward_data(gc0, c, -1, torch.float32)
    assert torch.allclose(hs1, hs0, rtol=1e-6, atol=1e-7) and torch.equal(gd1, gd0) and torch.equal(ge1, ge0)
    assert torch.allclose(gl1, gl0, rtol=2e-5, atol=1e-9), float((gl1 - gl0).abs().max())


def test_disc_loss_logits_twin_equals_softmax_around_the_twin():
    lib = load_oracle()
    sig = [C.c_void_p] * 6 + [C.c_int32] * 3 + [C.c_float, C.c_void_p, C.c_float, C.c_float] + [C.c_void_p] * 5 + [C.c_int64, C.c_void_p]
    lib.qo_disc_loss.argtypes = sig; lib.qo_disc_loss_logits.argtypes = sig
    b_lb, b_pi, b_ulb = 5, 7, 6
    B = 18
    r = np.random.default_rng(1)
    d, eps = r.standard_normal(B).astype(np.float32), r.standard_normal(B).astype(np.float32)
    z = (r.standard_normal((B, 5)) * 2).astype(np.float32)
    lab = r.integers(0, 5, b_lb).astype(np.int64); pe = r.standard_normal(b_pi).astype(np.float32)
    pc = np.abs(r.standard_normal((b_pi, 5))).astype(np.float32); info = np.array([0.7], np.float32)
    p = lambda x: x.ctypes.data
    def run(fn, c):
        gd, ge, gc, out = np.zeros(B, np.float32), np.zeros(B, np.float32), np.zeros((B, 5), np.float32), np.zeros(16, np.float32)
        assert fn(p(d), p(eps), p(c), p(lab), p(pe), p(pc), b_lb, b_pi, b_ulb, 1.3, p(info), 0.9, 0.4, p(gd), p(ge), p(gc), p(out), None, 0, None) == 0
        return gd, ge, gc, out
    gd1, ge1, gl1, o1 = run(lib.qo_disc_loss_logits, z)
    e = np.exp(z - z.max(1, keepdims=True)); sm = (e / e.sum(1, keepdims=True)).astype(np.float32)
    gd0, ge0, gc0, o0 = run(lib.qo_disc_loss, sm)
    gl0 = sm * (gc0 - (gc0 * sm).sum(1, keepdims=True))
    assert np.allclose(o1, o0, rtol=1e-5, atol=1e-6) and np.allclose(gd1, gd0) and np.allclose(ge1, ge0) and np.allclose(gl1, gl0, rtol=1e-4, atol=1e-7)


@pytest.mark.gpu
def test_amp_iteration_with_and_without_fused_heads():
    """one AMP iteration (80 discriminator steps, eager) through qa_disc_loss and through the eager head losses"""
    from tests.test_gpu_train import _make
    from quadrupedal_agility_amd.legged_gym.envs import task_registry
    res = []
    for fused in (True, False):
        torch.manual_seed(0)
        env, args, tcfg = _make(256, True)
        runner, _ = task_registry.make_alg_runner(env, name="go2_locomotion", args=args, train_cfg=tcfg, log_root=None)
        runner.alg.use_fused_loss = fused
        runner.learn(1, init_at_random_ep_len=True)
        res.append({k: v.clone() for k, v in runner.alg.disc.state_dict().items()})
        res.append(env.prior_parameters.clone())
    wa, pa, wb, pb = res
    assert torch.allclose(pa, pb, atol=1e-5)
    for k in wa:
        d = (wa[k] - wb[k]).abs()
        assert float((d > 2e-4 + 2e-3 * wb[k].abs()).float().mean()) < 0.01 and float(d.max()) < 5e-3, k


# ------------------------------------------------------------------ discriminator input preparation
def _prep_case(seed):
    g = torch.Generator().manual_seed(seed)
    bs = [torch.randn(n, 98, generator=g) * 3 for n in (50, 61, 7)]
    task = torch.zeros(2, 49); task[:, 3:9] = 1; task[:, 33:] = 1
    fm = (torch.arange(2, dtype=torch.float32) * 0.5 + 1).view(-1, 1).repeat(1, 49)
    mean = torch.randn(98, generator=g, dtype=torch.float64); var = torch.rand(98, generator=g, dtype=torch.float64) + 0.1
    return bs, task.view(-1).contiguous(), fm.view(-1).contiguous(), torch.tensor(0.37), mean, var


def _prep_eager(bs, task, fm, w, mean, var):
    outs = []
    for b in bs:
        x = b.clone().view(len(b), 2, 49)
        x[:, :, 3:9] *= w; x[:, :, 33:] *= w
        x = x.reshape(len(b), -1) * fm
        outs.append(torch.clamp((x - mean.float()) / torch.sqrt((var + 1e-4).float()), -10.0, 10.0))
    return torch.cat(outs)


def test_disc_prepare_oracle_matches_eager():
    bs, task, fm, w, mean, var = _prep_case(3)
    lib = load_oracle()
    lib.qo_disc_prepare.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32] + [C.c_void_p] * 5 + [C.c_float, C.c_float, C.c_void_p, C.c_void_p]
    n = [np.ascontiguousarray(b.numpy()) for b in bs]
    ptrs = (C.c_void_p * 3)(*[x.ctypes.data for x in n]); rows = (C.c_int64 * 3)(*[x.shape[0] for x in n])
    out = np.zeros((sum(x.shape[0] for x in n), 98), np.float32)
    tn, fn, wn, mn, vn = task.numpy(), fm.numpy(), np.array([0.37], np.float32), mean.numpy(), var.numpy()
    assert lib.qo_disc_prepare(ptrs, rows, 3, 98, tn.ctypes.data, fn.ctypes.data, wn.ctypes.data, mn.ctypes.data, vn.ctypes.data, 1e-4, 10.0, out.ctypes.data, None) == 0
    assert np.allclose(out, _prep_eager(bs, task, fm, w, mean, var).numpy(), rtol=1e-6, atol=1e-6)


@pytest.mark.gpu
def test_disc_prepare_hip_matches_eager():
    from quadrupedal_agility_amd.rsl_rl.algorithms import fused
    from quadrupedal_agility_amd.rsl_rl.utils.utils import TorchNormalizer
    bs, task, fm, w, mean, var = _prep_case(4)
    nz = TorchNormalizer(98, "cuda"); nz.mean.copy_(mean); nz.var.copy_(var)
    got = fused.disc_prepare([b.cuda() for b in bs], task.cuda(), fm.cuda(), w.cuda(), nz)
    assert torch.allclose(got.cpu(), _prep_eager(bs, task, fm, w, mean, var), rtol=1e-6, atol=1e-6)
    raw = fused.disc_prepare([b.cuda() for b in bs[:2]], task.cuda(), fm.cuda(), None, None)           # no task weight, no normaliser
    assert torch.allclose(raw.cpu(), torch.cat([b * fm for b in bs[:2]]), rtol=1e-6, atol=1e-6)


@pytest.mark.gpu
def test_kl_lr_rule_kernel_matches_the_reference_rule():
    """gail.py:367-379 on device scalars vs. the Python rule and the oracle's C twin, over all three branches and both clamps"""
    import ctypes as C
    from quadrupedal_agility_amd.rsl_rl.algorithms.fused import kl_lr_rule
    from tests.oracle_lib import load_oracle
    qo = load_oracle()
    cases = [(0.05, 1e-3), (0.004, 1e-3), (0.01, 1e-3), (0.0, 1e-3), (-1.0, 1e-3), (0.05, 1.2e-5), (0.001, 9e-3), (0.021, 5e-4), (0.0049, 5e-4)]
    for kl, lr0 in cases:
        want = max(1e-5, lr0 / 1.5) if kl > 0.02 else (min(1e-2, lr0 * 1.5) if 0.0 < kl < 0.005 else lr0)
        k, lr = torch.tensor(kl, device="cuda"), torch.tensor(lr0, device="cuda")
        kl_lr_rule(k, 0.01, lr)
        hk, hl = np.array([kl], np.float32), np.array([lr0], np.float32)
        assert qo.qo_kl_lr_rule(hk.ctypes.data, 0.01, 1.5, 1e-5, 1e-2, hl.ctypes.data, None) == 0
        assert abs(float(lr) - want) <= 1e-6 * want and abs(float(hl[0]) - want) <= 1e-6 * want, (kl, lr0)
        assert float(lr) == float(hl[0])


@pytest.mark.gpu
@pytest.mark.parametrize("mode,rows,cols", [(0, 24576, 29), (1, 24576, 4), (0, 1000, 29), (1, 257, 4), (0, 1, 3)])
def test_pair_loss_kernel_matches_torch_and_oracle(mode, rows, cols):
    """qa_pair_loss vs. the eager expressions of gail.py:346-358 (value and gradient) and vs. the oracle's C twin; b is a
    column slice of wider rows; a zero row exercises the norm's subgradient"""
    from quadrupedal_agility_amd.rsl_rl.algorithms.fused import pair_loss
    from tests.oracle_lib import load_oracle
    torch.manual_seed(mode * 10 + cols)
    a0 = torch.randn(rows, cols)
    wide = torch.randn(rows, cols + 7)
    if rows > 2:
        wide[2, 3:3 + cols] = a0[2]                      # identical rows: norm 0
    b0 = wide[:, 3:3 + cols]
    ref_a = a0.clone().requires_grad_(True)
    ref = ((ref_a - b0).norm(p=2, dim=1).mean() if mode == 0 else (ref_a - b0).pow(2).mean()) * 0.7
    ref.backward()
    a = a0.cuda().requires_grad_(True)
    loss = pair_loss(a, wide.cuda()[:, 3:3 + cols], mode) * 0.7
    loss.backward()
    assert abs(float(loss) - float(ref)) <= 2e-5 * max(1.0, abs(float(ref)))
    np.testing.assert_allclose(a.grad.cpu().numpy(), ref_a.grad.numpy(), rtol=2e-4, atol=1e-9)
    qo = load_oracle()
    g, out = np.zeros((rows, cols), np.float32), np.zeros(1, np.float32)
    an, wn = np.ascontiguousarray(a0.numpy()), np.ascontiguousarray(wide.numpy())
    assert qo.qo_pair_loss(an.ctypes.data, wn.ctypes.data + 12, rows, cols, cols + 7, mode, g.ctypes.data, out.ctypes.data, None, 0, None) == 0
    assert abs(out[0] * 0.7 - float(ref)) <= 2e-5 * max(1.0, abs(float(ref)))
    np.testing.assert_allclose(g * 0.7, ref_a.grad.numpy(), rtol=2e-4, atol=1e-9)


@pytest.mark.gpu
@pytest.mark.parametrize("rows", [1, 300, 3072, 24576])
def test_pair_losses_launch_equals_the_losses_one_by_one(rows):
    """qa_pair_losses (ABI 18): the regulariser's and the estimator's loss of a PPO step in one launch, the first gradient times a device scalar --
    bit-identical to qa_pair_loss twice + the multiply, over several launches in a row (the arrival counter resets itself)"""
    from quadrupedal_agility_amd.rsl_rl.algorithms import fused
    torch.manual_seed(rows)
    coef = torch.tensor(0.037, device="cuda")
    for rep in range(3):
        p, h = torch.randn(rows, 29, device="cuda"), torch.randn(rows, 29, device="cuda")
        if rows > 2:
            h[2] = p[2]                                   # a zero row: the norm's subgradient
        est, wide = torch.randn(rows, 4, device="cuda"), torch.randn(rows, 53, device="cuda")
        (l1, g1), (l2, g2) = fused.pair_losses_raw([(p, h, fused.PAIR_ROW_L2, coef), (est, wide[:, 7:11], fused.PAIR_MSE, None)])
        r1, q1 = fused.pair_loss_raw(p, h, fused.PAIR_ROW_L2)
        r2, q2 = fused.pair_loss_raw(est, wide[:, 7:11], fused.PAIR_MSE)
        assert torch.equal(l1, r1) and torch.equal(l2, r2) and torch.equal(g1, q1 * coef) and torch.equal(g2, q2), rep


@pytest.mark.gpu
def test_pair_losses_wide_rows_and_four_jobs():
    """rows wider than the LDS path takes (cols > 47: the lane-per-row walk) and the maximum of four jobs in one launch, against qa_pair_loss one by one"""
    from quadrupedal_agility_amd.rsl_rl.algorithms import fused
    torch.manual_seed(5)
    rows = 777
    coef = torch.tensor(1.7, device="cuda")
    mk = lambda c: (torch.randn(rows, c, device="cuda"), torch.randn(rows, c + 3, device="cuda")[:, 1:1 + c])
    jobs = [(*mk(64), fused.PAIR_ROW_L2, coef), (*mk(47), fused.PAIR_MSE, None), (*mk(48), fused.PAIR_ROW_L2, None), (*mk(1), fused.PAIR_MSE, coef)]
    got = fused.pair_losses_raw(jobs)
    for (a, b, mode, sc), (loss, grad) in zip(jobs, got):
        r, q = fused.pair_loss_raw(a, b, mode)
        assert torch.equal(loss, r) and torch.equal(grad, q * sc if sc is not None else q)


def test_pair_losses_twin_equals_the_single_twin():
    from quadrupedal_agility_amd._capi import QaPairJob
    lib = load_oracle()
    rng = np.random.default_rng(2)
    rows = 700
    p, h = rng.standard_normal((rows, 29)).astype(np.float32), rng.standard_normal((rows, 29)).astype(np.float32)
    est, wide = rng.standard_normal((rows, 4)).astype(np.float32), rng.standard_normal((rows, 53)).astype(np.float32)
    coef = np.array([0.25], np.float32)
    g1, g2, o1, o2 = np.zeros_like(p), np.zeros_like(est), np.zeros(1, np.float32), np.zeros(1, np.float32)
    jobs = (QaPairJob * 2)()
    jobs[0] = QaPairJob(p.ctypes.data, h.ctypes.data, rows, 29, 0, 29, coef.ctypes.data, g1.ctypes.data, o1.ctypes.data)
    jobs[1] = QaPairJob(est.ctypes.data, wide.ctypes.data + 28, rows, 4, 1, 53, None, g2.ctypes.data, o2.ctypes.data)
    lib.qo_pair_losses_scratch_bytes.restype = C.c_int64; lib.qo_pair_losses_scratch_bytes.argtypes = [C.c_void_p, C.c_int32]
    lib.qo_pair_losses.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p]
    nb = lib.qo_pair_losses_scratch_bytes(C.cast(jobs, C.c_void_p), 2)
    assert nb == 4 * (2 * 3 + 1)
    sc = np.zeros(nb, np.uint8)
    assert lib.qo_pair_losses(C.cast(jobs, C.c_void_p), 2, sc.ctypes.data, nb, None) == 0
    r1, r2, q1, q2 = np.zeros_like(p), np.zeros_like(est), np.zeros(1, np.float32), np.zeros(1, np.float32)
    lib.qo_pair_loss.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
    assert lib.qo_pair_loss(p.ctypes.data, h.ctypes.data, rows, 29, 29, 0, r1.ctypes.data, q1.ctypes.data, None, 0, None) == 0
    assert lib.qo_pair_loss(est.ctypes.data, wide.ctypes.data + 28, rows, 4, 53, 1, r2.ctypes.data, q2.ctypes.data, None, 0, None) == 0
    assert o1[0] == q1[0] and o2[0] == q2[0] and np.array_equal(g1, r1 * coef[0]) and np.array_equal(g2, r2)
    assert lib.qo_pair_losses(C.cast(jobs, C.c_void_p), 5, sc.ctypes.data, nb, None) != 0           # more than QA_PAIR_MAX_JOBS


def _stack_case(seed=0):
    """ragged tensors for qa_adam_stack_step: (numel, parts1, parts2, states) -- single elements, chunk boundaries (32 / 512), no parts, few parts, more than 16
    parts in either source (the 32-element chunk walk), one to three optimiser states"""
    shapes = [(1, 0, 0, 1), (31, 1, 0, 2), (32, 5, 3, 3), (33, 17, 0, 1), (511, 16, 20, 2), (512, 40, 3, 3), (513, 0, 0, 3), (2049, 2, 17, 1), (700, 64, 0, 2), (5000, 3, 0, 3)]
    r = np.random.default_rng(seed)
    T = []
    for n, p1, p2, ns in shapes:
        f = lambda *sh: r.standard_normal(sh).astype(np.float32)
        s1, s2 = n + r.integers(0, 5), n + r.integers(0, 5)
        T.append(dict(n=n, p1=p1, p2=p2, ns=ns, s1=int(s1), s2=int(s2), param=f(n), grad=f(n), src1=f(max(p1, 1), s1) * 0.1, src2=f(max(p2, 1), s2) * 0.1, tmp=np.zeros(n, np.float32),
                      m=[f(n) * 0.01 for _ in range(ns)], v=[np.abs(f(n)) * 0.01 for _ in range(ns)], step=[np.array([float(3 + k)], np.float32) for k in range(ns)],
                      lr=[np.array([1e-3 * (k + 1)], np.float32) for k in range(ns)], wd=[0.0, 1e-3, 1e-2][:ns], alpha2=0.37, reg=[0.0, 2e-4][n % 2]))
    return T


def _run_stack(lib, prefix, T, to_dev, ticket_ptr, stream):
    from quadrupedal_agility_amd._capi import QaAdamStackTensor
    fn = getattr(lib, prefix + "adam_stack_step")
    fn.argtypes = [C.c_void_p, C.c_int32, C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_void_p]
    recs = (QaAdamStackTensor * len(T))()
    dev = []
    for i, t in enumerate(T):
        d = {k: to_dev(t[k]) for k in ("param", "grad", "src1", "src2", "tmp")}
        d["m"], d["v"], d["step"], d["lr"] = ([to_dev(x) for x in t[k]] for k in ("m", "v", "step", "lr"))
        dev.append(d)
        rec = recs[i]
        ptr = lambda x: x.data_ptr() if hasattr(x, "data_ptr") else x.ctypes.data
        rec.param, rec.grad, rec.tmp, rec.numel, rec.num_states = ptr(d["param"]), ptr(d["grad"]), ptr(d["tmp"]), t["n"], t["ns"]
        if t["p1"]:
            rec.src1, rec.stride1, rec.parts1 = ptr(d["src1"]), t["s1"], t["p1"]
        if t["p2"]:
            rec.src2, rec.stride2, rec.parts2, rec.alpha2 = ptr(d["src2"]), t["s2"], t["p2"], t["alpha2"]
        rec.reg = t["reg"]
        for k in range(t["ns"]):
            e = rec.state[k]
            e.exp_avg, e.exp_avg_sq, e.step, e.lr, e.weight_decay = ptr(d["m"][k]), ptr(d["v"][k]), ptr(d["step"][k]), ptr(d["lr"][k]), t["wd"][k]
    for _ in range(2):          # two steps in a row: the arrival counter resets itself, the step counters advance
        assert fn(C.cast(recs, C.c_void_p), len(T), 0.9, 0.999, 1e-8, ticket_ptr, stream) == 0
    return dev


@pytest.mark.gpu
def test_adam_stack_step_ragged_tensors_match_the_twin():
    """qa_adam_stack_step over ragged sizes / part counts / state counts, HIP against the C twin (sums in double there: 2e-6 of the scale), two steps in a row"""
    from quadrupedal_agility_amd import _capi
    T = _stack_case()
    host = _run_stack(load_oracle(), "qo_", [dict(t) for t in T], lambda x: x.copy(), np.zeros(1, np.uint32).ctypes.data, None)
    ticket = torch.zeros(1, dtype=torch.int32, device="cuda")
    gpu = _run_stack(_capi.load_library(), "qa_", T, lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda(), ticket.data_ptr(), C.c_void_p(torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    assert int(ticket) == 0
    for i, (h, g, t) in enumerate(zip(host, gpu, T)):
        for key in ("param", "grad"):
            a, b = g[key].cpu().numpy(), h[key]
            assert np.allclose(a, b, rtol=2e-5, atol=2e-6 * (np.abs(b).max() + 1e-6)), (i, key, float(np.abs(a - b).max()))
        for k in range(t["ns"]):
            assert np.allclose(g["m"][k].cpu().numpy(), h["m"][k], rtol=2e-5, atol=1e-7) and np.allclose(g["v"][k].cpu().numpy(), h["v"][k], rtol=2e-5, atol=1e-9), (i, k)
            assert float(g["step"][k]) == float(h["step"][k][0]) == 3 + k + 2


@pytest.mark.gpu
def test_accumulate_scalars_is_stack_and_add():
    """qa_accumulate_scalars (ABI 18): scalars that live in different tensors onto an accumulator in one launch; anything else falls back to torch"""
    from quadrupedal_agility_amd.rsl_rl.algorithms import fused
    torch.manual_seed(0)
    out = torch.randn(8, device="cuda")
    a, b = torch.randn((), device="cuda"), torch.randn(1, device="cuda")
    parts = [out[1], out[2], out[3], out[4], a, b]
    acc, ref = torch.randn(6, device="cuda"), None
    ref = acc.clone()
    for _ in range(3):
        fused.accumulate_scalars(acc, parts)
        ref += torch.stack([x.reshape(()) for x in parts])
    assert torch.equal(acc, ref)
    acc64 = torch.zeros(6, dtype=torch.float64, device="cuda")         # not fp32: the torch expression
    fused.accumulate_scalars(acc64, [x.double() for x in parts])
    assert torch.allclose(acc64, torch.stack([x.reshape(()) for x in parts]).double())


def test_accumulate_scalars_twin():
    lib = load_oracle()
    lib.qo_accumulate_scalars.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]
    acc = np.arange(4, dtype=np.float32)
    xs = [np.array([v], np.float32) for v in (0.5, -1.0, 2.0)]
    ptrs = (C.c_void_p * 3)(*[x.ctypes.data for x in xs])
    assert lib.qo_accumulate_scalars(acc.ctypes.data, ptrs, 3, None) == 0
    assert np.array_equal(acc, np.array([0.5, 0.0, 4.0, 3.0], np.float32))
    assert lib.qo_accumulate_scalars(acc.ctypes.data, ptrs, 17, None) != 0


@pytest.mark.gpu
def test_gather_rows_kernel_is_exact():
    from quadrupedal_agility_amd.rsl_rl.algorithms.fused import gather_rows
    torch.manual_seed(3)
    n, rows = 5000, 1237
    srcs = [torch.randn(n, w, device="cuda") for w in (671, 12, 1, 1, 1, 1, 12, 12, 29)]
    srcs[2] = torch.randn(n, 5, device="cuda")[:, 1:2]                # strided source rows
    idx = torch.randint(0, n, (rows,), device="cuda")
    out = gather_rows(idx, srcs)
    for o, s_ in zip(out, srcs):
        assert torch.equal(o, s_[idx])


@pytest.mark.gpu
def test_overlapped_discriminator_and_ppo_updates_match_sequential(monkeypatch):
    """AMP config: discriminator steps replayed on a second stream beside the PPO steps vs. one loop after the other --
    no data flows between the two loops, so weights of all three networks agree to GEMM rounding after 4 iterations"""
    from tests.test_gpu_train import _make
    from quadrupedal_agility_amd.legged_gym.envs import task_registry
    res = []
    for overlap in (True, False):
        torch.manual_seed(0)
        env, args, tcfg = _make(256, True)
        runner, _ = task_registry.make_alg_runner(env, name="go2_locomotion", args=args, train_cfg=tcfg, log_root=None)
        runner.alg.overlap_updates = overlap
        runner.learn(4, init_at_random_ep_len=True)
        torch.cuda.synchronize()
        a = runner.alg
        assert a._disc_graph and a._ac_graph
        res.append([{k: v.clone() for k, v in m.state_dict().items()} for m in (a.actor_critic, a.estimator, a.disc)] + [float(a.lr_ac)])
    for wa, wb in zip(res[0][:3], res[1][:3]):
        for k in wa:
            assert torch.allclose(wa[k].float(), wb[k].float(), atol=5e-4, rtol=5e-3), k
    assert res[0][3] == res[1][3]


@pytest.mark.gpu
@pytest.mark.parametrize("n", [1, 300, 4096])
def test_rollout_post_amp_matches_predict_disc_reward(n):
    """qa_rollout_post_amp vs. Discriminator.predict_disc_reward's expressions (discriminator.py:88-118) + the time-out
    bootstrap and episode sums of the runner, and vs. the oracle's C twin"""
    import ctypes as C
    import torch.nn.functional as F
    from quadrupedal_agility_amd import _capi
    from tests.oracle_lib import load_oracle
    torch.manual_seed(n)
    dim_c, num_obs, stride = 5, 671, 680
    obs = torch.randn(n, stride)
    rew, values, d, eps, logits = torch.rand(n), torch.randn(n), torch.randn(n) * 1.5 + 0.5, torch.randn(n), torch.randn(n, dim_c) * 2
    reset = (torch.rand(n) < 0.2).long(); tout = ((torch.rand(n) < 0.5) & (reset > 0)).to(torch.uint8)
    cur = torch.randn(6, n)
    ci, cus, css, ct, dt, gamma = 0.35, 0.1, 0.25, 0.3, 0.02, 0.99
    # reference expressions
    label_eps = obs[:, num_obs - dim_c - 1]
    label_c = F.one_hot(torch.argmax(obs[:, num_obs - dim_c:num_obs], dim=-1), num_classes=dim_c).float()
    c = torch.clamp(torch.softmax(logits, -1), 1e-20, torch.inf)
    r_i = torch.clamp(1 - 0.25 * torch.square(d - 1), min=0) * dt
    r_us = -torch.abs(eps - label_eps) * dt
    r_ss = -F.cross_entropy(c, label_c, reduction="none") * dt
    total = ci * r_i + cus * r_us + css * r_ss + ct * rew
    want_rew = total + gamma * values * tout.float()
    want_fin = cur + torch.stack([total, r_i, r_us, r_ss, rew, torch.ones(n)])
    want_cur = want_fin * (reset == 0)
    for side in ("hip", "oracle"):
        if side == "hip":
            lib, pre, dev = _capi.load_library(), "qa_", "cuda"
        else:
            lib, pre, dev = load_oracle(), "qo_", "cpu"
        t = lambda x: x.to(dev).contiguous()
        a = [t(x) for x in (rew, reset, tout, values, d, eps, logits, obs)]
        st_r, st_d = torch.zeros(n, device=dev), torch.zeros(n, dtype=torch.uint8, device=dev)
        cur_d, fin, mask = t(cur).clone(), torch.zeros(6, n, device=dev), torch.zeros(n, dtype=torch.uint8, device=dev)
        P = lambda x: C.c_void_p(x.data_ptr())
        rc = getattr(lib, pre + "rollout_post_amp")(P(a[0]), P(a[1]), P(a[2]), P(a[3]), P(a[4]), P(a[5]), P(a[6]), dim_c, P(a[7]), stride, num_obs,
                                                     ci, cus, css, ct, dt, gamma, n, P(st_r), P(st_d), P(cur_d), P(fin), P(mask), None)
        assert rc == 0
        if dev == "cuda":
            torch.cuda.synchronize()
        np.testing.assert_allclose(st_r.cpu().numpy(), want_rew.numpy(), rtol=2e-5, atol=2e-6, err_msg=side)
        assert torch.equal(st_d.cpu(), (reset > 0).to(torch.uint8)) and torch.equal(mask.cpu(), (reset > 0).to(torch.uint8))
        np.testing.assert_allclose(fin.cpu().numpy(), want_fin.numpy(), rtol=2e-5, atol=2e-6, err_msg=side)
        np.testing.assert_allclose(cur_d.cpu().numpy(), want_cur.numpy(), rtol=2e-5, atol=2e-6, err_msg=side)


@pytest.mark.gpu
def test_discriminator_chain_matches_module():
    from quadrupedal_agility_amd.rsl_rl.algorithms.fused import PolicyChain
    from tests.test_gpu_train import _make
    from quadrupedal_agility_amd.legged_gym.envs import task_registry
    env, args, tcfg = _make(64, True)
    runner, _ = task_registry.make_alg_runner(env, name="go2_locomotion", args=args, train_cfg=tcfg, log_root=None)
    disc = runner.alg.disc
    with torch.no_grad():
        for p in disc.parameters():
            if p.dim() == 1:
                p.uniform_(-0.3, 0.3)
    chain = PolicyChain.describe_discriminator(disc)
    assert chain is not None
    x = torch.randn(777, disc.input_dim, device="cuda")
    with torch.inference_mode():
        chain.pack()
        d, eps, logits = chain.forward(x)
        rd, reps, rc = disc(x)
    torch.cuda.synchronize()
    np.testing.assert_allclose(d.cpu().numpy(), rd.cpu().numpy(), rtol=2e-4, atol=5e-5)
    np.testing.assert_allclose(eps.cpu().numpy(), reps.cpu().numpy(), rtol=2e-4, atol=5e-5)
    np.testing.assert_allclose(torch.softmax(logits, -1).cpu().numpy(), rc.cpu().numpy(), rtol=2e-4, atol=5e-6)


@pytest.mark.gpu
def test_clip_adam_pointer_table_path_with_many_tensors():
    """more than QA_ADAM_MAX_INLINE (64) gradient tensors: the device pointer table path (fewer: the pointers ride in the kernel
    arguments) -- both against torch.optim.Adam + clip_grad_norm_"""
    from quadrupedal_agility_amd.rsl_rl.algorithms.fused import ClipAdam
    torch.manual_seed(0)
    mk = lambda: torch.nn.ModuleList([torch.nn.Linear(7 + i % 5, 3 + i % 4) for i in range(40)]).cuda()      # 80 tensors
    a, b = mk(), mk()
    b.load_state_dict(a.state_dict())
    kw = dict(lr=torch.tensor(1e-3, device="cuda"), fused=True, capturable=True)
    oa, ob = torch.optim.Adam(a.parameters(), **kw), torch.optim.Adam(b.parameters(), **kw)
    stepper = ClipAdam(ob, 0.5)
    for k in range(5):
        g = torch.Generator(device="cuda").manual_seed(k)
        for p, q in zip(a.parameters(), b.parameters()):
            p.grad = torch.randn(p.shape, device="cuda", generator=g) * (2.0 if k % 2 else 0.05)
            q.grad = p.grad.clone()
        torch.nn.utils.clip_grad_norm_(a.parameters(), 0.5)
        oa.step()
        stepper.step()
    assert stepper._tab is not None and stepper._tab["n"] == 80 and stepper._tab["grad_ptrs"] is not None
    for p, q in zip(a.parameters(), b.parameters()):
        assert torch.allclose(p, q, rtol=2e-5, atol=2e-7)


@pytest.mark.gpu
def test_clip_adam_refuses_to_record_the_pointer_table_path():
    """with more than 64 gradient tensors the step uploads a shared pinned host table; recorded, that copy would read the host
    memory at replay time (another slot's or freed addresses).  The step refuses to be captured -- the callers then stay eager -- and the
    same object keeps stepping correctly outside a capture."""
    from quadrupedal_agility_amd.rsl_rl.algorithms.fused import ClipAdam
    torch.manual_seed(1)
    mk = lambda: torch.nn.ModuleList([torch.nn.Linear(5, 3) for _ in range(35)]).cuda()      # 70 tensors
    a, b = mk(), mk()
    b.load_state_dict(a.state_dict())
    kw = dict(lr=torch.tensor(1e-3, device="cuda"), fused=True, capturable=True)
    oa, ob = torch.optim.Adam(a.parameters(), **kw), torch.optim.Adam(b.parameters(), **kw)
    stepper = ClipAdam(ob, 1.0)

    def grads(k):
        g = torch.Generator(device="cuda").manual_seed(k)
        for p, q in zip(a.parameters(), b.parameters()):
            p.grad = torch.randn(p.shape, device="cuda", generator=g); q.grad = p.grad.clone()
    for k in (0, 1):         # the first step creates the optimiser state (torch path), the second builds the kernel's pointer tables
        grads(k); torch.nn.utils.clip_grad_norm_(a.parameters(), 1.0); oa.step(); stepper.step()
    grads(2)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    refused = None
    with torch.cuda.graph(graph):              # the refusal is caught INSIDE the capture: an exception leaving the context while capturing is replaced by HIP's
        try:
            stepper.step()
            refused = False
        except RuntimeError as e:
            refused = "stays eager" in str(e)
    torch.cuda.synchronize()
    assert refused is True
    torch.nn.utils.clip_grad_norm_(a.parameters(), 1.0); oa.step(); stepper.step()
    for p, q in zip(a.parameters(), b.parameters()):
        assert torch.allclose(p, q, rtol=2e-5, atol=2e-7)


@pytest.mark.gpu
@pytest.mark.parametrize("rows,o,k", [(24576, 1, 128), (24576, 12, 128), (49152, 18, 128), (3072, 3, 128), (5000, 29, 64), (2049, 32, 300)])
def test_narrow_wgrad_matches_torch_and_oracle(rows, o, k):
    """qa_narrow_wgrad: dW = gy^T x, db = sum gy for layers with <= 32 outputs (actor / critic heads), against torch in fp64 and the
    C twin; and through autograd: fused.narrow_linear == nn.Linear on outputs and all three gradients"""
    import ctypes as C
    from quadrupedal_agility_amd import _capi
    from quadrupedal_agility_amd.rsl_rl.algorithms import fused
    from tests.oracle_lib import load_oracle
    torch.manual_seed(rows + o)
    gy, x = torch.randn(rows, o), torch.randn(rows, k)
    ref_w, ref_b = (gy.double().t() @ x.double()), gy.double().sum(0)
    lib = _capi.load_library()
    gw, gb = torch.empty(o, k, device="cuda"), torch.empty(o, device="cuda")
    n = int(lib.qa_narrow_wgrad_scratch_bytes(rows, o, k)); scratch = torch.empty(n, dtype=torch.uint8, device="cuda")
    gyc, xc = gy.cuda(), x.cuda()
    assert lib.qa_narrow_wgrad(gyc.data_ptr(), xc.data_ptr(), rows, o, k, gw.data_ptr(), gb.data_ptr(), scratch.data_ptr(), n, None) == 0
    torch.cuda.synchronize()
    tol = 2e-5 * rows ** 0.5
    assert (gw.cpu().double() - ref_w).abs().max() < tol and (gb.cpu().double() - ref_b).abs().max() < tol
    orc = load_oracle()
    ow, ob = torch.empty(o, k), torch.empty(o)
    assert orc.qo_narrow_wgrad(gy.data_ptr(), x.data_ptr(), rows, o, k, ow.data_ptr(), ob.data_ptr(), None, 0, None) == 0
    assert (ow.double() - ref_w).abs().max() < 1e-4 and (gw.cpu() - ow).abs().max() < tol
    # autograd path
    lin = torch.nn.Linear(k, o).cuda()
    ref = torch.nn.Linear(k, o).cuda(); ref.load_state_dict(lin.state_dict())
    xa, xb = xc.clone().requires_grad_(True), xc.clone().requires_grad_(True)
    ya = fused.narrow_linear(lin, xa); yb = ref(xb)
    assert type(ya.grad_fn).__name__.startswith("_NarrowLinear") and torch.allclose(ya, yb, atol=1e-5)
    ya.backward(gyc); yb.backward(gyc)
    assert torch.allclose(xa.grad, xb.grad, atol=1e-5)
    assert (lin.weight.grad - ref.weight.grad).abs().max() < tol and (lin.bias.grad - ref.bias.grad).abs().max() < tol


@pytest.mark.gpu
def test_discriminator_step_backward_through_our_reductions_matches_autograd():
    """The discriminator's training forward (`forward_with_input_gradient`) with every batch reduction of its backward in our kernels
    (linear_relu = qa_elu_backward_bias with alpha 0, narrow heads at 921 rows, mask_times_row, proxy leaves for the penalty chain)
    against the same function through plain torch modules + autograd: heads, input gradient, and every parameter gradient."""
    from quadrupedal_agility_amd.rsl_rl.algorithms import fused
    from quadrupedal_agility_amd.rsl_rl.algorithms.discriminator import Discriminator

    class Env:
        task_obs_weight_decay, task_obs_weight = False, 1.0
    torch.manual_seed(3)
    d = Discriminator(Env(), 98, 49, 5, 0.02, "MSELoss", None, 1.0, 0.01, 0.2, 0.2, 2, 2, 0.0, [512, 256], "cuda").cuda()
    for m in d.trunk:
        if isinstance(m, torch.nn.Linear):
            torch.nn.init.normal_(m.bias, std=0.3)                  # zero-initialised biases would hide a wrong bias gradient's effect on the masks
    x = torch.randn(921, 98, device="cuda")
    rows = slice(614, None)
    gd, ge, gc, gg = torch.randn(921, 1, device="cuda"), torch.randn(921, 1, device="cuda"), torch.randn(921, 5, device="cuda"), torch.randn(307, 98, device="cuda")
    res = []
    for enabled in (True, False):
        keep = fused.ENABLED
        fused.ENABLED = enabled
        try:
            d.zero_grad(set_to_none=True)
            proxies = [] if enabled else None
            (dl, eps, c), g = d.forward_with_input_gradient(x, rows, clamp=False, proxies=proxies)
            if enabled:
                assert type(dl.grad_fn).__name__.startswith("_NarrowLinear")
            torch.autograd.backward([dl, eps, c, g], [gd, ge, gc, gg])
            if proxies:
                for w, q in proxies:
                    w.grad.add_(q.grad)
            res.append(([t.detach().clone() for t in (dl, eps, c, g)], {n: p.grad.detach().clone() for n, p in d.named_parameters()}))
        finally:
            fused.ENABLED = keep
    (oa, ga), (ob, gb) = res
    for a, b in zip(oa, ob):
        assert torch.allclose(a, b, atol=1e-5, rtol=1e-5)
    for n in ga:
        scale = float(gb[n].abs().max())
        assert float((ga[n] - gb[n]).abs().max()) <= 2e-5 * max(scale, 1.0) * 921 ** 0.5, n


@pytest.mark.gpu
def test_single_launch_adam_step_equals_the_two_launch_step_bit_for_bit():
    """without clipping, a scratch with the spare counter slot takes the one-launch path (the last workgroup writes the step counters);
    the same call with the short scratch takes finalize + update: identical parameters, moments, counters, over several steps"""
    from quadrupedal_agility_amd import _capi
    lib = _capi.load_library()
    g = torch.Generator(device="cuda").manual_seed(0)
    shapes = [(1024, 98), (1024,), (512, 1024), (512,), (1, 512), (1,)]
    ct, cs, cl = [], [], []
    for t, s in enumerate(shapes):
        n = int(np.prod(s))
        for s0 in range(0, n, 2048):
            ct.append(t); cs.append(s0); cl.append(min(2048, n - s0))
    i32 = lambda x: torch.tensor(x, dtype=torch.int32, device="cuda")
    ctd, csd, cld = i32(ct), i32(cs), i32(cl)
    wd = torch.full((len(shapes),), 1e-3, device="cuda"); lr = torch.tensor([1e-3], device="cuda")
    runs = []
    for extra in (0, 1):
        torch.manual_seed(1)
        p = [torch.randn(s, device="cuda") for s in shapes]
        m = [torch.zeros_like(x) for x in p]; v = [torch.zeros_like(x) for x in p]; st = [torch.zeros(1, device="cuda") for _ in p]
        scratch = torch.zeros(4 + len(ct) + extra, device="cuda")
        tab = lambda xs: torch.tensor([x.data_ptr() for x in xs], dtype=torch.int64, device="cuda")
        pt, mt, vt, stt = tab(p), tab(m), tab(v), tab(st)
        for k in range(5):
            gg = [torch.randn(s, device="cuda", generator=torch.Generator(device="cuda").manual_seed(10 + k)) for s in shapes]
            gt = tab(gg)
            pp = lambda t: C.c_void_p(t.data_ptr())
            rc = lib.qa_clip_adam_step(pp(pt), pp(gt), pp(mt), pp(vt), pp(stt), len(shapes), pp(ctd), pp(csd), pp(cld), len(ct), pp(wd), pp(lr),
                                       0.9, 0.999, 1e-8, 0.0, pp(scratch), scratch.numel(), None)
            assert rc == 0
            torch.cuda.synchronize()
        runs.append((p, m, v, st, scratch[:4].clone()))
    for a, b in zip(runs[0][:4], runs[1][:4]):
        for x, y in zip(a, b):
            assert torch.equal(x, y)
    assert torch.equal(runs[0][4], runs[1][4]) and float(runs[1][3][0]) == 5.0
