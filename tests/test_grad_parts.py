"""r5 (ABI 14): gradients left IN PARTS and finished by the optimiser's first pass (qa_clip_adam_step_reduce) or by one launch (qa_grad_reduce).

What they replace: the fixed-order finish launches behind every Linear layer's backward in the PPO minibatch step (gail.py:328-413 through
autograd: `sum(0)` of the bias gradient, the split-K reduction of the weight gradient).  CPU: the oracle's twins against numpy.  -m gpu: the
kernels against the twins, and a whole Linear+ELU stack stepped with and without deferral."""
import ctypes as C

import numpy as np
import pytest

from tests.oracle_lib import load_oracle

torch = pytest.importorskip("torch")


def _tab(arrs):
    return (C.c_void_p * len(arrs))(*[x.ctypes.data for x in arrs])


def _cases(seed=0):
    rng = np.random.default_rng(seed)
    # (numel, parts): split-K slabs of weights, many column-sum rows of biases, a ragged tail, a single part
    shapes = [(512 * 672, 8), (256, 384), (12, 384), (2048, 16), (2049, 3), (33, 17), (5, 1)]
    src = [rng.normal(0, 1, (p, n + 3)).astype(np.float32) for n, p in shapes]        # stride = numel + 3: parts are not packed
    return shapes, src


def test_grad_reduce_twin_matches_numpy():
    lib = load_oracle()
    lib.qo_grad_reduce.argtypes = [C.c_void_p] * 5 + [C.c_int32, C.c_void_p]
    shapes, src = _cases()
    dst = [np.zeros(n, np.float32) for n, _ in shapes]
    stride = np.array([n + 3 for n, _ in shapes], np.int64); parts = np.array([p for _, p in shapes], np.int32); numel = np.array([n for n, _ in shapes], np.int32)
    assert lib.qo_grad_reduce(_tab(dst), _tab(src), stride.ctypes.data, parts.ctypes.data, numel.ctypes.data, len(shapes), None) == 0
    for d, s, (n, p) in zip(dst, src, shapes):
        assert np.allclose(d, s[:, :n].astype(np.float64).sum(0), rtol=1e-6, atol=1e-6)
    bad = np.array([0] * len(shapes), np.int32)
    assert lib.qo_grad_reduce(_tab(dst), _tab(src), stride.ctypes.data, bad.ctypes.data, numel.ctypes.data, len(shapes), None) != 0


@pytest.mark.gpu
def test_grad_reduce_kernel_matches_twin_and_is_reproducible():
    from quadrupedal_agility_amd import _capi
    lib = _capi.load_library()
    shapes, src = _cases(1)
    dsrc = [torch.from_numpy(s).cuda() for s in src]
    outs = []
    for _ in range(2):
        ddst = [torch.full((n,), float("nan"), device="cuda") for n, _ in shapes]
        k = len(shapes)
        rc = lib.qa_grad_reduce((C.c_void_p * k)(*[d.data_ptr() for d in ddst]), (C.c_void_p * k)(*[s.data_ptr() for s in dsrc]),
                                (C.c_int64 * k)(*[n + 3 for n, _ in shapes]), (C.c_int32 * k)(*[p for _, p in shapes]), (C.c_int32 * k)(*[n for n, _ in shapes]),
                                k, C.c_void_p(torch.cuda.current_stream().cuda_stream))
        assert rc == 0, lib.qa_last_error()
        torch.cuda.synchronize()
        outs.append([d.cpu().numpy() for d in ddst])
    for a, b, s, (n, p) in zip(outs[0], outs[1], src, shapes):
        assert np.array_equal(a, b)                                              # fixed order: bit-reproducible
        ref = s[:, :n].astype(np.float64).sum(0)
        assert np.allclose(a, ref, rtol=2e-6, atol=2e-6 * np.sqrt(p)), (n, p)


@pytest.mark.gpu
def test_deferred_finishes_step_like_the_separate_launches():
    """a Linear+ELU stack of the PPO step's shapes: backward + clip + Adam with the finishes deferred into qa_clip_adam_step_reduce against the
    same step with qa_colsum_finish / qa_slab_sum launched by the backward functions -- and against plain PyTorch"""
    from quadrupedal_agility_amd.rsl_rl.algorithms import fused
    torch.manual_seed(0)
    rows = 8192
    mk = lambda: torch.nn.Sequential(torch.nn.Linear(96, 512), torch.nn.ELU(), torch.nn.Linear(512, 256), torch.nn.ELU(), torch.nn.Linear(256, 128), torch.nn.ELU()).cuda()
    nets = [mk() for _ in range(3)]
    for n in nets[1:]:
        n.load_state_dict(nets[0].state_dict())
    kw = dict(lr=torch.tensor(1e-3, device="cuda"), fused=True, capturable=True)
    opts = [torch.optim.Adam(n.parameters(), **kw) for n in nets]
    steppers = [fused.ClipAdam(o, 1.0) for o in opts[:2]]
    for k in range(4):
        g = torch.Generator(device="cuda").manual_seed(k)
        x = torch.randn(rows, 96, device="cuda", generator=g)
        gy = torch.randn(rows, 128, device="cuda", generator=g) * (0.05 if k % 2 else 1e-3)          # clipped and unclipped steps
        for o in opts:
            o.zero_grad(set_to_none=True)
        # deferred
        y = fused.mlp_forward(nets[0], x)
        with fused.deferred_grad_finishes():
            y.backward(gy)
        if k > 0:
            assert fused.pending_grads() == 6, "three weights + three biases should have been left in parts"
        steppers[0].step()
        assert fused.pending_grads() == 0
        # separate finish launches
        fused.mlp_forward(nets[1], x).backward(gy)
        assert fused.pending_grads() == 0
        steppers[1].step()
        # PyTorch
        nets[2](x).backward(gy)
        torch.nn.utils.clip_grad_norm_(nets[2].parameters(), 1.0)
        opts[2].step()
        if k > 0:      # the finished gradient was written where autograd put its tensor
            for p, q in zip(nets[0].parameters(), nets[1].parameters()):
                assert torch.allclose(p.grad, q.grad, rtol=1e-6, atol=1e-9)
    for p, q, r in zip(nets[0].parameters(), nets[1].parameters(), nets[2].parameters()):
        d = (p.detach() - q.detach()).abs().max().item()
        assert d <= 2e-7 + 1e-5 * q.detach().abs().max().item(), d          # same parts, same order; only the norm's chunking differs
        # against PyTorch: Adam's update is ill-conditioned where the clipped gradient cancels to ~eps (tests/test_fused_learner.py::_close_params)
        dr = (p.detach() - r.detach()).abs().reshape(-1)
        tol = 2e-6 + 1e-3 * r.detach().abs().reshape(-1)
        assert float((dr > tol).double().mean()) < 2e-3 and float(dr.max()) < 2 * 1e-3 * 4


@pytest.mark.gpu
def test_flush_finishes_gradients_for_readers_other_than_the_optimiser():
    from quadrupedal_agility_amd.rsl_rl.algorithms import fused
    torch.manual_seed(1)
    net = torch.nn.Sequential(torch.nn.Linear(64, 512), torch.nn.ELU(), torch.nn.Linear(512, 64), torch.nn.ELU()).cuda()
    ref = torch.nn.Sequential(torch.nn.Linear(64, 512), torch.nn.ELU(), torch.nn.Linear(512, 64), torch.nn.ELU()).cuda()
    ref.load_state_dict(net.state_dict())
    x = torch.randn(8192, 64, device="cuda"); gy = torch.randn(8192, 64, device="cuda")
    with fused.deferred_grad_finishes():
        fused.mlp_forward(net, x).backward(gy)
    assert fused.pending_grads() == 4
    fused.flush_pending_grads()
    assert fused.pending_grads() == 0
    ref(x).backward(gy)
    for p, q in zip(net.parameters(), ref.parameters()):
        assert torch.allclose(p.grad, q.grad, rtol=2e-4, atol=2e-4 * q.grad.abs().max().item())


def _twice_nets():
    torch.manual_seed(2)
    mk = lambda: torch.nn.Sequential(torch.nn.Linear(64, 512), torch.nn.ELU(), torch.nn.Linear(512, 64), torch.nn.ELU()).cuda()
    net, ref = mk(), mk()
    ref.load_state_dict(net.state_dict())
    return net, ref


def _close_grads(net, ref):
    for p, q in zip(net.parameters(), ref.parameters()):
        assert p.grad is not None and torch.isfinite(p.grad).all()
        assert torch.allclose(p.grad, q.grad, rtol=2e-4, atol=2e-4 * q.grad.abs().max().item()), (p.shape, (p.grad - q.grad).abs().max().item())


@pytest.mark.gpu
@pytest.mark.parametrize("side_stream", [False, True])
def test_a_module_used_twice_in_one_backward_keeps_both_gradients(side_stream):
    """the privileged encoder runs twice in a PPO minibatch step (`train_with_estimated_latent`: once for the regulariser, once
    inside the actor; gail.py:341-349 in the reference, where both terms reach its parameters).  The first backward node leaves its
    gradient in parts; the second arrival must find a FINISHED first gradient to be added to."""
    from quadrupedal_agility_amd.rsl_rl.algorithms import fused
    net, ref = _twice_nets()
    x1 = torch.randn(8192, 64, device="cuda"); x2 = torch.randn(8192, 64, device="cuda")
    g1 = torch.randn(8192, 64, device="cuda"); g2 = torch.randn(8192, 64, device="cuda") * 0.3
    cur = torch.cuda.current_stream()
    side = torch.cuda.Stream() if side_stream else cur
    side.wait_stream(cur)
    with torch.cuda.stream(side):
        y1 = fused.mlp_forward(net, x1)          # like the step's small-nets branch: forward (and therefore backward) on another stream
    y2 = fused.mlp_forward(net, x2)
    cur.wait_stream(side)
    with fused.deferred_grad_finishes():
        torch.autograd.backward([y1, y2], [g1, g2])
    assert fused.pending_grads() == 0, "a parameter with two gradients must not stay in parts"
    torch.autograd.backward([ref(x1), ref(x2)], [g1, g2])
    torch.cuda.synchronize()
    _close_grads(net, ref)
    # and the optimiser step that follows sees finished gradients (nothing left for qa_clip_adam_step_reduce to overwrite them with)
    fused.flush_pending_grads()
    _close_grads(net, ref)


@pytest.mark.gpu
def test_gradient_accumulation_over_backward_calls_and_preexisting_grads():
    from quadrupedal_agility_amd.rsl_rl.algorithms import fused
    net, ref = _twice_nets()
    xs = [torch.randn(8192, 64, device="cuda") for _ in range(3)]
    gs = [torch.randn(8192, 64, device="cuda") for _ in range(3)]
    for x, g in zip(xs, gs):          # three backward calls, no zero_grad between them: the first may defer, the later ones add to `.grad`
        with fused.deferred_grad_finishes():
            fused.mlp_forward(net, x).backward(g)
        ref(x).backward(g)
    fused.flush_pending_grads()
    torch.cuda.synchronize()
    _close_grads(net, ref)
    # a gradient that exists BEFORE the step (never deferred: autograd adds to it in place)
    with fused.deferred_grad_finishes():
        fused.mlp_forward(net, xs[0]).backward(gs[1])
    assert fused.pending_grads() == 0
    ref(xs[0]).backward(gs[1])
    _close_grads(net, ref)
