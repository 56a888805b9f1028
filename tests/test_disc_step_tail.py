"""qa_disc_step_tail: the logged sums of squares of a discriminator step (gradient penalty, logit regulariser, weight decay;
bbc/rsl_rl/algorithms/gail.py:486-504) assembled with the head statistics into the 11 values `update_ss_info_gail` returns (:520-533),
with the recorded step's accumulator and step counter.  CPU: the C twin against the eager expressions.  GPU: kernel vs twin."""
import ctypes as C

import numpy as np
import pytest
import torch

from tests.oracle_lib import load_oracle


def _case(seed, rows=1228, cols=98, shapes=((1024, 98), (512, 1024), (1, 512))):
    g = torch.Generator().manual_seed(seed)
    hs = torch.rand(16, generator=g)
    grad = torch.randn(rows, cols, generator=g) * 0.1
    ws = [torch.randn(*s, generator=g) * 0.05 for s in shapes]
    return hs, grad, ws


def _eager(hs, grad, ws):
    gp = grad.double().square().sum() / grad.shape[0]
    sq = torch.stack([w.double().square().sum() for w in ws])
    return torch.stack([hs[1].double(), hs[2].double(), hs[3].double(), hs[4].double(), gp, sq[-1], sq.sum(), hs[5].double(), hs[6].double(), hs[7].double(), hs[8].double()])


def _twin(hs, grad, ws, acc=None, step=None):
    lib = load_oracle()
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    hn, gn, wn = hs.numpy(), np.ascontiguousarray(grad.numpy()), [np.ascontiguousarray(w.numpy()) for w in ws]
    wp = (C.c_void_p * len(wn))(*[w.ctypes.data for w in wn]); cnt = (C.c_int64 * len(wn))(*[w.size for w in wn])
    out = np.zeros(11, np.float32); sc = np.zeros(16, np.uint8)
    assert lib.qo_disc_step_tail(p(hn), p(gn), gn.shape[0], gn.shape[1], wp, cnt, len(wn), p(out), p(acc) if acc is not None else None,
                                 p(step) if step is not None else None, p(sc), 16, None) == 0
    return out


@pytest.mark.parametrize("seed", [0, 1])
def test_twin_matches_the_eager_expressions(seed):
    hs, grad, ws = _case(seed)
    acc = np.full(11, 2.0, np.float32); step = np.array([5], np.int64)
    out = _twin(hs, grad, ws, acc, step)
    assert np.allclose(out, _eager(hs, grad, ws).numpy(), rtol=1e-6, atol=1e-7)
    assert np.allclose(acc, 2.0 + out) and step[0] == 6
    out2 = _twin(hs, grad, ws)                         # both optional
    assert np.array_equal(out, out2)


@pytest.mark.gpu
@pytest.mark.parametrize("rows,cols,shapes", [(1228, 98, ((1024, 98), (512, 1024), (1, 512))), (7, 3, ((5, 3),)), (1229, 97, ((3, 333), (1, 7), (2, 2), (1, 1)))])
def test_kernel_matches_twin_and_is_reproducible(rows, cols, shapes):
    from quadrupedal_agility_amd.rsl_rl.algorithms import fused
    hs, grad, ws = _case(2, rows, cols, shapes)
    want = _twin(hs, grad, ws)
    acc = torch.full((11,), 2.0, device="cuda"); step = torch.tensor([5], dtype=torch.int64, device="cuda")
    dws = [w.cuda() for w in ws]
    o1 = fused.disc_step_tail(hs.cuda(), grad.cuda(), dws, acc=acc, step=step)
    o2 = fused.disc_step_tail(hs.cuda(), grad.cuda(), dws)                       # the arrival counter was left at zero
    o3 = fused.disc_step_tail(hs.cuda(), grad.cuda()[:, :cols].contiguous(), dws, acc=acc, step=step)
    torch.cuda.synchronize()
    assert torch.equal(o1, o2) and torch.equal(o1, o3)
    assert np.allclose(o1.cpu().numpy(), want, rtol=2e-6, atol=1e-7)
    assert torch.allclose(acc.cpu(), 2.0 + 2 * o1.cpu()) and int(step.item()) == 7
