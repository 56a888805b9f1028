"""The image stem of the vision student's depth encoder (csrc/qa_conv.hip, the window GEMMs of csrc/qa_gemm.hip).

CPU: the oracle twins (oracle/qa_oracle.c) are pinned against torch's conv2d / max_pool2d / elu and their autograd -- i.e. against what
the reference's `DepthOnlyFCBackbone58x87.image_compression` (tsc/rsl_rl/modules/depth_backbone.py:63-75) computes.
GPU: the HIP kernels against the oracle through the C ABI, and the whole encoder (`depth_stem.image_compression` inside the module) against
the same module on aten / MIOpen, forward and parameter gradients."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from tests.oracle_lib import load_oracle

ALPHA = 1.0


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def _nhwc(t):
    return np.ascontiguousarray(t.detach().permute(0, 2, 3, 1).numpy())


def _stem_torch(img, w1, b1):
    pre = F.conv2d(img[:, None], w1, b1)
    pooled, idx = F.max_pool2d(pre, 2, 2, return_indices=True)
    return F.elu(pooled, ALPHA), pooled, idx, pre.shape[-1]


def _rand_stem(n, ih, iw, seed):
    g = torch.Generator().manual_seed(seed)
    img = torch.rand(n, ih, iw, generator=g) * 2 - 0.5
    w1 = (torch.rand(32, 1, 5, 5, generator=g) - 0.5) * 0.4
    b1 = (torch.rand(32, generator=g) - 0.5) * 0.2
    return img, w1, b1


@pytest.mark.parametrize("n,ih,iw", [(2, 58, 87), (1, 13, 18), (3, 24, 31)])
def test_oracle_stem_forward_and_backward_match_torch(n, ih, iw):
    lib = load_oracle()
    img, w1, b1 = _rand_stem(n, ih, iw, 1)
    w1.requires_grad_(True); b1.requires_grad_(True)
    y_t, pooled, idx, cw = _stem_torch(img, w1, b1)
    ph, pw = y_t.shape[2], y_t.shape[3]
    y = np.zeros((n, ph, pw, 32), np.float32); am = np.zeros((n, ph, pw, 32), np.uint8)
    imgs = np.ascontiguousarray(img.numpy()); w = np.ascontiguousarray(w1.detach().numpy()); b = np.ascontiguousarray(b1.detach().numpy())
    assert lib.qo_depth_stem_forward(_ptr(imgs), _ptr(w), _ptr(b), _ptr(y), _ptr(am), n, ih, iw, ALPHA, None) == 0
    assert np.allclose(y, _nhwc(y_t), atol=2e-6)
    # argmax: torch returns the flat index into the conv output plane
    iy, ix = (idx // cw).permute(0, 2, 3, 1).numpy(), (idx % cw).permute(0, 2, 3, 1).numpy()
    py, px = np.arange(ph)[None, :, None, None], np.arange(pw)[None, None, :, None]
    want = 2 * (iy - 2 * py) + (ix - 2 * px)
    agree = (want == am)
    assert agree.mean() > 0.9999         # fp32-vs-double near-ties may pick the other cell
    # backward through pool + ELU: gradient at the pooled pre-activation, then weight / bias gradient
    gout = torch.randn(y_t.shape, generator=torch.Generator().manual_seed(2))
    gw_t, gb_t = torch.autograd.grad(y_t, (w1, b1), gout)
    gpre = _nhwc(gout * torch.where(y_t > 0, torch.ones_like(y_t), y_t + ALPHA))
    # use torch's own argmax so that the comparison does not depend on the near-ties above
    am_t = np.ascontiguousarray(want.astype(np.uint8))
    gwb = np.zeros(832, np.float32); sc = np.zeros(16, np.uint8)
    assert lib.qo_depth_stem_backward(_ptr(imgs), _ptr(am_t), _ptr(gpre), _ptr(gwb), n, ih, iw, _ptr(sc), 16, None) == 0
    assert np.allclose(gwb[:800].reshape(32, 1, 5, 5), gw_t.numpy(), rtol=1e-4, atol=1e-4)
    assert np.allclose(gwb[800:], gb_t.numpy(), rtol=1e-4, atol=1e-4)


def _rand_conv(n, ih, iw, cin, cout, kh, kw, seed):
    g = torch.Generator().manual_seed(seed)
    x = F.elu(torch.randn(n, cin, ih, iw, generator=g))
    w = torch.randn(cout, cin, kh, kw, generator=g) * 0.1
    b = torch.randn(cout, generator=g) * 0.1
    return x, w, b


@pytest.mark.parametrize("n,ih,iw,cin,cout,kh,kw", [(2, 27, 41, 32, 64, 3, 3), (1, 20, 19, 16, 8, 2, 3)])
def test_oracle_window_gemms_match_torch_conv_autograd(n, ih, iw, cin, cout, kh, kw):
    lib = load_oracle()
    x, w, b = _rand_conv(n, ih, iw, cin, cout, kh, kw, 3)
    x.requires_grad_(True); w.requires_grad_(True); b.requires_grad_(True)
    y_t = F.elu(F.conv2d(x, w, b), ALPHA)
    oh, ow = y_t.shape[2:]
    xs, wk = _nhwc(x), np.ascontiguousarray(w.detach().permute(0, 2, 3, 1).numpy())
    bs = np.ascontiguousarray(b.detach().numpy())
    y = np.zeros((n, oh, ow, cout), np.float32)
    assert lib.qo_conv_nhwc_forward(_ptr(xs), _ptr(wk), _ptr(bs), _ptr(y), n, ih, iw, cin, kh, kw, cout, 1, ALPHA, None) == 0
    assert np.allclose(y, _nhwc(y_t), atol=3e-5)
    gout = torch.randn(y_t.shape, generator=torch.Generator().manual_seed(4))
    gx_t, gw_t, gb_t = torch.autograd.grad(y_t, (x, w, b), gout)
    g = _nhwc(gout)
    dy = np.zeros_like(y); dyp = np.full((n, oh + 2 * (kh - 1), ow + 2 * (kw - 1), cout), 7.0, np.float32)
    # the padded copy takes ONE pad: the test uses kh - 1 = kw - 1 or pads by the larger and slices
    pad = max(kh, kw) - 1
    dyp = np.full((n, oh + 2 * pad, ow + 2 * pad, cout), 7.0, np.float32)
    assert lib.qo_elu_backward_pad(_ptr(g), _ptr(y), _ptr(dy), _ptr(dyp), n, oh, ow, cout, pad, 1, ALPHA, None) == 0
    pre_t = gout * torch.where(y_t > 0, torch.ones_like(y_t), y_t + ALPHA)
    assert np.allclose(dy, _nhwc(pre_t), atol=1e-6)
    assert np.array_equal(dyp[:, pad:pad + oh, pad:pad + ow], dy) and dyp.sum(dtype=np.float64) == pytest.approx(dy.sum(dtype=np.float64), abs=1e-3)
    dyp = np.ascontiguousarray(dyp[:, pad - (kh - 1):pad + oh + (kh - 1), pad - (kw - 1):pad + ow + (kw - 1)])
    nb = lib.qo_conv_nhwc_backward_weight_scratch_bytes(n, ih, iw, cin, kh, kw, cout)
    sc = np.zeros(nb, np.uint8); gw = np.zeros((cout, kh, kw, cin), np.float32); gb = np.zeros(cout, np.float32)
    assert lib.qo_conv_nhwc_backward_weight(_ptr(xs), _ptr(dy), _ptr(gw), _ptr(gb), n, ih, iw, cin, kh, kw, cout, _ptr(sc), nb, None) == 0
    assert np.allclose(gw, gw_t.permute(0, 2, 3, 1).numpy(), rtol=1e-4, atol=1e-4)
    assert np.allclose(gb, gb_t.numpy(), rtol=1e-4, atol=1e-4)
    # input gradient (times the derivative of the ELU that produced x: x is an ELU output here)
    if cout >= 16 and not (cout & (cout - 1)):
        wf = np.ascontiguousarray(w.detach().flip(2, 3).permute(1, 2, 3, 0).numpy())
        gin = np.zeros((n, ih, iw, cin), np.float32)
        assert lib.qo_conv_nhwc_backward_input(_ptr(dyp), _ptr(wf), _ptr(xs), _ptr(gin), n, oh + 2 * (kh - 1), ow + 2 * (kw - 1), cout, kh, kw, cin, 1, ALPHA, None) == 0
        xd = x.detach()
        want = gx_t * torch.where(xd > 0, torch.ones_like(xd), xd + ALPHA)
        assert np.allclose(gin, _nhwc(want), rtol=1e-4, atol=1e-4)


def test_oracle_rejects_shapes_the_kernels_do_not_take():
    lib = load_oracle()
    z = np.zeros(4096, np.float32)
    assert lib.qo_conv_nhwc_forward(_ptr(z), _ptr(z), None, _ptr(z), 1, 27, 41, 24, 3, 3, 64, 1, ALPHA, None) != 0        # cin not a power of two
    assert lib.qo_conv_nhwc_forward(_ptr(z), _ptr(z), None, _ptr(z), 1, 10, 10, 32, 3, 3, 64, 1, ALPHA, None) != 0        # 64 output pixels
    assert lib.qo_depth_stem_forward(_ptr(z), _ptr(z), _ptr(z), _ptr(z), _ptr(z), 1, 58, 200, ALPHA, None) != 0


# ------------------------------------------------------------------ GPU
def _dev(a):
    return torch.from_numpy(a).cuda()


@pytest.mark.gpu
@pytest.mark.parametrize("n,ih,iw", [(3, 58, 87), (2, 24, 31), (70, 58, 87)])
def test_hip_stem_matches_oracle(n, ih, iw):
    from quadrupedal_agility_amd.tsc.rsl_rl.modules import depth_stem as ds
    lib = load_oracle()
    img, w1, b1 = _rand_stem(n, ih, iw, 5)
    ph, pw = (ih - 4) // 2, (iw - 4) // 2
    y = np.zeros((n, ph, pw, 32), np.float32); am = np.zeros((n, ph, pw, 32), np.uint8)
    imgs, w, b = (np.ascontiguousarray(t.numpy()) for t in (img, w1, b1))
    assert lib.qo_depth_stem_forward(_ptr(imgs), _ptr(w), _ptr(b), _ptr(y), _ptr(am), n, ih, iw, ALPHA, None) == 0
    yh, amh = ds.stem_forward_raw(img.cuda(), w1.cuda(), b1.cuda(), ALPHA)
    torch.cuda.synchronize()
    assert np.allclose(yh.cpu().numpy(), y, atol=3e-6)
    assert (amh.cpu().numpy() == am).mean() > 0.9999
    gpre = np.random.default_rng(6).standard_normal((n, ph, pw, 32)).astype(np.float32)
    amc = np.ascontiguousarray(amh.cpu().numpy())
    gwb = np.zeros(832, np.float32); sc = np.zeros(16, np.uint8)
    assert lib.qo_depth_stem_backward(_ptr(imgs), _ptr(amc), _ptr(gpre), _ptr(gwb), n, ih, iw, _ptr(sc), 16, None) == 0
    out = ds.stem_backward_raw(img.cuda(), amh, _dev(gpre))
    out2 = ds.stem_backward_raw(img.cuda(), amh, _dev(gpre))
    torch.cuda.synchronize()
    assert torch.equal(out, out2)                                   # fixed summation order
    assert np.allclose(out.cpu().numpy(), gwb, rtol=2e-4, atol=2e-4 * np.sqrt(n))


@pytest.mark.gpu
@pytest.mark.parametrize("n,ih,iw,cin,cout,kh,kw", [(3, 27, 41, 32, 64, 3, 3), (1, 20, 19, 16, 16, 2, 3), (37, 27, 41, 32, 64, 3, 3)])
def test_hip_window_gemms_match_oracle(n, ih, iw, cin, cout, kh, kw):
    from quadrupedal_agility_amd.tsc.rsl_rl.modules import depth_stem as ds
    lib = load_oracle()
    x, w, b = _rand_conv(n, ih, iw, cin, cout, kh, kw, 7)
    oh, ow = ih - kh + 1, iw - kw + 1
    xs, wk, bs = _nhwc(x), np.ascontiguousarray(w.permute(0, 2, 3, 1).numpy()), np.ascontiguousarray(b.numpy())
    y = np.zeros((n, oh, ow, cout), np.float32)
    assert lib.qo_conv_nhwc_forward(_ptr(xs), _ptr(wk), _ptr(bs), _ptr(y), n, ih, iw, cin, kh, kw, cout, 1, ALPHA, None) == 0
    yh = ds.conv_forward_raw(_dev(xs), _dev(wk), _dev(bs), 1, ALPHA)
    torch.cuda.synchronize()
    assert np.allclose(yh.cpu().numpy(), y, rtol=1e-5, atol=3e-5)
    g = np.random.default_rng(8).standard_normal(y.shape).astype(np.float32)
    if kh == kw:
        pad = kh - 1
        dy = np.zeros_like(y); dyp = np.zeros((n, oh + 2 * pad, ow + 2 * pad, cout), np.float32)
        assert lib.qo_elu_backward_pad(_ptr(g), _ptr(y), _ptr(dy), _ptr(dyp), n, oh, ow, cout, pad, 1, ALPHA, None) == 0
        dyh, dyph = ds.elu_backward_pad_raw(_dev(g), _dev(y), pad, 1, ALPHA)
        torch.cuda.synchronize()
        assert np.array_equal(dyh.cpu().numpy(), dy) and np.array_equal(dyph.cpu().numpy(), dyp)
        wf = np.ascontiguousarray(w.flip(2, 3).permute(1, 2, 3, 0).numpy())
        gin = np.zeros((n, ih, iw, cin), np.float32)
        assert lib.qo_conv_nhwc_backward_input(_ptr(dyp), _ptr(wf), _ptr(xs), _ptr(gin), n, oh + 2 * pad, ow + 2 * pad, cout, kh, kw, cin, 1, ALPHA, None) == 0
        ginh = ds.conv_backward_input_raw(_dev(dyp), _dev(wf), _dev(xs), 1, ALPHA)
        torch.cuda.synchronize()
        assert np.allclose(ginh.cpu().numpy(), gin, rtol=1e-4, atol=1e-4)
    else:
        dy = g
    nb = lib.qo_conv_nhwc_backward_weight_scratch_bytes(n, ih, iw, cin, kh, kw, cout)
    sc = np.zeros(nb, np.uint8); gw = np.zeros((cout, kh, kw, cin), np.float32); gb = np.zeros(cout, np.float32)
    assert lib.qo_conv_nhwc_backward_weight(_ptr(xs), _ptr(dy), _ptr(gw), _ptr(gb), n, ih, iw, cin, kh, kw, cout, _ptr(sc), nb, None) == 0
    gwh, gbh = ds.conv_backward_weight_raw(_dev(xs), _dev(dy), kh, kw)
    gwh2, gbh2 = ds.conv_backward_weight_raw(_dev(xs), _dev(dy), kh, kw)
    torch.cuda.synchronize()
    assert torch.equal(gwh, gwh2) and torch.equal(gbh, gbh2)
    tol = 2e-4 * np.sqrt(n * oh * ow)
    assert np.allclose(gwh.cpu().numpy(), gw, rtol=2e-4, atol=tol * 0.05)
    assert np.allclose(gbh.cpu().numpy(), gb, rtol=2e-4, atol=tol * 0.05)


@pytest.mark.gpu
@pytest.mark.parametrize("batch", [2, 256])
def test_depth_encoder_on_the_stem_equals_the_encoder_on_aten(batch):
    """the whole `DepthOnlyFCBackbone58x87` (reference parameter layout): output and every parameter gradient, stem kernels vs aten / MIOpen"""
    from quadrupedal_agility_amd.tsc.rsl_rl.modules import depth_stem as ds
    from quadrupedal_agility_amd.tsc.rsl_rl.modules.depth_backbone import DepthOnlyFCBackbone58x87
    torch.manual_seed(0)
    net = DepthOnlyFCBackbone58x87(53, 32, 512).cuda()
    img = (torch.rand(batch, 58, 87, device="cuda") - 0.5)
    gout = torch.randn(batch, 32, device="cuda")
    res = []
    for on in (True, False):
        ds.ENABLED = on
        try:
            net.zero_grad(set_to_none=True)
            out = net(img)
            out.backward(gout)
            res.append((out.detach().clone(), {k: p.grad.clone() for k, p in net.named_parameters()}))
        finally:
            ds.ENABLED = True
    (oa, ga), (ob, gb) = res
    assert torch.allclose(oa, ob, rtol=1e-4, atol=2e-5)
    for k in ga:
        scale = float(gb[k].abs().max()) + 1e-12
        assert float((ga[k] - gb[k]).abs().max()) <= 3e-4 * scale + 1e-6, k


@pytest.mark.gpu
def test_stem_is_skipped_under_no_grad_bookkeeping_and_for_other_layouts():
    from quadrupedal_agility_amd.tsc.rsl_rl.modules import depth_stem as ds
    from quadrupedal_agility_amd.tsc.rsl_rl.modules.depth_backbone import DepthOnlyFCBackbone58x87
    net = DepthOnlyFCBackbone58x87(53, 32, 512).cuda()
    assert ds.stem_matches(net.image_compression)
    two = DepthOnlyFCBackbone58x87(53, 32, 512, num_frames=2)
    assert not ds.stem_matches(two.image_compression)
    with torch.no_grad():
        out = net(torch.rand(4, 58, 87, device="cuda"))
    assert out.shape == (4, 32) and not out.requires_grad


@pytest.mark.gpu
def test_hip_entries_reject_what_the_twins_reject():
    """argument contract of the new entry points: same refusals on both sides of the ABI, and a readable qa_last_error()"""
    from quadrupedal_agility_amd import _capi
    lib, lo = _capi.load_library(), load_oracle()
    z = torch.zeros(1 << 16, device="cuda"); zc = np.zeros(1 << 16, np.float32)
    P = lambda t: C.c_void_p(t.data_ptr())
    cases = [dict(ih=27, iw=41, cin=24, kh=3, kw=3, cout=64),        # in channels not a power of two
             dict(ih=10, iw=10, cin=32, kh=3, kw=3, cout=64),        # 64 output pixels per image
             dict(ih=27, iw=41, cin=32, kh=3, kw=3, cout=6),         # out channels not a multiple of 4
             dict(ih=27, iw=41, cin=8, kh=3, kw=3, cout=64)]         # fewer than 16 in channels
    for c in cases:
        a = (1, c["ih"], c["iw"], c["cin"], c["kh"], c["kw"], c["cout"])
        assert lib.qa_conv_nhwc_forward(P(z), P(z), None, P(z), *a, 1, ALPHA, None) != 0 and b"qa_conv_nhwc_forward" in lib.qa_last_error()
        assert lo.qo_conv_nhwc_forward(_ptr(zc), _ptr(zc), None, _ptr(zc), *a, 1, ALPHA, None) != 0
        assert lib.qa_conv_nhwc_backward_weight_scratch_bytes(*a) == 0 and lo.qo_conv_nhwc_backward_weight_scratch_bytes(*a) == 0
    off = z[1:]                                                      # 4-byte aligned only
    assert lib.qa_conv_nhwc_forward(P(off), P(z), None, P(z), 1, 27, 41, 32, 3, 3, 64, 1, ALPHA, None) != 0
    nb = lib.qa_conv_nhwc_backward_weight_scratch_bytes(1, 27, 41, 32, 3, 3, 64)
    assert nb > 0 and lib.qa_conv_nhwc_backward_weight(P(z), P(z), P(z), P(z), 1, 27, 41, 32, 3, 3, 64, P(z), nb - 16, None) != 0
    assert lib.qa_depth_stem_forward(P(z), P(z), P(z), P(z), P(z), 1, 58, 200, ALPHA, None) != 0                   # wider than the LDS rows
    assert lib.qa_depth_stem_forward(P(z), P(z), P(z), P(z), P(z), 0, 58, 87, ALPHA, None) != 0                    # no images
    assert lib.qa_elu_backward_pad(P(z), P(z), P(z), P(z), 1, 5, 5, 6, 2, 1, ALPHA, None) != 0                      # channels not a multiple of 4
    torch.cuda.synchronize()
