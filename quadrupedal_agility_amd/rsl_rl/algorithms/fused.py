"""torch.autograd bindings of the fused learner kernels in libqa_sim.so (include/qa_sim.h, "learner kernels").

Only CUDA/ROCm tensors come here: the callers keep the plain PyTorch expression for CPU tensors (BASELINE config 1,
"PPO on CPU physics + CPU torch"), which is also the fp32 reference the kernels are tested against."""
import ctypes as C
import os

import torch

from quadrupedal_agility_amd import _capi


ENABLED = True      # tests flip this to compare against the unfused modules


def _ptr(t):
    return C.c_void_p(t.data_ptr())


def _f32c(t):
    return t if (t.dtype == torch.float32 and t.is_contiguous()) else t.contiguous().float()


class _PpoLoss(torch.autograd.Function):
    """loss, stats = f(mu (B,12), std (12), value (B,1) | fixed minibatch tensors); stats = [loss, surrogate, value,
    bound, entropy, kl, 0, 0] (means, no gradient).  One kernel pass computes the gradient too; backward only scales."""

    @staticmethod
    def forward(ctx, mu, std, value, actions, old_logp, old_mu, old_sigma, advantages, returns, target_values,
                clip, c_surr, c_value, c_bound, c_entropy, clipped_value):
        lib = _capi.load_library()
        mu_c, std_c, val_c = _f32c(mu), _f32c(std), _f32c(value)
        fixed = [_f32c(x) for x in (actions, old_logp, old_mu, old_sigma, advantages, returns, target_values)]
        B = mu_c.shape[0]
        assert mu_c.is_cuda and mu_c.shape == (B, 12) and std_c.numel() == 12 and val_c.numel() == B
        dmu = torch.empty_like(mu_c)
        dstd = torch.empty(12, dtype=torch.float32, device=mu_c.device)
        dvalue = torch.empty(B, dtype=torch.float32, device=mu_c.device)
        out = torch.empty(8, dtype=torch.float32, device=mu_c.device)
        nscratch = int(lib.qa_ppo_loss_scratch_bytes(B))
        scratch = torch.empty(nscratch, dtype=torch.uint8, device=mu_c.device)
        stream = C.c_void_p(torch.cuda.current_stream(mu_c.device).cuda_stream)
        rc = lib.qa_ppo_loss(_ptr(mu_c), _ptr(std_c), _ptr(val_c), *[_ptr(x) for x in fixed], B, 12, float(clip), float(c_surr),
                             float(c_value), float(c_bound), float(c_entropy), int(bool(clipped_value)),
                             _ptr(dmu), _ptr(dstd), _ptr(dvalue), _ptr(out), _ptr(scratch), nscratch, stream)
        if rc != 0:
            raise RuntimeError(f"qa_ppo_loss failed with code {rc}: {lib.qa_last_error().decode()}")
        ctx.save_for_backward(dmu, dstd, dvalue)
        ctx.value_shape = value.shape
        ctx.std_shape = std.shape
        ctx.mark_non_differentiable(out)
        return out[0].clone(), out

    @staticmethod
    def backward(ctx, g_loss, _g_stats):
        dmu, dstd, dvalue = ctx.saved_tensors
        return (dmu * g_loss, (dstd * g_loss).view(ctx.std_shape), (dvalue * g_loss).view(ctx.value_shape),
                None, None, None, None, None, None, None, None, None, None, None, None, None)


def ppo_loss_raw(mu, std, value, actions, old_logp, old_mu, old_sigma, advantages, returns, target_values, *, clip, c_surr,
                 c_value, c_bound, c_entropy, clipped_value=True):
    """qa_ppo_loss without the autograd wrapper: (stats[8], d loss/d mu (B,12), d loss/d std (12), d loss/d value (B)).  The
    caller feeds the gradients straight into autograd.backward() of the network outputs -- no scalar loss node, so none of
    the ones-fill / scale / accumulate launches a `loss.backward()` spends on it."""
    lib = _capi.load_library()
    mu_c, std_c, val_c = _f32c(mu.detach()), _f32c(std.detach()), _f32c(value.detach())
    fixed = [_f32c(x) for x in (actions, old_logp, old_mu, old_sigma, advantages, returns, target_values)]
    B = mu_c.shape[0]
    assert mu_c.is_cuda and mu_c.shape == (B, 12) and std_c.numel() == 12 and val_c.numel() == B
    dmu = torch.empty_like(mu_c)
    dstd = torch.empty(12, dtype=torch.float32, device=mu_c.device)
    dvalue = torch.empty(B, dtype=torch.float32, device=mu_c.device)
    out = torch.empty(8, dtype=torch.float32, device=mu_c.device)
    nscratch = int(lib.qa_ppo_loss_scratch_bytes(B))
    scratch = torch.empty(nscratch, dtype=torch.uint8, device=mu_c.device)
    rc = lib.qa_ppo_loss(_ptr(mu_c), _ptr(std_c), _ptr(val_c), *[_ptr(x) for x in fixed], B, 12, float(clip), float(c_surr),
                         float(c_value), float(c_bound), float(c_entropy), int(bool(clipped_value)),
                         _ptr(dmu), _ptr(dstd), _ptr(dvalue), _ptr(out), _ptr(scratch), nscratch,
                         C.c_void_p(torch.cuda.current_stream(mu_c.device).cuda_stream))
    if rc != 0:
        raise RuntimeError(f"qa_ppo_loss failed with code {rc}: {lib.qa_last_error().decode()}")
    return out, dmu, dstd, dvalue


def hybrid_ppo_loss_raw(logits, mean, std, value, actions, old_logp_d, old_logp_c, old_mu, old_sigma, advantages, returns, target_values, *,
                        clip, c_value, c_entropy, clipped_value=True):
    """qa_hybrid_ppo_loss (the task-level learner's categorical + Gaussian objective): (out[8], d loss/d logits (B,3), d mean (B,18),
    d std (18), d value (B)); None when the head widths are not the built ones (the caller keeps its eager expression)."""
    lib = _capi.load_library()
    lg, mu_c, std_c, val_c = (_f32c(x.detach()) for x in (logits, mean, std, value))
    B, nd, nc = lg.shape[0], lg.shape[1], mu_c.shape[1]
    if (nd, nc) != (3, 18):
        return None
    fixed = [_f32c(x) for x in (actions, old_logp_d, old_logp_c, old_mu, old_sigma, advantages, returns, target_values)]
    assert fixed[0].shape == (B, 1 + nc) and val_c.numel() == B
    dev = lg.device
    dlogits, dmean = torch.empty_like(lg), torch.empty_like(mu_c)
    dstd, dvalue, out = torch.empty(nc, device=dev), torch.empty(B, device=dev), torch.empty(8, device=dev)
    n = int(lib.qa_hybrid_ppo_loss_scratch_bytes(B))
    scratch = torch.empty(n, dtype=torch.uint8, device=dev)
    rc = lib.qa_hybrid_ppo_loss(_ptr(lg), _ptr(mu_c), _ptr(std_c), _ptr(val_c), *[_ptr(x) for x in fixed], B, nd, nc, float(clip), float(c_value),
                                float(c_entropy), int(bool(clipped_value)), _ptr(dlogits), _ptr(dmean), _ptr(dstd), _ptr(dvalue), _ptr(out),
                                _ptr(scratch), n, C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
    if rc != 0:
        raise RuntimeError(f"qa_hybrid_ppo_loss failed with code {rc}: {lib.qa_last_error().decode()}")
    return out, dlogits, dmean, dstd, dvalue


# ---- gradients left in parts (r5) ---------------------------------------------------------------------------------------------------
# The weight gradient of a wide Linear layer comes out of `weight_grad` as S slabs and its bias gradient out of qa_elu_backward_bias as one
# row of column sums per 64-row block; each used to be finished by its own fixed-order launch (qa_slab_reduce x 7, qa_colsum_finish x 12 per
# PPO minibatch step, ~5 us each, between GEMMs that wait for nothing they produce).  Inside `deferred_grad_finishes()` the backward
# functions hand autograd an UNWRITTEN gradient tensor and register the parts here, keyed by the parameter; ClipAdam.step() passes them to
# qa_clip_adam_step_reduce, whose first pass (the clipping norm) adds the parts in the same fixed order, writes the finished gradient into
# `param.grad` and squares it.  Whoever needs finished gradients before the optimiser runs (the data-parallel bucket, tests) calls
# `flush_pending_grads()` -- ONE launch for all of them.  Nothing may read `.grad` of a registered parameter before one of the two ran.
_DEFER = [False]
_PENDING = {}          # parameter data_ptr -> (parameter, parts tensor, number of parts, stride between parts in elements, placeholder alias, stream)
_RETIRED = []          # parts tensors of the step before: see _retire
_NO_DEFER = set()      # parameters that already received a second gradient in this deferral context: finished gradients only from here on


def _retire(parts):
    """The parts were written on the stream their backward node ran on (a branch stream of the PPO step) and are read by the optimiser's
    kernel on the CURRENT stream: the caching allocator must not hand the block back to the branch stream while that kernel can still be
    reading it.  Eager: record the consumer stream on the block.  Always: keep the tensor until the next step opens (by then every branch
    stream has been told to wait for the stream the optimiser ran on)."""
    if getattr(parts, "_qa_persistent", False):       # a buffer its owner keeps for good (train_chain): it never goes back to the allocator
        return
    if not torch.cuda.is_current_stream_capturing():
        parts.record_stream(torch.cuda.current_stream(parts.device))
    _RETIRED.append(parts)


class deferred_grad_finishes:
    def __enter__(self):
        self.prev = _DEFER[0]
        _DEFER[0] = ENABLED and os.environ.get("QA_DEFER_GRAD_FINISH", "1") != "0"
        del _RETIRED[:]
        _NO_DEFER.clear()
        return self

    def __exit__(self, *exc):
        _DEFER[0] = self.prev
        return False


def _defer_ok(param):
    """May this backward node leave `param`'s gradient in parts?  Only when the placeholder it hands autograd is the ONLY thing that can
    reach `.grad` before a finish runs: no gradient accumulated earlier (`.grad is None`: autograd would add the unwritten placeholder to
    it), no double-backward graph (autograd clones instead of adopting the placeholder), and no second gradient for the same parameter --
    a parameter used twice in one graph (the privileged encoder under `train_with_estimated_latent`: once for the regulariser, once inside
    the actor) or met again by a later backward call.  The second arrival is what `_second_gradient` handles."""
    if not (_DEFER[0] and param is not None and param.is_cuda and param.is_leaf and param.requires_grad):
        return False
    key = param.data_ptr()
    return key not in _PENDING and key not in _NO_DEFER and param.grad is None and not torch.is_grad_enabled()


def _gradient_arriving(*params):
    """every backward node of this module that produces a gradient for a Parameter says so BEFORE it decides about deferral"""
    for param in params:
        if param is not None and _PENDING and param.data_ptr() in _PENDING:
            _second_gradient(param)


def _second_gradient(param):
    """`param` has a gradient in parts and another one is about to be produced.  Autograd will ADD the two tensors, so the first must hold
    its finished value by then: the parts are summed NOW (one qa_grad_reduce launch) into the memory of the placeholder the first node
    returned -- which is either still waiting in the engine's input buffer (same backward call: the engine adds after this node returns, and
    orders that after everything launched here) or has become `.grad` (an earlier backward call: autograd adopts the placeholder's
    storage, or cloned it, in which case `.grad` itself is the destination).  From here on the parameter gets finished gradients only."""
    lib = _capi.load_library()
    p, parts, nparts, stride, alias, stream = _PENDING.pop(param.data_ptr())
    _NO_DEFER.add(param.data_ptr())
    dst = alias if (p.grad is None or p.grad.data_ptr() == alias.data_ptr()) else p.grad
    if not dst.is_contiguous() or dst.numel() != p.numel():
        raise RuntimeError("a gradient registered in parts cannot be finished: its destination is not a contiguous tensor of the parameter's size")
    cur = torch.cuda.current_stream(p.device)
    if stream is not None and stream != cur:
        cur.wait_stream(stream)          # the parts were written on the stream the first node ran on
    _check(lib.qa_grad_reduce((C.c_void_p * 1)(dst.data_ptr()), (C.c_void_p * 1)(parts.data_ptr()), (C.c_int64 * 1)(stride), (C.c_int32 * 1)(nparts),
                              (C.c_int32 * 1)(p.numel()), 1, C.c_void_p(cur.cuda_stream)), "qa_grad_reduce")
    _retire(parts)


def _register_parts(param, parts, nparts, stride, shape=None):
    """-> the placeholder to hand autograd for `param`'s gradient (unwritten; the finish writes where autograd keeps it: `.grad` adopts the
    placeholder's storage).  `_PENDING` keeps a second tensor object on the same storage, not the placeholder itself: autograd adopts a
    gradient only when nobody else holds the tensor object, and would clone it (one launch per parameter) otherwise."""
    buf = torch.empty(tuple(param.shape) if shape is None else shape, dtype=torch.float32, device=param.device)
    _PENDING[param.data_ptr()] = (param, parts, int(nparts), int(stride), buf, torch.cuda.current_stream(param.device))
    return buf.detach()


def register_grad_parts(param, parts, nparts, stride, grad):
    """A gradient computed OUTSIDE autograd, left in parts (train_chain.PpoTrainChain.backward): `grad` is the tensor the finish writes
    (the caller has made it `param.grad`), `parts` a buffer the caller keeps.  One product per parameter per step, assigned, never accumulated."""
    # a chain step's gradient REPLACES what a previous call left unfinished (the recorded update's warm-up pass runs the step without the
    # optimiser; a step whose capture failed is run again eagerly): these gradients are never accumulated, the newest product is the gradient
    _PENDING.pop(param.data_ptr(), None)
    parts._qa_persistent = True
    _PENDING[param.data_ptr()] = (param, parts, int(nparts), int(stride), grad, torch.cuda.current_stream(param.device))


def pending_grads():
    return len(_PENDING)


def flush_pending_grads(params=None):
    """finish the gradients still in parts (all of them, or those of `params`) into their `.grad` tensors: one launch (qa_grad_reduce)"""
    keys = list(_PENDING) if params is None else [p.data_ptr() for p in params if p.data_ptr() in _PENDING]
    if not keys:
        return
    lib = _capi.load_library()
    ent = [_PENDING.pop(k) for k in keys]
    for p, *_ in ent:
        if p.grad is None or not p.grad.is_contiguous():
            raise RuntimeError("a gradient registered in parts has no contiguous .grad tensor to be finished into")
    for i in range(0, len(ent), 64):
        e = ent[i:i + 64]
        n = len(e)
        dst = (C.c_void_p * n)(*[x[0].grad.data_ptr() for x in e]); src = (C.c_void_p * n)(*[x[1].data_ptr() for x in e])
        stride = (C.c_int64 * n)(*[x[3] for x in e]); parts = (C.c_int32 * n)(*[x[2] for x in e]); numel = (C.c_int32 * n)(*[x[0].numel() for x in e])
        _check(lib.qa_grad_reduce(dst, src, stride, parts, numel, n, C.c_void_p(torch.cuda.current_stream(e[0][0].device).cuda_stream)), "qa_grad_reduce")
    for e in ent:
        _retire(e[1])


class _LinearElu(torch.autograd.Function):
    """y = elu(x W^T + b).  Forward: addmm (hipBLASLt) + in-place ELU.  Backward: ONE kernel for the ELU derivative and
    the bias gradient (qa_elu_backward_bias), then the two GEMMs.  Saves the ELU output only (elu' = y + alpha for y <= 0)."""

    @staticmethod
    def forward(ctx, x, weight, bias, alpha):
        # r4 (profiles/r4_gemm_dma_bench.json, 24,576 rows): for the narrow trunk layers (256 -> 128) the hand-written GEMM with bias + ELU in its
        # epilogue is one 22 us launch against the library's 26 us product + a 6 us ELU pass; the wider products stay on the tuned library
        if (OWN_GEMM and OWN_FWD_NARROW and x.shape[0] >= 8192 and weight.shape[0] <= 128 and 64 < weight.shape[1] <= 256 and weight.is_contiguous()
                and bias.is_contiguous() and x.stride(1) == 1):
            y = linear_forward_raw(x, weight.detach(), bias.detach(), ACT_ELU, alpha)
        else:
            y = torch.addmm(bias, x, weight.t())
            torch.nn.functional.elu(y, alpha=alpha, inplace=True)
        ctx.save_for_backward(x, weight, y)
        ctx.alpha = alpha
        ctx.bias_param = bias
        return y

    @staticmethod
    def backward(ctx, gy):
        x, weight, y = ctx.saved_tensors
        bias = ctx.bias_param
        lib = _capi.load_library()
        _gradient_arriving(weight, bias)
        gy = _f32c(gy)
        rows, cols = y.shape
        g = torch.empty_like(y)
        gb = torch.empty(cols, dtype=torch.float32, device=y.device)
        nscratch = int(lib.qa_elu_backward_bias_scratch_bytes(rows, cols))
        scratch = torch.empty(nscratch, dtype=torch.uint8, device=y.device)
        stream = C.c_void_p(torch.cuda.current_stream(y.device).cuda_stream)
        defer_b = ctx.needs_input_grad[2] and _defer_ok(bias) and bias.numel() == cols
        rc = lib.qa_elu_backward_bias(_ptr(gy), _ptr(y), _ptr(g), None if defer_b else _ptr(gb), rows, cols, float(ctx.alpha), _ptr(scratch), nscratch, stream)
        if rc != 0:
            raise RuntimeError(f"qa_elu_backward_bias failed with code {rc}: {lib.qa_last_error().decode()}")
        if defer_b:           # the optimiser's first pass adds the (rows / 64) rows of column sums; autograd gets an unwritten placeholder
            gb = _register_parts(bias, scratch, (rows + 63) // 64, cols)
        gx = g.mm(weight) if ctx.needs_input_grad[0] else None
        gw = weight_grad(g, x, param=weight) if ctx.needs_input_grad[1] else None
        return gx, gw, (gb if ctx.needs_input_grad[2] else None), None


# the discriminator's trunk (3 x 1228 rows, 98 -> 1024 -> 512): its step is a chain of launch-latency-sized kernels, a launch saved is 5 us saved
RELU_OWN_MAX_ROWS = 8192


class _LinearRelu(torch.autograd.Function):
    """y = relu(x W^T + b) with the same backward kernel as _LinearElu (ELU with alpha = 0 IS ReLU: derivative 1 for y > 0, y + 0 = 0
    otherwise), i.e. the bias gradient is our fixed-order column sum, not torch's `sum(0)` (see profiles/r2_hipgraph_stale_reductions.md)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        if OWN_GEMM and x.shape[0] <= RELU_OWN_MAX_ROWS and weight.is_contiguous() and bias.is_contiguous():
            y = linear_forward_raw(x, weight.detach(), bias.detach(), ACT_RELU)      # bias + ReLU in the GEMM's epilogue: one launch instead of two
        else:
            y = torch.addmm(bias, x, weight.t())
            torch.relu_(y)
        ctx.save_for_backward(x, weight, y)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, weight, y = ctx.saved_tensors
        _gradient_arriving(weight)
        g, gb = _masked_colsum(gy, y)
        gx = g.mm(weight) if ctx.needs_input_grad[0] else None
        gw = weight_grad(g, x) if ctx.needs_input_grad[1] else None
        return gx, gw, (gb if ctx.needs_input_grad[2] else None)


def _masked_colsum(gy, y):
    """(gy * [y > 0], column sums of that): qa_elu_backward_bias with alpha = 0"""
    lib = _capi.load_library()
    gy, y = _f32c(gy), _f32c(y)
    rows, cols = y.shape
    g = torch.empty_like(y)
    gb = torch.empty(cols, dtype=torch.float32, device=y.device)
    nscratch = int(lib.qa_elu_backward_bias_scratch_bytes(rows, cols))
    scratch = torch.empty(nscratch, dtype=torch.uint8, device=y.device)
    rc = lib.qa_elu_backward_bias(_ptr(gy), _ptr(y), _ptr(g), _ptr(gb), rows, cols, 0.0, _ptr(scratch), nscratch,
                                  C.c_void_p(torch.cuda.current_stream(y.device).cuda_stream))
    if rc != 0:
        raise RuntimeError(f"qa_elu_backward_bias failed with code {rc}: {lib.qa_last_error().decode()}")
    return g, gb


def linear_relu(m, x):
    """relu(m(x)) for an nn.Linear; through _LinearRelu on ROCm tensors under autograd"""
    if ENABLED and x.is_cuda and torch.is_grad_enabled() and x.dim() == 2 and m.bias is not None and x.dtype == torch.float32:
        return _LinearRelu.apply(x, m.weight, m.bias)
    return torch.relu(m(x))


class _MaskTimesRow(torch.autograd.Function):
    """v = mask * w for a constant 0/1 mask (rows, n) and a row vector w (1, n) (the start of the gradient-penalty chain): the
    gradient of w is a column sum over the rows -- ours, not torch's broadcast reduction."""

    @staticmethod
    def forward(ctx, mask, w):
        ctx.save_for_backward(mask)
        return mask * w

    @staticmethod
    def backward(ctx, gv):
        (mask,) = ctx.saved_tensors
        _, gw = _masked_colsum(gv, mask)
        return None, gw.view(1, -1)


def mask_times_row(mask, w):
    if ENABLED and mask.is_cuda and torch.is_grad_enabled() and w.dim() == 2 and w.shape[0] == 1 and mask.dtype == torch.float32:
        return _MaskTimesRow.apply(mask, w)
    return mask * w


WGRAD_SLABS = 8


def weight_grad(g, x, param=None):
    """dW = g^T x for g (rows, n), x (rows, k).  The output is small (<= 512 x 671) and the reduction long (24,576 rows): the
    library's picks for that shape run split-K kernels on a few dozen workgroups (32 TFLOP/s for 128 x 256).  Splitting the rows
    into 8 slabs through ONE batched GEMM gives it 8x the workgroups; the 8 partial products are added in a fixed order
    (deterministic).  tools/gemm_bench.py: 128x256 50 -> 28 us, 256x512 78 -> 61, 512x671 200 -> 166, 512x101 54 -> 36.
    Tiny outputs (estimator, encoder) are faster through the plain product, whose operand order is F.linear's own backward."""
    rows, n = g.shape
    k = x.shape[1]
    S = WGRAD_SLABS
    if ENABLED and rows >= 8192 and rows % S == 0 and n * k >= 16384 and g.stride(1) == 1 and x.stride(1) == 1:
        slabs = torch.bmm(g.unflatten(0, (S, rows // S)).transpose(1, 2), x.unflatten(0, (S, rows // S)))
        if _defer_ok(param) and param.shape == (n, k) and param.is_contiguous():
            return _register_parts(param, slabs, S, n * k)   # the S slabs are added by the optimiser's first pass (deferred_grad_finishes)
        return slab_sum(slabs)   # ours, not torch's sum(0)
    return g.t().mm(x)


class _NarrowLinear(torch.autograd.Function):
    """y = x W^T + b for a layer with few outputs (critic head 128 -> 1, actor head 128 -> 12, ...).  Forward and the input gradient
    are the library's GEMMs; the weight + bias gradient -- an (out x in) product with a rows-long reduction the library cannot
    parallelise (60 us for 1 x 128 over 24,576 rows) -- is the streaming kernel qa_narrow_wgrad."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        return torch.addmm(bias, x, weight.t())

    @staticmethod
    def backward(ctx, gy):
        x, weight = ctx.saved_tensors
        _gradient_arriving(weight)
        gy = _f32c(gy)
        gx = gy.mm(weight) if ctx.needs_input_grad[0] else None
        if not (ctx.needs_input_grad[1] or ctx.needs_input_grad[2]):
            return gx, None, None
        lib = _capi.load_library()
        xc = _f32c(x)
        rows, k = xc.shape
        o = weight.shape[0]
        gw = torch.empty_like(weight, memory_format=torch.contiguous_format)
        gb = torch.empty(o, dtype=torch.float32, device=x.device)
        n = int(lib.qa_narrow_wgrad_scratch_bytes(rows, o, k))
        scratch = torch.empty(n, dtype=torch.uint8, device=x.device)
        rc = lib.qa_narrow_wgrad(_ptr(gy), _ptr(xc), rows, o, k, _ptr(gw), _ptr(gb), _ptr(scratch), n,
                                 C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream))
        if rc != 0:
            raise RuntimeError(f"qa_narrow_wgrad failed with code {rc}: {lib.qa_last_error().decode()}")
        return gx, gw, gb


NARROW_MAX_OUT = 32
NARROW_MIN_ROWS = 1          # every batch size: an eager step and its recording must run the same kernels (and a recording must not contain torch's sum(0))


def plain_linear(m, x):
    """`m(x)` for an nn.Linear that is not followed by ELU / ReLU, of ANY width, without torch's bias reduction under autograd: narrow
    layers through qa_narrow_wgrad (`narrow_linear`), wider ones as a one-layer _MlpChain (forward, input gradient and weight + bias
    gradient on csrc/qa_gemm.hip).  A recorded step may contain any Linear of an nn.Sequential this way (a scan-encoder output
    wider than NARROW_MAX_OUT used to fall back to F.linear, whose bias gradient is torch's sum(0) -- stale under hipGraph replay)."""
    if (ENABLED and OWN_GEMM and x.is_cuda and torch.is_grad_enabled() and x.dim() == 2 and x.dtype == torch.float32 and m.bias is not None
            and m.out_features > NARROW_MAX_OUT and m.weight.requires_grad):
        return _MlpChain.apply(x, ((ACT_NONE, 0.0),), m.weight, m.bias)
    return narrow_linear(m, x)


def narrow_linear(m, x, always=False):
    """`m(x)` for an nn.Linear head; through _NarrowLinear on ROCm tensors under autograd when the layer is narrow and the batch long
    (`always`: whatever NARROW_MIN_ROWS says)"""
    if (ENABLED and x.is_cuda and torch.is_grad_enabled() and x.dim() == 2 and m.bias is not None and m.out_features <= NARROW_MAX_OUT
            and (x.shape[0] >= NARROW_MIN_ROWS or always) and x.dtype == torch.float32 and m.weight.requires_grad):
        return _NarrowLinear.apply(x, m.weight, m.bias)
    return m(x)


def linear_elu(x, weight, bias, alpha=1.0):
    return _LinearElu.apply(x, weight, bias, alpha)


OWN_FWD_NARROW = os.environ.get("QA_OWN_FWD_NARROW", "0") == "1"      # measured neutral end to end (31.2 vs 31.0 ms): off by default
OWN_GEMM = os.environ.get("QA_OWN_GEMM", "1") != "0"      # the dense layers through csrc/qa_gemm.hip (0: library GEMMs + the r1/r2 kernels around them)
ACT_NONE, ACT_ELU, ACT_RELU = 0, 1, 2


def _check(rc, name):
    if rc != 0:
        raise RuntimeError(f"{name} failed with code {rc}: {_capi.load_library().qa_last_error().decode()}")


def _rows2d(t):
    """a (rows, cols) fp32 view whose rows are contiguous (a column slice of wider rows is fine: the kernels take a leading dimension)"""
    if t.dtype != torch.float32 or t.stride(1) != 1 or t.stride(0) < t.shape[1] or t.data_ptr() % 4:
        t = t.contiguous().float()
    return t


def linear_forward_raw(x, weight, bias, act, alpha=1.0, out=None):
    """act(x W^T + b) by qa_linear_forward (one launch, bias and activation in the GEMM's epilogue)"""
    lib = _capi.load_library()
    x = _rows2d(x)
    rows, k = x.shape
    n = weight.shape[0]
    y = out if out is not None else torch.empty(rows, n, dtype=torch.float32, device=x.device)
    _check(lib.qa_linear_forward(_ptr(x), x.stride(0), _ptr(weight), weight.stride(0), _ptr(bias) if bias is not None else None, _ptr(y), y.stride(0),
                                 rows, k, n, int(act), float(alpha), C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)), "qa_linear_forward")
    return y


_split_scratch = {}


def linear_forward_split_raw(x, weight, bias, act, alpha=1.0):
    """act(x W^T + b) by qa_linear_forward_split: the reduction dimension split over workgroups (few outputs, very long rows)"""
    lib = _capi.load_library()
    x = _rows2d(x)
    rows, k = x.shape
    n = weight.shape[0]
    nb = lib.qa_linear_forward_split_scratch_bytes(rows, k, n)
    if torch.cuda.is_current_stream_capturing():
        # a recording bakes the address in: allocate from the graph's pool (as linear_backward_weight_raw does), never from a cache whose
        # buffer a later, larger request replaces -- the replays would write partial sums into freed memory
        sc = torch.empty(max(nb, 16), dtype=torch.uint8, device=x.device)
    else:
        key = (x.device, torch.cuda.current_stream(x.device).cuda_stream)
        sc = _split_scratch.get(key)
        if sc is None or sc.numel() < nb:
            sc = torch.empty(nb, dtype=torch.uint8, device=x.device)
            _split_scratch[key] = sc
    y = torch.empty(rows, n, dtype=torch.float32, device=x.device)
    _check(lib.qa_linear_forward_split(_ptr(x), x.stride(0), _ptr(weight), weight.stride(0), _ptr(bias) if bias is not None else None, _ptr(y), n, rows, k, n,
                                       int(act), float(alpha), _ptr(sc), sc.numel(), C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)),
           "qa_linear_forward_split")
    return y


def linear_backward_input_raw(g, weight, y_prev, act_prev, alpha=1.0):
    """(g W) * act'(y_prev) by qa_linear_backward_input: the input gradient of a layer with the previous layer's activation derivative"""
    lib = _capi.load_library()
    g = _rows2d(g)
    rows, n = g.shape
    k = weight.shape[1]
    gin = torch.empty(rows, k, dtype=torch.float32, device=g.device)
    yp = _rows2d(y_prev) if (act_prev and y_prev is not None) else None
    _check(lib.qa_linear_backward_input(_ptr(g), g.stride(0), _ptr(weight), weight.stride(0), _ptr(yp) if yp is not None else None,
                                        yp.stride(0) if yp is not None else 0, _ptr(gin), k, rows, k, n, int(act_prev if yp is not None else 0), float(alpha),
                                        C.c_void_p(torch.cuda.current_stream(g.device).cuda_stream)), "qa_linear_backward_input")
    return gin


def linear_backward_weight_raw(g, x, want_bias=True):
    """(g^T x, column sums of g) by qa_linear_backward_weight (split over row slabs, fixed-order reduction)"""
    lib = _capi.load_library()
    g, x = _rows2d(g), _rows2d(x)
    rows, n = g.shape
    k = x.shape[1]
    gw = torch.empty(n, k, dtype=torch.float32, device=g.device)
    gb = torch.empty(n, dtype=torch.float32, device=g.device)
    nb = int(lib.qa_linear_backward_weight_scratch_bytes(rows, k, n))
    scratch = torch.empty(nb, dtype=torch.uint8, device=g.device)
    _check(lib.qa_linear_backward_weight(_ptr(g), g.stride(0), _ptr(x), x.stride(0), _ptr(gw), _ptr(gb), rows, k, n, _ptr(scratch), nb,
                                         C.c_void_p(torch.cuda.current_stream(g.device).cuda_stream)), "qa_linear_backward_weight")
    return gw, gb


OWN_MAX_IN, OWN_MAX_OUT_WIDE = 128, 128      # a layer goes through csrc/qa_gemm.hip when it is narrow: out <= 64, or in <= 128 and out <= 128
ALL_OWN = os.environ.get("QA_GEMM_ALL_OWN", "0") == "1"     # every layer through our kernels (measurements, tests)


OWN_LAYERS = os.environ.get("QA_OWN_LAYERS", "heads")      # which layers run on csrc/qa_gemm.hip: none | heads | small | narrow (= heads + small) | all


def own_layer(k, n):
    """Which dense layers run on the hand-written GEMMs (DESIGN.md 4.18 has the measurements behind the default)."""
    if ALL_OWN or OWN_LAYERS == "all":
        return True
    head = n <= 32 and k >= 100                                   # actor / critic / gait heads: 128 -> 12 / 1 / 3 / 18
    small = (n <= 64 or (k <= OWN_MAX_IN and n <= OWN_MAX_OUT_WIDE)) and not head     # estimator 57-128-64-4, privileged encoder 29-64-29
    return {"none": False, "heads": head, "small": small, "narrow": head or small}.get(OWN_LAYERS, False)


def pad16(k):
    return (int(k) + 15) // 16 * 16


def pad_k():
    """First layers whose input width is not a multiple of 16 (critic 671, actor 101) run on rows the CALLER has zero-padded to whole 16-wide
    k-tiles (rollout storage / minibatch copy: 672; the actor's concatenation: 112) against a zero-padded copy of the weight: every operand
    of the wide products is then 16-byte aligned with whole k-tiles, i.e. eligible for the LDS-DMA GEMM (csrc/qa_gemm.hip).  Only when
    every layer runs on our kernels (QA_OWN_LAYERS=all): the library path multiplies by the unpadded weight."""
    return ENABLED and OWN_GEMM and (ALL_OWN or OWN_LAYERS == "all") and os.environ.get("QA_PAD_K", "1") != "0"


def slab_sum(parts):
    """sum over the leading dimension of a contiguous (S, ...) tensor by qa_slab_sum (fixed order) -- not torch's sum(0)"""
    lib = _capi.load_library()
    S = parts.shape[0]
    out = torch.empty(parts.shape[1:], dtype=torch.float32, device=parts.device)
    n = out.numel()
    _check(lib.qa_slab_sum(_ptr(parts), n, S, n, _ptr(out), C.c_void_p(torch.cuda.current_stream(parts.device).cuda_stream)), "qa_slab_sum")
    return out


class _MlpChain(torch.autograd.Function):
    """y = act_L(W_L ... act_1(W_1 x + b_1) ... + b_L) for a stack of Linear(+ELU/ReLU) layers, every product one of the hand-written
    fp32-MFMA GEMMs of csrc/qa_gemm.hip.  Forward: one launch per layer (bias + activation in the epilogue), activations saved.
    Backward, per layer from the top: weight + bias gradient (one split launch + its fixed-order reduction), then the input gradient with
    the layer below's activation derivative in the epilogue -- no elementwise or reduction kernel between the GEMMs, and in particular
    none of torch's batch reductions (profiles/r2_hipgraph_stale_reductions.md)."""

    @staticmethod
    def forward(ctx, x, spec, *params):
        h = _rows2d(x)
        acts = [h]
        # zero-padded input rows (`pad_k`): the first weight is padded to the same width, once per forward (1.4 MB for the critic)
        kpad = h.shape[1] - params[0].shape[1]
        assert 0 <= kpad < 16, (h.shape, params[0].shape)
        w0 = torch.nn.functional.pad(params[0].detach(), (0, kpad)) if kpad else params[0]
        for i, (act, alpha) in enumerate(spec):
            h = linear_forward_raw(h, w0 if i == 0 else params[2 * i], params[2 * i + 1], act, alpha)
            acts.append(h)
        ctx.spec, ctx.kpad = spec, kpad
        ctx.save_for_backward(*acts, *params, *([w0] if kpad else []))
        return h

    @staticmethod
    def backward(ctx, gy):
        spec, kpad = ctx.spec, ctx.kpad
        L = len(spec)
        saved = ctx.saved_tensors
        acts, params = saved[:L + 1], saved[L + 1:L + 1 + 2 * L]
        _gradient_arriving(*params)
        w0p = saved[-1] if kpad else None
        g = _rows2d(gy)
        act_top, alpha_top = spec[-1]
        if act_top != ACT_NONE:        # the chain ends in an activation (encoders): its derivative has no GEMM above it to ride on
            g, _ = _masked_colsum(g, acts[L]) if act_top == ACT_RELU else _elu_bwd(g, acts[L], alpha_top)
        grads = [None] * (2 * L)
        gx = None
        for i in range(L - 1, -1, -1):
            w = params[2 * i]
            need_w, need_b = ctx.needs_input_grad[2 + 2 * i], ctx.needs_input_grad[3 + 2 * i]
            if need_w or need_b:
                gw, gb = linear_backward_weight_raw(g, acts[i])
                if i == 0 and kpad:             # the padded columns' gradient is exactly zero (their inputs are): drop them
                    gw = gw[:, :w.shape[1]].contiguous()
                grads[2 * i], grads[2 * i + 1] = (gw if need_w else None), (gb if need_b else None)
            if i > 0:
                g = linear_backward_input_raw(g, w, acts[i], spec[i - 1][0], spec[i - 1][1])
            elif ctx.needs_input_grad[0]:
                gx = linear_backward_input_raw(g, w0p if kpad else w, None, ACT_NONE)       # (rows, padded k): zero in the padded columns
        return (gx, None, *grads)


def _elu_bwd(gy, y, alpha):
    lib = _capi.load_library()
    gy, y = _f32c(gy), _f32c(y)
    rows, cols = y.shape
    g = torch.empty_like(y)
    gb = torch.empty(cols, dtype=torch.float32, device=y.device)
    nscratch = int(lib.qa_elu_backward_bias_scratch_bytes(rows, cols))
    scratch = torch.empty(nscratch, dtype=torch.uint8, device=y.device)
    _check(lib.qa_elu_backward_bias(_ptr(gy), _ptr(y), _ptr(g), _ptr(gb), rows, cols, float(alpha), _ptr(scratch), nscratch,
                                    C.c_void_p(torch.cuda.current_stream(y.device).cuda_stream)), "qa_elu_backward_bias")
    return g, gb


def _chain_spec(mods):
    """[(Linear, act code, alpha), ...] when `mods` is a plain stack of Linear layers each optionally followed by ELU / ReLU, else None"""
    out, i = [], 0
    while i < len(mods):
        m = mods[i]
        if not (isinstance(m, torch.nn.Linear) and m.bias is not None):
            return None
        nxt = mods[i + 1] if i + 1 < len(mods) else None
        if isinstance(nxt, torch.nn.ELU):
            out.append((m, ACT_ELU, float(nxt.alpha))); i += 2
        elif isinstance(nxt, torch.nn.ReLU):
            out.append((m, ACT_RELU, 0.0)); i += 2
        else:
            out.append((m, ACT_NONE, 0.0)); i += 1
    return out or None


def _flatten(mods):
    flat = []
    for m in mods:
        flat.extend(list(m) if isinstance(m, torch.nn.Sequential) else [m])
    return flat


def mlp_chain(mods, x):
    """Run a stack of modules (Sequentials are flattened, so a trunk and its head are ONE stack) on a 2-D ROCm tensor under autograd.
    Consecutive narrow layers (`own_layer`) form one _MlpChain of hand-written GEMMs; the wide trunk layers go through `linear_elu` /
    `linear_relu` (library GEMM + our ELU-backward / bias / slab-sum kernels).  Anything the stack does not cover (other activations, CPU
    tensors, no_grad) takes the modules as they are."""
    flat = _flatten(mods)
    spec = _chain_spec(flat) if (ENABLED and OWN_GEMM and x.is_cuda and torch.is_grad_enabled() and x.dim() == 2 and x.dtype == torch.float32) else None
    if spec is None:
        for m in mods:
            if isinstance(m, torch.nn.Sequential):
                x = mlp_forward(m, x)
            elif isinstance(m, torch.nn.Linear):
                x = plain_linear(m, x)
            else:
                x = m(x)
        return x
    i = 0
    while i < len(spec):
        m, act, alpha = spec[i]
        if own_layer(m.in_features, m.out_features):
            j = i
            while j < len(spec) and own_layer(spec[j][0].in_features, spec[j][0].out_features):
                j += 1
            params = []
            for mm, _, _ in spec[i:j]:
                params += [mm.weight, mm.bias]
            x = _MlpChain.apply(x, tuple((a, al) for _, a, al in spec[i:j]), *params)
            i = j
        else:
            x = linear_elu(x, m.weight, m.bias, alpha) if act == ACT_ELU else (_LinearRelu.apply(x, m.weight, m.bias) if act == ACT_RELU else plain_linear(m, x))
            i += 1
    return x


def mlp_forward(seq, x):
    """Run an nn.Sequential of Linear / ELU modules; on ROCm tensors with gradients enabled through the hand-written GEMM chain
    (`mlp_chain`), or with QA_OWN_GEMM=0 every Linear+ELU pair through `linear_elu` (library GEMMs).  Anything else (CPU tensors,
    other activations, no_grad) takes the modules as they are."""
    mods = list(seq)
    if not (x.is_cuda and torch.is_grad_enabled()):
        return seq(x)
    if ENABLED and OWN_GEMM and x.dim() == 2 and x.dtype == torch.float32 and _chain_spec(mods) is not None:
        return mlp_chain(mods, x)
    i = 0
    while i < len(mods):
        m = mods[i]
        nxt = mods[i + 1] if i + 1 < len(mods) else None
        if isinstance(m, torch.nn.Linear) and isinstance(nxt, torch.nn.ELU) and m.bias is not None and x.dim() == 2:
            x = linear_elu(x, m.weight, m.bias, float(nxt.alpha))
            i += 2
        elif isinstance(m, torch.nn.Linear) and isinstance(nxt, torch.nn.ReLU) and m.bias is not None and x.dim() == 2:
            x = _LinearRelu.apply(x, m.weight, m.bias)
            i += 2
        elif isinstance(m, torch.nn.Linear):
            x = plain_linear(m, x)
            i += 1
        else:
            x = m(x)
            i += 1
    return x


ELEMENTWISE_ACTIVATIONS = (torch.nn.ELU, torch.nn.ReLU, torch.nn.Tanh, torch.nn.SELU, torch.nn.LeakyReLU, torch.nn.Sigmoid, torch.nn.GELU, torch.nn.SiLU,
                           torch.nn.Identity, torch.nn.Flatten)


def recordable(*nets):
    """May a training step of these networks be recorded into a hipGraph?  Only when every trainable parameter's gradient comes out of
    OUR kernels: torch's batch reductions (the bias gradient of F.linear / conv: `sum(0)`) go stale or unwritten under hipGraph replay on
    this ROCm (profiles/r2_hipgraph_stale_reductions.md).  Whitelist: nn.Linear WITH a bias inside nn.Sequential stacks run through
    `mlp_forward` / `mlp_chain` (any width, any elementwise activation after it), the history encoder's Conv1d layers (evaluated as
    window-gather + `linear_elu` when their activation is ELU), bare parameters (`std`).  Anything else -> stay eager."""
    if not (ENABLED and OWN_GEMM):
        return False
    from quadrupedal_agility_amd.rsl_rl.modules.actor_critic import StateHistoryEncoder
    for net in nets:
        if net is None:
            continue
        for mod in net.modules():
            own = list(mod.parameters(recurse=False))
            if not any(p.requires_grad for p in own):
                continue
            if isinstance(mod, torch.nn.Linear):
                if mod.bias is None:
                    return False
            elif isinstance(mod, torch.nn.Conv1d):
                pass          # only inside StateHistoryEncoder (checked below)
            elif own and not all(p.dim() <= 1 for p in own):
                return False
        for mod in net.modules():
            if isinstance(mod, StateHistoryEncoder) and not isinstance(mod.activation_fn, torch.nn.ELU) and any(p.requires_grad for p in mod.parameters()):
                return False
            if isinstance(mod, torch.nn.Conv1d) and not any(isinstance(h, StateHistoryEncoder) and mod in list(h.modules()) for h in net.modules()):
                if any(p.requires_grad for p in mod.parameters()):
                    return False
            if isinstance(mod, (torch.nn.GRU, torch.nn.LSTM, torch.nn.Conv2d, torch.nn.BatchNorm1d, torch.nn.BatchNorm2d, torch.nn.LayerNorm)) and \
                    any(p.requires_grad for p in mod.parameters()):
                return False
    return True


# autograd nodes whose backward is one of torch's batch reductions (bias gradient of F.linear / addmm / conv, broadcast gradients, loss means,
# batch-norm statistics): exactly what must not end up in a recorded step (profiles/r2_hipgraph_stale_reductions.md)
_TORCH_BATCH_REDUCTION_NODES = ("AddmmBackward", "LinearBackward", "SumBackward", "MeanBackward", "ConvolutionBackward", "NativeBatchNormBackward",
                                "MiopenBatchNormBackward", "CudnnBatchNormBackward", "NormBackward", "LinalgVectorNormBackward", "NativeLayerNormBackward")


def assert_recordable_graph(outputs, what="step"):
    """`recordable()` judges the MODULES; this judges the autograd graph a step actually built (a bare `self.fc(x)` in a custom
    network passes the module whitelist and still puts torch's `sum(0)` into the recording).  Called while a step is being captured, on the
    tensors whose gradients start the backward pass; raises -- the capture blocks then keep the step eager -- when a node of the graph is one
    of torch's batch reductions."""
    seen, stack, bad = set(), [t.grad_fn for t in outputs if torch.is_tensor(t) and t.grad_fn is not None], []
    while stack:
        fn = stack.pop()
        if fn is None or fn in seen:
            continue
        seen.add(fn)
        name = type(fn).__name__
        if name.startswith(_TORCH_BATCH_REDUCTION_NODES):
            bad.append(name)
        stack.extend(f for f, _ in fn.next_functions)
    if bad:
        raise RuntimeError(f"{what}: the autograd graph contains torch batch reductions {sorted(set(bad))}; not recorded (they go stale under hipGraph replay)")


def ppo_loss(mu, std, value, actions, old_logp, old_mu, old_sigma, advantages, returns, target_values, *, clip,
             c_surr, c_value, c_bound, c_entropy, clipped_value=True):
    return _PpoLoss.apply(mu, std, value, actions, old_logp, old_mu, old_sigma, advantages, returns, target_values,
                          clip, c_surr, c_value, c_bound, c_entropy, clipped_value)


def ppo_loss_reference(mu, std, value, actions, old_logp, old_mu, old_sigma, advantages, returns, target_values, *, clip,
                       c_surr, c_value, c_bound, c_entropy, clipped_value=True):
    """The same objective as eager PyTorch ops, written as gail.py:333-403 writes it (any device).  Returns
    (loss, stats[6]) with stats = [loss, surrogate, value, bound, entropy, kl]."""
    from torch.distributions import Normal
    dist = Normal(mu, mu * 0.0 + std, validate_args=False)
    logp = dist.log_prob(actions).sum(dim=-1)
    sigma = dist.stddev
    entropy = dist.entropy().sum(dim=-1)
    with torch.no_grad():
        kl = torch.sum(torch.log(sigma / old_sigma + 1.0e-5) +
                       (torch.square(old_sigma) + torch.square(old_mu - mu)) / (2.0 * torch.square(sigma)) - 0.5, dim=-1).mean()
    adv = torch.squeeze(advantages)
    ratio = torch.exp(logp - torch.squeeze(old_logp))
    surrogate = torch.max(-adv * ratio, -adv * torch.clamp(ratio, 1.0 - clip, 1.0 + clip)).mean()
    if clipped_value:
        v_clip = target_values + (value - target_values).clamp(-clip, clip)
        value_loss = torch.max((value - returns).pow(2), (v_clip - returns).pow(2)).mean()
    else:
        value_loss = (returns - value).pow(2).mean()
    b_loss = (torch.clamp(mu + 1.0, max=0.0) ** 2 + torch.clamp(mu - 1.0, min=0.0) ** 2).sum(dim=-1).mean()
    ent = entropy.mean()
    loss = c_surr * surrogate + c_value * value_loss + c_bound * b_loss - c_entropy * ent
    return loss, torch.stack([loss.detach(), surrogate.detach(), value_loss.detach(), b_loss.detach(), ent.detach(), kl])


def normalizer_update(mean, var, count, batches):
    """RunningMeanStd update of the device-resident double moments with 1..4 fp32 batches, one launch (qa_normalizer_update)."""
    lib = _capi.load_library()
    bs = [_f32c(b.detach()) for b in batches]
    k, d = len(bs), mean.shape[0]
    assert 1 <= k <= 4 and all(b.dim() == 2 and b.shape[1] == d for b in bs) and mean.dtype == torch.float64
    ptrs = (C.c_void_p * k)(*[b.data_ptr() for b in bs])
    rows = (C.c_int64 * k)(*[b.shape[0] for b in bs])
    stream = C.c_void_p(torch.cuda.current_stream(mean.device).cuda_stream)
    rc = lib.qa_normalizer_update(ptrs, rows, k, d, _ptr(mean), _ptr(var), _ptr(count), stream)
    if rc != 0:
        raise RuntimeError(f"qa_normalizer_update failed with code {rc}: {lib.qa_last_error().decode()}")


def normalizer_apply(x, mean, var, epsilon, clip):
    lib = _capi.load_library()
    xc = _f32c(x)
    y = torch.empty_like(xc)
    d = mean.shape[0]
    assert xc.shape[-1] == d
    stream = C.c_void_p(torch.cuda.current_stream(xc.device).cuda_stream)
    rc = lib.qa_normalizer_apply(_ptr(xc), _ptr(y), xc.numel() // d, d, _ptr(mean), _ptr(var), float(epsilon), float(clip), stream)
    if rc != 0:
        raise RuntimeError(f"qa_normalizer_apply failed with code {rc}: {lib.qa_last_error().decode()}")
    return y


PAIR_ROW_L2, PAIR_MSE = 0, 1


class _PairLoss(torch.autograd.Function):
    """loss = f(a, b) with b constant: mean row L2 norm of a - b (PAIR_ROW_L2) or mean squared difference (PAIR_MSE).  Loss and
    d loss / d a come out of one pass (qa_pair_loss); backward only scales.  b may be a column slice of wider rows."""

    @staticmethod
    def forward(ctx, a, b, mode):
        a = _f32c(a)
        assert b.dtype == torch.float32 and b.stride(1) == 1 and a.shape == b.shape
        rows, cols = a.shape
        lib = _capi.load_library()
        grad = torch.empty_like(a)
        out = torch.empty((), dtype=torch.float32, device=a.device)
        nscratch = int(lib.qa_pair_loss_scratch_bytes(rows))
        scratch = torch.empty(nscratch, dtype=torch.uint8, device=a.device)
        rc = lib.qa_pair_loss(_ptr(a), _ptr(b), rows, cols, b.stride(0), int(mode), _ptr(grad), _ptr(out), _ptr(scratch), nscratch,
                              C.c_void_p(torch.cuda.current_stream(a.device).cuda_stream))
        if rc != 0:
            raise RuntimeError(f"qa_pair_loss failed with code {rc}: {lib.qa_last_error().decode()}")
        ctx.save_for_backward(grad)
        return out

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return grad * g, None, None


def pair_loss(a, b, mode):
    return _PairLoss.apply(a, b, mode)


def pair_loss_raw(a, b, mode):
    """qa_pair_loss without the autograd wrapper: (loss (0-d), d loss / d a)"""
    a = _f32c(a.detach())
    assert b.dtype == torch.float32 and b.stride(1) == 1 and a.shape == b.shape
    rows, cols = a.shape
    lib = _capi.load_library()
    grad = torch.empty_like(a)
    out = torch.empty((), dtype=torch.float32, device=a.device)
    nscratch = int(lib.qa_pair_loss_scratch_bytes(rows))
    scratch = torch.empty(nscratch, dtype=torch.uint8, device=a.device)
    rc = lib.qa_pair_loss(_ptr(a), _ptr(b), rows, cols, b.stride(0), int(mode), _ptr(grad), _ptr(out), _ptr(scratch), nscratch,
                          C.c_void_p(torch.cuda.current_stream(a.device).cuda_stream))
    if rc != 0:
        raise RuntimeError(f"qa_pair_loss failed with code {rc}: {lib.qa_last_error().decode()}")
    return out, grad


_pair_scratch = {}


def pair_losses_raw(jobs):
    """several qa_pair_loss problems in ONE launch (qa_pair_losses, ABI 18).  jobs: [(a, b, mode, grad_scale)] with grad_scale a 0-d ROCm tensor or None;
    -> [(loss (0-d, unscaled), d loss / d a times grad_scale)].  The launches of one call site must follow each other on one stream (they share the
    arrival counter's word)."""
    lib = _capi.load_library()
    n = len(jobs)
    recs = (_capi.QaPairJob * n)()
    outs, keep = [], []
    dev = jobs[0][0].device
    for i, (a, b, mode, scale) in enumerate(jobs):
        a = _f32c(a.detach())
        assert b.dtype == torch.float32 and b.stride(1) == 1 and a.shape == b.shape
        grad = torch.empty_like(a)
        out = torch.empty((), dtype=torch.float32, device=dev)
        if scale is not None:
            assert torch.is_tensor(scale) and scale.dtype == torch.float32 and scale.numel() == 1 and scale.device == dev
        r = recs[i]
        r.a, r.b, r.rows, r.cols, r.mode, r.b_stride = a.data_ptr(), b.data_ptr(), a.shape[0], a.shape[1], int(mode), b.stride(0)
        r.grad_scale, r.grad_a, r.out = (scale.data_ptr() if scale is not None else None), grad.data_ptr(), out.data_ptr()
        outs.append((out, grad)); keep.append(a)
    nb = int(lib.qa_pair_losses_scratch_bytes(C.cast(recs, C.c_void_p), n))
    key = (dev, nb)
    sc = _pair_scratch.get(key)
    if sc is None:
        sc = _pair_scratch[key] = torch.zeros(nb, dtype=torch.uint8, device=dev)          # the arrival counter: zeroed ONCE
    _check(lib.qa_pair_losses(C.cast(recs, C.c_void_p), n, _ptr(sc), nb, C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)), "qa_pair_losses")
    return outs


def accumulate_scalars(acc, scalars):
    """acc[i] += scalars[i] (0-d / 1-element fp32 ROCm tensors, possibly views of different kernels' outputs) in ONE launch (qa_accumulate_scalars) instead of
    torch.stack + add_; falls back to those for anything else"""
    n = len(scalars)
    if not (ENABLED and acc.is_cuda and acc.dtype == torch.float32 and acc.is_contiguous() and acc.numel() >= n and n <= 16
            and all(torch.is_tensor(x) and x.dtype == torch.float32 and x.numel() == 1 and x.device == acc.device for x in scalars)):
        acc[:n].add_(torch.stack([x.reshape(()) for x in scalars]))
        return
    lib = _capi.load_library()
    _check(lib.qa_accumulate_scalars(_ptr(acc), (C.c_void_p * n)(*[x.data_ptr() for x in scalars]), n, C.c_void_p(torch.cuda.current_stream(acc.device).cuda_stream)),
           "qa_accumulate_scalars")


def gather_rows(idx, srcs, dsts=None, block_dev=None):
    """[src[idx] for src in srcs] for row-major fp32 tensors (N, w) in ONE launch; idx (rows) int64 on the device.  `dsts`
    (dense (rows, w) tensors) are allocated when not given.  With block_dev (0-d int64 device tensor) idx is a (blocks, rows)
    table and row block *block_dev is used."""
    rows = idx.shape[-1]
    n = len(srcs)
    flat = [x if x.dim() == 2 else x.reshape(x.shape[0], -1) for x in srcs]
    assert idx.dtype == torch.int64 and idx.is_contiguous() and all(x.dtype == torch.float32 and x.stride(1) == 1 for x in flat)
    if dsts is None:
        dsts = [torch.empty(rows, x.shape[1], dtype=torch.float32, device=idx.device) for x in flat]
    lib = _capi.load_library()
    src = (C.c_void_p * n)(*[x.data_ptr() for x in flat])
    dst = (C.c_void_p * n)(*[d.data_ptr() for d in dsts])
    strides = (C.c_int64 * n)(*[x.stride(0) for x in flat])
    widths = (C.c_int32 * n)(*[x.shape[1] for x in flat])
    rc = lib.qa_gather_rows(_ptr(idx), _ptr(block_dev) if block_dev is not None else None, rows, n, src, strides, widths, dst,
                            C.c_void_p(torch.cuda.current_stream(idx.device).cuda_stream))
    if rc != 0:
        raise RuntimeError(f"qa_gather_rows failed with code {rc}: {lib.qa_last_error().decode()}")
    return dsts


def kl_lr_rule(kl, desired_kl, lr, factor=1.5, lr_min=1e-5, lr_max=1e-2):
    """lr (0-d ROCm tensor) <- the KL-adaptive rule of gail.py:367-379 applied to kl (0-d ROCm tensor); one launch"""
    lib = _capi.load_library()
    rc = lib.qa_kl_lr_rule(_ptr(kl), float(desired_kl), float(factor), float(lr_min), float(lr_max), _ptr(lr),
                           C.c_void_p(torch.cuda.current_stream(lr.device).cuda_stream))
    if rc != 0:
        raise RuntimeError(f"qa_kl_lr_rule failed with code {rc}: {lib.qa_last_error().decode()}")


class ClipAdam:
    """clip_grad_norm_(max_norm) + optimizer.step() of ONE torch.optim.Adam as three launches (qa_clip_adam_step).

    The torch optimizer stays the owner of the state (`exp_avg`, `exp_avg_sq`, `step` tensors), so `state_dict()` /
    `load_state_dict()` and with them model.pt are unchanged; this object only replaces the arithmetic of a step.  It
    falls back to the PyTorch calls whenever something is not as the kernel needs it (CPU tensors, state not created
    yet, groups with different learning rates, non-contiguous gradients)."""

    CHUNK = 2048
    SMALL_CHUNK = 32
    WIDE_PARTS = 16         # QA_REDUCE_WIDE of csrc/qa_learner.hip

    def __init__(self, optimizer, max_norm=None):
        self.opt = optimizer
        self.max_norm = float(max_norm) if max_norm else 0.0
        self._tab = None

    def _torch_step(self, params):
        if self.max_norm > 0:
            torch.nn.utils.clip_grad_norm_(params, self.max_norm)
        self.opt.step()

    def _build(self, items):
        opt = self.opt
        dev = items[0][0].device
        g0 = opt.param_groups[0]
        i64 = lambda xs: torch.tensor(xs, dtype=torch.int64, device=dev)
        st = [opt.state[p] for p, _ in items]
        ct, cs, cl = [], [], []
        for t, (p, _) in enumerate(items):
            n = p.numel()
            # small tensors (biases, std, heads) in 32-element chunks: a gradient that arrives as MANY parts (the 384 row-block column sums
            # of a bias gradient, qa_clip_adam_step_reduce) is added by the eight thread rows of a workgroup over a chunk of <= 32 elements
            chunk = self.SMALL_CHUNK if n <= self.CHUNK else self.CHUNK
            for s0 in range(0, n, chunk):
                ct.append(t); cs.append(s0); cl.append(min(chunk, n - s0))
        i32 = lambda xs: torch.tensor(xs, dtype=torch.int32, device=dev)
        tab = dict(items=[p for p, _ in items], n=len(items), params=i64([p.data_ptr() for p, _ in items]),
                   exp_avg=i64([s["exp_avg"].data_ptr() for s in st]), exp_avg_sq=i64([s["exp_avg_sq"].data_ptr() for s in st]),
                   steps=i64([s["step"].data_ptr() for s in st]), state_refs=st, exp_avg_ptrs=[s["exp_avg"].data_ptr() for s in st],
                   grads=torch.zeros(len(items), dtype=torch.int64, device=dev),
                   grads_host=torch.zeros(len(items), dtype=torch.int64).pin_memory(), grad_ptrs=None,
                   chunk_tensor=i32(ct), chunk_start=i32(cs), chunk_len=i32(cl), num_chunks=len(ct),
                   wd=torch.tensor([g["weight_decay"] for _, g in items], dtype=torch.float32, device=dev),
                   scratch=torch.zeros(5 + len(ct), dtype=torch.float32, device=dev),      # [4 + chunks]: the single-launch (no clipping) step's arrival counter
                   betas=g0["betas"], eps=g0["eps"], lr_dev=None, lr_val=None)
        self._tab = tab

    def _prepare(self, for_pair=False):
        """-> None (nothing to step), False (the PyTorch calls must take this step) or dict(tab, params, ptrs, pend, lr_dev, inline)"""
        opt = self.opt
        items = [(p, g) for g in opt.param_groups for p in g["params"] if p.grad is not None]
        params = [p for p, _ in items]
        if not items:
            return None
        g0 = opt.param_groups[0]
        ok = (ENABLED and params[0].is_cuda and not g0.get("amsgrad", False) and not g0.get("maximize", False)
              and all(("exp_avg" in opt.state.get(p, {})) and torch.is_tensor(opt.state[p]["step"]) and opt.state[p]["step"].is_cuda for p in params)
              and all(g["lr"] is g0["lr"] or (not torch.is_tensor(g["lr"]) and not torch.is_tensor(g0["lr"]) and g["lr"] == g0["lr"]) for _, g in items)
              and all(g["betas"] == g0["betas"] and g["eps"] == g0["eps"] for _, g in items)
              and all(p.is_contiguous() and p.grad.is_contiguous() and p.dtype == torch.float32 and p.grad.dtype == torch.float32 for p in params))
        # gradients still in parts (deferred_grad_finishes): added by this step's first pass when the kernel can take them, else finished now
        pend = [_PENDING.get(p.data_ptr()) for p in params] if _PENDING else None
        if pend is not None and any(e is not None for e in pend):
            can = (ok and self.max_norm > 0 and len(params) <= 64 and
                   all(e is None or e[2] <= self.WIDE_PARTS or p.numel() <= self.CHUNK for p, e in zip(params, pend)))
            if not can:
                if for_pair:
                    return False                # (the caller steps the two optimisers one after the other: that path finishes the parts it cannot take)
                flush_pending_grads(params)
                pend = None
        else:
            pend = None
        if not ok:
            return False
        tab = self._tab
        if (tab is None or len(tab["items"]) != len(params) or any(a is not b for a, b in zip(tab["items"], params))
                or any(opt.state[p] is not s or s["exp_avg"].data_ptr() != e for p, s, e in zip(params, tab["state_refs"], tab["exp_avg_ptrs"]))):
            self._build(items)          # first use, or load_state_dict() replaced the state tensors
        t = self._tab
        ptrs = [p.grad.data_ptr() for p in params]
        lr = g0["lr"]
        if torch.is_tensor(lr):
            lr_dev = lr if lr.dtype == torch.float32 else None
            if lr_dev is None:
                return False
        else:
            if t["lr_dev"] is None:
                t["lr_dev"] = torch.zeros((), dtype=torch.float32, device=params[0].device)
            if t["lr_val"] != float(lr):
                t["lr_dev"].fill_(float(lr)); t["lr_val"] = float(lr)
            lr_dev = t["lr_dev"]
        return dict(tab=t, params=params, ptrs=ptrs, pend=pend, lr_dev=lr_dev, inline=len(ptrs) <= 64)      # QA_ADAM_MAX_INLINE: the pointers ride in the kernel arguments

    def step(self):
        pre = self._prepare()
        if pre is None:
            return
        if pre is False:
            opt = self.opt
            return self._torch_step([p for g in opt.param_groups for p in g["params"] if p.grad is not None])
        t, params, ptrs, pend, lr_dev, inline = pre["tab"], pre["params"], pre["ptrs"], pre["pend"], pre["lr_dev"], pre["inline"]
        if not inline and torch.cuda.is_current_stream_capturing():
            # The table path uploads ONE shared pinned host table; a captured copy of it reads the host memory at REPLAY time, so every
            # recording but the last would step with the last recording's (or freed) gradient addresses.  Refuse: the
            # callers' capture blocks catch this and keep the step eager.
            raise RuntimeError(f"ClipAdam: {len(ptrs)} gradient tensors (> 64) cannot ride in the kernel arguments, and the pointer-table "
                               "upload cannot be recorded into a hipGraph; this step stays eager")
        if not inline and ptrs != t["grad_ptrs"]:       # autograd allocated new gradient tensors: refresh the pointer table
            if t.get("copied") is not None:
                t["copied"].synchronize()               # the pinned staging buffer may still be waiting for its last async copy
            t["grads_host"].copy_(torch.tensor(ptrs, dtype=torch.int64))
            t["grads"].copy_(t["grads_host"], non_blocking=True)
            t["copied"] = torch.cuda.Event()
            t["copied"].record()
            t["grad_ptrs"] = ptrs
        lib = _capi.load_library()
        stream = C.c_void_p(torch.cuda.current_stream(params[0].device).cuda_stream)
        if pend is not None:
            n = len(params)
            src = (C.c_void_p * n)(*[(e[1].data_ptr() if e is not None else 0) for e in pend])
            stride = (C.c_int64 * n)(*[(e[3] if e is not None else 0) for e in pend])
            parts = (C.c_int32 * n)(*[(e[2] if e is not None else 0) for e in pend])
            rc = lib.qa_clip_adam_step_reduce(_ptr(t["params"]), (C.c_void_p * n)(*ptrs), _ptr(t["exp_avg"]), _ptr(t["exp_avg_sq"]), _ptr(t["steps"]), t["n"],
                                              _ptr(t["chunk_tensor"]), _ptr(t["chunk_start"]), _ptr(t["chunk_len"]), t["num_chunks"], _ptr(t["wd"]),
                                              _ptr(lr_dev), float(t["betas"][0]), float(t["betas"][1]), float(t["eps"]), self.max_norm,
                                              _ptr(t["scratch"]), t["scratch"].numel(), src, stride, parts, stream)
            for p in params:
                e = _PENDING.pop(p.data_ptr(), None)
                if e is not None:
                    _retire(e[1])
            if rc != 0:
                raise RuntimeError(f"qa_clip_adam_step_reduce failed with code {rc}: {lib.qa_last_error().decode()}")
            return
        fn, garg = (lib.qa_clip_adam_step_hostgrads, (C.c_void_p * len(ptrs))(*ptrs)) if inline else (lib.qa_clip_adam_step, _ptr(t["grads"]))
        rc = fn(_ptr(t["params"]), garg, _ptr(t["exp_avg"]), _ptr(t["exp_avg_sq"]), _ptr(t["steps"]), t["n"],
                                   _ptr(t["chunk_tensor"]), _ptr(t["chunk_start"]), _ptr(t["chunk_len"]), t["num_chunks"], _ptr(t["wd"]),
                                   _ptr(lr_dev), float(t["betas"][0]), float(t["betas"][1]), float(t["eps"]), self.max_norm,
                                   _ptr(t["scratch"]), t["scratch"].numel(), stream)
        if rc != 0:
            raise RuntimeError(f"qa_clip_adam_step failed with code {rc}: {lib.qa_last_error().decode()}")


class ClipAdamPair:
    """`first.step(); kl rule on second's learning rate; second.step()` of two ClipAdam objects as THREE launches instead of seven
    (qa_clip_adam_pair_step, ABI 18): the tail of a PPO minibatch step (gail.py:359-362, 367-379, 405-408).  `step()` returns False when the pair launch
    cannot take the step (state not created yet, a fallback condition of either optimiser, more than 64 tensors together): the caller then steps them one
    after the other as before."""

    def __init__(self, first, second):
        self.first, self.second = first, second
        self._tab = None

    def warm(self):
        """build the merged tables now (gradients must exist, nothing is launched): call before recording a step whose first paired launch would
        otherwise happen inside the capture"""
        return self.step(dry=True)

    def step(self, kl=None, desired_kl=0.0, factor=1.5, lr_min=1e-5, lr_max=1e-2, dry=False):
        a, b = self.first, self.second
        self.why_not = None
        if not (ENABLED and a.max_norm > 0 and b.max_norm > 0):
            self.why_not = "an optimiser without clipping"
            return False
        ga, gb = a.opt.param_groups[0], b.opt.param_groups[0]
        if ga["betas"] != gb["betas"] or ga["eps"] != gb["eps"] or not (torch.is_tensor(gb["lr"]) and gb["lr"].dtype == torch.float32 and gb["lr"].is_cuda):
            self.why_not = "different betas / eps, or the second learning rate is not a device scalar"
            return False
        if kl is not None and not (torch.is_tensor(kl) and kl.is_cuda and kl.dtype == torch.float32 and kl.numel() == 1):
            self.why_not = "kl is not a device fp32 scalar"
            return False
        pa = a._prepare(for_pair=True)
        if not pa:
            self.why_not = f"the first optimiser cannot take the fused step ({pa})"
            return False
        pb = b._prepare(for_pair=True)
        if not pb or len(pa["ptrs"]) + len(pb["ptrs"]) > 64:
            self.why_not = f"the second optimiser cannot take the fused step ({pb if not pb else 'more than 64 tensors together'})"
            return False
        ta, tb = pa["tab"], pb["tab"]
        m = self._tab
        if m is None or m["ta"] is not ta or m["tb"] is not tb:
            if torch.cuda.is_current_stream_capturing():
                # the merged tables are a dozen small device copies: recorded, they would replay with every step (measured: 12 launches, 56 us of a
                # task-level step).  `warm()` builds them before a capture; without it this recording keeps the two steps apart
                self.why_not = "the merged tables do not exist yet and a capture is running (call warm() first)"
                return False
            cat = lambda k: torch.cat([ta[k], tb[k]])
            m = self._tab = dict(ta=ta, tb=tb, params=cat("params"), exp_avg=cat("exp_avg"), exp_avg_sq=cat("exp_avg_sq"), steps=cat("steps"),
                                 chunk_tensor=torch.cat([ta["chunk_tensor"], tb["chunk_tensor"] + ta["n"]]), chunk_start=cat("chunk_start"), chunk_len=cat("chunk_len"),
                                 wd=cat("wd"), n=ta["n"] + tb["n"], num_chunks=ta["num_chunks"] + tb["num_chunks"],
                                 scratch=torch.zeros(ta["num_chunks"] + tb["num_chunks"] + 9, dtype=torch.float32, device=ta["params"].device))
        if dry:
            return True
        ptrs = pa["ptrs"] + pb["ptrs"]
        n = len(ptrs)
        pend = (pa["pend"] or [None] * len(pa["ptrs"])) + (pb["pend"] or [None] * len(pb["ptrs"]))
        src = (C.c_void_p * n)(*[(e[1].data_ptr() if e is not None else 0) for e in pend])
        stride = (C.c_int64 * n)(*[(e[3] if e is not None else 0) for e in pend])
        parts = (C.c_int32 * n)(*[(e[2] if e is not None else 0) for e in pend])
        pair = _capi.QaAdamPair(ta["n"], ta["num_chunks"], pb["lr_dev"].data_ptr(), b.max_norm, kl.data_ptr() if kl is not None else None,
                                float(desired_kl), float(factor), float(lr_min), float(lr_max))
        lib = _capi.load_library()
        params = pa["params"] + pb["params"]
        rc = lib.qa_clip_adam_pair_step(_ptr(m["params"]), (C.c_void_p * n)(*ptrs), _ptr(m["exp_avg"]), _ptr(m["exp_avg_sq"]), _ptr(m["steps"]), m["n"],
                                        _ptr(m["chunk_tensor"]), _ptr(m["chunk_start"]), _ptr(m["chunk_len"]), m["num_chunks"], _ptr(m["wd"]),
                                        _ptr(pa["lr_dev"]), float(ta["betas"][0]), float(ta["betas"][1]), float(ta["eps"]), a.max_norm,
                                        _ptr(m["scratch"]), m["scratch"].numel(), src, stride, parts, C.byref(pair),
                                        C.c_void_p(torch.cuda.current_stream(params[0].device).cuda_stream))
        for p in params:
            e = _PENDING.pop(p.data_ptr(), None)
            if e is not None:
                _retire(e[1])
        if rc != 0:
            raise RuntimeError(f"qa_clip_adam_pair_step failed with code {rc}: {lib.qa_last_error().decode()}")
        return True


class StackedAdam:
    """SEVERAL torch.optim.Adam optimisers that are stepped one after the other from the same gradients (the discriminator's three, gail.py:518-520,
    each over the trunk plus one head, gail.py:107-132: a trunk parameter has three sets of moments) as ONE launch (qa_adam_stack_step, ABI 18).  The
    launch also puts each gradient together -- parts of the loss's product, alpha * parts of the penalty's product, reg * weight -- and writes it to
    `.grad`, which is what qa_grad_reduce, two multi-tensor adds, an add and three optimiser launches did before.  The torch optimisers stay the owners
    of the state; `ready()` says whether the launch can take the step (else the caller steps them as before)."""

    def __init__(self, optimizers, lib=None, prefix="qa_"):
        self.opts = list(optimizers)
        self._lib, self._prefix = lib, prefix
        self._lr = {}
        self._ticket = None
        self._keep = None

    def _members(self):
        """param -> [(optimizer index, group)], optimiser order"""
        out = {}
        for k, o in enumerate(self.opts):
            for g in o.param_groups:
                for p in g["params"]:
                    out.setdefault(p, []).append((k, g))
        return out

    def ready(self):
        if not ENABLED or not self.opts or not all(type(o) is torch.optim.Adam for o in self.opts):
            return False
        g0 = self.opts[0].param_groups[0]
        mem = self._members()
        if len(mem) > _capi.ADAM_STACK_MAX_TENSORS or any(len(v) > _capi.ADAM_STACK_MAX_STATES for v in mem.values()):
            return False
        for o in self.opts:
            for g in o.param_groups:
                if g.get("amsgrad", False) or g.get("maximize", False) or g["betas"] != g0["betas"] or g["eps"] != g0["eps"]:
                    return False
                if torch.is_tensor(g["lr"]) and (g["lr"].dtype != torch.float32 or g["lr"].numel() != 1):
                    return False
                for p in g["params"]:
                    st = o.state.get(p, {})
                    if ("exp_avg" not in st or not torch.is_tensor(st["step"]) or st["step"].dtype != torch.float32 or st["step"].device != p.device
                            or p.dtype != torch.float32 or not p.is_contiguous()):
                        return False
        return True

    def _lr_dev(self, k, g, dev):
        lr = g["lr"]
        if torch.is_tensor(lr):
            return lr
        key = (k, id(g))
        e = self._lr.get(key)
        if e is None:
            e = self._lr[key] = [torch.zeros((), dtype=torch.float32, device=dev), None]
        if e[1] != float(lr):
            e[0].fill_(float(lr)); e[1] = float(lr)
        return e[0]

    def step(self, sources, reg=None):
        """sources: {param: dict(grad=, src1=, stride1=, parts1=, src2=, stride2=, parts2=, alpha2=, tmp=)} -- `grad` a contiguous fp32 tensor of the
        parameter's size (it becomes / is `.grad`), the rest optional (tensors for src / tmp: a parameter's first part starts at the tensor's first
        element); reg: {param: coefficient}.  Every parameter of the optimisers needs an entry."""
        mem = self._members()
        items = list(mem.items())
        n = len(items)
        dev = items[0][0].device
        if self._ticket is None:
            self._ticket = torch.zeros(1, dtype=torch.int32, device=dev)
        recs = (_capi.QaAdamStackTensor * n)()
        keep = []
        for i, (p, ms) in enumerate(items):
            src = sources[p]
            g = src["grad"]
            assert g.is_contiguous() and g.dtype == torch.float32 and g.numel() == p.numel()
            r = recs[i]
            r.param, r.grad, r.numel, r.num_states = p.data_ptr(), g.data_ptr(), p.numel(), len(ms)
            if src.get("parts1", 0):
                r.src1, r.stride1, r.parts1 = src["src1"].data_ptr(), int(src["stride1"]), int(src["parts1"])
            if src.get("parts2", 0):
                r.src2, r.stride2, r.parts2, r.alpha2, r.tmp = src["src2"].data_ptr(), int(src["stride2"]), int(src["parts2"]), float(src["alpha2"]), src["tmp"].data_ptr()
                assert src["tmp"].numel() >= p.numel()
            r.reg = float(reg.get(p, 0.0)) if reg else 0.0
            for s_, (k, grp) in enumerate(ms):
                st = self.opts[k].state[p]
                lr = self._lr_dev(k, grp, dev)
                e = r.state[s_]
                e.exp_avg, e.exp_avg_sq, e.step, e.lr, e.weight_decay = st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), st["step"].data_ptr(), lr.data_ptr(), float(grp["weight_decay"])
                keep.append(lr)
            p.grad = g
        self._keep = keep
        g0 = self.opts[0].param_groups[0]
        lib = self._lib or _capi.load_library()
        stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream) if dev.type == "cuda" else None
        rc = getattr(lib, self._prefix + "adam_stack_step")(C.cast(recs, C.c_void_p), n, float(g0["betas"][0]), float(g0["betas"][1]), float(g0["eps"]), _ptr(self._ticket), stream)
        if rc != 0:
            raise RuntimeError(f"adam_stack_step failed with code {rc}" + (f": {lib.qa_last_error().decode()}" if self._prefix == "qa_" else ""))


class _DiscLoss(torch.autograd.Function):
    """loss, stats = f(d (B,1), eps (B,1), c (B,5) | labels, policy latents): the head losses of the discriminator step
    (MSELoss variant) and their gradient in one pass (qa_disc_loss); backward only scales the stored gradients.
    stats = [loss, ss, info, disc, us, acc_lb, acc_pi, acc_exp, acc_ulb, pred_mean(5), 0, 0]."""

    @staticmethod
    def forward(ctx, d, eps, c, label_lb, policy_eps, policy_c, b_lb, b_pi, b_ulb, c_ss, info_coef_dev, c_disc, c_us):
        lib = _capi.load_library()
        dc, ec, cc = _f32c(d), _f32c(eps), _f32c(c)
        pe, pc = _f32c(policy_eps), _f32c(policy_c)
        lab = label_lb if (label_lb.dtype == torch.int64 and label_lb.is_contiguous()) else label_lb.contiguous().long()
        B = b_lb + b_pi + b_ulb
        assert dc.numel() == B and ec.numel() == B and cc.shape == (B, 5) and lab.numel() == b_lb and pe.numel() == b_pi and pc.shape == (b_pi, 5)
        assert info_coef_dev.is_cuda and info_coef_dev.dtype == torch.float32
        gd = torch.empty(B, dtype=torch.float32, device=dc.device); ge = torch.empty_like(gd); gc = torch.empty_like(cc)
        out = torch.empty(16, dtype=torch.float32, device=dc.device)
        n = int(lib.qa_disc_loss_scratch_bytes(B))
        scratch = torch.empty(n, dtype=torch.uint8, device=dc.device)
        stream = C.c_void_p(torch.cuda.current_stream(dc.device).cuda_stream)
        rc = lib.qa_disc_loss(_ptr(dc), _ptr(ec), _ptr(cc), _ptr(lab), _ptr(pe), _ptr(pc), b_lb, b_pi, b_ulb, float(c_ss), _ptr(info_coef_dev),
                              float(c_disc), float(c_us), _ptr(gd), _ptr(ge), _ptr(gc), _ptr(out), _ptr(scratch), n, stream)
        if rc != 0:
            raise RuntimeError(f"qa_disc_loss failed with code {rc}: {lib.qa_last_error().decode()}")
        ctx.save_for_backward(gd, ge, gc)
        ctx.shapes = (d.shape, eps.shape, c.shape)
        ctx.mark_non_differentiable(out)
        return out[0].clone(), out

    @staticmethod
    def backward(ctx, g_loss, _g_stats):
        gd, ge, gc = ctx.saved_tensors
        sd, se, sc = ctx.shapes
        return ((gd * g_loss).view(sd), (ge * g_loss).view(se), (gc * g_loss).view(sc)) + (None,) * 10


def disc_loss_raw(d, eps, c, label_lb, policy_eps, policy_c, b_lb, b_pi, b_ulb, *, c_ss, info_coef_dev, c_disc, c_us, from_logits=False):
    """qa_disc_loss without the autograd wrapper: (stats[16], d loss/d d, d loss/d eps, d loss/d c) shaped like d, eps, c.  from_logits (qa_disc_loss_logits,
    ABI 18): `c` holds the class LOGITS -- the softmax and its backward run inside the launch, the last result is d loss / d logits"""
    lib = _capi.load_library()
    dc, ec, cc = _f32c(d.detach()), _f32c(eps.detach()), _f32c(c.detach())
    pe, pc = _f32c(policy_eps), _f32c(policy_c)
    lab = label_lb if (label_lb.dtype == torch.int64 and label_lb.is_contiguous()) else label_lb.contiguous().long()
    B = b_lb + b_pi + b_ulb
    assert dc.numel() == B and ec.numel() == B and cc.shape == (B, 5) and lab.numel() == b_lb and pe.numel() == b_pi and pc.shape == (b_pi, 5)
    gd = torch.empty(B, dtype=torch.float32, device=dc.device); ge = torch.empty_like(gd); gc = torch.empty_like(cc)
    out = torch.empty(16, dtype=torch.float32, device=dc.device)
    n = int(lib.qa_disc_loss_scratch_bytes(B))
    scratch = torch.empty(n, dtype=torch.uint8, device=dc.device)
    fn = lib.qa_disc_loss_logits if from_logits else lib.qa_disc_loss
    rc = fn(_ptr(dc), _ptr(ec), _ptr(cc), _ptr(lab), _ptr(pe), _ptr(pc), b_lb, b_pi, b_ulb, float(c_ss), _ptr(info_coef_dev),
            float(c_disc), float(c_us), _ptr(gd), _ptr(ge), _ptr(gc), _ptr(out), _ptr(scratch), n,
            C.c_void_p(torch.cuda.current_stream(dc.device).cuda_stream))
    if rc != 0:
        raise RuntimeError(f"qa_disc_loss failed with code {rc}: {lib.qa_last_error().decode()}")
    return out, gd.view(d.shape), ge.view(eps.shape), gc.view(c.shape)


def disc_sample_prepare(tables, block_dev, srcs, eps_src, c_src, label_table, task_mask, frame_mult, task_weight_dev, normalizer, outs=None):
    """minibatch `block_dev` of the update's index tables, read from (labelled expert, policy ring, unlabelled expert) straight into the
    prepared (3 mb, dim) matrix + the policy rows' (eps, c) + the labelled rows' classes: qa_disc_sample_prepare, one launch.
    `tables` = (t_lb, t_pi, t_ulb) int64 (steps, mb); returns (x_all, eps, c, label)."""
    lib = _capi.load_library()
    dev = srcs[0].device
    mb = [t.shape[1] for t in tables]
    d = srcs[0].shape[1]
    assert all(x.dtype == torch.float32 and x.is_contiguous() and x.dim() == 2 and x.shape[1] == d for x in srcs)
    assert all(t.dtype == torch.int64 and t.is_contiguous() for t in tables) and label_table.dtype == torch.int64 and label_table.is_contiguous()
    cd = c_src.shape[-1]
    if outs is None:
        outs = (torch.empty(sum(mb), d, device=dev), torch.empty(mb[1], 1, device=dev), torch.empty(mb[1], cd, device=dev), torch.empty(mb[0], dtype=torch.int64, device=dev))
    x_all, eps, c, label = outs
    io = _capi.QaDiscSampleIo()
    for b in range(3):
        io.src[b] = srcs[b].data_ptr(); io.index[b] = tables[b].data_ptr(); io.rows[b] = mb[b]
    io.eps_src, io.c_src, io.eps_out, io.c_out = eps_src.data_ptr(), c_src.data_ptr(), eps.data_ptr(), c.data_ptr()
    io.label_src, io.label_out, io.block_dev = label_table.data_ptr(), label.data_ptr(), block_dev.data_ptr()
    mean = _ptr(normalizer.mean) if normalizer is not None else None
    var = _ptr(normalizer.var) if normalizer is not None else None
    _check(lib.qa_disc_sample_prepare(C.byref(io), d, cd, _ptr(task_mask), _ptr(frame_mult), _ptr(task_weight_dev) if task_weight_dev is not None else None, mean, var,
                                      float(normalizer.epsilon) if normalizer is not None else 0.0, float(normalizer.clip_obs) if normalizer is not None else 0.0,
                                      _ptr(x_all), C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)), "qa_disc_sample_prepare")
    return x_all, eps, c, label


_tail_scratch = {}


def disc_step_tail(head_stats, input_grad, weights, acc=None, step=None, prior=None, prior_soft_coef=0.0):
    """the 11 logged values of a discriminator step by qa_disc_step_tail (sums of squares of the penalty's input gradient and of the
    regularised weights + the head statistics), optionally added to `acc`, `step` bumped and the class prior's EMA stepped, in ONE launch"""
    lib = _capi.load_library()
    dev = head_stats.device
    sc = _tail_scratch.get(dev)
    if sc is None:
        sc = torch.zeros(int(lib.qa_disc_step_tail_scratch_bytes()), dtype=torch.uint8, device=dev)      # holds the arrival counter: zeroed ONCE
        _tail_scratch[dev] = sc
    g = _f32c(input_grad)
    ws = [w.detach() for w in weights]
    assert all(w.dtype == torch.float32 and w.is_contiguous() for w in ws)
    wp = (C.c_void_p * len(ws))(*[w.data_ptr() for w in ws])
    wn = (C.c_int64 * len(ws))(*[w.numel() for w in ws])
    out = torch.empty(11, dtype=torch.float32, device=dev)
    _check(lib.qa_disc_step_tail(_ptr(head_stats), _ptr(g), g.shape[0], g.numel() // g.shape[0], wp, wn, len(ws), _ptr(out), _ptr(acc) if acc is not None else None,
                                 _ptr(step) if step is not None else None, _ptr(prior) if prior is not None else None, prior.numel() if prior is not None else 0,
                                 float(prior_soft_coef), _ptr(sc), sc.numel(), C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)),
           "qa_disc_step_tail")
    return out


def disc_loss(d, eps, c, label_lb, policy_eps, policy_c, b_lb, b_pi, b_ulb, *, c_ss, info_coef_dev, c_disc, c_us):
    return _DiscLoss.apply(d, eps, c, label_lb, policy_eps, policy_c, b_lb, b_pi, b_ulb, c_ss, info_coef_dev, c_disc, c_us)


def disc_loss_reference(d, eps, c, label_lb, policy_eps, policy_c, b_lb, b_pi, b_ulb, *, c_ss, info_coef, c_disc, c_us):
    """The same head losses as eager PyTorch ops, as gail.py:452-520 writes them (any device)."""
    import torch.nn.functional as F
    pred_c_lb = c[:b_lb]
    logits_pi, e_pi, pred_c = d[b_lb:b_lb + b_pi], eps[b_lb:b_lb + b_pi], c[b_lb:b_lb + b_pi]
    logits_exp, pred_c_ulb = d[b_lb + b_pi:], c[b_lb + b_pi:]
    ss = F.cross_entropy(pred_c_lb, label_lb)
    info = torch.mean(-torch.sum(pred_c_ulb * torch.log(pred_c_ulb + 1e-20), dim=-1))
    disc = 0.5 * (F.mse_loss(logits_pi, -torch.ones_like(logits_pi)) + F.mse_loss(logits_exp, torch.ones_like(logits_exp)))
    us = F.l1_loss(e_pi, policy_eps.view_as(e_pi))
    loss = c_ss * ss + info_coef * info + c_disc * disc + c_us * us
    with torch.no_grad():
        acc = [(torch.argmax(pred_c_lb, -1) == label_lb).float().mean(), (logits_pi < 0).float().mean(), (logits_exp > 0).float().mean(),
               (torch.argmax(pred_c, -1) == torch.argmax(policy_c, -1)).float().mean()]
        stats = torch.stack([loss.detach(), ss.detach(), info.detach(), disc.detach(), us.detach()] + acc)
        stats = torch.cat([stats, pred_c_ulb.mean(0), torch.zeros(2, device=d.device)])
    return loss, stats


def disc_prepare(batches, task_mask, frame_mult, task_weight_dev, normalizer=None):
    """[x_0; x_1; x_2] -> one (sum rows, dim) matrix with the discriminator's task weighting, frame weighting and (optionally)
    the running-moment normalisation + clip applied: one launch (qa_disc_prepare)."""
    lib = _capi.load_library()
    bs = [_f32c(b.detach()) for b in batches]
    k, d = len(bs), bs[0].shape[1]
    assert 1 <= k <= 3 and all(b.dim() == 2 and b.shape[1] == d for b in bs)
    ptrs = (C.c_void_p * k)(*[b.data_ptr() for b in bs])
    rows = (C.c_int64 * k)(*[b.shape[0] for b in bs])
    out = torch.empty(sum(b.shape[0] for b in bs), d, dtype=torch.float32, device=bs[0].device)
    stream = C.c_void_p(torch.cuda.current_stream(out.device).cuda_stream)
    mean = _ptr(normalizer.mean) if normalizer is not None else None
    var = _ptr(normalizer.var) if normalizer is not None else None
    rc = lib.qa_disc_prepare(ptrs, rows, k, d, _ptr(task_mask), _ptr(frame_mult), _ptr(task_weight_dev) if task_weight_dev is not None else None,
                             mean, var, float(normalizer.epsilon) if normalizer is not None else 0.0,
                             float(normalizer.clip_obs) if normalizer is not None else 0.0, _ptr(out), stream)
    if rc != 0:
        raise RuntimeError(f"qa_disc_prepare failed with code {rc}: {lib.qa_last_error().decode()}")
    return out


_GAE_SCRATCH = {}


def gae(rewards, values, dones, last_values, returns, advantages, gamma, lam, normalize=True):
    """qa_gae on (T, N[, 1]) ROCm tensors without an engine handle (the task-level learner has no simulator of its own)."""
    T, N = rewards.shape[0], rewards.shape[1]
    for x in (rewards, values, dones, last_values, returns, advantages):
        assert x.is_cuda and x.is_contiguous()
    assert dones.dtype == torch.uint8
    scratch = _GAE_SCRATCH.get(rewards.device)
    if scratch is None:
        scratch = _GAE_SCRATCH[rewards.device] = torch.zeros(1024, dtype=torch.float32, device=rewards.device)
    lib = _capi.load_library()
    rc = lib.qa_gae(_ptr(rewards), _ptr(values), _ptr(dones), _ptr(last_values), _ptr(returns), _ptr(advantages), T, N, float(gamma),
                    float(lam), int(bool(normalize)), _ptr(scratch), C.c_void_p(torch.cuda.current_stream(rewards.device).cuda_stream))
    if rc != 0:
        raise RuntimeError(f"qa_gae failed with code {rc}: {lib.qa_last_error().decode()}")


class PolicyChain:
    """SSInfoGAIL.act's network half -- Estimator.forward, the privileged-latent encoder, the actor trunk + head and the
    critic trunk + head (gail.py:176-197) -- as ONE launch: `qa_mlp_forward` carries each 16-row tile through every
    layer with the activations in LDS.  `describe()` turns the modules into the op list of include/qa_sim.h; `pack()`
    repacks the weights (once per rollout: they only change in update()); `forward(obs)` -> (mean (N,A), value (N,1)).

    Both actor variants are described (privileged encoder; history encoder with its Conv1d stack as structured layers), and
    `describe_discriminator` covers Discriminator.forward for the rollout's reward."""

    BUF_COLS = _capi.MLP_BUF_COLS

    def __init__(self, ops, params, num_actions, out_widths=None):
        self.n_ops = len(ops)
        self.ops = (_capi.QaMlpOp * self.n_ops)(*ops)
        self.params = params                      # per op: (weight, bias) tensors (or callables building them) or None
        self.num_actions = num_actions
        self.out_widths = list(out_widths) if out_widths is not None else [num_actions, 1]
        self.packed = None
        self._out = {}

    # ---------------------------------------------------------------- description (host logic, no device access)
    @staticmethod
    def k_blocks(k, n):
        """16-wide k-blocks a layer is stored with (qa_policy.hip mlp_kb): rounded up to the kernel's prefetch depth"""
        per = ((n + 15) // 16 + 7) // 8
        pf = 4 if per > 2 else 8
        return ((k + 15) // 16 + pf - 1) // pf * pf

    @staticmethod
    def _linears(seq):
        """[(Linear, elu?)] of an nn.Sequential of Linear / ELU(alpha=1) modules, None for anything else"""
        import torch.nn as nn
        mods = list(seq) if isinstance(seq, nn.Sequential) else [seq]
        out = []
        for m in mods:
            if isinstance(m, nn.Linear):
                out.append([m, 0])
            elif isinstance(m, nn.ELU) and m.alpha == 1.0 and out and out[-1][1] == 0:
                out[-1][1] = 1
            elif isinstance(m, nn.ReLU) and out and out[-1][1] == 0:
                out[-1][1] = 2
            else:
                return None
        return out

    @classmethod
    def describe_discriminator(cls, disc):
        """Discriminator.forward (bbc/rsl_rl/algorithms/discriminator.py:48-69) on prepared inputs x (N, 98): ReLU trunk, then the
        three heads as global outputs (logit (N,1), epsilon (N,1), class LOGITS (N,dim_c): the softmax is the consumer's)."""
        trunk = cls._linears(disc.trunk)
        if trunk is None or any(a != 2 for _, a in trunk) or disc.input_dim > cls.BUF_COLS[0] or disc.dim_c > 8:
            return None
        ops, params, woff, src, scol, k = [], [], 0, 0, 0, disc.input_dim
        bufs = {0: 1, 1: 2, 2: 1}
        for lin, act in trunk:
            dst = bufs[src]
            n = lin.out_features
            if lin.in_features != k or n > cls.BUF_COLS[dst]:
                return None
            nt, kb = (n + 15) // 16, cls.k_blocks(k, n)
            ops.append(_capi.QaMlpOp(kind=_capi.MLP_LAYER, src_buf=src, src_col=0, dst_buf=dst, dst_col=0, k=k, n=n, act=act, out_index=0,
                                     w_off=woff, b_off=woff + nt * kb * 256))
            params.append((lin.weight, lin.bias))
            woff += nt * kb * 256 + nt * 16
            src, k = dst, n
        for oi, head in enumerate((disc.linear, disc.encoder_eps, disc.classifier)):
            n = head.out_features
            nt, kb = (n + 15) // 16, cls.k_blocks(k, n)
            ops.append(_capi.QaMlpOp(kind=_capi.MLP_LAYER, src_buf=src, src_col=0, dst_buf=-1, dst_col=0, k=k, n=n, act=0, out_index=oi,
                                     w_off=woff, b_off=woff + nt * kb * 256))
            params.append((head.weight, head.bias))
            woff += nt * kb * 256 + nt * 16
        self = cls(ops, params, 0, out_widths=[1, 1, disc.dim_c])
        self.packed_floats = woff
        return self

    @classmethod
    def describe(cls, actor_critic, estimator, use_estimator, hist_encoding=False, with_critic=True, estimate_col=None):
        """-> PolicyChain, or None when the modules are not plain Linear/ELU stacks that fit the kernel's buffers.
        hist_encoding=True describes the variant whose latent comes from the history encoder (rollouts of every
        dagger_update_freq-th iteration, play.py, the exported policy): its per-frame Linear and its two Conv1d layers are
        each ONE layer op over the t-major window buffer, with the structured matrix (block-diagonal / banded) built from the
        module's parameters at pack() time; Flatten's channel-major order is folded into the output layer's columns."""
        import torch.nn as nn
        ac = actor_critic
        sl = ac._sl
        n_prop, n_exp, n_lat = ac.num_prop, ac.num_explicit, ac.num_latent
        hist0, cmd0 = sl[3].start, sl[4].start
        n_obs = ac.num_critic_obs
        n_cmd = n_obs - cmd0
        n_in = n_prop + n_exp + n_lat + n_cmd
        stacks = dict(actor=cls._linears(ac.actor_trunk))
        if with_critic:
            stacks["critic"] = cls._linears(ac.critic_trunk)
        use_priv = ac.train_with_estimated_latent and not hist_encoding and not isinstance(ac.priv_encoder, nn.Identity)
        use_hist = ac.train_with_estimated_latent and hist_encoding
        if use_priv:
            stacks["priv"] = cls._linears(ac.priv_encoder)
        if use_estimator:
            stacks["est"] = cls._linears(estimator.estimator)
        if any(v is None for v in stacks.values()) or n_obs > cls.BUF_COLS[0] or n_in > cls.BUF_COLS[3] or cmd0 + n_cmd != n_obs:
            return None
        stacks["actor"] = stacks["actor"] + [[ac.actor_head, 0]]
        if with_critic:
            stacks["critic"] = stacks["critic"] + [[ac.critic_head, 0]]
        ops, params, woff = [], [], [0]
        Z = 3
        lds_end = sum(16 * (c + 4) for c in cls.BUF_COLS)

        def copy(src, scol, dst, dcol, n):
            ops.append(_capi.QaMlpOp(kind=_capi.MLP_COPY, src_buf=src, src_col=scol, dst_buf=dst, dst_col=dcol, k=0, n=n))
            params.append(None)

        def layer(src, scol, k, dst, dcol, n, act, oi, w, bias):
            if dst == src or (dst >= 0 and dcol + n > cls.BUF_COLS[dst]) or scol % 4 or scol + k > cls.BUF_COLS[src] + 4:
                return False
            nt, kb = (n + 15) // 16, cls.k_blocks(k, n)
            base = sum(16 * (c + 4) for c in cls.BUF_COLS[:src])
            if base + 15 * (cls.BUF_COLS[src] + 4) + scol + 16 * kb > lds_end:
                return False            # the padded k-blocks would be read from beyond the kernel's LDS
            w_off = woff[0]; b_off = w_off + nt * kb * 256; woff[0] = b_off + nt * 16
            ops.append(_capi.QaMlpOp(kind=_capi.MLP_LAYER, src_buf=src, src_col=scol, dst_buf=dst, dst_col=dcol, k=k, n=n, act=act,
                                     out_index=oi, w_off=w_off, b_off=b_off))
            params.append((w, bias))
            return True

        def chain(layers, src, scol, k, final, reserved):
            """layers through scratch buffers; `final` = ("buf", b, col) or ("out", index)"""
            for i, (lin, act) in enumerate(layers):
                if lin.in_features != k:
                    return False
                n = lin.out_features
                if i == len(layers) - 1:
                    dst, dcol, oi = (final[1], final[2], 0) if final[0] == "buf" else (-1, 0, final[1])
                else:
                    fit = [b for b in (3, 2, 1) if b != src and b not in reserved and cls.BUF_COLS[b] >= n]
                    if not fit:
                        return False
                    dst, dcol, oi = fit[0], 0, 0
                if not layer(src, scol, k, dst, dcol, n, act, oi, lin.weight, lin.bias):
                    return False
                src, scol, k = dst, 0, n
            return True

        ok = True
        lat_col = n_prop + n_exp
        # estimate_col: where the estimator's output is written in the OBSERVATION row (default: the explicit privileged columns
        # right after proprio).  The task-level tree's act_bbc writes it at num_prop + num_scan of the 671-wide behaviour row
        # (tsc/rsl_rl/algorithms/ppo.py:127-137), i.e. into the history block, and leaves the explicit columns as they came: the
        # estimate then goes into the history window buffer before the history encoder reads it (and is moot without that encoder).
        est_in_obs = use_estimator and estimate_col is not None and estimate_col != n_prop
        if est_in_obs and not (hist0 <= estimate_col and estimate_col + n_exp <= cmd0):
            return None
        if use_hist:        # first: it borrows buffer 3 (the actor-input buffer Z) for the first convolution's output
            he = ac.history_encoder
            convs = [m for m in he.conv_layers if isinstance(m, nn.Conv1d)]
            elu = isinstance(he.activation_fn, nn.ELU) and he.activation_fn.alpha == 1.0
            if len(convs) != 2 or not elu or any(c.padding[0] or c.dilation[0] != 1 or c.bias is None for c in convs):
                return None
            enc, outl, T = he.encoder[0], he.linear_output[0], he.tsteps
            (c1, c2) = convs
            C0, C1, C2 = enc.out_features, c1.out_channels, c2.out_channels
            T1 = (T - c1.kernel_size[0]) // c1.stride[0] + 1
            T2 = (T1 - c2.kernel_size[0]) // c2.stride[0] + 1
            if (enc.in_features != n_prop or hist0 + T * n_prop != cmd0 or outl.in_features != C2 * T2 or outl.out_features != n_lat
                    or T * n_prop > cls.BUF_COLS[1] or T * C0 > cls.BUF_COLS[2] or T1 * C1 > cls.BUF_COLS[3] or T2 * C2 > cls.BUF_COLS[1]):
                return None

            def conv_matrix(conv, t_in, t_out):
                """(t_out C_out, t_in C_in): row p C_out + o, column (s p + j) C_in + c  <-  conv.weight[o, c, j]"""
                def build():
                    w = conv.weight
                    co, ci, k = w.shape
                    s_ = conv.stride[0]
                    full = torch.zeros(t_out * co, t_in * ci, dtype=w.dtype, device=w.device)
                    blk = w.permute(0, 2, 1).reshape(co, k * ci)
                    for p_ in range(t_out):
                        full[p_ * co:(p_ + 1) * co, s_ * p_ * ci:(s_ * p_ + k) * ci] = blk
                    return full
                return build

            rep = lambda b_, r: (lambda: b_.repeat(r))
            copy(0, hist0, 1, 0, T * n_prop)
            if est_in_obs:
                ok &= chain(stacks["est"], 0, 0, n_prop, ("buf", 1, estimate_col - hist0), {1}) and stacks["est"][-1][0].out_features == n_exp
            ok &= layer(1, 0, T * n_prop, 2, 0, T * C0, 1, 0, lambda: torch.block_diag(*([enc.weight] * T)).contiguous(), rep(enc.bias, T))
            ok &= layer(2, 0, T * C0, 3, 0, T1 * C1, 1, 0, conv_matrix(c1, T, T1), rep(c1.bias, T1))
            ok &= layer(3, 0, T1 * C1, 1, 0, T2 * C2, 1, 0, conv_matrix(c2, T1, T2), rep(c2.bias, T2))
            ok &= layer(1, 0, T2 * C2, Z, lat_col, n_lat, 1, 0,
                        lambda: outl.weight.view(n_lat, C2, T2).permute(0, 2, 1).reshape(n_lat, T2 * C2).contiguous(), outl.bias)
        copy(0, 0, Z, 0, n_prop)
        if use_estimator and not est_in_obs:
            ok &= chain(stacks["est"], 0, 0, n_prop, ("buf", Z, n_prop), {Z}) and stacks["est"][-1][0].out_features == n_exp
        else:
            copy(0, n_prop, Z, n_prop, n_exp)
        if use_priv:
            copy(0, lat_col, 2, 0, n_lat)                            # to column 0: a layer's source must be 16-byte aligned
            ok &= chain(stacks["priv"], 2, 0, n_lat, ("buf", Z, lat_col), {Z}) and stacks["priv"][-1][0].out_features == n_lat
        elif not use_hist:
            copy(0, lat_col, Z, lat_col, n_lat)
        copy(0, cmd0, Z, lat_col + n_lat, n_cmd)
        ok &= chain(stacks["actor"], Z, 0, n_in, ("out", 0), set())
        if with_critic:
            ok &= chain(stacks["critic"], 0, 0, n_obs, ("out", 1), set()) and stacks["critic"][-1][0].out_features == 1
        if not ok or len(ops) > _capi.MLP_MAX_OPS:
            return None
        self = cls(ops, params, stacks["actor"][-1][0].out_features)
        self.packed_floats = woff[0]
        return self

    @classmethod
    def describe_task_level(cls, actor_critic, estimator, use_estimator, part="all"):
        """The task-level teacher's networks of one env step (tsc/rsl_rl/algorithms/ppo.py:101-125 `act`: Estimator.forward on the first 57
        proprioception entries written over the privileged-explicit columns, Actor.forward -- scan encoder (last layer tanh), privileged
        encoder, trunk -- with the two heads `actor_d` / `actor_c`, and the critic on the TRUE 800-wide row; tsc/rsl_rl/modules/
        actor_critic.py:59-284) as one qa_mlp_forward chain: -> (gait logits (N, nd), parameter means (N, nd * nc), value (N, 1)).
        Privileged-encoder variant only (19 of 20 iterations; the history-encoder rollouts keep the module path).  None when the modules do
        not fit the kernel (other activations, widths beyond the LDS buffers).
        `part`: "all" (one launch, three outputs), "actor" (-> logits, means) or "critic" (-> value): the two halves as separate programs, for
        SplitTeacherChain (few row tiles: the halves run side by side instead of one after the other)."""
        import torch.nn as nn
        assert part in ("all", "actor", "critic")
        ac = actor_critic
        actor = ac.actor
        a, n_scan, n_exp, n_lat = actor.num_prop, actor.num_scan, actor.num_priv_explicit, actor.num_priv_latent
        scan0, exp0, lat0 = a, a + n_scan, a + n_scan + n_exp

        def stack(seq, tanh_last=False):
            mods = list(seq) if isinstance(seq, nn.Sequential) else None
            if mods is None:
                return None
            if tanh_last and mods and isinstance(mods[-1], nn.Tanh):
                body = cls._linears(nn.Sequential(*mods[:-1]))
                if body is None or body[-1][1] != 0:
                    return None
                body[-1][1] = 3
                return body
            return cls._linears(seq)
        crit = stack(ac.critic)
        st = dict(scan=stack(actor.scan_encoder, tanh_last=True) if actor.if_scan_encode else None, priv=stack(actor.priv_encoder), trunk=stack(actor.actor_trunk), critic=crit)
        if use_estimator:
            st["est"] = stack(estimator.estimator)
        if not actor.if_scan_encode or any(v is None for v in st.values()) or isinstance(actor.priv_encoder, nn.Identity):
            return None
        n_obs = crit[0][0].in_features
        n_in = a + st["scan"][-1][0].out_features + n_exp + n_lat
        Z = 2                                               # the trunk's input is assembled in scratch buffer 2
        if n_obs > cls.BUF_COLS[0] or n_in > cls.BUF_COLS[Z] or st["trunk"][0][0].in_features != n_in or lat0 + n_lat > n_obs:
            return None
        ops, params, woff = [], [], [0]
        lds_end = sum(16 * (c + 4) for c in cls.BUF_COLS)

        def copy(src, scol, dst, dcol, n):
            ops.append(_capi.QaMlpOp(kind=_capi.MLP_COPY, src_buf=src, src_col=scol, dst_buf=dst, dst_col=dcol, k=0, n=n))
            params.append(None)

        def layer(src, scol, k, dst, dcol, n, act, oi, w, bias):
            if dst == src or (dst >= 0 and dcol + n > cls.BUF_COLS[dst]) or scol % 4 or scol + k > cls.BUF_COLS[src] + 4:
                return False
            nt, kb = (n + 15) // 16, cls.k_blocks(k, n)
            base = sum(16 * (c + 4) for c in cls.BUF_COLS[:src])
            if base + 15 * (cls.BUF_COLS[src] + 4) + scol + 16 * kb > lds_end or nt > 32:
                return False
            w_off = woff[0]; b_off = w_off + nt * kb * 256; woff[0] = b_off + nt * 16
            ops.append(_capi.QaMlpOp(kind=_capi.MLP_LAYER, src_buf=src, src_col=scol, dst_buf=dst, dst_col=dcol, k=k, n=n, act=act, out_index=oi, w_off=w_off, b_off=b_off))
            params.append((w, bias))
            return True

        def chain(layers, src, scol, k, final, reserved):
            for i, (lin, act) in enumerate(layers):
                if lin.in_features != k:
                    return False
                n = lin.out_features
                if i == len(layers) - 1:
                    dst, dcol, oi = (final[1], final[2], 0) if final[0] == "buf" else (-1, 0, final[1])
                else:
                    fit = [b for b in (3, 2, 1) if b != src and b not in reserved and cls.BUF_COLS[b] >= n]
                    if not fit:
                        return False
                    dst, dcol, oi = fit[0], 0, 0
                if not layer(src, scol, k, dst, dcol, n, act, oi, lin.weight, lin.bias):
                    return False
                src, scol, k = dst, 0, n
            return True

        ok = st["critic"][-1][0].out_features == 1
        if part != "actor":
            ok = ok and chain(st["critic"], 0, 0, n_obs, ("out", 2 if part == "all" else 0), set())      # first: it may use every scratch buffer
        if part == "critic":
            if not ok or len(ops) > _capi.MLP_MAX_OPS:
                return None
            self = cls(ops, params, 1, out_widths=[1])
            self.packed_floats = woff[0]
            return self
        copy(0, 0, Z, 0, a)
        copy(0, scan0, 1, 0, n_scan)                        # a layer's source column must be 16-byte aligned: the scan starts at column 65
        ok = ok and chain(st["scan"], 1, 0, n_scan, ("buf", Z, a), {Z})
        lat_dst = a + st["scan"][-1][0].out_features
        if use_estimator:
            ok = ok and chain(st["est"], 0, 0, st["est"][0][0].in_features, ("buf", Z, lat_dst), {Z}) and st["est"][-1][0].out_features == n_exp
        else:
            copy(0, exp0, Z, lat_dst, n_exp)
        copy(0, lat0, 1, 0, n_lat)
        ok = ok and chain(st["priv"], 1, 0, n_lat, ("buf", Z, lat_dst + n_exp), {Z}) and st["priv"][-1][0].out_features == n_lat
        ok = ok and chain(st["trunk"], Z, 0, n_in, ("buf", 3, 0), set())
        k = st["trunk"][-1][0].out_features
        ok = ok and layer(3, 0, k, -1, 0, actor.actor_d.out_features, 0, 0, actor.actor_d.weight, actor.actor_d.bias)
        ok = ok and layer(3, 0, k, -1, 0, actor.actor_c.out_features, 0, 1, actor.actor_c.weight, actor.actor_c.bias)
        if not ok or len(ops) > _capi.MLP_MAX_OPS or k > cls.BUF_COLS[3]:
            return None
        self = cls(ops, params, actor.actor_c.out_features, out_widths=[actor.actor_d.out_features, actor.actor_c.out_features] + ([1] if part == "all" else []))
        self.packed_floats = woff[0]
        return self

    # ---------------------------------------------------------------- device side
    def _ptr_arrays(self):
        """device pointers of every layer's (weight, bias); entries of `params` may be callables that build the tensor from the
        module's parameters (the structured matrices of the history encoder) -- the results stay referenced in self._built"""
        self._built = [tuple((t() if callable(t) else t) for t in p) if p else None for p in self.params]
        w = (C.c_void_p * self.n_ops)(*[(p[0].data_ptr() if p else None) for p in self._built])
        b = (C.c_void_p * self.n_ops)(*[(p[1].data_ptr() if p and p[1] is not None else None) for p in self._built])
        return w, b

    def pack(self):
        w, b = self._ptr_arrays()
        dev = next(p[0] for p in self._built if p).device
        if self.packed is None:
            self.packed = torch.zeros(self.packed_floats, dtype=torch.float32, device=dev)
        for p in self._built:
            if p and not (p[0].is_cuda and p[0].is_contiguous() and p[0].dtype == torch.float32):
                raise RuntimeError("PolicyChain: parameters must be contiguous fp32 ROCm tensors")
        lib = _capi.load_library()
        rc = lib.qa_mlp_pack(self.ops, self.n_ops, w, b, _ptr(self.packed), self.packed_floats, C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
        if rc != 0:
            raise RuntimeError(f"qa_mlp_pack failed with code {rc}: {lib.qa_last_error().decode()}")

    def forward(self, obs):
        """-> one (N, width) tensor per global output of the chain (policy chains: action mean, value).  The tensors are
        persistent per batch size: the next call overwrites them."""
        n = obs.shape[0]
        assert obs.is_cuda and obs.dtype == torch.float32 and obs.stride(1) == 1 and self.packed is not None
        out = self._out.get(n)
        if out is None:
            bufs = [torch.zeros(n, w, device=obs.device) for w in self.out_widths]
            k = len(bufs)
            out = self._out[n] = (bufs, (C.c_void_p * k)(*[b.data_ptr() for b in bufs]), (C.c_int64 * k)(*self.out_widths))
        lib = _capi.load_library()
        rc = lib.qa_mlp_forward(_ptr(obs), obs.stride(0), n, obs.shape[1], self.ops, self.n_ops, _ptr(self.packed), out[1], out[2], len(out[0]),
                                C.c_void_p(torch.cuda.current_stream(obs.device).cuda_stream))
        if rc != 0:
            raise RuntimeError(f"qa_mlp_forward failed with code {rc}: {lib.qa_last_error().decode()}")
        return tuple(out[0])


class SplitTeacherChain:
    """The task-level teacher's chain as TWO programs -- actor side (estimator, encoders, trunk, heads) and critic -- launched side by side, the
    critic on a second stream.  A chain kernel's time is one 16-row tile's serial walk through its layers whatever the number of tiles, so with
    few tiles (1024 envs = 64 workgroups on 256 CUs: the per-GPU share of BASELINE configs[3]) two half-depth launches next to each other take the
    longer half's time instead of the sum.  Same ops per output as the one-launch chain: the outputs are bit-identical.  Capturable (the side
    stream forks from and joins the calling stream inside forward)."""

    def __init__(self, actor_chain, critic_chain):
        self.actor, self.critic = actor_chain, critic_chain
        self._side = None

    @classmethod
    def describe(cls, actor_critic, estimator, use_estimator):
        a = PolicyChain.describe_task_level(actor_critic, estimator, use_estimator, part="actor")
        c = PolicyChain.describe_task_level(actor_critic, estimator, use_estimator, part="critic")
        return cls(a, c) if (a is not None and c is not None) else None

    def pack(self):
        self.actor.pack(); self.critic.pack()

    def forward(self, obs):
        cur = torch.cuda.current_stream(obs.device)
        if self._side is None:
            self._side = torch.cuda.Stream(device=obs.device)
        self._side.wait_stream(cur)                 # the observation row and the packed weights are ready on `cur`
        with torch.cuda.stream(self._side):
            (value,) = self.critic.forward(obs)
        logits, mean = self.actor.forward(obs)
        cur.wait_stream(self._side)
        return logits, mean, value
