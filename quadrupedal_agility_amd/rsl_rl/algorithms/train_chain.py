"""The networks of one PPO minibatch step as TWO chain launches (include/qa_sim.h, ABI 17): forward with saved activations, then the
input-gradient path on transposed weights; weight gradients stay GEMMs.

What it replaces: SSInfoGAIL.update_actor_critic's forward and backward passes (bbc/rsl_rl/algorithms/gail.py:328-413 through autograd):
estimator, privileged encoder, actor trunk + head, critic trunk + head -- 13 Linear layers, i.e. ~25 launches forward (GEMMs, ELUs,
concatenations) and ~45 backward (ELU', two GEMMs and a bias reduction per layer) over three streams.  With few rows per minibatch (the
per-GPU share of the 8-GPU job: 512 envs = 3,072 rows) every one of those is launch-latency sized and the step is a serial chain of them
(profiles/r5_final_ppo_step_kernel_sequence.txt: 95 launches, ~0.5 ms for 14 GFLOP).  `qa_mlp_forward` already walks a 16-row tile through
the whole network with the activations in LDS for the rollout; here the same launch also SAVES every hidden layer's output to a (rows, 2160)
tape, and a second program walks the tile back from the heads' gradients: each input-gradient layer multiplies by the forward layer's
transposed matrix and scales by the activation derivative taken from the tape (ELU: y > 0 ? 1 : y + 1), writing the gradient at every
layer's pre-activation to a second tape.  The 13 weight (+ bias) gradients are `qa_linear_backward_weight` products of tape columns, left in
parts for the optimiser's first pass (fused.deferred parts).

Only used below `MAX_ROWS` rows per step: at 24,576 rows the library's GEMMs run the 512-wide trunk layers at 0.7-0.85 of the MFMA peak and
a 16-row tile walk does not."""
import ctypes as C
import os

import torch

from quadrupedal_agility_amd import _capi
from quadrupedal_agility_amd.rsl_rl.algorithms import fused

MAX_ROWS = int(os.environ.get("QA_TRAIN_CHAIN_MAX_ROWS", "8192"))      # measured (r6, profiles/r6_chain_steps.txt): 3,072-row steps 30 % faster, 6,144-row steps 17 % faster than the three-stream autograd step; 12,288 rows and up: slower
ENABLED = os.environ.get("QA_TRAIN_CHAIN", "1") != "0"
DISC_ENABLED = os.environ.get("QA_DISC_TRAIN_CHAIN", "1") != "0"


def _ptr(t):
    return C.c_void_p(t.data_ptr())


def _pad4(n):
    return (int(n) + 3) // 4 * 4


def _ok(rc, name, prog):
    if rc != 0:
        msg = _capi.load_library().qa_last_error().decode() if prog.prefix == "qa_" else "(an injected library: no error text)"
        raise RuntimeError(f"{prog.prefix}{name} failed with code {rc}: {msg}")


class _Program:
    """an op list + its packed weights (fused.PolicyChain's pack / launch halves, with explicit output tables)"""

    def __init__(self, lib=None, prefix="qa_"):
        self.ops, self.params, self.woff = [], [], 0
        self.packed = None
        self.lib, self.prefix = lib, prefix          # injection point of the tests (another library exporting the same entry points under another prefix); the default is libqa_sim.so, and it is required

    def _fn(self, name):
        return getattr(self.lib or _capi.load_library(), self.prefix + name)

    def _stream(self, t):
        return C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream) if t.is_cuda else None

    def _add(self, op, param):
        self.ops.append(op); self.params.append(param)

    def copy(self, src, scol, dst, dcol, n, save=None):
        kw = dict(flags=_capi.MLP_F_SAVE, out_index=save[0], out_col=save[1]) if save else {}
        self._add(_capi.QaMlpOp(kind=_capi.MLP_COPY, src_buf=src, src_col=scol, dst_buf=dst, dst_col=dcol, k=0, n=n, **kw), None)

    def load(self, aux, dst, dcol, n, save=None):
        """scratch buffer <- columns of a global tensor (aux = (output slot, first column))"""
        kw = dict(flags=_capi.MLP_F_SAVE, out_index=save[0], out_col=save[1]) if save else {}
        self._add(_capi.QaMlpOp(kind=_capi.MLP_LOAD, src_buf=0, src_col=0, dst_buf=dst, dst_col=dcol, k=0, n=n, aux_index=aux[0], aux_col=aux[1], **kw), None)

    def grad(self, src, scol, dst, dcol, n, act=0, aux=None, add=False, save=None):
        flags = (_capi.MLP_F_ADD if add else 0) | (_capi.MLP_F_SAVE if save else 0)
        kw = dict(out_index=save[0], out_col=save[1]) if save else {}
        if aux:
            kw.update(aux_index=aux[0], aux_col=aux[1])
        self._add(_capi.QaMlpOp(kind=_capi.MLP_GRAD, src_buf=src, src_col=scol, dst_buf=dst, dst_col=dcol, k=0, n=n, act=act, flags=flags, **kw), None)

    def layer(self, src, scol, k, dst, dcol, n, act, weight, bias, out=None, save=None, aux=None, transposed=False):
        """dst >= 0: into scratch buffer dst (and, with `save` = (output, column), to that global output too); dst = -1: to `out` = (output, column)"""
        cols = _capi.MLP_BUF_COLS
        assert dst != src and scol % 4 == 0 and scol + k <= cols[src] + 4 and (dst < 0 or dcol + n <= cols[dst]), (src, scol, k, dst, dcol, n)
        nt, kb = (n + 15) // 16, fused.PolicyChain.k_blocks(k, n)
        base = sum(16 * (c + 4) for c in cols[:src])
        assert base + 15 * (cols[src] + 4) + scol + 16 * kb <= sum(16 * (c + 4) for c in cols), "padded k-blocks beyond the kernel's LDS"
        w_off = self.woff; b_off = w_off + nt * kb * 256; self.woff = b_off + nt * 16
        flags = (_capi.MLP_F_SAVE if (save and dst >= 0) else 0) | (_capi.MLP_F_TRANSPOSED if transposed else 0)
        oi, oc = (out if dst < 0 else save) or (0, 0)
        ai, acol = aux or (0, 0)
        self._add(_capi.QaMlpOp(kind=_capi.MLP_LAYER, src_buf=src, src_col=scol, dst_buf=dst, dst_col=dcol, k=k, n=n, act=act, out_index=oi, flags=flags,
                                w_off=w_off, b_off=b_off, out_col=oc, aux_index=ai, aux_col=acol), (weight, bias))

    def finish(self):
        assert len(self.ops) <= _capi.MLP_MAX_OPS, len(self.ops)
        self.n_ops = len(self.ops)
        self.c_ops = (_capi.QaMlpOp * self.n_ops)(*self.ops)
        self.w_ptrs = (C.c_void_p * self.n_ops)(*[(p[0].data_ptr() if p else None) for p in self.params])
        self.b_ptrs = (C.c_void_p * self.n_ops)(*[(p[1].data_ptr() if p and p[1] is not None else None) for p in self.params])
        for p in self.params:
            if p and not ((p[0].is_cuda or self.prefix != "qa_") and p[0].is_contiguous() and p[0].dtype == torch.float32):
                raise RuntimeError("train chain: parameters must be contiguous fp32 ROCm tensors")
        dev = next(p[0] for p in self.params if p).device
        self.packed = torch.zeros(self.woff, dtype=torch.float32, device=dev)

    def pack(self):
        _ok(self._fn("mlp_pack")(self.c_ops, self.n_ops, self.w_ptrs, self.b_ptrs, _ptr(self.packed), self.woff, self._stream(self.packed)), "mlp_pack", self)

    def launch(self, x, x_cols, outs):
        k = len(outs)
        ptrs = (C.c_void_p * k)(*[o.data_ptr() for o in outs]); strides = (C.c_int64 * k)(*[o.stride(0) for o in outs])
        _ok(self._fn("mlp_forward")(_ptr(x), x.stride(0), x.shape[0], x_cols, self.c_ops, self.n_ops, _ptr(self.packed), ptrs, strides, k, self._stream(x)),
            "mlp_forward", self)


class _Packer:
    """qa_mlp_pack for SEVERAL programs in one launch: their layers as one op list over one packed buffer (each program's `packed` becomes its slice).
    The weights change with every optimiser step, so a chain step repacks before it runs: one launch instead of one per program."""

    def __init__(self, programs):
        p0 = programs[0]
        self.prog = p0
        total = sum(p.woff for p in programs)
        self.buf = torch.zeros(total, dtype=torch.float32, device=p0.packed.device)
        ops, w, b, off = [], [], [], 0
        for p in programs:
            p.packed = self.buf[off:off + p.woff]
            for op, param in zip(p.ops, p.params):
                if op.kind == _capi.MLP_LAYER:
                    o2 = _capi.QaMlpOp.from_buffer_copy(op)
                    o2.w_off += off; o2.b_off += off
                    ops.append(o2); w.append(param[0].data_ptr()); b.append(param[1].data_ptr() if param[1] is not None else None)
            off += p.woff
        assert off == total
        M = _capi.MLP_MAX_OPS            # (a pack call takes at most this many layers: the task-level step's 30 go in two)
        self.calls = [((_capi.QaMlpOp * len(ops[i:i + M]))(*ops[i:i + M]), len(ops[i:i + M]), (C.c_void_p * len(ops[i:i + M]))(*w[i:i + M]),
                       (C.c_void_p * len(ops[i:i + M]))(*b[i:i + M])) for i in range(0, len(ops), M)]
        self.total = total

    def pack(self):
        for ops, n, w, b in self.calls:
            _ok(self.prog._fn("mlp_pack")(ops, n, w, b, _ptr(self.buf), self.total, self.prog._stream(self.buf)), "mlp_pack", self.prog)


class _Sides:
    """Side streams for launches that do not depend on each other (the weight-gradient products of a step; the discriminator's penalty path beside
    its objective): forked from the current stream, joined back into it.  Recorded into a step's hipGraph they are parallel branches.  Every buffer
    the chains touch is persistent, so no block changes hands between streams."""

    def __init__(self, device, n):
        self.streams = [torch.cuda.Stream(device=device) for _ in range(n)] if torch.device(device).type == "cuda" else []
        self.open = set()

    def fork(self, i):
        s = self.streams[i]
        s.wait_stream(torch.cuda.current_stream(s.device))
        self.open.add(i)
        return torch.cuda.stream(s)

    def join(self):
        """the current stream waits for every side stream forked since the last join (only those: a stream that took no part in a recording
        must not be waited for inside it)"""
        for i in sorted(self.open):
            torch.cuda.current_stream(self.streams[i].device).wait_stream(self.streams[i])
        self.open.clear()


# Measured (profiles/r6_chain_side_streams.txt): as parallel branches of a step's hipGraph the side streams LOSE -- every fork / join is a cross-branch
# dependency the graph executes with a 10-140 us gap (512-env PPO step 517 -> 621 us, config 3's iteration 53.4 -> 64.2 ms).  Off by default; the
# independent products go side by side inside ONE launch instead (qa_linear_backward_weight_batch).
SIDE_STREAMS = os.environ.get("QA_TRAIN_CHAIN_SIDES", "0") == "1"


class PpoTrainChain:
    """forward(obs) -> (est, mu, value, priv_latent); backward(g_est, dmu, dvalue, g_priv) -> every parameter's `.grad` (wide ones in parts,
    registered with fused's deferred finishes).  Buffers are persistent per instance: one instance per minibatch size, recordable."""

    ELU, DELU = 1, _capi.MLP_ACT_ELU_GRAD

    @classmethod
    def describe(cls, ac, estimator, rows, lib=None, prefix="qa_"):
        import torch.nn as nn
        lin = fused.PolicyChain._linears
        st = dict(actor=lin(ac.actor_trunk), critic=lin(ac.critic_trunk), est=lin(estimator.estimator),
                  priv=None if isinstance(ac.priv_encoder, nn.Identity) else lin(ac.priv_encoder))
        if not ENABLED or rows > MAX_ROWS or any(v is None for v in st.values()) or not ac.train_with_estimated_latent or ac.fixed_std:
            return None
        if (any(a != 1 for k in ("actor", "critic", "priv") for _, a in st[k]) or any(a != 1 for _, a in st["est"][:-1]) or st["est"][-1][1] != 0
                or len(st["actor"]) != 3 or len(st["critic"]) != 3 or len(st["priv"]) != 2 or len(st["est"]) != 3):
            return None
        sl = ac._sl
        n_prop, n_exp, n_lat = ac.num_prop, ac.num_explicit, ac.num_latent
        cmd0, n_obs = sl[4].start, ac.num_critic_obs
        n_cmd, lat_col = n_obs - cmd0, n_prop + n_exp
        n_in = n_prop + n_exp + n_lat + n_cmd
        cols = _capi.MLP_BUF_COLS
        a_w, c_w, p_w, e_w = ([l.out_features for l, _ in st[k]] for k in ("actor", "critic", "priv", "est"))
        n_act = ac.actor_head.out_features
        if (n_obs > cols[0] or n_in > cols[3] or a_w[0] > cols[1] or a_w[1] > cols[2] or a_w[2] > cols[3] or c_w[0] > cols[1] or c_w[1] > cols[2] or c_w[2] > cols[3]
                or p_w[0] > cols[1] or p_w[1] != n_lat or e_w[0] > cols[3] or e_w[1] > cols[2] or e_w[2] != n_exp or st["actor"][0][0].in_features != n_in
                or st["critic"][0][0].in_features != n_obs or st["est"][0][0].in_features != n_prop or st["priv"][0][0].in_features != n_lat
                or ac.critic_head.out_features != 1 or n_in > cols[2]):
            return None
        self = cls()
        self.ac, self.estimator, self.rows = ac, estimator, rows
        self.dims = dict(n_prop=n_prop, n_exp=n_exp, n_lat=n_lat, n_cmd=n_cmd, lat_col=lat_col, cmd0=cmd0, n_obs=n_obs, n_in=n_in, n_act=n_act)
        # ---- the tape of saved activations (forward) and of pre-activation gradients (backward): column offsets, each a multiple of 4
        t, off = {}, 0
        for name, w in (("ain", n_in), ("p1", p_w[0]), ("a1", a_w[0]), ("a2", a_w[1]), ("a3", a_w[2]), ("c1", c_w[0]), ("c2", c_w[1]), ("c3", c_w[2]),
                        ("e1", e_w[0]), ("e2", e_w[1])):
            t[name] = (off, w); off += _pad4(w)
        self.tape_cols, self.t = off, t
        g, off = {}, 0
        for name, w in (("c3", c_w[2]), ("c2", c_w[1]), ("c1", c_w[0]), ("a3", a_w[2]), ("a2", a_w[1]), ("a1", a_w[0]), ("p2", n_lat), ("p1", p_w[0]),
                        ("e2", e_w[1]), ("e1", e_w[0])):
            g[name] = (off, w); off += _pad4(w)
        self.gtape_cols, self.g = off, g
        T, MU, VAL, EST = 0, 1, 2, 3          # output slots of the forward launch
        E, D = cls.ELU, cls.DELU
        (a1, a2, a3), (c1, c2, c3), (p1, p2), (e1, e2, e3) = ([l for l, _ in st[k]] for k in ("actor", "critic", "priv", "est"))
        ain = t["ain"][0]
        # ---- forward program (what PolicyChain.describe builds for the rollout, plus the saves; the actor's explicit-state columns are the
        # OBSERVATION's: update_actor_critic feeds the true privileged state, only act() substitutes the estimate)
        f = _Program(lib, prefix)
        Z = 3
        f.copy(0, 0, Z, 0, lat_col, save=(T, ain))                                     # proprio + explicit
        f.copy(0, lat_col, 2, 0, n_lat)                                                # (a layer's source starts 16-byte aligned)
        f.layer(2, 0, n_lat, 1, 0, p_w[0], E, p1.weight, p1.bias, save=(T, t["p1"][0]))
        f.layer(1, 0, p_w[0], Z, lat_col, n_lat, E, p2.weight, p2.bias, save=(T, ain + lat_col))      # = the privileged latent, in the actor's input row
        f.copy(Z, lat_col, 2, 0, n_lat, save=(4, 0))                                   # ... and once more as a dense (rows, n_lat) tensor for the regulariser's kernel
        f.copy(0, cmd0, Z, lat_col + n_lat, n_cmd, save=(T, ain + lat_col + n_lat))
        f.layer(Z, 0, n_in, 1, 0, a_w[0], E, a1.weight, a1.bias, save=(T, t["a1"][0]))
        f.layer(1, 0, a_w[0], 2, 0, a_w[1], E, a2.weight, a2.bias, save=(T, t["a2"][0]))
        f.layer(2, 0, a_w[1], Z, 0, a_w[2], E, a3.weight, a3.bias, save=(T, t["a3"][0]))
        f.layer(Z, 0, a_w[2], -1, 0, n_act, 0, ac.actor_head.weight, ac.actor_head.bias, out=(MU, 0))
        f.layer(0, 0, n_obs, 1, 0, c_w[0], E, c1.weight, c1.bias, save=(T, t["c1"][0]))
        f.layer(1, 0, c_w[0], 2, 0, c_w[1], E, c2.weight, c2.bias, save=(T, t["c2"][0]))
        f.layer(2, 0, c_w[1], Z, 0, c_w[2], E, c3.weight, c3.bias, save=(T, t["c3"][0]))
        f.layer(Z, 0, c_w[2], -1, 0, 1, 0, ac.critic_head.weight, ac.critic_head.bias, out=(VAL, 0))
        f.layer(0, 0, n_prop, Z, 0, e_w[0], E, e1.weight, e1.bias, save=(T, t["e1"][0]))
        f.layer(Z, 0, e_w[0], 2, 0, e_w[1], E, e2.weight, e2.bias, save=(T, t["e2"][0]))
        f.layer(2, 0, e_w[1], -1, 0, n_exp, 0, e3.weight, e3.bias, out=(EST, 0))
        f.finish()
        # ---- backward program.  Input tile = d loss / d action mean; the other three gradients are LOADED from their own tensors (the objectives'
        # kernels write them where they like).  outputs: 0 = gradient tape (written), 1 = activation tape, 2 = d value, 3 = d estimate, 4 = d latent (read)
        G, A, DV, DE, DP = 0, 1, 2, 3, 4
        b = _Program(lib, prefix)
        b.load((DV, 0), 1, 0, 1)
        b.layer(1, 0, 1, Z, 0, c_w[2], D, ac.critic_head.weight, None, save=(G, g["c3"][0]), aux=(A, t["c3"][0]), transposed=True)
        b.layer(Z, 0, c_w[2], 2, 0, c_w[1], D, c3.weight, None, save=(G, g["c2"][0]), aux=(A, t["c2"][0]), transposed=True)
        b.layer(2, 0, c_w[1], 1, 0, c_w[0], D, c2.weight, None, save=(G, g["c1"][0]), aux=(A, t["c1"][0]), transposed=True)
        b.layer(0, 0, n_act, Z, 0, a_w[2], D, ac.actor_head.weight, None, save=(G, g["a3"][0]), aux=(A, t["a3"][0]), transposed=True)
        b.layer(Z, 0, a_w[2], 2, 0, a_w[1], D, a3.weight, None, save=(G, g["a2"][0]), aux=(A, t["a2"][0]), transposed=True)
        b.layer(2, 0, a_w[1], 1, 0, a_w[0], D, a2.weight, None, save=(G, g["a1"][0]), aux=(A, t["a1"][0]), transposed=True)
        b.layer(1, 0, a_w[0], 2, 0, n_in, 0, a1.weight, None, transposed=True)                    # d loss / d actor input (only its latent columns are needed)
        b.load((DP, 0), Z, 0, n_lat)                                                                # the regulariser's gradient at the privileged latent ...
        b.grad(2, lat_col, Z, 0, n_lat, act=D, aux=(A, ain + lat_col), add=True, save=(G, g["p2"][0]))   # ... + the actor's, times ELU' of the latent
        b.layer(Z, 0, n_lat, 2, 0, p_w[0], D, p2.weight, None, save=(G, g["p1"][0]), aux=(A, t["p1"][0]), transposed=True)
        b.load((DE, 0), 1, 0, n_exp)
        b.layer(1, 0, n_exp, Z, 0, e_w[1], D, e3.weight, None, save=(G, g["e2"][0]), aux=(A, t["e2"][0]), transposed=True)
        b.layer(Z, 0, e_w[1], 2, 0, e_w[0], D, e2.weight, None, save=(G, g["e1"][0]), aux=(A, t["e1"][0]), transposed=True)
        b.finish()
        self.fwd, self.bwd = f, b
        self._packer = _Packer([f, b])
        dev = a1.weight.device
        z = lambda *s: torch.zeros(*s, dtype=torch.float32, device=dev)
        self.tape, self.gtape = z(rows, self.tape_cols), z(rows, self.gtape_cols)
        self.mu, self.value, self.est, self.priv = z(rows, n_act), z(rows, 1), z(rows, n_exp), z(rows, n_lat)
        # ---- weight-gradient products: (parameter pair, gradient columns, input columns); `obs` inputs are resolved per call
        tc = lambda name: ("tape", t[name][0], t[name][1])
        gcol = lambda name: ("gtape", g[name][0], g[name][1])
        self.wgrads = [
            (c1, gcol("c1"), ("obs", 0, n_obs)), (c2, gcol("c2"), tc("c1")), (c3, gcol("c3"), tc("c2")), (ac.critic_head, ("dvalue", 0, 1), tc("c3")),
            (a1, gcol("a1"), ("tape", ain, n_in)), (a2, gcol("a2"), tc("a1")), (a3, gcol("a3"), tc("a2")), (ac.actor_head, ("dmu", 0, n_act), tc("a3")),
            (p1, gcol("p1"), ("obs", lat_col, n_lat)), (p2, gcol("p2"), tc("p1")),
            (e1, gcol("e1"), ("obs", 0, n_prop)), (e2, gcol("e2"), tc("e1")), (e3, ("g_est", 0, n_exp), tc("e2"))]
        self._wg = []
        for lin_, _, _ in self.wgrads:
            n, k = lin_.out_features, lin_.in_features
            nb = int(f._fn("linear_backward_weight_batch_scratch_bytes")(rows, k, n))
            lay = (C.c_int64 * 5)()
            _ok(f._fn("linear_backward_weight_batch_layout")(rows, k, n, lay), "linear_backward_weight_batch_layout", f)
            scratch = torch.zeros(nb // 4 + 4, dtype=torch.float32, device=dev)
            gw, gb = torch.zeros_like(lin_.weight), torch.zeros_like(lin_.bias)
            self._wg.append((scratch, nb, [int(v) for v in lay], gw, gb))
        self.sides = _Sides(dev, 1) if (SIDE_STREAMS and prefix == "qa_") else None          # (the estimator's optimiser beside the actor-critic's: gail._ac_apply)
        return self

    def pack(self):
        self._packer.pack()

    def forward(self, obs):
        assert obs.shape[0] == self.rows and obs.stride(1) == 1 and obs.shape[1] >= self.dims["n_obs"]
        self._obs = obs
        self.fwd.launch(obs, self.dims["n_obs"], [self.tape, self.mu, self.value, self.est, self.priv])
        return self.est, self.mu, self.value, self.priv

    def backward(self, g_est, dmu, dvalue, g_priv, defer=True):
        """the four gradients at the chains' outputs -> `.grad` of all 26 parameters.  With `defer` the weight / bias gradients stay in parts,
        registered with fused's deferred finishes: ClipAdam.step() (or fused.flush_pending_grads()) adds them."""
        d = self.dims
        f32c = lambda t, shape: (t if (t.dtype == torch.float32 and t.is_contiguous()) else t.contiguous().float()).view(shape)
        dmu, dvalue, g_est, g_priv = f32c(dmu, (self.rows, d["n_act"])), f32c(dvalue, (self.rows, 1)), f32c(g_est, (self.rows, d["n_exp"])), f32c(g_priv, (self.rows, d["n_lat"]))
        gin = dmu
        self.bwd.launch(dmu, d["n_act"], [self.gtape, self.tape, dvalue, g_est, g_priv])
        src = {"tape": self.tape, "gtape": self.gtape, "dmu": dmu, "dvalue": dvalue, "g_est": g_est, "obs": self._obs}
        defer = defer and gin.is_cuda and fused.ENABLED and os.environ.get("QA_DEFER_GRAD_FINISH", "1") != "0"
        # the 13 weight (+ bias) gradient products of the step in ONE call: qa_linear_backward_weight_batch groups them by operand alignment into at
        # most four launches of a few thousand workgroups (one after the other they were 13 launches of 16-190 workgroups, 215 of the step's 517 us)
        descs = (_capi.QaWgradDesc * len(self.wgrads))()
        for i, ((lin_, (gs, g0, gn), (xs, x0, xk)), (scratch, nb, lay, gw, gb)) in enumerate(zip(self.wgrads, self._wg)):
            gt, xt = src[gs], src[xs]
            descs[i] = _capi.QaWgradDesc(gt.data_ptr() + 4 * g0, gt.stride(0), xt.data_ptr() + 4 * x0, xt.stride(0), None if defer else gw.data_ptr(),
                                         None if defer else gb.data_ptr(), self.rows, xk, gn, scratch.data_ptr(), nb)
        _ok(self.bwd._fn("linear_backward_weight_batch")(descs, len(self.wgrads), self.bwd._stream(gin)), "linear_backward_weight_batch", self.bwd)
        for (lin_, _, _), (scratch, nb, lay, gw, gb) in zip(self.wgrads, self._wg):
            lin_.weight.grad, lin_.bias.grad = gw, gb
            if defer:
                fused.register_grad_parts(lin_.weight, scratch, lay[0], lay[1], gw)
                fused.register_grad_parts(lin_.bias, scratch[lay[4]:], lay[2], lay[3], gb)


class DiscTrainChain:
    """One discriminator step's networks (SSInfoGAIL.update_ss_info_gail, bbc/rsl_rl/algorithms/gail.py:415-541; Discriminator.forward,
    discriminator.py:48-69) as THREE chain launches instead of ~30 launch-latency-sized GEMM / ReLU / mask kernels in a row:

      forward   x (R, 98) -> h1 -> h2 (ReLU, saved) -> logit, epsilon, class logits                                     all R = 3 x minibatch rows
      penalty   the gradient penalty's whole path on the unlabelled expert rows U (gail.py:487-492 gets d logit / d x by double backward; for this
                piecewise-linear trunk it is a chain of masked products, see Discriminator.forward_with_input_gradient):
                  v2 = m2 * w_out,  v1 = (v2 W2) * m1,  g = v1 W1          the penalty's argument (per row: P = c |g|^2 / |U|)
                  u1 = (g W1^T) * m1,  u2 = (u1 W2^T) * m2                 its way back: dP/dW1 = a v1^T g, dP/dW2 = a v2^T u1, dP/dw_out = a colsum(u2),
                                                                           a = 2 c / |U| -- g enters linearly, so the SAME tile walks there and back
      backward  [d logit | d eps | d class logits] (R, 8) -> gh2 = (. Wh) * m2 -> gh1 = (gh2 W2) * m1, Wh = the three heads' weights stacked

    m_l = [h_l > 0] comes from the saved activations (act 5).  Weight gradients: three products over the R rows (trunk 1, trunk 2, the stacked heads
    -- the heads' `.grad` are row views of one (8, 256) product) and three over the U rows, the latter added with the factor a."""

    @classmethod
    def describe(cls, disc, rows, n_u, lib=None, prefix="qa_"):
        lins = disc._relu_trunk()
        cols = _capi.MLP_BUF_COLS
        if not ENABLED or lins is None or len(lins) != 2 or rows > MAX_ROWS:
            return None
        l1, l2 = lins
        k0, w1, w2 = l1.in_features, l1.out_features, l2.out_features
        heads = (disc.linear, disc.encoder_eps, disc.classifier)
        nh = sum(h.out_features for h in heads)
        if (k0 > cols[2] or w1 > cols[1] or w2 > cols[2] or nh > 8 or disc.linear.out_features != 1 or disc.encoder_eps.out_features != 1
                or any(h.in_features != w2 for h in heads) or l2.in_features != w1 or any(h.bias is None for h in heads) or n_u > rows):
            return None
        self = cls()
        self.disc, self.rows, self.n_u, self.dims = disc, rows, n_u, (k0, w1, w2, nh)
        self.heads = heads
        dev = l1.weight.device
        z = lambda *s: torch.zeros(*s, dtype=torch.float32, device=dev)
        H1, H2 = 0, _pad4(w1)                                   # activation tape columns
        self.tape = z(rows, H1 + _pad4(w1) + _pad4(w2)); self.tcol = {"h1": H1, "h2": H2}
        V2, V1, U1, U2 = 0, _pad4(w2), _pad4(w2) + _pad4(w1), _pad4(w2) + 2 * _pad4(w1)
        self.vu = z(n_u, U2 + _pad4(w2)); self.vcol = {"v2": V2, "v1": V1, "u1": U1, "u2": U2}
        GH2, GH1 = 0, _pad4(w2)
        self.gt = z(rows, GH1 + _pad4(w1)); self.gcol = {"gh2": GH2, "gh1": GH1}
        self.g = z(n_u, k0)                                      # d logit / d x on the unlabelled rows (dense: qa_disc_step_tail reads it as it is)
        self.gin = z(rows, 8)
        self.ones = z(n_u, 4); self.ones[:, 0] = 1.0
        self.d, self.eps, self.logits = z(rows, 1), z(rows, 1), z(rows, disc.classifier.out_features)
        self.wh, self.bh_dummy = z(8, w2), None                  # the heads' weights stacked (refreshed per step: they change with every optimiser step)
        R = 2                                                    # ReLU
        DR = _capi.MLP_ACT_RELU_GRAD
        f = _Program(lib, prefix)
        f.layer(0, 0, k0, 1, 0, w1, R, l1.weight, l1.bias, save=(0, H1))
        f.layer(1, 0, w1, 2, 0, w2, R, l2.weight, l2.bias, save=(0, H2))
        f.layer(2, 0, w2, -1, 0, 1, 0, disc.linear.weight, disc.linear.bias, out=(1, 0))
        f.layer(2, 0, w2, -1, 0, 1, 0, disc.encoder_eps.weight, disc.encoder_eps.bias, out=(2, 0))
        f.layer(2, 0, w2, -1, 0, disc.classifier.out_features, 0, disc.classifier.weight, disc.classifier.bias, out=(3, 0))
        f.finish()
        p = _Program(lib, prefix)                                # outputs: 0 = vu (written), 1 = the U rows of the activation tape (read), 2 = g
        p.layer(0, 0, 1, 2, 0, w2, DR, disc.linear.weight, None, save=(0, V2), aux=(1, H2), transposed=True)
        p.layer(2, 0, w2, 1, 0, w1, DR, l2.weight, None, save=(0, V1), aux=(1, H1), transposed=True)
        p.layer(1, 0, w1, 2, 0, k0, 0, l1.weight, None, save=(2, 0), transposed=True)
        p.layer(2, 0, k0, 1, 0, w1, DR, l1.weight, None, save=(0, U1), aux=(1, H1))
        p.layer(1, 0, w1, -1, 0, w2, DR, l2.weight, None, out=(0, U2), aux=(1, H2))
        p.finish()
        b = _Program(lib, prefix)   # input tile = d loss / d logit; outputs: 0 = gradient tape (written), 1 = activation tape, 2 = d eps, 3 = d class logits (read), 4 = the
                                    # three side by side as one (R, 8) matrix (written: the stacked heads' weight-gradient product reads it)
        b.copy(0, 0, 1, 0, 1, save=(4, 0))
        b.load((2, 0), 1, 1, 1, save=(4, 1))
        b.load((3, 0), 1, 2, nh - 2, save=(4, 2))
        b.layer(1, 0, 8, 2, 0, w2, DR, self.wh, None, save=(0, GH2), aux=(1, H2), transposed=True)
        b.layer(2, 0, w2, -1, 0, w1, DR, l2.weight, None, out=(0, GH1), aux=(1, H1), transposed=True)
        b.finish()
        self.fwd, self.pen, self.bwd = f, p, b
        self._packer = _Packer([f, p, b])
        # opt-in side streams (QA_TRAIN_CHAIN_SIDES=1; measured slower): 3 = what `beside()` is handed.  (The three optimisers were once forked onto 1 / 2 as
        # well -- they share the trunk's parameters and must run one after the other: removed in r6.)
        self.sides = _Sides(dev, 4) if (SIDE_STREAMS and prefix == "qa_") else None
        # weight-gradient products: (rows, g tensor, g col, n, x tensor, x col, k)
        self._wg = {}
        for name, (r, n, k) in dict(w1=(rows, w1, k0), w2=(rows, w2, w1), wh=(rows, 8, w2), p1=(n_u, w1, k0), p2=(n_u, w2, w1), p3=(n_u, w2, 1)).items():
            nb = int(f._fn("linear_backward_weight_batch_scratch_bytes")(r, k, n))
            self._wg[name] = (torch.zeros(nb // 4 + 4, dtype=torch.float32, device=dev), nb, z(n, k), z(n))
        return self

    def pack(self):
        with torch.no_grad():
            torch.cat([h.weight for h in self.heads], dim=0, out=self.wh[:self.dims[3]])
        self._packer.pack()

    def _layouts(self):
        """product name -> qa_linear_backward_weight_batch_layout (parts, weight slab stride, bias parts, bias stride, bias slabs' offset; floats)"""
        if getattr(self, "_lay", None) is None:
            k0, w1, w2, nh = self.dims
            R, U = self.rows, self.n_u
            self._lay = {}
            for name, (r, n, k) in dict(w1=(R, w1, k0), w2=(R, w2, w1), wh=(R, 8, w2), p1=(U, w1, k0), p2=(U, w2, w1), p3=(U, w2, 1)).items():
                lay = (C.c_int64 * 5)()
                _ok(self.fwd._fn("linear_backward_weight_batch_layout")(r, k, n, lay), "linear_backward_weight_batch_layout", self.fwd)
                self._lay[name] = [int(v) for v in lay]
        return self._lay

    def forward(self, x):
        """-> logit (R, 1), epsilon (R, 1), class LOGITS (R, dim_c); then `penalty_gradient()`"""
        assert x.shape[0] == self.rows and x.stride(1) == 1 and x.shape[1] >= self.dims[0]
        self._x = x
        self.fwd.launch(x, self.dims[0], [self.tape, self.d, self.eps, self.logits])
        return self.d, self.eps, self.logits

    def penalty_gradient(self):
        """d logit / d x on the unlabelled expert rows (the LAST n_u rows of x), (n_u, input_dim)"""
        self.pen.launch(self.ones, 4, [self.vu, self.tape[self.rows - self.n_u:], self.g])
        return self.g

    def wait_penalty(self):
        """before the first reader of `penalty_gradient()`'s result on the calling stream"""
        return None          # (the penalty path runs on the calling stream: nothing to wait for)

    def beside(self, fn):
        """run `fn` on a side stream from here; `backward()` joins it"""
        if self.sides:
            with self.sides.fork(3):
                fn()
        else:
            fn()

    def backward(self, g_d, g_eps, g_logits, penalty_coef, finish=True):
        """gradients at the three heads' outputs (R rows) + the penalty's coefficient c (loss term c * mean_U |g|^2) -> `.grad` of the 10 parameters.
        finish=False leaves the six products IN PARTS and returns {parameter: record} for fused.StackedAdam.step, whose launch adds the parts (the
        penalty's with the factor a), writes `.grad` and steps the optimisers: no reduction launch, no multi-tensor adds here."""
        k0, w1, w2, nh = self.dims
        l1, l2 = self.disc._relu_trunk()
        gin = self.gin
        f32c = lambda t, shape: (t if (t.dtype == torch.float32 and t.is_contiguous()) else t.contiguous().float()).view(shape)
        g_d, g_eps, g_logits = f32c(g_d, (self.rows, 1)), f32c(g_eps, (self.rows, 1)), f32c(g_logits, (self.rows, nh - 2))
        self.bwd.launch(g_d, 1, [self.gt, self.tape, g_eps, g_logits, gin])
        R, U = self.rows, self.n_u
        # six products -- trunk 1, trunk 2, the stacked heads over the R rows; the penalty's three over the U rows -- in one call (<= 4 launches + one
        # reduction for all of them)
        jobs = [("w1", R, self.gt, self.gcol["gh1"], w1, self._x, 0, k0), ("w2", R, self.gt, self.gcol["gh2"], w2, self.tape, self.tcol["h1"], w1),
                ("wh", R, gin, 0, 8, self.tape, self.tcol["h2"], w2), ("p1", U, self.vu, self.vcol["v1"], w1, self.g, 0, k0),
                ("p2", U, self.vu, self.vcol["v2"], w2, self.vu, self.vcol["u1"], w1), ("p3", U, self.vu, self.vcol["u2"], w2, self.ones, 0, 1)]
        descs = (_capi.QaWgradDesc * len(jobs))()
        for i, (name, rows, g, g0, n, x, x0, k) in enumerate(jobs):
            scratch, nb, gw, gb = self._wg[name]
            descs[i] = _capi.QaWgradDesc(g.data_ptr() + 4 * g0, g.stride(0), x.data_ptr() + 4 * x0, x.stride(0), gw.data_ptr() if finish else None,
                                         gb.data_ptr() if finish else None, rows, k, n, scratch.data_ptr(), nb)
        _ok(self.fwd._fn("linear_backward_weight_batch")(descs, len(jobs), self.fwd._stream(gin)), "linear_backward_weight_batch", self.fwd)
        (gw1, gb1), (gw2, gb2), (gwh, gbh) = (self._wg[k_][2:] for k_ in ("w1", "w2", "wh"))
        t1, t2, t3 = (self._wg[k_][2] for k_ in ("p1", "p2", "p3"))
        a = 2.0 * float(penalty_coef) / U
        if not finish:
            lay = self._layouts()

            def rec(grad, name, col0=0, bias=False, pen=None, tmp=None):
                L = lay[name]
                sc = self._wg[name][0]
                d = dict(grad=grad, src1=sc[(L[4] if bias else 0) + col0:], stride1=L[3] if bias else L[1], parts1=L[2] if bias else L[0])
                if pen is not None:
                    P = lay[pen]
                    d.update(src2=self._wg[pen][0], stride2=P[1], parts2=P[0], alpha2=a, tmp=tmp)
                return d
            out = {l1.weight: rec(gw1, "w1", pen="p1", tmp=t1), l1.bias: rec(gb1, "w1", bias=True),
                   l2.weight: rec(gw2, "w2", pen="p2", tmp=t2), l2.bias: rec(gb2, "w2", bias=True)}
            r = 0
            for h in self.heads:
                n = h.out_features
                first = h is self.disc.linear
                out[h.weight] = rec(gwh[r:r + n], "wh", col0=r * w2, pen="p3" if first else None, tmp=t3 if first else None)
                out[h.bias] = rec(gbh[r:r + n], "wh", col0=r, bias=True)
                r += n
            return out
        with torch.no_grad():
            torch._foreach_add_([gw1, gw2, gwh[0]], [t1, t2, t3.view(-1)], alpha=a)
        l1.weight.grad, l1.bias.grad, l2.weight.grad, l2.bias.grad = gw1, gb1, gw2, gb2
        r = 0
        for h in self.heads:
            n = h.out_features
            h.weight.grad, h.bias.grad = gwh[r:r + n], gbh[r:r + n]
            r += n


class TscTrainChain:
    """PpoTrainChain's counterpart for the task-level learner's minibatch step (tsc/rsl_rl/algorithms/ppo.py:222-262 through autograd; modules
    tsc/rsl_rl/modules/actor_critic.py:59-284): estimator 57-128-64-4, scan encoder 132-128-64-32 (tanh), privileged encoder 29-64-29, trunk
    130-512-256-128 with the categorical (3) and the Gaussian (18) head, critic 800-512-256-128-1 -- 17 Linear layers.  forward(obs) -> (est, logits,
    mean, value, priv_latent); backward(g_est, dlogits, dmean, dvalue, g_priv) -> `.grad` of the 34 parameters (the two heads' as row views of one
    stacked (21, 128) product).  6,144-row steps (1024 envs per GPU, BASELINE configs[3]'s share) are inside the row limit."""

    @classmethod
    def describe(cls, ac, estimator, rows, est_in, lib=None, prefix="qa_"):
        import torch.nn as nn
        actor = ac.actor
        lin = fused.PolicyChain._linears

        def stack(seq, tanh_last=False):
            mods = list(seq) if isinstance(seq, nn.Sequential) else None
            if mods is None:
                return None
            if tanh_last:
                if not isinstance(mods[-1], nn.Tanh):
                    return None
                body = lin(nn.Sequential(*mods[:-1]))
                if body is None or body[-1][1] != 0:
                    return None
                body[-1][1] = 3
                return body
            return lin(seq)
        if not ENABLED or rows > MAX_ROWS or not actor.if_scan_encode or isinstance(actor.priv_encoder, nn.Identity) or not isinstance(ac.std, nn.Parameter):
            return None
        st = dict(scan=stack(actor.scan_encoder, True), priv=stack(actor.priv_encoder), trunk=stack(actor.actor_trunk), critic=stack(ac.critic), est=stack(estimator.estimator))
        if any(v is None for v in st.values()):
            return None
        acts = {k: [a for _, a in v] for k, v in st.items()}
        if (acts["scan"] != [1, 1, 3] or acts["priv"] != [1, 1] or acts["trunk"] != [1, 1, 1] or acts["critic"] != [1, 1, 1, 0] or acts["est"] != [1, 1, 0]):
            return None
        a, n_scan, n_exp, n_lat = actor.num_prop, actor.num_scan, actor.num_priv_explicit, actor.num_priv_latent
        scan0, exp0, lat0 = a, a + n_scan, a + n_scan + n_exp
        (s1, s2, s3), (p1, p2), (t1, t2, t3), (c1, c2, c3, c4), (e1, e2, e3) = ([l for l, _ in st[k]] for k in ("scan", "priv", "trunk", "critic", "est"))
        hd, hc = actor.actor_d, actor.actor_c
        n_obs, n_lats = c1.in_features, s3.out_features
        n_in = a + n_lats + n_exp + n_lat
        nh = hd.out_features + hc.out_features
        cols = _capi.MLP_BUF_COLS
        sw, pw, tw, cw, ew = ([l.out_features for l in ls] for ls in ((s1, s2, s3), (p1, p2), (t1, t2, t3), (c1, c2, c3, c4), (e1, e2, e3)))
        ok = (n_obs <= cols[0] and lat0 + n_lat <= n_obs and t1.in_features == n_in and n_in <= cols[2] and tw[0] <= cols[1] and tw[1] <= cols[2] and tw[2] <= cols[3]
              and cw[0] <= cols[1] and cw[1] <= cols[2] and cw[2] <= cols[3] and cw[3] == 1 and sw[0] <= cols[3] and sw[1] <= cols[1] and n_scan <= cols[1]
              and pw[0] <= cols[3] and pw[1] == n_lat and ew[0] <= cols[3] and ew[1] <= cols[2] and ew[2] == n_exp and e1.in_features == est_in and est_in <= a
              and s1.in_features == n_scan and p1.in_features == n_lat and nh <= 24 and hd.in_features == tw[2] and hc.in_features == tw[2] and _pad4(nh) <= cols[3])
        if not ok:
            return None
        self = cls()
        self.ac, self.estimator, self.rows = ac, estimator, rows
        self.dims = dict(a=a, n_scan=n_scan, n_exp=n_exp, n_lat=n_lat, n_obs=n_obs, n_in=n_in, nh=nh, nd=hd.out_features, nc=hc.out_features, est_in=est_in,
                         scan0=scan0, exp0=exp0, lat0=lat0)
        t, off = {}, 0
        for name, w in (("tin", n_in), ("s1", sw[0]), ("s2", sw[1]), ("p1", pw[0]), ("t1", tw[0]), ("t2", tw[1]), ("t3", tw[2]), ("c1", cw[0]), ("c2", cw[1]), ("c3", cw[2]),
                        ("e1", ew[0]), ("e2", ew[1])):
            t[name] = (off, w); off += _pad4(w)
        self.tape_cols, self.t = off, t
        g, off = {}, 0
        for name, w in (("t3", tw[2]), ("t2", tw[1]), ("t1", tw[0]), ("s3", sw[2]), ("s2", sw[1]), ("s1", sw[0]), ("p2", n_lat), ("p1", pw[0]), ("c3", cw[2]), ("c2", cw[1]),
                        ("c1", cw[0]), ("e2", ew[1]), ("e1", ew[0])):
            g[name] = (off, w); off += _pad4(w)
        self.gtape_cols, self.g = off, g
        E, D, TANH, DTANH = 1, _capi.MLP_ACT_ELU_GRAD, 3, _capi.MLP_ACT_TANH_GRAD
        T, LG, MU, VAL, EST, PRIV = 0, 1, 2, 3, 4, 5
        tin = t["tin"][0]
        c_lats, c_exp, c_lat = a, a + n_lats, a + n_lats + n_exp          # columns of the trunk's input row
        f = _Program(lib, prefix)
        # critic first: it may use every scratch buffer
        f.layer(0, 0, n_obs, 1, 0, cw[0], E, c1.weight, c1.bias, save=(T, t["c1"][0]))
        f.layer(1, 0, cw[0], 2, 0, cw[1], E, c2.weight, c2.bias, save=(T, t["c2"][0]))
        f.layer(2, 0, cw[1], 3, 0, cw[2], E, c3.weight, c3.bias, save=(T, t["c3"][0]))
        f.layer(3, 0, cw[2], -1, 0, 1, 0, c4.weight, c4.bias, out=(VAL, 0))
        f.layer(0, 0, est_in, 3, 0, ew[0], E, e1.weight, e1.bias, save=(T, t["e1"][0]))
        f.layer(3, 0, ew[0], 2, 0, ew[1], E, e2.weight, e2.bias, save=(T, t["e2"][0]))
        f.layer(2, 0, ew[1], -1, 0, n_exp, 0, e3.weight, e3.bias, out=(EST, 0))
        # the trunk's input row in buffer 2: proprioception | scan latent | true privileged-explicit state | privileged latent
        f.copy(0, 0, 2, 0, a, save=(T, tin))
        f.copy(0, scan0, 1, 0, n_scan)                                                # (a layer's source starts 16-byte aligned)
        f.layer(1, 0, n_scan, 3, 0, sw[0], E, s1.weight, s1.bias, save=(T, t["s1"][0]))
        f.layer(3, 0, sw[0], 1, 0, sw[1], E, s2.weight, s2.bias, save=(T, t["s2"][0]))
        f.layer(1, 0, sw[1], 2, c_lats, n_lats, TANH, s3.weight, s3.bias, save=(T, tin + c_lats))
        f.copy(0, exp0, 2, c_exp, n_exp, save=(T, tin + c_exp))
        f.copy(0, lat0, 1, 0, n_lat)
        f.layer(1, 0, n_lat, 3, 0, pw[0], E, p1.weight, p1.bias, save=(T, t["p1"][0]))
        f.layer(3, 0, pw[0], 2, c_lat, n_lat, E, p2.weight, p2.bias, save=(T, tin + c_lat))
        f.copy(2, c_lat, 3, 0, n_lat, save=(PRIV, 0))                                 # the privileged latent once more, dense, for the regulariser's kernel
        f.layer(2, 0, n_in, 1, 0, tw[0], E, t1.weight, t1.bias, save=(T, t["t1"][0]))
        f.layer(1, 0, tw[0], 2, 0, tw[1], E, t2.weight, t2.bias, save=(T, t["t2"][0]))
        f.layer(2, 0, tw[1], 3, 0, tw[2], E, t3.weight, t3.bias, save=(T, t["t3"][0]))
        f.layer(3, 0, tw[2], -1, 0, hd.out_features, 0, hd.weight, hd.bias, out=(LG, 0))
        f.layer(3, 0, tw[2], -1, 0, hc.out_features, 0, hc.weight, hc.bias, out=(MU, 0))
        f.finish()
        # ---- backward.  Input tile = d loss / d gait logits; outputs: 0 = gradient tape (written), 1 = activation tape, 2 = d mean, 3 = d latent, 4 = d value, 5 = d
        # estimate (read), 6 = [d logits | d mean] side by side (written: the stacked heads' weight-gradient product reads it)
        G, A, DM, DP, DV, DE, GH = 0, 1, 2, 3, 4, 5, 6
        dev = t1.weight.device
        self.wh = torch.zeros(_pad4(nh), tw[2], dtype=torch.float32, device=dev)      # the two heads' weights stacked; refreshed by pack()
        nd = hd.out_features
        b = _Program(lib, prefix)
        b.copy(0, 0, 3, 0, nd, save=(GH, 0))
        b.load((DM, 0), 3, nd, hc.out_features, save=(GH, nd))
        b.layer(3, 0, _pad4(nh), 1, 0, tw[2], D, self.wh, None, save=(G, g["t3"][0]), aux=(A, t["t3"][0]), transposed=True)
        b.layer(1, 0, tw[2], 2, 0, tw[1], D, t3.weight, None, save=(G, g["t2"][0]), aux=(A, t["t2"][0]), transposed=True)
        b.layer(2, 0, tw[1], 1, 0, tw[0], D, t2.weight, None, save=(G, g["t1"][0]), aux=(A, t["t1"][0]), transposed=True)
        b.layer(1, 0, tw[0], 2, 0, n_in, 0, t1.weight, None, transposed=True)                         # d loss / d trunk input: its scan-latent and latent columns go on
        b.grad(2, c_lats, 3, 0, n_lats, act=DTANH, aux=(A, tin + c_lats), save=(G, g["s3"][0]))
        b.layer(3, 0, n_lats, 1, 0, sw[1], D, s3.weight, None, save=(G, g["s2"][0]), aux=(A, t["s2"][0]), transposed=True)
        b.layer(1, 0, sw[1], 3, 0, sw[0], D, s2.weight, None, save=(G, g["s1"][0]), aux=(A, t["s1"][0]), transposed=True)
        b.load((DP, 0), 1, 0, n_lat)
        b.grad(2, c_lat, 1, 0, n_lat, act=D, aux=(A, tin + c_lat), add=True, save=(G, g["p2"][0]))
        b.layer(1, 0, n_lat, 3, 0, pw[0], D, p2.weight, None, save=(G, g["p1"][0]), aux=(A, t["p1"][0]), transposed=True)
        b.load((DV, 0), 1, 0, 1)
        b.layer(1, 0, 1, 3, 0, cw[2], D, c4.weight, None, save=(G, g["c3"][0]), aux=(A, t["c3"][0]), transposed=True)
        b.layer(3, 0, cw[2], 2, 0, cw[1], D, c3.weight, None, save=(G, g["c2"][0]), aux=(A, t["c2"][0]), transposed=True)
        b.layer(2, 0, cw[1], 1, 0, cw[0], D, c2.weight, None, save=(G, g["c1"][0]), aux=(A, t["c1"][0]), transposed=True)
        b.load((DE, 0), 3, 0, n_exp)
        b.layer(3, 0, n_exp, 2, 0, ew[1], D, e3.weight, None, save=(G, g["e2"][0]), aux=(A, t["e2"][0]), transposed=True)
        b.layer(2, 0, ew[1], 3, 0, ew[0], D, e2.weight, None, save=(G, g["e1"][0]), aux=(A, t["e1"][0]), transposed=True)
        b.finish()
        self.fwd, self.bwd = f, b
        self._packer = _Packer([f, b])
        z = lambda *s: torch.zeros(*s, dtype=torch.float32, device=dev)
        self.tape, self.gtape, self.gh = z(rows, self.tape_cols), z(rows, self.gtape_cols), z(rows, _pad4(nh))
        self.logits, self.mean, self.value, self.est, self.priv = z(rows, nd), z(rows, hc.out_features), z(rows, 1), z(rows, n_exp), z(rows, n_lat)
        tc = lambda name: ("tape", t[name][0], t[name][1])
        gcol = lambda name: ("gtape", g[name][0], g[name][1])
        # (module or None for the stacked heads, gradient columns, input columns)
        self.wgrads = [(c1, gcol("c1"), ("obs", 0, n_obs)), (c2, gcol("c2"), tc("c1")), (c3, gcol("c3"), tc("c2")), (c4, ("dvalue", 0, 1), tc("c3")),
                       (t1, gcol("t1"), ("tape", tin, n_in)), (t2, gcol("t2"), tc("t1")), (t3, gcol("t3"), tc("t2")), (None, ("gh", 0, nh), tc("t3")),
                       (s1, gcol("s1"), ("obs", scan0, n_scan)), (s2, gcol("s2"), tc("s1")), (s3, gcol("s3"), tc("s2")),
                       (p1, gcol("p1"), ("obs", lat0, n_lat)), (p2, gcol("p2"), tc("p1")),
                       (e1, gcol("e1"), ("obs", 0, est_in)), (e2, gcol("e2"), tc("e1")), (e3, ("g_est", 0, n_exp), tc("e2"))]
        self._wg = []
        for lin_, (_, _, n), (_, _, k) in self.wgrads:
            nb = int(f._fn("linear_backward_weight_batch_scratch_bytes")(rows, k, n))
            lay = (C.c_int64 * 5)()
            _ok(f._fn("linear_backward_weight_batch_layout")(rows, k, n, lay), "linear_backward_weight_batch_layout", f)
            self._wg.append((torch.zeros(nb // 4 + 4, dtype=torch.float32, device=dev), nb, [int(v) for v in lay], z(n, k), z(n)))
        return self

    def pack(self):
        with torch.no_grad():
            torch.cat([self.ac.actor.actor_d.weight, self.ac.actor.actor_c.weight], dim=0, out=self.wh[:self.dims["nh"]])
        self._packer.pack()

    def forward(self, obs):
        assert obs.shape[0] == self.rows and obs.stride(1) == 1 and obs.shape[1] >= self.dims["n_obs"]
        self._obs = obs
        self.fwd.launch(obs, self.dims["n_obs"], [self.tape, self.logits, self.mean, self.value, self.est, self.priv])
        return self.est, self.logits, self.mean, self.value, self.priv

    def backward(self, g_est, dlogits, dmean, dvalue, g_priv, defer=True):
        d = self.dims
        f32c = lambda x, shape: (x if (x.dtype == torch.float32 and x.is_contiguous()) else x.contiguous().float()).view(shape)
        dlogits, dmean = f32c(dlogits, (self.rows, d["nd"])), f32c(dmean, (self.rows, d["nc"]))
        dvalue, g_est, g_priv = f32c(dvalue, (self.rows, 1)), f32c(g_est, (self.rows, d["n_exp"])), f32c(g_priv, (self.rows, d["n_lat"]))
        self.bwd.launch(dlogits, d["nd"], [self.gtape, self.tape, dmean, g_priv, dvalue, g_est, self.gh])
        src = {"tape": self.tape, "gtape": self.gtape, "gh": self.gh, "dvalue": dvalue, "g_est": g_est, "obs": self._obs}
        defer = defer and dlogits.is_cuda and fused.ENABLED and os.environ.get("QA_DEFER_GRAD_FINISH", "1") != "0"
        descs = (_capi.QaWgradDesc * len(self.wgrads))()
        for i, ((lin_, (gs, g0, gn), (xs, x0, xk)), (scratch, nb, lay, gw, gb)) in enumerate(zip(self.wgrads, self._wg)):
            gt, xt = src[gs], src[xs]
            descs[i] = _capi.QaWgradDesc(gt.data_ptr() + 4 * g0, gt.stride(0), xt.data_ptr() + 4 * x0, xt.stride(0), None if defer else gw.data_ptr(),
                                         None if defer else gb.data_ptr(), self.rows, xk, gn, scratch.data_ptr(), nb)
        _ok(self.bwd._fn("linear_backward_weight_batch")(descs, len(self.wgrads), self.bwd._stream(dlogits)), "linear_backward_weight_batch", self.bwd)
        hd, hc = self.ac.actor.actor_d, self.ac.actor.actor_c
        for (lin_, (_, _, n), (_, _, k)), (scratch, nb, lay, gw, gb) in zip(self.wgrads, self._wg):
            # the stacked heads' product: each head's gradient is a run of its rows -- of the finished product and of every part alike
            pieces = [(lin_, 0, n)] if lin_ is not None else [(hd, 0, d["nd"]), (hc, d["nd"], d["nc"])]
            for m, r0, rn in pieces:
                m.weight.grad, m.bias.grad = gw[r0:r0 + rn], gb[r0:r0 + rn]
                if defer:
                    fused.register_grad_parts(m.weight, scratch[r0 * k:], lay[0], lay[1], m.weight.grad)
                    fused.register_grad_parts(m.bias, scratch[lay[4] + r0:], lay[2], lay[3], m.bias.grad)
