"""qa_mlp_pack / qa_mlp_forward (policy inference as one launch) against the modules they stand for.

CPU: PolicyChain.describe() + the oracle's plain-C twin (qo_mlp_*) vs. Estimator / ActorCritic evaluated by torch --
pins the op list (host logic) and the twin.  GPU: the HIP kernel vs. the twin and vs. torch on the device."""
import ctypes as C

import numpy as np
import pytest
import torch

from quadrupedal_agility_amd import _capi
from quadrupedal_agility_amd.rsl_rl.algorithms.fused import PolicyChain
from quadrupedal_agility_amd.rsl_rl.modules import ActorCritic, Estimator

DIMS = dict(num_prop=57, num_hist=10, num_explicit=4, num_latent=29, num_command=11)


def modules(seed=0, priv_dims=(64,), hidden=(512, 256, 128), est_hidden=(128, 64)):
    torch.manual_seed(seed)
    n_obs = 57 + 4 + 29 + 570 + 11
    ac = ActorCritic(57 + 4 + 29 + 11, n_obs, 12, actor_hidden_dims=list(hidden), critic_hidden_dims=list(hidden),
                     priv_encoder_dims=list(priv_dims), activation="elu", train_with_estimated_latent=True, **DIMS)
    est = Estimator(57, 4, hidden_dims=list(est_hidden))
    with torch.no_grad():
        for p in list(ac.parameters()) + list(est.parameters()):
            if p.dim() == 1:
                p.uniform_(-0.2, 0.2)          # biases are zero-initialised in the actor trunk: make them count
    return ac, est, n_obs


def torch_reference(ac, est, obs, use_estimator, hist_encoding=False):
    with torch.no_grad():
        x = torch.cat([obs[:, :57], est(obs[:, :57]), obs[:, 61:]], dim=-1) if use_estimator else obs
        return ac._actor_mean(x, hist_encoding), ac.evaluate(obs)


def run_oracle(chain, obs):
    from tests.oracle_lib import load_oracle
    lib = load_oracle()
    n = obs.shape[0]
    with torch.no_grad():
        w, b = chain._ptr_arrays()          # evaluates the structured-matrix providers of the history-encoder variant
    packed = np.zeros(chain.packed_floats, np.float32)
    assert lib.qo_mlp_packed_floats(chain.ops, chain.n_ops) == chain.packed_floats
    assert lib.qo_mlp_pack(chain.ops, chain.n_ops, w, b, packed.ctypes.data, packed.size, None) == 0
    mean, value = np.zeros((n, 12), np.float32), np.zeros((n, 1), np.float32)
    x = np.ascontiguousarray(obs.numpy())
    outs = (C.c_void_p * 2)(mean.ctypes.data, value.ctypes.data)
    strides = (C.c_int64 * 2)(12, 1)
    assert lib.qo_mlp_forward(x.ctypes.data, x.shape[1], n, x.shape[1], chain.ops, chain.n_ops, packed.ctypes.data, outs, strides, 2, None) == 0
    return mean, value


@pytest.mark.parametrize("use_estimator", [True, False])
def test_description_and_twin_match_torch_modules(use_estimator):
    ac, est, n_obs = modules()
    chain = PolicyChain.describe(ac, est, use_estimator)
    assert chain is not None and chain.n_ops == (16 if use_estimator else 14)
    obs = torch.randn(37, n_obs)
    mean, value = run_oracle(chain, obs)
    rm, rv = torch_reference(ac, est, obs, use_estimator)
    np.testing.assert_allclose(mean, rm.numpy(), rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(value, rv.numpy(), rtol=1e-4, atol=2e-5)


@pytest.mark.parametrize("use_estimator,with_critic", [(True, True), (False, False)])
def test_history_encoder_variant_matches_torch_modules(use_estimator, with_critic):
    """the variant of DAgger rollouts / play.py / the exported policy: per-frame Linear, two Conv1d and Flatten of the history
    encoder as four layer ops with structured matrices"""
    ac, est, n_obs = modules(seed=5)
    chain = PolicyChain.describe(ac, est, use_estimator, hist_encoding=True, with_critic=with_critic)
    assert chain is not None and chain.n_ops <= _capi.MLP_MAX_OPS
    obs = torch.randn(21, n_obs)
    mean, value = run_oracle(chain, obs)
    rm, rv = torch_reference(ac, est, obs, use_estimator, hist_encoding=True)
    np.testing.assert_allclose(mean, rm.numpy(), rtol=1e-4, atol=2e-5)
    if with_critic:
        np.testing.assert_allclose(value, rv.numpy(), rtol=1e-4, atol=2e-5)


def test_description_refuses_what_the_kernel_cannot_hold():
    ac, est, _ = modules(hidden=(1024, 256, 128))                    # 1024 > widest scratch buffer
    assert PolicyChain.describe(ac, est, True) is None
    ac, est, _ = modules()
    ac.actor_trunk[1] = torch.nn.Tanh()
    assert PolicyChain.describe(ac, est, True) is None
    ac, est, _ = modules(priv_dims=())                                # Identity encoder: latent copied through
    chain = PolicyChain.describe(ac, est, True)
    assert chain is not None
    obs = torch.randn(5, 671)
    mean, value = run_oracle(chain, obs)
    rm, rv = torch_reference(ac, est, obs, True)
    np.testing.assert_allclose(mean, rm.numpy(), rtol=1e-4, atol=2e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("n", [1, 16, 100, 4096])
def test_hip_chain_matches_twin_and_torch(n):
    ac, est, n_obs = modules(seed=3)
    obs = torch.randn(n, n_obs) * 1.5
    chain_cpu = PolicyChain.describe(ac, est, True)
    om, ov = run_oracle(chain_cpu, obs[:256])
    ac, est = ac.cuda(), est.cuda()
    chain = PolicyChain.describe(ac, est, True)
    chain.pack()
    g = obs.cuda()
    mean, value = chain.forward(g)
    torch.cuda.synchronize()
    rm, rv = torch_reference(ac, est, g, True)
    np.testing.assert_allclose(mean.cpu().numpy(), rm.cpu().numpy(), rtol=2e-4, atol=5e-5)
    np.testing.assert_allclose(value.cpu().numpy(), rv.cpu().numpy(), rtol=2e-4, atol=5e-5)
    np.testing.assert_allclose(mean.cpu().numpy()[:256], om, rtol=2e-4, atol=5e-5)
    np.testing.assert_allclose(value.cpu().numpy()[:256], ov, rtol=2e-4, atol=5e-5)
    # weights change -> repack -> new outputs; strided input rows (a view into a wider arena)
    with torch.no_grad():
        ac.actor_head.weight.mul_(0.5); ac.actor_head.bias.mul_(0.5)
    chain.pack()
    wide = torch.zeros(n, n_obs + 9, device="cuda"); wide[:, :n_obs] = g
    m2, _ = chain.forward(wide[:, :n_obs])
    np.testing.assert_allclose(m2.cpu().numpy(), 0.5 * rm.cpu().numpy(), rtol=2e-4, atol=5e-5)


def _strands(lib, prefix, chain, max_strands):
    out = (C.c_int32 * chain.n_ops)()
    ns = getattr(lib, prefix + "mlp_strands")(chain.ops, chain.n_ops, max_strands, out)
    return ns, list(out)


def test_strand_partition_of_the_chains_host_side():
    """r5 / ABI 15: qa_mlp_strands (host-only, no GPU needed) against the oracle's restatement on every chain the runners describe, and what the
    partition has to mean: an op and the last writer of every scratch column it reads are in one strand; the critic runs beside the actor."""
    from quadrupedal_agility_amd import _capi
    from tests.oracle_lib import load_oracle
    qa, qo = _capi.load_library(), load_oracle()
    ac, est, _ = modules(seed=1)
    tac, test_ = tsc_modules(seed=1)[:2]
    chains = [PolicyChain.describe(ac, est, True), PolicyChain.describe(ac, est, False), PolicyChain.describe(ac, est, True, hist_encoding=True),
              PolicyChain.describe(ac, est, True, hist_encoding=True, with_critic=False), PolicyChain.describe_task_level(tac, test_, True),
              PolicyChain.describe_task_level(tac, test_, True, part="critic")]
    assert all(c is not None for c in chains)
    for ci, ch in enumerate(chains):
        for ms in (1, 2, 3, 4):
            ns, st = _strands(qa, "qa_", ch, ms)
            ns_o, st_o = _strands(qo, "qo_", ch, ms)
            assert (ns, st) == (ns_o, st_o), (ci, ms)
            assert 1 <= ns <= ms and set(st) == set(range(ns))
            ops = [ch.ops[i] for i in range(ch.n_ops)]
            for i, o in enumerate(ops):                       # flow dependences stay inside a strand
                rd = o.k if o.kind == _capi.MLP_LAYER else o.n
                if o.src_buf <= 0:
                    continue
                for c in range(o.src_col, o.src_col + rd):
                    w = [j for j in range(i) if ops[j].dst_buf == o.src_buf and ops[j].dst_col <= c < ops[j].dst_col + ops[j].n]
                    if w:
                        assert st[w[-1]] == st[i], (ci, ms, i, w[-1])
    # SSInfoGAIL.act: the critic's four layers (the ops that end in global output 1) are one strand, the actor's head (output 0) the other
    ns, st = _strands(qa, "qa_", chains[0], 2)
    ops = [chains[0].ops[i] for i in range(chains[0].n_ops)]
    head = {o.out_index: st[i] for i, o in enumerate(ops) if o.kind == _capi.MLP_LAYER and o.dst_buf < 0}
    assert ns == 2 and head[0] != head[1]
    cost = [sum((o.k * o.n if o.kind == _capi.MLP_LAYER else 0) for i, o in enumerate(ops) if st[i] == g) for g in range(2)]
    assert max(cost) < 0.75 * sum(cost)                        # the critic is ~68 % of the chain
    assert _strands(qa, "qa_", chains[3], 4)[0] == 1          # the actor alone is one dependence chain: nothing to split


def _groups(lib, prefix, chain, x_cols):
    st = (C.c_int32 * chain.n_ops)()
    base, stride, lds = (C.c_int32 * 8)(), (C.c_int32 * 8)(), C.c_int32(0)
    rc = getattr(lib, prefix + "mlp_groups")(chain.ops, chain.n_ops, x_cols, st, base, stride, C.byref(lds))
    return rc, list(st), list(base), list(stride), lds.value


def test_two_group_plan_of_the_chains_host_side():
    """r5 / ABI 16: qa_mlp_groups (host-only) -- the LDS plan of the launch that runs a chain's two strands side by side inside every workgroup --
    against the oracle's restatement, and what the plan has to guarantee: the two groups' scratch buffers and the input tile are pairwise
    disjoint, every column an op touches lies inside its buffer's row, every layer's padded k-blocks are read from inside the launch's LDS,
    and the whole fits a CU (160 KB).  SSInfoGAIL.act's chains (both actor variants) must qualify: that is the 4096-env rollout's launch."""
    from tests.oracle_lib import load_oracle
    qa, qo = _capi.load_library(), load_oracle()
    ac, est, n_obs = modules(seed=1)
    tac, test_ = tsc_modules(seed=1)[:2]
    chains = [(PolicyChain.describe(ac, est, True), n_obs), (PolicyChain.describe(ac, est, False), n_obs), (PolicyChain.describe(ac, est, True, hist_encoding=True), n_obs),
              (PolicyChain.describe(ac, est, True, hist_encoding=True, with_critic=False), n_obs), (PolicyChain.describe_task_level(tac, test_, True), 800)]
    applies = []
    for ci, (ch, x_cols) in enumerate(chains):
        got, ref = _groups(qa, "qa_", ch, x_cols), _groups(qo, "qo_", ch, x_cols)
        assert got[0] == ref[0] and got[0] in (0, 1), ci
        applies.append(got[0])
        if _strands(qa, "qa_", ch, 2)[0] != 2:
            assert got[0] == 0
            continue
        assert got == ref, ci
        rc, st, base, stride, lds = got
        ops = [ch.ops[i] for i in range(ch.n_ops)]
        # regions (start, end, group, first op, last op).  Two regions may only overlap inside ONE group, and then only if the buffers' lifetimes do
        # not (a buffer first touched after another's last use is laid over it: the critic's third layer writes where its first layer's output was)
        regions = [(0, 16 * stride[0], -1, -1, ch.n_ops)]                # the shared input tile
        for g in range(2):
            assert base[4 * g] == 0 and stride[4 * g] == stride[0] >= x_cols and stride[4 * g] % 32 == 4
            for b in range(1, 4):
                if stride[4 * g + b]:
                    assert stride[4 * g + b] % 32 == 4
                    touch = [i for i, o in enumerate(ops) if st[i] == g and b in (o.src_buf, o.dst_buf)]
                    regions.append((base[4 * g + b], base[4 * g + b] + 16 * stride[4 * g + b], g, touch[0], touch[-1]))
        assert max(r[1] for r in regions) <= lds, ci
        for x in range(len(regions)):
            for y in range(x + 1, len(regions)):
                A, B = regions[x], regions[y]
                if A[0] < B[1] and B[0] < A[1]:
                    assert A[2] == B[2] >= 0 and (A[4] < B[3] or B[4] < A[3]), (ci, A, B)
        for i, o in enumerate(ops):
            g = st[i]
            rd = o.k if o.kind == _capi.MLP_LAYER else o.n
            assert o.src_col + rd <= stride[4 * g + o.src_buf], (ci, i)
            if o.dst_buf > 0:
                assert o.dst_col + o.n <= stride[4 * g + o.dst_buf], (ci, i)
            if o.kind == _capi.MLP_LAYER:
                assert base[4 * g + o.src_buf] + 15 * stride[4 * g + o.src_buf] + o.src_col + 16 * PolicyChain.k_blocks(o.k, o.n) <= lds, (ci, i)
        assert (lds * 4 <= 160 * 1024 - 256) == bool(rc)
    assert applies == [1, 1, 1, 0, 1]          # the actor alone (no critic) is one strand; everything else the runners launch at full size qualifies


@pytest.mark.gpu
@pytest.mark.parametrize("hist", [False, True])
def test_two_groups_in_a_workgroup_equal_the_one_group_kernel(hist):
    """r5 / ABI 16: at a tile per CU and more (> 2048 rows) the chain's two strands run side by side inside every workgroup (waves 0-3 | waves 4-7,
    own scratch buffers, own barrier; csrc/qa_policy.hip qa_mlp_forward_groups_kernel).  Same arithmetic per output element: bit-identical to the
    one-group kernel on the same rows (4096 rows; 5000 rows with a ragged last tile; 16,384 rows: more tiles than CUs)."""
    lib = _capi.load_library()
    ac, est, n_obs = modules(seed=6)
    ac, est = ac.cuda(), est.cuda()
    chain = PolicyChain.describe(ac, est, True, hist_encoding=hist) if hist else PolicyChain.describe(ac, est, True)
    chain.pack()
    assert _groups(lib, "qa_", chain, n_obs)[0] == 1
    x = (torch.randn(16384, n_obs) * 1.5).cuda()
    prev = lib.qa_mlp_set_groups(1)          # (the two-group launch ships off: measured 2.5 % slower than the one-group kernel)
    try:
        one = {rows: [t.clone() for t in chain.forward(x[:rows].clone())] for rows in (4096, 5000, 16384)}
        lib.qa_mlp_set_groups(2)
        for rows in (4096, 5000, 16384):
            two = chain.forward(x[:rows].clone())
            torch.cuda.synchronize()
            for a, b in zip(two, one[rows]):
                assert torch.equal(a, b), rows
    finally:
        lib.qa_mlp_set_groups(prev)


@pytest.mark.gpu
@pytest.mark.parametrize("hist", [False, True])
def test_few_rows_split_over_strands_equal_the_same_rows_in_a_full_launch(hist):
    """r5: a launch with few 16-row tiles gives each tile to up to 4 workgroups that run the chain's independent strands (critic | estimator ->
    actor) side by side (csrc/qa_policy.hip mlp_strands); at 4096 rows (256 tiles) nothing is split.  Same arithmetic per op: the first rows
    of a full launch and the same rows launched alone (512 rows: 32 tiles x strands; 1040: 65 tiles, a ragged last tile, up to 3 strands;
    2048: 2 strands) must agree bit for bit."""
    ac, est, n_obs = modules(seed=5)
    ac, est = ac.cuda(), est.cuda()
    chain = PolicyChain.describe(ac, est, True, hist_encoding=hist) if hist else PolicyChain.describe(ac, est, True)
    chain.pack()
    g = (torch.randn(4096, n_obs) * 1.5).cuda()
    full = [t.clone() for t in chain.forward(g)]
    for rows in (1, 512, 1040, 2048):
        part = chain.forward(g[:rows].clone())
        torch.cuda.synchronize()
        for a, b in zip(part, full):
            assert torch.equal(a, b[:rows]), rows


@pytest.mark.gpu
@pytest.mark.parametrize("n", [1, 100, 4096])
def test_hip_chain_history_encoder_variant(n):
    ac, est, n_obs = modules(seed=9)
    ac, est = ac.cuda(), est.cuda()
    obs = torch.randn(n, n_obs, device="cuda") * 1.5
    for use_est, with_critic in ((True, True), (False, False)):
        chain = PolicyChain.describe(ac, est, use_est, hist_encoding=True, with_critic=with_critic)
        with torch.inference_mode():
            chain.pack()
            mean, value = chain.forward(obs)
        torch.cuda.synchronize()
        rm, rv = torch_reference(ac, est, obs, use_est, hist_encoding=True)
        np.testing.assert_allclose(mean.cpu().numpy(), rm.cpu().numpy(), rtol=2e-4, atol=5e-5)
        if with_critic:
            np.testing.assert_allclose(value.cpu().numpy(), rv.cpu().numpy(), rtol=2e-4, atol=5e-5)


@pytest.mark.gpu
def test_hip_chain_rejects_malformed_ops():
    lib = _capi.load_library()
    x = torch.zeros(16, 671, device="cuda"); packed = torch.zeros(1 << 20, device="cuda")
    op = _capi.QaMlpOp(kind=_capi.MLP_LAYER, src_buf=0, src_col=61, dst_buf=1, dst_col=0, k=29, n=64, act=1, w_off=0, b_off=4096)
    ops = (_capi.QaMlpOp * 1)(op)
    rc = lib.qa_mlp_forward(x.data_ptr(), 671, 16, 671, ops, 1, packed.data_ptr(), None, None, 0, None)
    assert rc == -1 and b"malformed" in lib.qa_last_error()            # source column not a multiple of 4


@pytest.mark.gpu
@pytest.mark.parametrize("amp", [False, True])
def test_rollout_through_the_chain_matches_the_gemm_path(amp):
    """Two identically seeded runners, one acting through qa_mlp_forward and one through the GEMMs: the first step's action
    mean / value (before the environments can diverge through contact flips) agree to rounding, later steps stay close."""
    from quadrupedal_agility_amd.legged_gym.envs import task_registry
    from tests.test_gpu_train import _make
    got = {}
    for fused in (True, False):
        env, args, t = _make(256, amp)
        torch.manual_seed(7)
        runner, _ = task_registry.make_alg_runner(env, name="go2_locomotion", args=args, train_cfg=t, log_root=None)
        runner.use_fused_policy = fused
        runner.use_rollout_graph = False
        if not hasattr(runner, "_obs_cur"):
            runner._alloc_rollout_state()
        torch.manual_seed(11)
        with torch.inference_mode():
            runner._rollout_steps(False, False, recorded=False)
        torch.cuda.synchronize()
        st = runner.alg.storage
        got[fused] = (st.mu.cpu().clone(), st.values.cpu().clone(), st.actions_log_prob.cpu().clone())
        assert bool(runner._chain and runner._chain.get(False)) == fused
    (m1, v1, l1), (m0, v0, l0) = got[True], got[False]
    np.testing.assert_allclose(m1[0], m0[0], rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(v1[0], v0[0], rtol=1e-4, atol=2e-5)
    if not amp:      # config 3: the fused rollout draws its action noise from the engine's Philox stream, the unfused one from
        close = np.isclose(m1.numpy(), m0.numpy(), rtol=1e-2, atol=1e-2).all(axis=-1)      # torch's generator: only step 0 is comparable
        assert close.mean() > 0.9


def test_discriminator_description_and_twin_match_the_module():
    """PolicyChain.describe_discriminator (ReLU trunk + three heads as global outputs) through the oracle's C twin vs. the module"""
    import types
    from quadrupedal_agility_amd.rsl_rl.algorithms.discriminator import Discriminator
    from tests.oracle_lib import load_oracle
    torch.manual_seed(2)
    env = types.SimpleNamespace(task_obs_weight_decay=False)
    disc = Discriminator(env, 98, 49, 5, 0.02, "MSELoss", None, 0.35, 0.1, 0.25, 0.3, 2, 2, 0.0, [512, 256], "cpu")
    with torch.no_grad():
        for p in disc.parameters():
            if p.dim() == 1:
                p.uniform_(-0.3, 0.3)
    chain = PolicyChain.describe_discriminator(disc)
    assert chain is not None and chain.out_widths == [1, 1, 5] and [int(o.act) for o in chain.ops] == [2, 2, 0, 0, 0]
    lib = load_oracle()
    x = torch.randn(33, 98)
    with torch.no_grad():
        w, b = chain._ptr_arrays()
    packed = np.zeros(chain.packed_floats, np.float32)
    assert lib.qo_mlp_pack(chain.ops, chain.n_ops, w, b, packed.ctypes.data, packed.size, None) == 0
    outs = [np.zeros((33, k), np.float32) for k in (1, 1, 5)]
    ptrs = (C.c_void_p * 3)(*[o.ctypes.data for o in outs]); strides = (C.c_int64 * 3)(1, 1, 5)
    xn = np.ascontiguousarray(x.numpy())
    assert lib.qo_mlp_forward(xn.ctypes.data, 98, 33, 98, chain.ops, chain.n_ops, packed.ctypes.data, ptrs, strides, 3, None) == 0
    with torch.no_grad():
        d, eps, c = disc(x)
    np.testing.assert_allclose(outs[0], d.numpy(), rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(outs[1], eps.numpy(), rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(torch.softmax(torch.from_numpy(outs[2]), -1).numpy(), c.numpy(), rtol=1e-4, atol=2e-6)


# ------------------------------------------------------------------------------------------ the task-level teacher (r4)
def tsc_modules(seed=0):
    from quadrupedal_agility_amd.tsc.rsl_rl.modules.actor_critic import ActorCriticTSC
    torch.manual_seed(seed)
    n_prop, n_aux, n_scan, n_lat, n_exp, hist = 65, 8, 132, 29, 4, 10
    n_obs = n_prop + n_scan + n_lat + n_exp + hist * (n_prop - n_aux)
    ac = ActorCriticTSC(n_prop, n_aux, n_scan, n_obs, n_lat, n_exp, hist, 6, 3, scan_encoder_dims=[128, 64, 32], actor_hidden_dims=[512, 256, 128],
                        critic_hidden_dims=[512, 256, 128], priv_encoder_dims=[64], activation="elu")
    est = Estimator(n_prop - n_aux, n_exp, hidden_dims=[128, 64])
    with torch.no_grad():
        for p in list(ac.parameters()) + list(est.parameters()):
            if p.dim() == 1:
                p.uniform_(-0.2, 0.2)
    return ac, est, n_obs


def tsc_reference(ac, est, obs, use_estimator):
    """tsc/rsl_rl/algorithms/ppo.py:101-125: the policy sees the estimated privileged states, the critic the true row"""
    with torch.no_grad():
        x = obs.clone()
        if use_estimator:
            x[:, 65 + 132:65 + 132 + 4] = est(x[:, :57])
        emb = ac.actor(x, False)
        return ac.actor.actor_d(emb), ac.actor.actor_c(emb), ac.evaluate(obs)


def run_oracle_outputs(chain, obs):
    from tests.oracle_lib import load_oracle
    lib = load_oracle()
    n = obs.shape[0]
    with torch.no_grad():
        w, b = chain._ptr_arrays()
    packed = np.zeros(chain.packed_floats, np.float32)
    assert lib.qo_mlp_packed_floats(chain.ops, chain.n_ops) == chain.packed_floats
    assert lib.qo_mlp_pack(chain.ops, chain.n_ops, w, b, packed.ctypes.data, packed.size, None) == 0
    outs_np = [np.zeros((n, wd), np.float32) for wd in chain.out_widths]
    x = np.ascontiguousarray(obs.numpy())
    k = len(outs_np)
    outs = (C.c_void_p * k)(*[o.ctypes.data for o in outs_np])
    strides = (C.c_int64 * k)(*chain.out_widths)
    assert lib.qo_mlp_forward(x.ctypes.data, x.shape[1], n, x.shape[1], chain.ops, chain.n_ops, packed.ctypes.data, outs, strides, k, None) == 0
    return outs_np


@pytest.mark.parametrize("use_estimator", [True, False])
def test_task_level_description_matches_the_modules(use_estimator):
    """PolicyChain.describe_task_level: estimator -> explicit columns, scan encoder (tanh), privileged encoder, trunk, both heads, critic on the
    true 800-wide row -- the op list through the C twin against the modules"""
    ac, est, n_obs = tsc_modules()
    assert n_obs == 800
    chain = PolicyChain.describe_task_level(ac, est, use_estimator)
    assert chain is not None and chain.n_ops <= _capi.MLP_MAX_OPS and chain.out_widths == [6, 18, 1]
    obs = torch.randn(29, n_obs)
    logits, mean, value = run_oracle_outputs(chain, obs)
    rl, rm, rv = tsc_reference(ac, est, obs, use_estimator)
    np.testing.assert_allclose(logits, rl.numpy(), rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(mean, rm.numpy(), rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(value, rv.numpy(), rtol=1e-4, atol=2e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("rows", [1, 100, 1024, 8192])
def test_task_level_chain_on_the_gpu(rows):
    ac, est, n_obs = tsc_modules(seed=3)
    obs = torch.randn(rows, n_obs)
    want = run_oracle_outputs(PolicyChain.describe_task_level(ac, est, True), obs[:64])
    ac, est = ac.cuda(), est.cuda()
    chain = PolicyChain.describe_task_level(ac, est, True)
    chain.pack()
    got = chain.forward(obs.cuda())
    torch.cuda.synchronize()
    rl, rm, rv = tsc_reference(ac, est, obs.cuda(), True)
    for g, w, r in zip(got, want, (rl, rm, rv)):
        np.testing.assert_allclose(g[:64].cpu().numpy(), w[:rows], rtol=2e-4, atol=5e-5)
        np.testing.assert_allclose(g.cpu().numpy(), r.cpu().numpy(), rtol=2e-4, atol=5e-5)


@pytest.mark.parametrize("use_estimator", [True, False])
def test_task_level_chain_halves_equal_the_whole(use_estimator):
    """describe_task_level(part="actor" / "critic") -- the two programs of fused.SplitTeacherChain: the same ops per output as the one-launch
    chain (the twin's outputs are bit-identical), fewer ops each"""
    ac, est, n_obs = tsc_modules(seed=5)
    whole = PolicyChain.describe_task_level(ac, est, use_estimator)
    actor = PolicyChain.describe_task_level(ac, est, use_estimator, part="actor")
    critic = PolicyChain.describe_task_level(ac, est, use_estimator, part="critic")
    assert actor.out_widths == [6, 18] and critic.out_widths == [1] and actor.n_ops + critic.n_ops == whole.n_ops
    obs = torch.randn(37, n_obs)
    logits, mean, value = run_oracle_outputs(whole, obs)
    l2, m2 = run_oracle_outputs(actor, obs)
    (v2,) = run_oracle_outputs(critic, obs)
    assert np.array_equal(logits, l2) and np.array_equal(mean, m2) and np.array_equal(value, v2)


@pytest.mark.gpu
@pytest.mark.parametrize("rows", [1, 1024, 2048])
def test_split_teacher_chain_equals_the_one_launch_chain(rows):
    """fused.SplitTeacherChain (critic on a second stream beside the actor side): bit-identical outputs, eagerly and replayed from a hipGraph"""
    from quadrupedal_agility_amd.rsl_rl.algorithms.fused import SplitTeacherChain
    ac, est, n_obs = tsc_modules(seed=4)
    ac, est = ac.cuda(), est.cuda()
    whole = PolicyChain.describe_task_level(ac, est, True); whole.pack()
    split = SplitTeacherChain.describe(ac, est, True); split.pack()
    obs = torch.randn(rows, n_obs, device="cuda")
    want = [t.clone() for t in whole.forward(obs)]
    got = [t.clone() for t in split.forward(obs)]
    torch.cuda.synchronize()
    assert all(torch.equal(g, w) for g, w in zip(got, want))
    # recorded: new observations in the same buffer, replay, compare with the one-launch chain on them
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        split.forward(obs)
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        outs = split.forward(obs)
    obs.copy_(torch.randn(rows, n_obs, device="cuda"))
    g.replay()
    torch.cuda.synchronize()
    got = [t.clone() for t in outs]
    want = [t.clone() for t in whole.forward(obs)]
    torch.cuda.synchronize()
    assert all(torch.equal(a, b) for a, b in zip(got, want))
