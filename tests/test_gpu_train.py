"""-m gpu: the host mirror driving the real HIP engine: LeggedRobot views, the runner's sync-free rollout, the fused GAE
against the reference's golden vectors, a short training run (loss finite, weights move, checkpoint round trip)."""
import os

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _make(n, amp, mesh="plane"):
    from quadrupedal_agility_amd.legged_gym.envs import task_registry
    from quadrupedal_agility_amd.legged_gym.envs.go2.go2_locomotion_config import Go2LocomotionCfg, Go2LocomotionCfgAlgo
    from quadrupedal_agility_amd.legged_gym.utils import get_args
    cfg = Go2LocomotionCfg(); cfg.env.num_envs = n; cfg.terrain.mesh_type = mesh; cfg.env.mocap_state_init = amp; cfg.seed = 1
    t = Go2LocomotionCfgAlgo(); t.runner.amp_enabled = amp; t.runner.num_preload_transitions = 5000; t.algorithm.disc_replay_buffer_size = 50000
    args = get_args(["--device", "gpu"])
    env, _ = task_registry.make_env("go2_locomotion", args=args, env_cfg=cfg)
    return env, args, t


def test_hip_gae_matches_reference_golden():
    from quadrupedal_agility_amd.sim import QaSim
    from tests.oracle_lib import go2_cfg
    h = QaSim(go2_cfg(16))
    g = np.load(os.path.join(GOLD, "gae.npz"))
    for i in range(int(g["num_cases"])):
        tr = lambda k: torch.tensor(np.ascontiguousarray(g[f"c{i}_{k}"][..., 0])).cuda()
        rew, val, done, last = tr("rewards"), tr("values"), tr("dones"), tr("last")
        ret, adv = torch.zeros_like(rew), torch.zeros_like(rew)
        h.gae(rew, val, done, last, ret, adv, 0.99, 0.95, normalize=True)
        torch.cuda.synchronize()
        assert np.allclose(ret.cpu().numpy(), g[f"c{i}_returns"][..., 0], atol=1e-6)
        assert np.allclose(adv.cpu().numpy(), g[f"c{i}_advantages"][..., 0], atol=1e-5)


def test_env_views_alias_the_device_arena():
    env, _, _ = _make(64, False)
    obs, _ = env.reset()
    assert obs.is_cuda and obs.shape == (64, 671) and obs.data_ptr() == env.sim.t["OBS"].data_ptr()
    assert env.dof_pos.data_ptr() == env.sim.t["DOF_STATE"].data_ptr()
    env.sync_reset_ids = True
    out = env.step(torch.zeros(64, 12, device="cuda"))
    assert len(out) == 7 and out[5].dtype == torch.int64 and out[6].shape[1] == 49
    assert torch.isfinite(out[0]).all() and torch.isfinite(out[2]).all()
    # standing still for 2 s: nobody falls, feet carry the weight
    for _ in range(100):
        env.step(torch.zeros(64, 12, device="cuda"))
    fz = env.contact_forces[:, :, 2].sum(1)
    alive = env.episode_length_buf > 50
    assert alive.float().mean() > 0.9
    m = (15.019 + env.mass_params_tensor[:, 0]) * 9.81
    assert torch.allclose(fz[alive], m[alive], rtol=0.1)


@pytest.mark.parametrize("amp", [False, True])
def test_short_training_run_on_gpu(tmp_path, amp):
    from quadrupedal_agility_amd.legged_gym.envs import task_registry
    torch.manual_seed(0)
    env, args, tcfg = _make(512, amp)
    runner, _ = task_registry.make_alg_runner(env, name="go2_locomotion", args=args, train_cfg=tcfg, log_root=str(tmp_path))
    before = {k: v.clone() for k, v in runner.alg.actor_critic.state_dict().items()}
    runner.learn(3, init_at_random_ep_len=True)
    after = runner.alg.actor_critic.state_dict()
    assert all(torch.isfinite(v).all() for v in after.values())
    assert any(not torch.equal(before[k], after[k]) for k in before)
    assert runner.last_perf["fps"] > 1e4
    # the non-DAgger rollouts were replayed from a hipGraph; the device-side step counter tracks the host's
    assert runner._graph is not None and not runner._graph_failed
    assert int(env._step_ctr.item()) == env.common_step_counter == 1 + 3 * 24
    assert runner.alg.storage.step == 0 and torch.isfinite(runner.alg.storage.advantages).all()
    path = os.path.join(runner.log_dir, "model.pt")
    ck = torch.load(path, weights_only=False)
    assert type(ck["disc_normalizer"]).__name__ == "Normalizer"
    runner.load(path)
    # training goes on after a load: recorded launches and the optimizer pointer tables are rebuilt around the new state
    assert runner.alg._ac_graph is None and runner._graphs == {}
    # ... and the adaptive-KL learning rate is still ONE device tensor shared by the LR rule, ClipAdam and the recorded steps
    # (Optimizer.load_state_dict replaces the group's lr by a copy; load() re-links it)
    alg = runner.alg
    assert all(g["lr"] is alg._lr_ac for g in alg.optim_ac.param_groups) and alg.optim_ac.param_groups[0]["capturable"]
    lr_loaded = float(alg._lr_ac)
    assert lr_loaded == pytest.approx(float(ck["optim_ac"]["param_groups"][0]["lr"]))
    alg._lr_ac.fill_(3.3e-4)                                # whatever the rule writes must be what the optimiser reads
    assert float(alg.optim_ac.param_groups[0]["lr"]) == pytest.approx(3.3e-4)
    alg._lr_ac.fill_(lr_loaded)
    runner.learn(3, init_at_random_ep_len=False)          # one eager update, then recorded again
    assert all(g["lr"] is alg._lr_ac for g in alg.optim_ac.param_groups)
    assert runner.alg._ac_graph not in (None, False)
    assert all(torch.isfinite(v).all() for v in runner.alg.actor_critic.state_dict().values())
    st = runner.alg.optim_ac.state_dict()["state"]
    assert float(st[0]["step"]) == 20 * 6        # 3 iterations before the checkpoint + 3 after, 20 Adam steps each


def test_trimesh_course_env_and_training(tmp_path):
    """the reference's default terrain (go2_locomotion_config.py:62-64: trimesh + measure_heights) as a height field"""
    from quadrupedal_agility_amd.legged_gym.envs import task_registry
    np.random.seed(0); torch.manual_seed(0)
    env, args, tcfg = _make(512, False, mesh="trimesh")
    assert env.sim.cfg.terrain_type == 1 and env.height_samples.shape == (1600, 1600)
    assert torch.equal(env.sim.t["HEIGHT_SAMPLES"], env.height_samples)
    env.reset()
    for _ in range(60):
        env.step(torch.zeros(512, 12, device="cuda"))
    # the kernel's scan sample is point 94 of the reference's 17 x 11 scan (legged_robot.py:1209-1228), evaluated before resets
    fresh = env.episode_length_buf > 0
    full = env._get_heights()
    assert torch.equal(env.sim.t["SCAN_HEIGHT"][fresh], full[fresh, 94])
    assert (full[:, 94] != 0).float().mean() > 0.5
    alive = env.episode_length_buf > 40
    assert alive.float().mean() > 0.8
    clearance = env.root_states[alive, 2] - env.sim.t["SCAN_HEIGHT"][alive]
    assert clearance.min() > 0.15 and clearance.max() < 0.5
    fz = env.contact_forces[alive][:, :, 2].sum(1)
    m = (15.019 + env.mass_params_tensor[alive, 0]) * 9.81
    assert (torch.abs(fz - m) < 0.25 * m).float().mean() > 0.9          # slopes up to 0.4: normal force ~ weight
    runner, _ = task_registry.make_alg_runner(env, name="go2_locomotion", args=args, train_cfg=tcfg, log_root=str(tmp_path))
    runner.learn(2, init_at_random_ep_len=True)
    assert all(torch.isfinite(v).all() for v in runner.alg.actor_critic.state_dict().values())
    assert runner._graph is not None and not runner._graph_failed


def test_logged_episode_statistics_survive_graph_replays(tmp_path):
    """state that recorded rollouts read across replays must live in persistent buffers (a rebinding of
    `_episode_means` once let replays read freed memory: negative 'collision counts' in 2-5 % of the logged iterations)"""
    import json
    from quadrupedal_agility_amd.legged_gym.envs import task_registry
    torch.manual_seed(0)
    env, args, tcfg = _make(256, False)
    p0 = env._episode_means.data_ptr()
    env.reset(); env.step(torch.zeros(256, 12, device="cuda"))
    assert env._episode_means.data_ptr() == p0
    runner, _ = task_registry.make_alg_runner(env, name="go2_locomotion", args=args, train_cfg=tcfg, log_root=str(tmp_path))
    runner.learn(60, init_at_random_ep_len=True)
    assert env._episode_means.data_ptr() == p0
    rows = [json.loads(l) for l in open(os.path.join(runner.log_dir, "scalars.jsonl"))]
    for tag in ("Episode/rew_collision", "Episode/rew_torques", "Episode/rew_dof_acc", "Episode/rew_tracking_lin_vel"):
        v = np.array([r["value"] for r in rows if r["tag"] == tag])
        assert len(v) == 60 and np.isfinite(v).all() and (v >= 0).all(), tag          # un-scaled sums of non-negative terms


def test_dagger_iterations_are_recorded_too(tmp_path):
    """iterations 0, 20, 40 act through the history encoder and run the DAgger regression: eager, recorded, replayed"""
    from quadrupedal_agility_amd.legged_gym.envs import task_registry
    torch.manual_seed(0)
    env, args, tcfg = _make(256, False)
    runner, _ = task_registry.make_alg_runner(env, name="go2_locomotion", args=args, train_cfg=tcfg, log_root=None)
    assert runner.dagger_update_freq == 20
    he0 = {k: v.clone() for k, v in runner.alg.actor_critic.history_encoder.state_dict().items()}
    runner.learn(41, init_at_random_ep_len=True)
    assert set(runner._graphs) == {False, True} and not runner._graph_failed
    assert runner.alg._dagger_graph not in (None, False) and runner.alg._dagger_calls == 3
    he1 = runner.alg.actor_critic.history_encoder.state_dict()
    assert all(torch.isfinite(v).all() for v in he1.values()) and any(not torch.equal(he0[k], he1[k]) for k in he0)
    assert int(env._step_ctr.item()) == env.common_step_counter == 1 + 41 * 24


def test_recorded_dagger_sessions_equal_eager_sessions_on_new_data():
    """The DAgger regression replayed from its hipGraph against the same steps launched eagerly, from the same weights / Adam state /
    permutation, on rollouts that CHANGE between sessions.  Regression for the r2 finding that torch's two-stage `sum(0)` (the bias
    gradients of the 61440 x 30 and 18432 x 10 layers) returns its first result on every later replay: the history encoder now runs
    through our own fixed-order kernels (profiles/r2_hipgraph_stale_reductions.md), so the two paths agree to rounding."""
    from quadrupedal_agility_amd.legged_gym.envs import task_registry
    torch.manual_seed(0)
    env, args, tcfg = _make(1024, False)                     # 6144-row minibatches: the size at which the stale reduction was found
    runner, _ = task_registry.make_alg_runner(env, name="go2_locomotion", args=args, train_cfg=tcfg, log_root=None)
    runner.learn(21, init_at_random_ep_len=True)             # DAgger at 0 (eager) and 20 (recorded + replayed)
    a = runner.alg
    assert a._dagger_graph not in (None, False)
    hp = list(a.actor_critic.history_encoder.parameters())
    st = a.optim_hist_encoder.state

    def snap():
        return [p.detach().clone() for p in hp], [{k: v.clone() for k, v in st[p].items()} for p in hp]

    def restore(s):
        with torch.no_grad():
            for p, q in zip(hp, s[0]): p.copy_(q)
            for p, d in zip(hp, s[1]):
                for k, v in d.items(): st[p][k].copy_(v)

    graph = a._dagger_graph
    for session in range(3):
        a.storage.observations.copy_(torch.randn_like(a.storage.observations) * (0.5 + session))
        start, rng = snap(), torch.cuda.get_rng_state()
        a.storage.step = a.storage.num_transitions_per_env
        loss_g = a.update_dagger(); got = [p.detach().clone() for p in hp]
        restore(start); torch.cuda.set_rng_state(rng)
        a.storage.step = a.storage.num_transitions_per_env
        a._dagger_graph = False
        loss_e = a.update_dagger(); want = [p.detach().clone() for p in hp]
        a._dagger_graph = graph
        assert loss_g == pytest.approx(loss_e, rel=1e-5)
        for (n, _), x, y in zip(a.actor_critic.history_encoder.named_parameters(), got, want):
            assert torch.allclose(x, y, rtol=1e-5, atol=1e-7), (session, n, float((x - y).abs().max()))
        assert any(not torch.equal(x, y) for x, y in zip(got, start[0]))       # the session did train


@pytest.mark.parametrize("rollout_graph", ["1", "0"])
def test_recorded_discriminator_flow_equals_eager_flow_with_phase_syncs(monkeypatch, rollout_graph):
    """Config 3, the device drained at the phase boundaries (what a run with a log directory does): recorded discriminator steps
    against eager steps fed the same sample tables must give the SAME discriminator, optimiser state, normaliser and policy, bit for
    bit, over several updates.  Regression for the r2 finding (profiles/r2_cfg3_fast_path_vs_eager_bisect.md): torch's `sum(0)` left
    the first trunk bias's gradient buffer unwritten under replay -> garbage gradient -> Adam second moment = inf -> 512 biases
    stopped training and the style reward came out ~20 % high.  The step's batch reductions are our own kernels now."""
    from quadrupedal_agility_amd.legged_gym.envs import task_registry
    monkeypatch.setenv("QA_OVERLAP_UPDATES", "0")
    monkeypatch.setenv("QA_PHASE_TIMING", "1"); monkeypatch.setenv("QA_ROLLOUT_GRAPH", rollout_graph)
    res = []
    for recorded in (True, False):
        torch.manual_seed(0)
        env, args, tcfg = _make(512, True)
        runner, _ = task_registry.make_alg_runner(env, name="go2_locomotion", args=args, train_cfg=tcfg, log_root=None)
        runner.phase_timing = True
        a = runner.alg
        a.eager_from_tables = True
        if not recorded:
            a._disc_graph = False
        runner.learn(5, init_at_random_ep_len=True)
        torch.cuda.synchronize()
        assert (a._disc_graph not in (None, False)) == recorded
        state = [p.detach().clone() for p in a.disc.parameters()] + [p.detach().clone() for p in a.actor_critic.parameters()]
        adam = [st[k].clone() for o in (a.optim_d, a.optim_q_eps, a.optim_q_c) for st in o.state.values() for k in ("exp_avg", "exp_avg_sq")]
        assert all(torch.isfinite(t).all() for t in adam)
        res.append((state, adam, a.disc_normalizer.mean.clone(), env.prior_parameters.clone()))
    (sa, aa, na, pa), (sb, ab, nb, pb) = res
    assert all(torch.equal(x, y) for x, y in zip(sa, sb)) and all(torch.equal(x, y) for x, y in zip(aa, ab))
    assert torch.equal(na, nb) and torch.equal(pa, pb)
