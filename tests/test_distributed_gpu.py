"""-m gpu: the data-parallel learner path with world_size 2 on ONE GPU (two processes sharing cuda:0, `gloo` moving the
CUDA buckets; production is one process per GPU over RCCL).  What it adds to the CPU gloo tests: the real HIP env per
rank, the recorded rollout, and the PPO step recorded as two hipGraphs around the gradient collective with TWO ranks
feeding that collective."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q, amp, iters, backend="gloo"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dev = rank if backend == "nccl" else 0              # RCCL: one process per GPU; gloo: two processes share cuda:0
    torch.cuda.set_device(dev)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(f"cuda:{dev}"))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    from quadrupedal_agility_amd.legged_gym.envs import task_registry
    from quadrupedal_agility_amd.legged_gym.envs.go2.go2_locomotion_config import Go2LocomotionCfg, Go2LocomotionCfgAlgo
    from quadrupedal_agility_amd.legged_gym.utils import get_args
    cfg = Go2LocomotionCfg(); cfg.env.num_envs = 256; cfg.terrain.mesh_type = "plane"; cfg.env.mocap_state_init = amp
    cfg.seed = 1                                        # ONE job of 512 envs: rank r owns envs [256 r, 256 (r + 1)), draws keyed by the global env id
    cfg.env.env_id_offset, cfg.env.num_envs_global = 256 * rank, 256 * world
    t = Go2LocomotionCfgAlgo(); t.runner.amp_enabled = amp; t.runner.num_preload_transitions = 5000; t.algorithm.disc_replay_buffer_size = 50000
    args = get_args(["--device", "gpu", "--device_id", str(dev)])
    torch.manual_seed(100 + rank)                      # different initial weights per rank: the broadcast must fix that
    env, _ = task_registry.make_env("go2_locomotion", args=args, env_cfg=cfg)
    runner, _ = task_registry.make_alg_runner(env, name="go2_locomotion", args=args, train_cfg=t, log_root=None)
    assert runner.distributed
    runner.learn(iters, init_at_random_ep_len=True)
    torch.cuda.synchronize()
    a = runner.alg
    flat = torch.cat([p.detach().flatten() for m in (a.actor_critic, a.estimator, a.disc) for p in m.parameters()]).cpu()
    two_graphs = isinstance(a._ac_graph, list) and all(gb is not None for _, gb, _ in a._ac_graph)
    norm = torch.cat([a.disc_normalizer.mean, a.disc_normalizer.var, a.disc_normalizer.count.reshape(1)]).cpu().numpy() if amp else np.zeros(1)
    q.put((rank, flat.numpy(), float(a.lr_ac), bool(two_graphs), bool(torch.isfinite(flat).all()), env.root_states[:, :3].cpu().numpy().copy(), norm))
    dist.destroy_process_group()


@pytest.mark.parametrize("amp", [False, True])
def test_two_ranks_on_one_gpu_keep_replicas_identical(amp):
    ctx = mp.get_context("spawn"); q = ctx.Queue(); port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q, amp, 5)) for r in range(2)]
    [p.start() for p in ps]
    out = sorted([q.get(timeout=900) for _ in ps], key=lambda t: t[0])
    [p.join(120) for p in ps]
    _check_replicas(out, amp)


def _check_replicas(out, amp):
    (_, w0, lr0, g0, f0, pos0, n0), (_, w1, lr1, g1, f1, pos1, n1) = out
    assert f0 and f1
    assert g0 and g1                                   # the PPO step ran as two recorded launches around the collective
    assert np.array_equal(w0, w1)                      # same broadcast start + same averaged gradients -> bit-identical replicas
    assert lr0 == lr1                                  # the KL mean rode in the bucket: both ranks took the same LR branch
    assert not np.allclose(pos0, pos1)                 # the ranks simulate different envs (disjoint global env ids)
    if amp:                                            # the discriminator-input normaliser folded the GLOBAL batch moments on both ranks
        assert np.array_equal(n0, n1) and n0[-1] > 1000


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="RCCL needs one GPU per rank: run on a multi-GPU node")
def test_two_ranks_over_rccl():
    """the production path: one process per GPU, gradient buckets over RCCL (xGMI)"""
    ctx = mp.get_context("spawn"); q = ctx.Queue(); port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q, False, 4, "nccl")) for r in range(2)]
    [p.start() for p in ps]
    out = sorted([q.get(timeout=900) for _ in ps], key=lambda t: t[0])
    [p.join(120) for p in ps]
    _check_replicas(out, False)


def _tsc_worker(rank, world, port, q, vision, iters):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from quadrupedal_agility_amd.legged_gym.utils.cfg_to_c import class_to_dict
    from quadrupedal_agility_amd.tsc.legged_gym.envs.base import legged_robot as lr
    from quadrupedal_agility_amd.tsc.legged_gym.envs.go2.go2_agility_config import Go2AgilityCfg, Go2AgilityCfgPPO
    from quadrupedal_agility_amd.tsc.rsl_rl.runners import OnPolicyRunner
    n = 128
    cfg = Go2AgilityCfg()
    cfg.env.num_envs, cfg.seed, cfg.course_seed = n, 1, 1 + rank
    cfg.env.env_id_offset, cfg.env.num_envs_global = rank * n, world * n
    cfg.env.episode_length_s = 1.0
    cfg.depth.use_camera = vision
    tcfg = class_to_dict(Go2AgilityCfgPPO())
    tcfg["depth_encoder"]["if_depth"] = vision
    torch.manual_seed(100 + rank)                      # different initial weights per rank: the broadcast must fix that
    env = lr.LeggedRobot(cfg, sim_device="cuda:0")
    runner = OnPolicyRunner(env, tcfg, log_dir=None, device="cuda:0")
    assert runner.distributed
    runner.learn(iters, init_at_random_ep_len=not vision)
    torch.cuda.synchronize()
    a = runner.alg
    mods = (a.depth_encoder, a.depth_actor) if vision else (a.actor_critic, a.estimator)
    flat = torch.cat([p.detach().flatten() for m in mods for p in m.parameters()]).cpu()
    two_graphs = (not vision) and isinstance(a._graph, tuple) and a._graph[1] is not None
    q.put((rank, flat.numpy(), float(a.learning_rate), bool(two_graphs), bool(torch.isfinite(flat).all()), env.root_states[:, :3].cpu().numpy().copy()))
    dist.destroy_process_group()


@pytest.mark.parametrize("vision", [False, True])
def test_task_level_two_ranks_on_one_gpu_keep_replicas_identical(vision):
    """BASELINE configs[3] / [4] are 8-GPU jobs of the task-level tree: two ranks (sharing cuda:0, gloo moving the CUDA buckets) train
    the teacher -- recorded rollout, the update as TWO hipGraphs around the gradient collective -- and the depth student (DAgger + BYOL
    gradients averaged); replicas stay identical, the ranks simulate different envs"""
    ctx = mp.get_context("spawn"); q = ctx.Queue(); port = _free_port()
    ps = [ctx.Process(target=_tsc_worker, args=(r, 2, port, q, vision, 4 if not vision else 2)) for r in range(2)]
    [p.start() for p in ps]
    out = sorted([q.get(timeout=300) for _ in ps], key=lambda t: t[0])
    [p.join(120) for p in ps]
    (_, w0, lr0, g0, f0, pos0), (_, w1, lr1, g1, f1, pos1) = out
    assert f0 and f1 and np.array_equal(w0, w1) and lr0 == lr1
    if not vision:
        assert g0 and g1                               # the update really ran as two graphs around the all-reduce
    assert not np.allclose(pos0, pos1)


def test_bench_gpus_2_runs_without_a_launcher_on_the_shared_gpu_harness():
    """`python bench.py --gpus 2` exactly as the driver types it, on a one-GPU box with QA_BENCH_SHARED_GPU=1: bench.py starts
    its two ranks, they share cuda:0 and move the buckets over gloo, rank 0 prints the line."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["QA_BENCH_SHARED_GPU"] = "1"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--num_envs", "512", "--no_cpu_baseline"],
                       env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["ranks"] == 2 and d["n_gpus"] == 1 and "shared_gpu_harness" in d and d["config"]["num_envs_per_gpu"] == 256 and d["config"]["num_envs_total"] == 512
    assert d["collective"].startswith("gloo") and d["value"] > 0 and d["scaling"] == "strong"
