"""Dev tool: host cost vs. total cost of replaying the recorded discriminator step, PPO step and rollout (AMP config)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_gpu_train import _make
from quadrupedal_agility_amd.legged_gym.envs import task_registry
env, args, t = _make(4096, True); t.algorithm.disc_replay_buffer_size = 1000000
runner, _ = task_registry.make_alg_runner(env, name="go2_locomotion", args=args, train_cfg=t, log_root=None)
runner.learn(4, init_at_random_ep_len=True)
torch.cuda.synchronize()
a = runner.alg
for name, g, n in (("disc step", a._disc_graph, 80), ("ppo step", a._ac_graph[0][0], 20), ("rollout", runner._graph, 3)):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        g.replay()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"{name}: host {1e3 * (t1 - t0) / n:.3f} ms per replay, host+gpu {1e3 * (t2 - t0) / n:.3f} ms per replay ({n} replays)")
