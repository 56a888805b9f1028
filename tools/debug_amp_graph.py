import sys, torch
sys.path.insert(0, ".")
from tests.test_gpu_train import _make
from quadrupedal_agility_amd.legged_gym.envs import task_registry
torch.manual_seed(0)
env, args, tcfg = _make(512, True)
runner, _ = task_registry.make_alg_runner(env, name="go2_locomotion", args=args, train_cfg=tcfg, log_root=None)
a = runner.alg
for it in range(4):
    runner.learn(1, init_at_random_ep_len=(it == 0))
    st = a.storage
    f = lambda x: bool(torch.isfinite(x).all())
    print(it, "graph", runner._graph is not None, "obs", f(st.observations), "rew", f(st.rewards), float(st.rewards.abs().max()), "val", f(st.values), "adv", f(st.advantages),
          "act", f(st.actions), "norm mean", f(a.disc_normalizer.mean), "var", f(a.disc_normalizer.var), float(a.disc_normalizer.var.min()), "count", float(a.disc_normalizer.count),
          "replay", f(a.disc_storage.states[:a.disc_storage.num_samples]), a.disc_storage.num_samples, "w", all(f(p) for p in a.actor_critic.parameters()), "disc w", all(f(p) for p in a.disc.parameters()))
