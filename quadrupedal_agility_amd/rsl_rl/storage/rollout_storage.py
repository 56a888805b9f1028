"""(T, N, .) rollout slabs, GAE, minibatch gather.

Interface of bbc/rsl_rl/storage/rollout_storage.py:7-157.  Differences, all on purpose:
  * the reference stores the 671-float observation twice per step (actor + critic copies are the
    same values, legged_robot.py:321): here `privileged_observations` aliases `observations`
    (half the HBM traffic of the rollout, 264 MB instead of 528 MB at N=4096);
  * `compute_returns` is the fused HIP kernel `qa_gae` (2 launches) instead of 24 x 6 eager ops;
  * dones are kept as uint8 (T, N, 1) like the reference.
"""
import torch


class RolloutStorage:
    class Transition:
        def __init__(self):
            self.observations = None
            self.critic_observations = None
            self.actions = None
            self.rewards = None
            self.dones = None
            self.values = None
            self.actions_log_prob = None
            self.action_mean = None
            self.action_sigma = None
            self.hidden_states = None

        def clear(self):
            self.__init__()

    def __init__(self, num_envs, num_transitions_per_env, obs_shape, privileged_obs_shape, actions_shape, device="cpu",
                 gae_fn=None):
        self.device = device
        self.obs_shape, self.privileged_obs_shape, self.actions_shape = obs_shape, privileged_obs_shape, actions_shape
        T, N = num_transitions_per_env, num_envs
        z = lambda *s: torch.zeros(T, N, *s, device=device)
        # rows padded with zeros to whole 16-float k-tiles (671 -> 672): `observations` is the reference's (T, N, obs) view of it, the padded slab is
        # what the learner's first-layer GEMMs read (16-byte aligned rows; fused.pad_k)
        width = int(obs_shape[-1])
        self._obs_padded = torch.zeros(T, N, *obs_shape[:-1], (width + 15) // 16 * 16, device=device)
        self.observations = self._obs_padded[..., :width]
        self.privileged_observations = self.observations if privileged_obs_shape[0] is not None else None
        self.rewards, self.values, self.returns, self.advantages, self.actions_log_prob = z(1), z(1), z(1), z(1), z(1)
        self.actions, self.mu, self.sigma = z(*actions_shape), z(*actions_shape), z(*actions_shape)
        self.dones = torch.zeros(T, N, 1, device=device, dtype=torch.uint8)
        self.num_transitions_per_env, self.num_envs = T, N
        self.saved_hidden_states_a = self.saved_hidden_states_c = None
        self.step = 0
        self._gae_fn = gae_fn

    def add_transitions(self, tr):
        if self.step >= self.num_transitions_per_env:
            raise AssertionError("Rollout buffer overflow")
        t = self.step
        self.observations[t].copy_(tr.observations)
        self.actions[t].copy_(tr.actions)
        self.rewards[t].copy_(tr.rewards.view(-1, 1))
        self.dones[t].copy_(tr.dones.view(-1, 1))
        self.values[t].copy_(tr.values)
        self.actions_log_prob[t].copy_(tr.actions_log_prob.view(-1, 1))
        self.mu[t].copy_(tr.action_mean)
        self.sigma[t].copy_(tr.action_sigma)
        self.step += 1

    def clear(self):
        self.step = 0

    def compute_returns(self, last_values, gamma, lam):
        """rollout_storage.py:97-111.  With a fused kernel available (GPU) it is used; the eager form below is the
        same arithmetic and only serves storage objects created without an engine (unit tests of the learner)."""
        if self._gae_fn is not None:
            self._gae_fn(self.rewards, self.values, self.dones, last_values.reshape(-1).contiguous(), self.returns,
                         self.advantages, gamma, lam)
            return
        adv = 0
        for t in reversed(range(self.num_transitions_per_env)):
            nxt = last_values if t == self.num_transitions_per_env - 1 else self.values[t + 1]
            alive = 1.0 - self.dones[t].float()
            delta = self.rewards[t] + alive * gamma * nxt - self.values[t]
            adv = delta + alive * gamma * lam * adv
            self.returns[t] = adv + self.values[t]
        a = self.returns - self.values
        self.advantages.copy_((a - a.mean()) / (a.std() + 1e-8))      # in place: recorded PPO steps hold this buffer's address

    def get_statistics(self):
        done = self.dones.clone()
        done[-1] = 1
        flat = done.permute(1, 0, 2).reshape(-1, 1)
        idx = torch.cat((flat.new_tensor([-1], dtype=torch.int64), flat.nonzero(as_tuple=False)[:, 0]))
        return (idx[1:] - idx[:-1]).float().mean(), self.rewards.mean()

    def mini_batch_generator(self, num_mini_batches, num_epochs=8, perm=None):
        """One permutation reused for all epochs, contiguous index slices (rollout_storage.py:122-157).  `perm`: a given permutation instead of
        a drawn one (checker hook: two learners stepping on the same minibatches)."""
        batch = self.num_envs * self.num_transitions_per_env
        mb = batch // num_mini_batches
        if perm is None:
            perm = torch.randperm(num_mini_batches * mb, requires_grad=False, device=self.device)
        flat = [x.flatten(0, 1) for x in (self.observations, self.actions, self.values, self.advantages, self.returns,
                                          self.actions_log_prob, self.mu, self.sigma)]
        for _ in range(num_epochs):
            for i in range(num_mini_batches):
                idx = perm[i * mb:(i + 1) * mb]
                obs, act, val, adv, ret, logp, mu, sig = (x[idx] for x in flat)
                yield obs, obs, act, val, adv, ret, logp, mu, sig, (None, None), None
