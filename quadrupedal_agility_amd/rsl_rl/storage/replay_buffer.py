"""Ring buffer of policy discriminator observations + the latents they were generated under
(bbc/rsl_rl/storage/replay_buffer.py:5-48).  Sampling indices come from torch on the buffer's device
(the reference uses numpy's global RNG on the host and indexes a device tensor with it)."""
import torch


class ReplayBuffer:
    def __init__(self, obs_dim, dim_c, history_len, buffer_size, device):
        self.states = torch.zeros(buffer_size, history_len * obs_dim, device=device)
        self.latent_eps = torch.zeros(buffer_size, 1, device=device)
        self.latent_c = torch.zeros(buffer_size, dim_c, device=device)
        self.buffer_size = buffer_size
        self.device = device
        self.step = 0
        self.num_samples = 0

    def insert(self, states, latent_eps, latent_c):
        """Ring insert of n rows (a single env step, or a whole recorded rollout at once).  More rows than the ring holds: only
        the newest buffer_size of them are kept; otherwise the rows go to (step + i) % size, so any number of wraps is fine."""
        n = states.shape[0]
        if n >= self.buffer_size:
            keep = slice(n - self.buffer_size, n)
            self.states.copy_(states[keep]); self.latent_eps.copy_(latent_eps[keep]); self.latent_c.copy_(latent_c[keep])
            self.step, self.num_samples = 0, self.buffer_size
            return
        first = min(n, self.buffer_size - self.step)
        for dst, src in ((self.states, states), (self.latent_eps, latent_eps), (self.latent_c, latent_c)):
            dst[self.step:self.step + first].copy_(src[:first])
            if first < n:
                dst[:n - first].copy_(src[first:])
        self.num_samples = min(self.buffer_size, max(self.step + n, self.num_samples))
        self.step = (self.step + n) % self.buffer_size

    def feed_forward_generator(self, num_mini_batch, mini_batch_size):
        for _ in range(num_mini_batch):
            idx = torch.randint(0, self.num_samples, (mini_batch_size,), device=self.device)
            yield self.states[idx], self.latent_eps[idx], self.latent_c[idx]
