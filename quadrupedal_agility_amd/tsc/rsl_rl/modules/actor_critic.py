"""Task-level actor-critic of the TSC tree: a hybrid policy (one categorical gait choice + a Gaussian parameter vector per
gait) over [proprioception | height scan | privileged explicit | privileged latent | history].

Same parameter names/shapes as tsc/rsl_rl/modules/actor_critic.py:59-284 (ActorCriticTSC: `actor.{priv_encoder,
history_encoder, scan_encoder, actor_trunk, actor_d, actor_c}`, `critic.{0,2,4,6}`, `std`) and :286-450 (ActorCriticBBC,
the frozen low-level policy: same layout as the BBC tree's ActorCritic with num_prop = n_proprio - n_auxiliary), so a
reference checkpoint loads into these classes.  Built from the BBC tree's blocks (`_mlp`, the conv-as-GEMM history
encoder, the fused Linear+ELU backward on ROCm tensors)."""
import torch
import torch.nn as nn
from torch.distributions import Categorical, Normal

from quadrupedal_agility_amd.rsl_rl.modules.actor_critic import StateHistoryEncoder, _head, _mlp, _run, get_activation


class Actor(nn.Module):
    """obs -> trunk embedding.  Observation layout: prop | scan | priv_explicit | priv_latent | ... | history (last
    num_hist * (num_prop - num_auxiliary) entries)."""

    def __init__(self, num_prop, num_auxiliary, num_scan, num_actions_d, num_actions_c, scan_encoder_dims, actor_hidden_dims,
                 priv_encoder_dims, num_priv_latent, num_priv_explicit, num_hist, activation, tanh_encoder_output=False):
        super().__init__()
        self.num_prop, self.num_auxiliary, self.num_scan, self.num_hist = num_prop, num_auxiliary, num_scan, num_hist
        self.num_actions_d, self.num_actions_c = num_actions_d, num_actions_c
        self.num_priv_latent, self.num_priv_explicit = num_priv_latent, num_priv_explicit
        self.if_scan_encode = scan_encoder_dims is not None and num_scan > 0
        if len(priv_encoder_dims) > 0:
            self.priv_encoder = _mlp([num_priv_latent] + list(priv_encoder_dims) + [num_priv_latent], activation, last_act=True)
        else:
            self.priv_encoder = nn.Identity()
        self.history_encoder = StateHistoryEncoder(activation, num_prop - num_auxiliary, num_hist, num_priv_latent)
        if self.if_scan_encode:
            dims = [num_scan] + list(scan_encoder_dims)
            layers = []
            for i in range(len(dims) - 1):
                layers += [nn.Linear(dims[i], dims[i + 1]), nn.Tanh() if i == len(dims) - 2 else activation]
            self.scan_encoder = nn.Sequential(*layers)
            self.scan_encoder_output_dim = scan_encoder_dims[-1]
        else:
            self.scan_encoder = nn.Identity()
            self.scan_encoder_output_dim = num_scan
        n_in = num_prop + self.scan_encoder_output_dim + num_priv_explicit + num_priv_latent
        self.actor_trunk = _mlp([n_in] + list(actor_hidden_dims), activation, last_act=True)
        self.actor_d = nn.Linear(actor_hidden_dims[-1], num_actions_d)
        self.actor_c = nn.Linear(actor_hidden_dims[-1], num_actions_d * num_actions_c)

    def forward(self, obs, hist_encoding: bool, eval=False, scandots_latent=None):
        a, b = self.num_prop, self.num_prop + self.num_scan
        if self.if_scan_encode:
            scan_latent = _run(self.scan_encoder, obs[:, a:b]) if scandots_latent is None else scandots_latent
            prop_scan = torch.cat([obs[:, :a], scan_latent], dim=1)
        else:
            prop_scan = obs[:, :b]
        explicit = obs[:, b:b + self.num_priv_explicit]
        latent = self.infer_hist_latent(obs) if hist_encoding else self.infer_priv_latent(obs)
        return _run(self.actor_trunk, torch.cat([prop_scan, explicit, latent], dim=1))

    def infer_priv_latent(self, obs):
        s = self.num_prop + self.num_scan + self.num_priv_explicit
        return _run(self.priv_encoder, obs[:, s:s + self.num_priv_latent])

    def infer_hist_latent(self, obs):
        n = self.num_prop - self.num_auxiliary
        return self.history_encoder(obs[:, -self.num_hist * n:].reshape(-1, self.num_hist, n))

    def infer_scandots_latent(self, obs):
        return _run(self.scan_encoder, obs[:, self.num_prop:self.num_prop + self.num_scan])


class ActorCriticTSC(nn.Module):
    is_recurrent = False

    def __init__(self, num_prop, num_auxiliary, num_scan, num_critic_obs, num_priv_latent, num_priv_explicit, num_hist,
                 num_actions_d, num_actions_c, scan_encoder_dims=[256, 256, 256], actor_hidden_dims=[256, 256, 256],
                 critic_hidden_dims=[256, 256, 256], activation="elu", init_noise_std=1.0, fixed_std=False,
                 device=torch.device("cpu"), **kwargs):
        super().__init__()
        self.kwargs = kwargs
        act = get_activation(activation)
        self.num_actions_d = num_actions_d
        self.actor = Actor(num_prop, num_auxiliary, num_scan, num_actions_d, num_actions_c, scan_encoder_dims, actor_hidden_dims,
                           kwargs["priv_encoder_dims"], num_priv_latent, num_priv_explicit, num_hist, act,
                           tanh_encoder_output=kwargs.get("tanh_encoder_output", False))
        self.critic = _mlp([num_critic_obs] + list(critic_hidden_dims) + [1], act, last_act=False)
        std = init_noise_std * torch.ones(num_actions_d * num_actions_c)
        self.std = std.clone().to(device) if fixed_std else nn.Parameter(std)
        self.distribution_d = None
        self.distribution_c = None

    def reset(self, dones=None):
        pass

    def forward(self):
        raise NotImplementedError

    @property
    def action_mean(self):
        return self.distribution_c.mean

    @property
    def action_std(self):
        return self.distribution_c.stddev

    @property
    def entropy_c(self):
        return self.distribution_c.entropy().mean(dim=-1)      # mean over the parameter vector, as the reference (:245-246)

    @property
    def entropy_d(self):
        return self.distribution_d.entropy()

    def _distributions(self, observations, hist_encoding):
        emb = self.actor(observations, hist_encoding)
        self.distribution_d = Categorical(probs=torch.softmax(_head(self.actor.actor_d, emb), dim=-1), validate_args=False)
        mean = _head(self.actor.actor_c, emb)
        self.distribution_c = Normal(mean, mean * 0.0 + self.std, validate_args=False)

    def act(self, observations, hist_encoding=False, **kwargs):
        """hybrid action (B, 1 + num_actions_d * num_actions_c): [gait index, parameter vector of every gait]"""
        self._distributions(observations, hist_encoding)
        if observations.is_cuda:
            # torch.multinomial validates its input with a host read (`.item()`), which stalls the stream every env step and cannot be
            # recorded into a hipGraph; this is its own one-sample fast path (argmax of p / Exp(1)) without the check
            p = self.distribution_d.probs
            a_d = torch.argmax(p / torch.empty_like(p).exponential_(1.0), dim=-1)
            mean = self.distribution_c.mean                  # torch.normal(mean, std) checks std >= 0 with a host read too
            a_c = mean + self.distribution_c.stddev * torch.randn_like(mean)
        else:
            a_d, a_c = self.distribution_d.sample(), self.distribution_c.sample()
        return torch.cat([a_d.unsqueeze(-1), a_c], dim=-1)

    def get_actions_log_prob_d(self, actions):
        return self.distribution_d.log_prob(actions)

    def get_actions_log_prob_c(self, actions):
        return self.distribution_c.log_prob(actions).sum(dim=-1)

    def act_inference(self, observations, hist_encoding=False, eval=False, scandots_latent=None, **kwargs):
        emb = self.actor(observations, hist_encoding, eval, scandots_latent)
        a_d = torch.argmax(torch.softmax(self.actor.actor_d(emb), dim=-1), dim=-1)
        return torch.cat([a_d.unsqueeze(-1), self.actor.actor_c(emb)], dim=-1)

    def evaluate(self, critic_observations, **kwargs):
        return _run(self.critic, critic_observations)

    def reset_std(self, std, num_actions, device):
        self.std.data = (std * torch.ones(num_actions, device=device)).data


class ActorCriticBBC(nn.Module):
    """The behaviour controller the task level drives, frozen: obs_bbc -> 12 joint targets (mean of the BBC policy)."""
    is_recurrent = False

    def __init__(self, num_actor_obs, num_critic_obs, num_actions, num_prop, num_auxiliary, num_hist, num_explicit, num_latent,
                 num_command, actor_hidden_dims=[256, 256, 256], critic_hidden_dims=[256, 256, 256], priv_encoder_dims=[256, 256],
                 activation="elu", init_noise_std=1.0, fixed_std=False, train_with_estimated_latent=True, **kwargs):
        super().__init__()
        act = get_activation(activation)
        self.num_actor_obs, self.num_critic_obs = num_actor_obs, num_critic_obs
        self.train_with_estimated_latent = train_with_estimated_latent
        self.num_prop, self.num_explicit, self.num_latent = num_prop - num_auxiliary, num_explicit, num_latent
        self.num_hist, self.num_command = num_hist, num_command
        a = self.num_prop; b = a + num_explicit; c = b + num_latent; d = c + num_hist * self.num_prop
        self._sl = (slice(0, a), slice(a, b), slice(b, c), slice(c, d), slice(d, None))     # as the BBC tree's ActorCritic: PolicyChain.describe reads it
        if len(priv_encoder_dims) > 0:
            self.priv_encoder = _mlp([num_latent] + list(priv_encoder_dims) + [num_latent], act, last_act=True)
        else:
            self.priv_encoder = nn.Identity()
        self.history_encoder = StateHistoryEncoder(act, self.num_prop, num_hist, num_latent)
        self.actor_trunk = _mlp([num_actor_obs] + list(actor_hidden_dims), act, last_act=True)
        self.actor_head = nn.Linear(actor_hidden_dims[-1], num_actions)
        self.critic_trunk = _mlp([num_critic_obs] + list(critic_hidden_dims), act, last_act=True)
        self.critic_head = nn.Linear(critic_hidden_dims[-1], 1)
        for m in self.actor_trunk.modules():
            if isinstance(m, nn.Linear) and m.bias is not None:
                nn.init.zeros_(m.bias)
        self.fixed_std = fixed_std
        std = init_noise_std * torch.ones(num_actions)
        self.std = std.clone() if fixed_std else nn.Parameter(std)
        self.distribution = None

    def reset(self, dones=None):
        pass

    def _mean(self, observations, hist_encoding):
        a = self.num_prop; b = a + self.num_explicit; c = b + self.num_latent; d = c + self.num_hist * self.num_prop
        latent = observations[:, b:c]
        if self.train_with_estimated_latent:
            latent = self.infer_hist_latent(observations[:, c:d]) if hist_encoding else self.infer_priv_latent(latent)
        x = torch.cat([observations[:, :a], observations[:, a:b], latent, observations[:, d:]], dim=-1)
        return self.actor_head(_run(self.actor_trunk, x))

    def update_distribution(self, observations, hist_encoding: bool):
        mean = self._mean(observations, hist_encoding)
        self.distribution = Normal(mean, mean * 0.0 + self.std.to(mean.device), validate_args=False)

    @property
    def action_mean(self):
        return self.distribution.mean

    @property
    def action_std(self):
        return self.distribution.stddev

    @property
    def entropy(self):
        return self.distribution.entropy().sum(dim=-1)

    def act(self, observations, hist_encoding=False, **kwargs):
        self.update_distribution(observations, hist_encoding)
        return self.distribution.sample()

    def get_actions_log_prob(self, actions):
        return self.distribution.log_prob(actions).sum(dim=-1)

    def act_inference(self, observations, hist_encoding=True):
        return self._mean(observations, hist_encoding)

    def infer_priv_latent(self, obs):
        return _run(self.priv_encoder, obs)

    def infer_hist_latent(self, obs):
        return self.history_encoder(obs.reshape(-1, self.num_hist, self.num_prop))

    def evaluate(self, critic_observations, **kwargs):
        return self.critic_head(_run(self.critic_trunk, critic_observations))
