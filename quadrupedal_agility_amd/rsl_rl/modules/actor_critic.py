"""Actor-critic with privileged-latent encoder and 1-D-conv history encoder.

Same parameter names/shapes as bbc/rsl_rl/modules/actor_critic.py:9-225 (that is the model.pt
contract): std, priv_encoder.{0,2}, history_encoder.{encoder.0, conv_layers.{0,2},
linear_output.0}, actor_trunk.{0,2,4}, actor_head, critic_trunk.{0,2,4}, critic_head.
Sub-modules are created in the reference's order so a given torch seed yields the same init.
"""
import torch
import torch.nn as nn
from torch.distributions import Normal

_ACTIVATIONS = {"elu": nn.ELU, "selu": nn.SELU, "relu": nn.ReLU, "crelu": nn.ReLU, "lrelu": nn.LeakyReLU,
                "tanh": nn.Tanh, "sigmoid": nn.Sigmoid}


def _head(m, x):
    """a narrow output layer (actor / critic head): its weight gradient is a streaming kernel on ROCm (algorithms/fused.py)"""
    if x.is_cuda and torch.is_grad_enabled():
        from quadrupedal_agility_amd.rsl_rl.algorithms import fused
        return fused.narrow_linear(m, x)
    return m(x)


def _run_head(trunk, head, x):
    """head(trunk(x)); on ROCm tensors under autograd trunk and head are ONE chain of hand-written GEMMs (algorithms/fused.py)"""
    if x.is_cuda and torch.is_grad_enabled() and isinstance(trunk, nn.Sequential):
        from quadrupedal_agility_amd.rsl_rl.algorithms import fused
        if fused.ENABLED and fused.OWN_GEMM:
            return fused.mlp_chain([trunk, head], x)
    return _head(head, _run(trunk, x))


def _run(seq, x):
    """nn.Sequential forward; Linear+ELU pairs on ROCm tensors under autograd use the fused backward (algorithms/fused.py)"""
    if x.is_cuda and isinstance(seq, nn.Sequential) and torch.is_grad_enabled():
        from quadrupedal_agility_amd.rsl_rl.algorithms import fused
        if fused.ENABLED:
            return fused.mlp_forward(seq, x)
    return seq(x)


def get_activation(act_name):
    if act_name not in _ACTIVATIONS:
        print("invalid activation function!")
        return None
    return _ACTIVATIONS[act_name]()


def _mlp(sizes, act, last_act):
    layers = []
    for i in range(len(sizes) - 1):
        layers.append(nn.Linear(sizes[i], sizes[i + 1]))
        if i < len(sizes) - 2 or last_act:
            layers.append(act)
    return nn.Sequential(*layers)


class StateHistoryEncoder(nn.Module):
    """(B, T, n_prop) -> latent: per-frame linear to 30 channels, two temporal convs, linear out."""
    _CONV = {10: [(4, 2), (2, 1)], 20: [(6, 2), (4, 2)], 50: [(8, 4), (5, 1), (5, 1)]}

    def __init__(self, activation_fn, input_size, tsteps, output_size, tanh_encoder_output=False):
        super().__init__()
        if tsteps not in self._CONV:
            raise ValueError("tsteps must be 10, 20 or 50")
        self.activation_fn = activation_fn
        self.tsteps = tsteps
        ch = 10
        self.encoder = nn.Sequential(nn.Linear(input_size, 3 * ch), activation_fn)
        chans = [3 * ch, 2 * ch, ch, ch]
        convs = []
        for i, (k, s) in enumerate(self._CONV[tsteps]):
            convs += [nn.Conv1d(chans[i], chans[i + 1], kernel_size=k, stride=s), activation_fn]
        convs.append(nn.Flatten())
        self.conv_layers = nn.Sequential(*convs)
        self.linear_output = nn.Sequential(nn.Linear(ch * 3, output_size), activation_fn)

    def forward(self, obs):
        """Same arithmetic as encoder -> Conv1d stack -> Flatten -> linear_output, but the temporal convolutions are
        evaluated as window-gather + GEMM: a (B=24576, C=30, T=10) Conv1d goes through MIOpen's im2col/igemm paths
        on ROCm (0.5 ms per call, 0.2 s of solver search on first use -- profiles/r1_bench_kernel_stats_v0*), while the
        equivalent (B*T_out, k*C) x (k*C, C_out) matmul is one small hipBLASLt GEMM.  Parameters stay in the Conv1d
        modules, so state_dict names/shapes are unchanged."""
        b = obs.shape[0]
        fused = _fused_for(obs, self.activation_fn)
        if fused is not None:
            # training on ROCm: every layer is Linear+ELU through fused.linear_elu, whose bias gradient is our own fixed-order
            # column sum.  torch's tall-skinny `sum(0)` (61440 x 30, 18432 x 10, ...) is a two-stage reduction whose second stage
            # does not re-run under hipGraph replay on this ROCm -- the recorded DAgger step trained these biases on a stale
            # gradient (profiles/r2_hipgraph_stale_reductions.md)
            alpha = self.activation_fn.alpha
            enc, out = self.encoder[0], self.linear_output[0]
            x = fused.linear_elu(obs.reshape(b * self.tsteps, -1), enc.weight, enc.bias, alpha).reshape(b, self.tsteps, -1)
            for m in self.conv_layers:
                if isinstance(m, nn.Conv1d):
                    win, t_out = _conv1d_windows(x, m)
                    x = fused.linear_elu(win, m.weight.reshape(m.out_channels, -1), m.bias, alpha).reshape(b, t_out, m.out_channels)
            x = x.permute(0, 2, 1).reshape(b, -1)
            return fused.linear_elu(x, out.weight, out.bias, alpha)
        x = self.encoder(obs.reshape(b * self.tsteps, -1)).reshape(b, self.tsteps, -1)      # (B, T, C) channels last
        for m in self.conv_layers:
            if isinstance(m, nn.Conv1d):
                x = _conv1d_channels_last(x, m)
            elif not isinstance(m, nn.Flatten):
                x = m(x)
        x = x.permute(0, 2, 1).reshape(b, -1)           # Flatten of (B, C, T): channel-major
        return self.linear_output(x)


def _fused_for(x, act):
    """algorithms/fused.py when `x` is a ROCm tensor under autograd and the activation is ELU, else None"""
    if x.is_cuda and torch.is_grad_enabled() and isinstance(act, nn.ELU) and x.dtype == torch.float32:
        from quadrupedal_agility_amd.rsl_rl.algorithms import fused
        if fused.ENABLED:
            return fused
    return None


def _conv1d_windows(x, conv):
    """x (B, T, C_in) -> ((B * T_out, C_in * k) window rows in the weight's (c, k) order, T_out)"""
    k, s = conv.kernel_size[0], conv.stride[0]
    b, t, c = x.shape
    t_out = (t - k) // s + 1
    return x.unfold(1, k, s).reshape(b * t_out, c * k), t_out


def _conv1d_channels_last(x, conv):
    """x (B, T, C_in) -> (B, T_out, C_out) for a Conv1d(C_in, C_out, k, stride) without padding/dilation."""
    win, t_out = _conv1d_windows(x, conv)                # rows of the (B, T_out, C_in, k) unfold view
    w = conv.weight.reshape(conv.out_channels, -1)       # (C_out, C_in * k), same (c, k) order as the window
    return torch.addmm(conv.bias, win, w.t()).reshape(x.shape[0], t_out, conv.out_channels)


class ActorCritic(nn.Module):
    is_recurrent = False

    def __init__(self, num_actor_obs, num_critic_obs, num_actions, num_prop, num_hist, num_explicit, num_latent,
                 num_command, actor_hidden_dims=[256, 256, 256], critic_hidden_dims=[256, 256, 256],
                 priv_encoder_dims=[256, 256], activation="elu", init_noise_std=1.0, fixed_std=False,
                 train_with_estimated_latent=False, **kwargs):
        if kwargs:
            print("ActorCritic.__init__ got unexpected arguments, which will be ignored: " + str(list(kwargs.keys())))
        super().__init__()
        act = get_activation(activation)
        self.num_actor_obs, self.num_critic_obs = num_actor_obs, num_critic_obs
        self.train_with_estimated_latent = train_with_estimated_latent
        self.num_prop, self.num_explicit, self.num_latent = num_prop, num_explicit, num_latent
        self.num_hist, self.num_command = num_hist, num_command
        # slices of the observation row (legged_robot.py:261-331)
        a, b, c = num_prop, num_prop + num_explicit, num_prop + num_explicit + num_latent
        d = c + num_hist * num_prop
        self._sl = (slice(0, a), slice(a, b), slice(b, c), slice(c, d), slice(d, d + num_command))      # explicit end: rows may carry zero padding

        if len(priv_encoder_dims) > 0:
            self.priv_encoder = _mlp([num_latent] + list(priv_encoder_dims) + [num_latent], act, last_act=True)
        else:
            self.priv_encoder = nn.Identity()
        self.history_encoder = StateHistoryEncoder(act, num_prop, num_hist, num_latent)
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

    def forward(self):
        raise NotImplementedError

    @property
    def action_mean(self):
        return self.distribution.mean

    @property
    def action_std(self):
        return self.distribution.stddev

    @property
    def entropy(self):
        return self.distribution.entropy().sum(dim=-1)

    def infer_priv_latent(self, obs):
        return _run(self.priv_encoder, obs)

    def infer_hist_latent(self, obs):
        return self.history_encoder(obs.view(-1, self.num_hist, self.num_prop))

    def _actor_mean(self, observations, hist_encoding):
        prop, explicit, latent, hist, command = (observations[:, s] for s in self._sl)
        if self.train_with_estimated_latent:
            latent = self.infer_hist_latent(hist) if hist_encoding else self.infer_priv_latent(latent)
        parts = [prop, explicit, latent, command]
        if observations.is_cuda and torch.is_grad_enabled():
            from quadrupedal_agility_amd.rsl_rl.algorithms import fused
            k = sum(p.shape[1] for p in parts)
            if fused.pad_k() and k % 16 and isinstance(self.actor_trunk, nn.Sequential):
                parts.append(prop.new_zeros(prop.shape[0], fused.pad16(k) - k))      # whole 16-wide k-tiles for the first layer's GEMM (fused.pad_k)
        x = torch.cat(parts, dim=-1)
        return _run_head(self.actor_trunk, self.actor_head, x)

    def update_distribution(self, observations, hist_encoding: bool):
        mean = self._actor_mean(observations, hist_encoding)
        # validate_args=False: the reference means to switch validation off (`Normal.set_default_validate_args = False`,
        # actor_critic.py:148, which as written is a no-op); each validation is a device->host sync per call
        self.distribution = Normal(mean, mean * 0.0 + self.std.to(mean.device), validate_args=False)

    def act(self, observations, hist_encoding=False, **kwargs):
        self.update_distribution(observations, hist_encoding)
        d = self.distribution
        with torch.no_grad():      # == Normal.sample(); torch.normal(tensor, tensor) checks std >= 0 on the host (2 syncs per call)
            return d.loc + d.scale * torch.randn_like(d.loc)

    def get_actions_log_prob(self, actions):
        return self.distribution.log_prob(actions).sum(dim=-1)

    def act_inference(self, observations, hist_encoding=True):
        return self._actor_mean(observations, hist_encoding)

    def evaluate(self, critic_observations, **kwargs):
        return _run_head(self.critic_trunk, self.critic_head, critic_observations)
