"""Semi-supervised InfoGAIL discriminator: trunk 98->512->256 (ReLU) with three heads
(logit 1, gait classifier 5, epsilon encoder 1).  Parameter names as in
bbc/rsl_rl/algorithms/discriminator.py:36-46; reward mapping :71-118."""
import torch
import torch.nn as nn
import torch.nn.functional as F

DISC_LOGIT_INIT_SCALE = 1.0


class Discriminator(nn.Module):
    def __init__(self, env, input_dim, num_disc_obs, dim_c, dt, disc_loss_function, reward_i_normalizer, reward_i_coef,
                 reward_us_coef, reward_ss_coef, reward_t_coef, disc_history_len, disc_obs_len, obs_disc_weight_step,
                 hidden_units, device):
        super().__init__()
        self.device, self.env = device, env
        self.input_dim, self.num_disc_obs, self.dim_c, self.dt = input_dim, num_disc_obs, dim_c, dt
        self.disc_loss_function = disc_loss_function
        self.reward_i_normalizer = reward_i_normalizer
        self.disc_history_len, self.disc_obs_len, self.obs_disc_weight_step = disc_history_len, disc_obs_len, obs_disc_weight_step
        self.reward_i_coef, self.reward_us_coef = reward_i_coef, reward_us_coef
        self.reward_ss_coef, self.reward_t_coef = reward_ss_coef, reward_t_coef
        layers, d = [], input_dim
        for h in hidden_units:
            layers += [nn.Linear(d, h), nn.ReLU()]
            d = h
        self.trunk = nn.Sequential(*layers)
        self.linear = nn.Linear(d, 1)
        self.classifier = nn.Linear(d, dim_c)
        self.encoder_eps = nn.Linear(d, 1)
        for m in self.trunk.modules():
            if isinstance(m, nn.Linear) and m.bias is not None:
                nn.init.zeros_(m.bias)
        nn.init.uniform_(self.linear.weight, -DISC_LOGIT_INIT_SCALE, DISC_LOGIT_INIT_SCALE)
        nn.init.zeros_(self.linear.bias)
        self.train()
        mult = torch.arange(disc_obs_len, dtype=torch.float32) * obs_disc_weight_step + 1
        self.register_buffer("_frame_mult", mult.view(1, -1, 1).repeat(1, 1, num_disc_obs).view(1, -1), persistent=False)
        task = torch.zeros(disc_obs_len, num_disc_obs)
        task[:, 3:9] = 1.0; task[:, 33:] = 1.0                 # the dims prepare_input() scales by task_obs_weight
        self.register_buffer("_task_mask", task.view(-1).contiguous(), persistent=False)

    def forward(self, x):
        x = self.trunk(x)
        c = torch.softmax(self.classifier(x), -1)
        return self.linear(x), self.encoder_eps(x), torch.clamp(c, 1e-20, torch.inf)

    def _relu_trunk(self):
        mods = list(self.trunk)
        ok = len(mods) % 2 == 0 and all(isinstance(mods[i], nn.Linear) and isinstance(mods[i + 1], nn.ReLU) for i in range(0, len(mods), 2))
        return [mods[i] for i in range(0, len(mods), 2)] if ok else None

    def forward_with_input_gradient(self, x, rows, clamp=True, proxies=None):
        """Heads on all rows of x plus d logit / d x on the row slice `rows` -- the quantity the gradient penalty squares
        (gail.py:487-492 obtains it with autograd.grad(..., create_graph=True)).  For this piecewise-linear trunk it is
            g = W_1^T diag(m_1) W_2^T diag(m_2) ... w_out,      m_l = [layer l is active]
        evaluated as a chain of small GEMMs on the slice only; the masks are constants (ReLU has no curvature), so
        differentiating g w.r.t. the weights with ordinary autograd gives exactly the double-backward result, without
        making the input a leaf, without a second-order graph over the whole batch."""
        from quadrupedal_agility_amd.rsl_rl.algorithms import fused
        lins = self._relu_trunk()
        assert lins is not None
        h, masks = x, []
        # every reduction over the batch in this step's backward (bias gradients of the trunk and the heads, the first link of the
        # penalty chain) goes through our fixed-order kernels: torch's `sum(0)` left the first trunk bias's gradient buffer
        # unwritten under hipGraph replay (garbage -> Adam second moment = inf -> the 512 biases stopped training)
        for lin in lins:
            h = fused.linear_relu(lin, h)
            masks.append(torch.sign(h[rows].detach()))   # h >= 0 after the ReLU: sign = [h > 0] as floats, one launch instead of two (a constant: detached)
        head = lambda m: fused.narrow_linear(m, h, always=True)
        c = torch.softmax(head(self.classifier), -1)
        heads = (head(self.linear), head(self.encoder_eps), torch.clamp(c, 1e-20, torch.inf) if clamp else c)     # clamp=False: qa_disc_loss clamps
        # `proxies` (a list to fill): the chain reads each weight through a fresh leaf that shares its storage, so that every real
        # parameter enters the autograd graph ONCE; the caller adds proxy.grad to param.grad after backward().  Without it the
        # engine sums the two contributions of a weight in its input buffer on the AccumulateGrad node's stream -- under a
        # recorded step that is a second branch of the hipGraph (profiles/r2_cfg3_fast_path_vs_eager_bisect.md).
        def leaf(w):
            if proxies is None:
                return w
            q = w.detach().requires_grad_(True)
            proxies.append((w, q))
            return q
        v = fused.mask_times_row(masks[-1], leaf(self.linear.weight))      # (rows, H_last): d logit / d (last pre-activation)
        for l in range(len(lins) - 1, 0, -1):
            v = (v @ leaf(lins[l].weight)) * masks[l - 1]
        return heads, v @ leaf(lins[0].weight)                  # (rows, input_dim)

    def prepare_input(self, obs_disc, task_obs_weight):
        """(B, disc_obs_len, 49) -> (B, 98): task dims weighted (discriminator.py:77-87)."""
        if self.env.task_obs_weight_decay:
            obs_disc = obs_disc.clone()
            obs_disc[:, :, 3:9] *= task_obs_weight
            obs_disc[:, :, 33:] *= task_obs_weight
        return obs_disc[:, -self.disc_obs_len:, :].reshape(len(obs_disc), -1) * self._frame_mult

    def predict_disc_reward(self, reward_t, obs, obs_disc, normalizer=None):
        label_eps = obs[:, -self.dim_c - 1].unsqueeze(-1)
        label_c = F.one_hot(torch.argmax(obs[:, -self.dim_c:], dim=-1), num_classes=self.dim_c)
        x = self.prepare_input(obs_disc, getattr(self.env, "task_obs_weight_dev", None) if getattr(self.env, "task_obs_weight_dev", None) is not None else self.env.task_obs_weight)
        with torch.no_grad():
            self.eval()
            if normalizer is not None:
                x = normalizer.normalize_torch(x, self.device)
            d, eps, c = self.forward(x)
            if self.disc_loss_function == "BCEWithLogitsLoss":
                reward_i = -torch.log(torch.clamp(1 - 1 / (1 + torch.exp(-d)), min=0.0001))
            elif self.disc_loss_function == "MSELoss":
                reward_i = torch.clamp(1 - 0.25 * torch.square(d - 1), min=0)
            elif self.disc_loss_function == "WassersteinLoss":
                reward_i = self.reward_i_normalizer.normalize_torch(d, self.device)
                self.reward_i_normalizer.update(d.cpu().numpy())
            else:
                raise ValueError("Unexpected style reward mapping specified")
            reward_us = -torch.abs(eps - label_eps)
            # the reference feeds the already-softmaxed probabilities to CrossEntropyLoss (a second log-softmax)
            reward_ss = -F.cross_entropy(c, label_c.to(c.dtype), reduction="none").unsqueeze(1)
            reward_i, reward_us, reward_ss = reward_i * self.dt, reward_us * self.dt, reward_ss * self.dt
            rewards = (self.reward_i_coef * reward_i + self.reward_us_coef * reward_us +
                       self.reward_ss_coef * reward_ss + self.reward_t_coef * reward_t)
        return rewards.squeeze(), reward_i.squeeze(), reward_us.squeeze(), reward_ss.squeeze(), reward_t.squeeze()

    def get_disc_logit_weights(self):
        return torch.flatten(self.linear.weight)

    def get_disc_weights(self):
        w = [torch.flatten(m.weight) for m in self.trunk.modules() if isinstance(m, nn.Linear)]
        w.append(torch.flatten(self.linear.weight))
        return w
