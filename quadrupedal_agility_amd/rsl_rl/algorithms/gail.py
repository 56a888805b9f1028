"""SSInfoGAIL: PPO (clipped surrogate + clipped value + privileged-latent regulariser + estimator)
and the semi-supervised InfoGAIL discriminator update.

Follows bbc/rsl_rl/algorithms/gail.py (act :176-197, process_env_step :199-212, update :231-326,
update_actor_critic :328-413, update_ss_info_gail :415-541, update_dagger :543-575) term by term;
constructor signature and attribute names are the reference's.  What is different is mechanical:
  * no host sync inside the minibatch loops: loss scalars are accumulated on the device and read
    once per iteration; the adaptive-KL learning rate lives in a device tensor (Adam capturable);
    the discriminator normaliser's running moments stay on the device;
  * `grad_sync` (set by the runner when world_size > 1) all-reduces one flat gradient bucket per
    optimiser step and the scalars that must agree across ranks (KL mean) over RCCL;
  * `amp_enabled=False` skips the discriminator update (BASELINE config 2; the reference has no
    such switch).
"""
import json
import os

import torch
import torch.nn as nn
import torch.nn.functional as F
import torch.optim as optim

from quadrupedal_agility_amd.rsl_rl.algorithms import fused as fused_mod
from quadrupedal_agility_amd.rsl_rl.algorithms.fused import ClipAdam, disc_loss, disc_prepare, ppo_loss
from quadrupedal_agility_amd.rsl_rl.storage import ReplayBuffer, RolloutStorage


def _vec11(out):
    return out if torch.is_tensor(out) else torch.stack(out)


class LossReadout:
    """The 17 loss / accuracy means of one update() (same order as the reference's return tuple, gail.py:318-326), still on
    the device: the ONE host read of the update happens when somebody looks at a value (the logger), not at the end of
    update() -- a run without a log directory never waits for the GPU and the host can enqueue the next rollout while the
    last minibatch step is still running."""

    def __init__(self, dev_values):
        self._dev, self._host = dev_values, None

    def _values(self):
        if self._host is None:
            self._host = tuple(self._dev.tolist())
        return self._host

    def __iter__(self):
        return iter(self._values())

    def __getitem__(self, i):
        return self._values()[i]

    def __len__(self):
        return int(self._dev.shape[0])


class SSInfoGAIL:
    def __init__(self, env, actor_critic, discriminator, estimator, estimator_paras, motion_loader, disc_normalizer,
                 disc_history_len, disc_obs_len, num_disc_obs, obs_disc_weight_step, disc_loss_function=None,
                 num_learning_epochs=1, num_mini_batches=1, clip_param=0.2, gamma=0.998, lam=0.95,
                 surrogate_loss_coef=1.0, value_loss_coef=5.0, entropy_coef=0.0, bounds_loss_coef=10.0, disc_coef=5.0,
                 disc_logit_reg=0.05, disc_grad_penalty=0.2, disc_weight_decay=0.0001, lr_ac=1e-3, lr_disc=1e-3,
                 lr_q=1e-3, max_grad_norm=1.0, use_clipped_value_loss=False, schedule="fixed", desired_kl=0.01,
                 device="cpu", disc_replay_buffer_size=100000, min_std=None, us_coef=1.0, ss_coef=4.0,
                 prior_soft_coef=1e-3, info_max_coef=2.0, begin_rim=100, priv_reg_coef_schedual=[0, 0.1, 0, 1],
                 priv_reg_coef_schedual_resume=[0, 0.1, 0, 1], amp_enabled=True):
        self.device, self.env = device, env
        self.desired_kl, self.schedule = desired_kl, schedule
        self.lr_disc, self.lr_q, self.min_std = lr_disc, lr_q, min_std
        self.dim_c = env.dim_c
        self.disc_loss_function = disc_loss_function
        self.disc_history_len, self.disc_obs_len = disc_history_len, disc_obs_len
        self.num_disc_obs, self.obs_disc_weight_step = num_disc_obs, obs_disc_weight_step
        self.amp_enabled = amp_enabled
        self._on_gpu = torch.device(device).type == "cuda"

        self.disc = discriminator.to(device)
        self.disc_storage = ReplayBuffer(env.num_obs_disc, self.dim_c, disc_obs_len, disc_replay_buffer_size, device) if amp_enabled else None
        self.motion_loader = motion_loader
        self.disc_normalizer = disc_normalizer
        c = env.cfg.env
        self.num_prop, self.num_explicit, self.num_latent = c.num_prop, c.num_explicit, c.num_latent
        self.num_hist, self.num_command = c.history_len, c.num_command

        self.actor_critic = actor_critic.to(device)
        self.transition = None
        self.storage = None
        self.estimator = estimator
        # adaptive learning rate: a device tensor on GPU (no host round trip per minibatch), a float on CPU
        self._lr_ac = torch.tensor(float(lr_ac), device=device) if self._on_gpu else float(lr_ac)
        adam = dict(fused=True, capturable=True) if self._on_gpu else {}   # one multi-tensor kernel per step, tensor lr, recordable
        self.optim_ac = optim.Adam([{"params": self.actor_critic.parameters(), "name": "actor_critic"}], lr=self._lr_ac, **adam)
        self.optim_hist_encoder = optim.Adam(self.actor_critic.history_encoder.parameters(), lr=estimator_paras["learning_rate"], **adam)
        self.optim_estimator = optim.Adam(self.estimator.parameters(), lr=estimator_paras["learning_rate"], **adam)
        self.priv_reg_coef_schedual = priv_reg_coef_schedual
        self.priv_reg_counter = 0
        self.train_with_estimated_explicit = estimator_paras["train_with_estimated_explicit"]

        def groups(head, name):   # per-group weight_decay 1e-3 is live for Adam (gail.py:107-124); 'momentum' is an inert key
            return [{"params": self.disc.trunk.parameters(), "weight_decay": 1e-3, "momentum": 0.9, "name": "trunk"},
                    {"params": head.parameters(), "weight_decay": 1e-3, "momentum": 0.9, "name": name}]
        if disc_loss_function == "WassersteinLoss":
            self.optim_d = optim.RMSprop(groups(self.disc.linear, "head"), lr=lr_disc)
        else:
            self.optim_d = optim.Adam(groups(self.disc.linear, "head"), lr=lr_disc, **adam)
        self.optim_q_eps = optim.Adam(groups(self.disc.encoder_eps, "encoder_eps"), lr=lr_q, **adam)
        self.optim_q_c = optim.Adam(groups(self.disc.classifier, "classifier"), lr=lr_q, **adam)

        self._step_ac = ClipAdam(self.optim_ac, max_grad_norm)
        self._step_estimator = ClipAdam(self.optim_estimator, max_grad_norm)
        self._step_hist_encoder = ClipAdam(self.optim_hist_encoder, max_grad_norm)
        self._step_pair = (fused_mod.ClipAdamPair(self._step_estimator, self._step_ac)
                           if (self._on_gpu and max_grad_norm and os.environ.get("QA_ADAM_PAIR", "1") != "0") else None)
        self._step_disc = [ClipAdam(o, None) if isinstance(o, optim.Adam) else o for o in (self.optim_d, self.optim_q_eps, self.optim_q_c)]
        self._disc_stack = (fused_mod.StackedAdam([self.optim_d, self.optim_q_eps, self.optim_q_c])
                            if (self._on_gpu and isinstance(self.optim_d, optim.Adam) and os.environ.get("QA_DISC_STACKED_ADAM", "1") != "0") else None)
        self.clip_param, self.num_learning_epochs, self.num_mini_batches = clip_param, num_learning_epochs, num_mini_batches
        self.surrogate_loss_coef, self.value_loss_coef, self.entropy_coef = surrogate_loss_coef, value_loss_coef, entropy_coef
        self.bounds_loss_coef, self.disc_coef, self.disc_logit_reg = bounds_loss_coef, disc_coef, disc_logit_reg
        self.disc_grad_penalty, self.disc_weight_decay = disc_grad_penalty, disc_weight_decay
        self.gamma, self.lam, self.max_grad_norm = gamma, lam, max_grad_norm
        self.use_clipped_value_loss = use_clipped_value_loss
        self.us_coef, self.ss_coef, self.prior_soft_coef = us_coef, ss_coef, prior_soft_coef
        self.info_max_coef_on, self.info_max_coef = 0, info_max_coef
        self.learning_steps, self.begin_rim = 0, begin_rim
        self.grad_sync = None          # callable(list_of_params, extra_scalars) -> None, installed for world_size > 1
        self.use_fused_loss = True     # GPU: PPO objective + gradient as one HIP kernel (qa_ppo_loss); False = eager PyTorch ops
        self._ac_graph, self._recording_ac, self._priv_coef_dev = None, False, None
        self._disc_stream, self._recording_disc = None, False
        self.overlap_updates = os.environ.get("QA_OVERLAP_UPDATES", "1") != "0"     # GPU: discriminator steps on a second stream beside the PPO steps
        self.eager_from_tables = False      # test hook: eager discriminator steps on the sample tables the recorded path draws (update())
        self._dagger_graph, self._dagger_calls = None, 0
        self._task_w_dev = None
        # recordings wait for one eager update since construction / checkpoint load: optimizer state and the pointer
        # tables of the fused optimizer steps must exist before a capture (building them copies from pageable host memory)
        self._warm_updates, self._dagger_warm = 0, 0
        # GPU: the PPO / discriminator / DAgger steps replay hipGraphs -- but only when every batch reduction of their backward is one of
        # our kernels, i.e. Linear+ELU networks, a ReLU discriminator trunk and the fused objectives: torch's own `sum(0)` goes stale or
        # unwritten under replay on this ROCm (profiles/r2_hipgraph_stale_reductions.md).  Anything else stays eager.
        self.use_update_graph = self._recordable_networks()
        self._disc_graph = None
        self._info_max_dev = torch.zeros((), device=device) if self._on_gpu else None
        # PPO step: critic / actor / small nets on three streams (config 2: update 32.2 -> 28.9 ms).  Not with the discriminator: its
        # 80 recorded steps already run beside the PPO steps on their own stream, and three more streams of GEMMs starve them
        # (config 3: update 61 -> 90 ms with both)
        self.branch_streams = (self._on_gpu and os.environ.get("QA_PPO_BRANCHES", "1") != "0" and
                               (not (self.amp_enabled and self.overlap_updates) or os.environ.get("QA_PPO_BRANCHES_AMP", "0") == "1"))
        self._branch = None

    # ---- lr_ac is read by the logger and by checkpoints as a float
    @property
    def lr_ac(self):
        return float(self._lr_ac)

    @lr_ac.setter
    def lr_ac(self, v):
        if self._on_gpu:
            self._lr_ac.fill_(float(v))
        else:
            self._lr_ac = float(v)
            for g in self.optim_ac.param_groups:
                g["lr"] = self._lr_ac

    def init_storage(self, num_envs, num_transitions_per_env, actor_obs_shape, critic_obs_shape, action_shape, gae_fn=None):
        self.transition = RolloutStorage.Transition()
        self.storage = RolloutStorage(num_envs, num_transitions_per_env, actor_obs_shape, critic_obs_shape, action_shape,
                                      self.device, gae_fn=gae_fn)

    def test_mode(self):
        self.actor_critic.eval()

    def train_mode(self):
        self.actor_critic.train()

    # ------------------------------------------------------------------ rollout side
    def act(self, obs, critic_obs, hist_encoding=False, chain=None):
        """gail.py:176-197.  `chain` (fused.PolicyChain of the same actor variant) evaluates the estimator, encoder,
        actor and critic in one launch; the sampling and the transition record are the same either way."""
        tr = self.transition
        if chain is not None:           # described for this actor variant (privileged / history encoder) by the caller
            ac = self.actor_critic
            mean, value = chain.forward(obs)
            ac.distribution = torch.distributions.Normal(mean, mean * 0.0 + ac.std, validate_args=False)
            with torch.no_grad():
                tr.actions = mean + ac.distribution.scale * torch.randn_like(mean)
            tr.values = value
            tr.actions_log_prob = ac.get_actions_log_prob(tr.actions).detach()
            tr.action_mean, tr.action_sigma = mean, ac.distribution.scale.detach()
            tr.observations, tr.critic_observations = obs, critic_obs
            return tr.actions
        if self.train_with_estimated_explicit:
            a, b = self.num_prop, self.num_prop + self.num_explicit
            est = self.estimator(obs[:, :a])
            obs_in = torch.cat([obs[:, :a], est, obs[:, b:]], dim=-1)     # estimated root height / base lin vel
        else:
            obs_in = obs
        tr.actions = self.actor_critic.act(obs_in, hist_encoding).detach()
        tr.values = self.actor_critic.evaluate(critic_obs).detach()
        tr.actions_log_prob = self.actor_critic.get_actions_log_prob(tr.actions).detach()
        tr.action_mean = self.actor_critic.action_mean.detach()
        tr.action_sigma = self.actor_critic.action_std.detach()
        tr.observations = obs
        tr.critic_observations = critic_obs
        return tr.actions

    def act_mean_value(self, obs, critic_obs, hist_encoding=False):
        """The GEMM half of act(): action mean and value; sampling, log-prob and the storage writes are qa_rollout_act."""
        if self.train_with_estimated_explicit:
            a, b = self.num_prop, self.num_prop + self.num_explicit
            obs_in = torch.cat([obs[:, :a], self.estimator(obs[:, :a]), obs[:, b:]], dim=-1)
        else:
            obs_in = obs
        return self.actor_critic._actor_mean(obs_in, hist_encoding), self.actor_critic.evaluate(critic_obs)

    def process_env_step(self, rewards, dones, infos, obs_disc_history_buf, disc_stage=None):
        tr = self.transition
        tr.rewards = rewards.clone()
        tr.dones = dones
        if "time_outs" in infos:        # bootstrap on time-outs (gail.py:203-205)
            tr.rewards += self.gamma * torch.squeeze(tr.values * infos["time_outs"].unsqueeze(1).to(self.device), 1)
        if self.amp_enabled:
            flat = obs_disc_history_buf.view(obs_disc_history_buf.shape[0], -1)
            if disc_stage is not None:          # recorded rollout: the ring position is a host variable, so stage and insert after replay
                t = self.storage.step
                disc_stage[0][t].copy_(flat); disc_stage[1][t].copy_(self.env.latent_eps); disc_stage[2][t].copy_(self.env.latent_c)
            else:
                self.disc_storage.insert(flat, self.env.latent_eps, self.env.latent_c)
        self.storage.add_transitions(tr)
        self.actor_critic.reset(dones)

    def compute_returns(self, last_critic_obs):
        last_values = self.actor_critic.evaluate(last_critic_obs.detach()).detach()
        self.storage.compute_returns(last_values, self.gamma, self.lam)

    # ------------------------------------------------------------------ update
    def update(self, tables=None):
        """One iteration's PPO and discriminator steps (gail.py:231-326).  `tables` is a checker hook (tools/learner_lockstep.py): the sample tables
        an update otherwise draws itself, on this learner's device -- dict(perm (num_mini_batches x minibatch,) int64: the rollout permutation;
        pi / lb / ulb (80, minibatch) int64: per discriminator step the rows of the replay ring, the labelled and the unlabelled expert set).
        Two learners given the same tables step on the same samples in the same order, whatever path (recorded launches, eager GPU, torch CPU)."""
        self.learning_steps += 1
        if self.learning_steps >= self.begin_rim:
            self.info_max_coef_on = min(self.info_max_coef * (self.learning_steps - self.begin_rim) / 10000, self.info_max_coef)
        dev = self.device
        n_ac = self.num_learning_epochs * self.num_mini_batches
        n_d = n_ac * 4
        ac_recorded = (self._on_gpu and self.use_update_graph and self._warm_updates >= 1 and self._ac_graph is not False
                       and self.desired_kl is not None and self.schedule == "adaptive")
        # The discriminator steps read the replay ring, the mocap clips and their own networks; the PPO steps read the rollout
        # storage and theirs: no data flows between the two loops inside update() (the reference runs them one after the other,
        # gail.py:274-289).  Once both steps exist as recorded launches, the 80 tiny discriminator steps (launch-latency-bound,
        # a few CUs each) are replayed on a second stream WHILE the 20 GEMM-bound PPO steps run.
        if (ac_recorded and self.amp_enabled and self.overlap_updates and self.grad_sync is None and self._ac_graph and self._disc_graph):
            main = torch.cuda.current_stream()
            # drawn first so that the generator is consumed in the same order as when the loops run one after the other
            perm = (tables["perm"] if tables is not None else
                    torch.randperm(self.storage.num_envs * self.storage.num_transitions_per_env // self.num_mini_batches * self.num_mini_batches, device=dev))
            self._disc_stream.wait_stream(main)
            with torch.cuda.stream(self._disc_stream):
                mb = self.storage.num_envs * self.storage.num_transitions_per_env // n_d
                acc_d_dev = self._disc_updates_recorded(n_d, mb, tables)
            acc_ac = self._ac_updates_recorded(perm)
            main.wait_stream(self._disc_stream)
            self._clamp_std()
            self.storage.clear()
            self.priv_reg_counter += 1
            self._warm_updates += 1
            return LossReadout(torch.cat([acc_ac / n_ac, acc_d_dev / n_d]))
        if ac_recorded:
            acc_ac = self._ac_updates_recorded(tables["perm"] if tables is not None else None)
        else:
            acc_ac = torch.zeros(6, device=dev)
            for sample in self.storage.mini_batch_generator(self.num_mini_batches, self.num_learning_epochs, perm=tables["perm"] if tables is not None else None):
                acc_ac += torch.stack(self.update_actor_critic(sample))
        acc_d = torch.zeros(11, device=dev)
        if self.amp_enabled:
            mb = self.storage.num_envs * self.storage.num_transitions_per_env // n_d
            if self._on_gpu and self.use_update_graph and self.grad_sync is None and self._warm_updates >= 1 and self._disc_graph is not False:
                acc_d = self._disc_updates_recorded(n_d, mb, tables).clone()
                self._clamp_std()
            elif tables is not None or (self.eager_from_tables and self._on_gpu):
                # the eager steps on exactly the samples the recorded path would draw (same generator calls): what the recorded-vs-eager
                # regression test compares the replays with (tests/test_gpu_train.py)
                ml, rb = self.motion_loader, self.disc_storage
                if tables is not None:
                    tabs = [tables["pi"], tables["lb"], tables["ulb"]]
                    assert all(t.shape == (n_d, mb) for t in tabs) and int(tabs[0].max()) < rb.num_samples
                else:
                    tabs = [torch.zeros(n_d, mb, dtype=torch.int64, device=dev) for _ in range(3)]
                    nsd = torch.full((), float(rb.num_samples), device=dev)
                    tabs[0].copy_((torch.rand(tabs[0].shape, device=dev) * nsd).long())
                    torch.randint(0, ml.preloaded_s_lb.shape[0], tabs[1].shape, device=dev, out=tabs[1])
                    torch.randint(0, ml.preloaded_s_ulb.shape[0], tabs[2].shape, device=dev, out=tabs[2])
                for k in range(n_d):
                    i_pi, i_lb, i_ulb = tabs[0][k], tabs[1][k], tabs[2][k]
                    acc_d += _vec11(self.update_ss_info_gail((rb.states[i_pi], rb.latent_eps[i_pi], rb.latent_c[i_pi]),
                                                                  (ml.preloaded_s_lb[i_lb], ml.preloaded_label[i_lb]), ml.preloaded_s_ulb[i_ulb]))
            else:
                gens = zip(self.disc_storage.feed_forward_generator(n_d, mb),
                           self.motion_loader.feed_forward_generator_lb(n_d, mb),
                           self.motion_loader.feed_forward_generator_ulb(n_d, mb))
                for s_pi, s_lb, s_ulb in gens:
                    acc_d += _vec11(self.update_ss_info_gail(s_pi, s_lb, s_ulb))
        self.storage.clear()
        self.priv_reg_counter += 1
        self._warm_updates += 1
        return LossReadout(torch.cat([acc_ac / n_ac, acc_d / n_d]))

    def _recordable_networks(self):
        other = (nn.SELU, nn.ReLU, nn.LeakyReLU, nn.Tanh, nn.Sigmoid, nn.GELU, nn.SiLU)
        plain = not any(isinstance(m, other) for net in (self.actor_critic, self.estimator) for m in net.modules())
        disc_ok = (not self.amp_enabled) or (self.disc._relu_trunk() is not None and self.disc_loss_function == "MSELoss")
        return bool(fused_mod.ENABLED and plain and disc_ok)

    def _clamp_std(self):
        """gail.py:522-523 (inside every discriminator step there; idempotent, and nothing reads std in between)"""
        if not self.actor_critic.fixed_std and self.min_std is not None:
            self.actor_critic.std.data.clamp_(min=self.min_std)       # in place: recorded rollouts keep reading this buffer

    def _priv_reg_coef_now(self):
        s0, s1, t0, t1 = self.priv_reg_coef_schedual
        stage = min(max(self.priv_reg_counter - t0, 0) / t1, 1)
        return stage * (s1 - s0) + s0

    def _ac_updates_recorded(self, perm=None):
        """The 20 PPO minibatch steps of an iteration are ~270 launches each and the host cannot issue them faster than
        the GPU retires them (the iteration time followed the host's launch rate, 59-78 ms, not the GPU's 57 ms of kernel
        time).  One step -- minibatch gather from a device index buffer, both forwards, the fused objective, both
        backward passes, clipping, the KL-adaptive learning rate and the Adam updates -- is recorded into a hipGraph once
        and replayed; per step the host only copies the next 24,576 indices into the index buffer.
        Data-parallel runs record the step as TWO graphs around the gradient collective, which stays an ordinary
        launch: [gather .. backward, gradients + KL packed into a persistent bucket] -> all-reduce(bucket) ->
        [scale + unpack, clip, LR rule, Adam]."""
        dev, st = self.device, self.storage
        batch = st.num_envs * st.num_transitions_per_env
        mb = batch // self.num_mini_batches
        nmb = self.num_mini_batches
        if self._ac_graph is None:
            try:
                self._acc_ac = torch.zeros(6, device=dev)
                self._priv_coef_dev = torch.zeros((), device=dev)
                flat = [x.flatten(0, 1) for x in (st.observations, st.actions, st.values, st.advantages, st.returns,
                                                  st.actions_log_prob, st.mu, st.sigma)]
                # (r6: also when the steps of this size run as chain launches: the first layers' weight-gradient products then read the 671 observation
                # columns of 672-wide rows in 16-byte pieces, qa_linear_backward_weight_batch)
                if (fused_mod.pad_k() or self._train_chain_rows(mb) is not None) and getattr(st, "_obs_padded", None) is not None:
                    flat[0] = st._obs_padded.flatten(0, 1)      # the minibatch copy keeps the zero padding (672 columns): 16-byte aligned GEMM rows
                sync = self.grad_sync
                all_params = list(self.estimator.parameters()) + list(self.actor_critic.parameters())

                # The history encoder is not trained by these steps (its latent enters the regulariser detached; its own
                # regression is update_dagger), so its output for a sample is the same in all 5 epochs: evaluate it ONCE per
                # update for the whole rollout (~0.6 ms) instead of once per minibatch step (20 x 0.16 ms) and gather rows.
                self._hist_latent_all = torch.zeros(batch, self.num_latent, device=dev)
                self._hist_cols = slice(self.num_prop + self.num_explicit + self.num_latent,
                                        self.num_prop + self.num_explicit + self.num_latent + self.num_hist * self.num_prop)
                # The reference draws ONE permutation per update and reuses its 4 slices in all 5 epochs (rollout_storage.py:122-157): the
                # rollout is gathered into permuted order ONCE per update (one qa_gather_rows launch over all 98,304 rows, 0.17 ms) and
                # minibatch i is rows [i mb, (i+1) mb) of that copy -- a recorded step per minibatch slot (they share one memory pool and never
                # run concurrently) instead of a 43 us gather at the head of each of the 20 steps (r3: -0.7 ms per iteration).
                self._gather_srcs = flat + [self._hist_latent_all]
                self._perm_bufs = [torch.empty(nmb * mb, x.shape[1] if x.dim() == 2 else 1, device=dev) for x in self._gather_srcs]

                def front(i):
                    obs, act, val, adv, ret, logp, mu, sig, hl = (b[i * mb:(i + 1) * mb] for b in self._perm_bufs)
                    return self._ac_forward_backward((obs, obs, act, val, adv, ret, logp, mu, sig, (None, None), None, hl))

                torch.cuda.synchronize()
                for o in (self.optim_ac, self.optim_estimator):
                    o.zero_grad(set_to_none=True)
                from quadrupedal_agility_amd.rsl_rl.runners.on_policy_runner import _no_gc
                self._recording_ac = True
                graphs = []
                pool = torch.cuda.graph_pool_handle()
                try:
                    # r6: steps that run as chain launches are short (~0.37 ms), and two consecutive replays are ~8.5 us apart on the device: the
                    # nmb slots of an epoch as ONE recording (one replay per epoch)
                    epoch_graph = sync is None and self._train_chain_rows(mb) is not None and os.environ.get("QA_STEP_UNROLL", "1") != "0"
                    if epoch_graph:
                        g = torch.cuda.CUDAGraph()
                        with _no_gc(), torch.cuda.graph(g, pool=pool):
                            for i in range(nmb):
                                stats, kl = front(i)
                                self._ac_apply(kl)
                                fused_mod.accumulate_scalars(self._acc_ac, stats)       # (one launch; was torch.stack + add_)
                                for o in (self.optim_ac, self.optim_estimator):
                                    o.zero_grad(set_to_none=True)
                        graphs.append((g, None, None))
                    for i in range(0 if epoch_graph else nmb):
                        if sync is None:
                            g = torch.cuda.CUDAGraph()
                            with _no_gc(), torch.cuda.graph(g, pool=pool):
                                stats, kl = front(i)
                                self._ac_apply(kl)
                                fused_mod.accumulate_scalars(self._acc_ac, stats)
                            graphs.append((g, None, None))
                        else:
                            ga, gb = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
                            with _no_gc(), torch.cuda.graph(ga, pool=pool):
                                stats, kl = front(i)
                                grads = [p.grad for p in all_params if p.grad is not None]
                                packed = grads + [kl.detach().reshape(1)]
                                bucket = torch._utils._flatten_dense_tensors(packed)       # lives in the graphs' pool
                                stats_tmp = torch.stack(stats)
                            with _no_gc(), torch.cuda.graph(gb, pool=pool):
                                bucket.div_(sync.world)
                                parts = torch._utils._unflatten_dense_tensors(bucket, packed)
                                torch._foreach_copy_(grads, list(parts[:len(grads)]))
                                self._ac_apply(parts[-1].reshape(()))
                                self._acc_ac.add_(stats_tmp)
                            graphs.append((ga, gb, bucket))
                        for o in (self.optim_ac, self.optim_estimator):       # the next slot's recording creates its gradients afresh
                            o.zero_grad(set_to_none=True)
                    self._ac_graph = graphs
                finally:
                    self._recording_ac = False
            except Exception as e:      # never fatal
                print(f"[ppo update graph] capture failed, staying eager: {e}")
                self._ac_graph = False
                torch.cuda.synchronize()
                acc = torch.zeros(6, device=dev)
                for sample in st.mini_batch_generator(self.num_mini_batches, self.num_learning_epochs):
                    acc += torch.stack(self.update_actor_critic(sample))
                return acc
        self._priv_coef_dev.fill_(float(self._priv_reg_coef_now()))
        self._acc_ac.zero_()
        with torch.no_grad():
            self._hist_latent_all.copy_(self.actor_critic.infer_hist_latent(st.observations.flatten(0, 1)[:, self._hist_cols]))
        if perm is None:
            perm = torch.randperm(nmb * mb, device=dev)   # one permutation for all epochs (rollout_storage.py:122-157)
        fused_mod.gather_rows(perm, self._gather_srcs, dsts=self._perm_bufs)
        for _ in range(self.num_learning_epochs):
            for i in range(len(self._ac_graph)):            # nmb recordings of one step each, or one of the whole epoch
                ga, gb, bucket = self._ac_graph[i]
                ga.replay()
                if gb is not None:
                    self.grad_sync.all_reduce_(bucket)
                    gb.replay()
        return self._acc_ac.clone()

    def _disc_updates_recorded(self, n_steps, mb, tables=None):
        """The discriminator steps are tiny (3 x 1228 x 98 inputs, ~300 launches each, double backward) and therefore
        launch-bound: one step is recorded into a hipGraph -- sampling included, with the replay-buffer fill level and the
        info-max coefficient read from device scalars -- and replayed n_steps times."""
        dev = self.device
        if self._disc_graph is None:
            try:
                self._n_samples_dev = torch.zeros((), device=dev)
                self._acc_d = torch.zeros(11, device=dev)
                ml, rb = self.motion_loader, self.disc_storage
                n_lb, n_ulb = ml.preloaded_s_lb.shape[0], ml.preloaded_s_ulb.shape[0]

                # the sample indices of all n_steps steps are drawn once per update into tables (3 launches per update instead of
                # 5 per step); a recorded step reads its row block through a device-side step counter (qa_gather_rows)
                self._d_tables = [torch.zeros(n_steps, mb, dtype=torch.int64, device=dev) for _ in range(4)]      # i_pi, i_lb, i_ulb, labels
                self._d_step = torch.zeros((), dtype=torch.int64, device=dev)
                flat2 = lambda x: x.reshape(x.shape[0], -1)

                def one_step():
                    t_pi, t_lb, t_ulb, t_lab = self._d_tables
                    if self.use_fused_loss and self._fused_prep_ok(flat2(rb.states).shape[1]):
                        # sampling + preparation of the step's three batches in ONE launch (was: 3 row gathers, an index_select, the prepare launch)
                        w = self._task_weight_dev() if self.env.task_obs_weight_decay else None
                        x_all, e_pi, c_pi, label = fused_mod.disc_sample_prepare(
                            (t_lb, t_pi, t_ulb), self._d_step, (flat2(ml.preloaded_s_lb), flat2(rb.states), flat2(ml.preloaded_s_ulb)), flat2(rb.latent_eps),
                            flat2(rb.latent_c), t_lab, self.disc._task_mask, self.disc._frame_mult.view(-1), w, self.disc_normalizer)
                        out = self.update_ss_info_gail((None, e_pi, c_pi), (None, label), None, acc=self._acc_d, step=self._d_step,
                                                       prepared=(x_all, t_lb.shape[1], t_pi.shape[1]))
                        if not self._tail_folded:
                            self._acc_d.add_(_vec11(out))
                            self._d_step.add_(1)
                        return
                    if self.use_fused_loss:
                        s_pi, e_pi, c_pi = fused_mod.gather_rows(t_pi, [flat2(rb.states), flat2(rb.latent_eps), flat2(rb.latent_c)], block_dev=self._d_step)
                        (s_lb,) = fused_mod.gather_rows(t_lb, [flat2(ml.preloaded_s_lb)], block_dev=self._d_step)
                        (s_ulb,) = fused_mod.gather_rows(t_ulb, [flat2(ml.preloaded_s_ulb)], block_dev=self._d_step)
                    else:
                        sel = lambda tb: tb.index_select(0, self._d_step.view(1)).view(-1)
                        i_pi, i_lb, i_ulb = sel(t_pi), sel(t_lb), sel(t_ulb)
                        s_pi, e_pi, c_pi, s_lb, s_ulb = rb.states[i_pi], rb.latent_eps[i_pi], rb.latent_c[i_pi], ml.preloaded_s_lb[i_lb], ml.preloaded_s_ulb[i_ulb]
                    label = t_lab.index_select(0, self._d_step.view(1)).view(-1)
                    out = self.update_ss_info_gail((s_pi, e_pi, c_pi), (s_lb, label), s_ulb, acc=self._acc_d, step=self._d_step)
                    if not self._tail_folded:
                        self._acc_d.add_(_vec11(out))
                        self._d_step.add_(1)
                self._n_samples_dev.fill_(float(rb.num_samples))
                self._info_max_dev.fill_(float(self.info_max_coef_on))
                torch.cuda.synchronize()
                for o in (self.optim_d, self.optim_q_eps, self.optim_q_c):
                    o.zero_grad(set_to_none=True)
                g = torch.cuda.CUDAGraph()
                from quadrupedal_agility_amd.rsl_rl.runners.on_policy_runner import _no_gc
                # recorded on a stream of its own: the library GEMM workspace is keyed by the stream a launch is recorded on,
                # and this step is replayed CONCURRENTLY with the PPO step's recording (update(), "overlap")
                if self._disc_stream is None:
                    # (r6: a high-priority stream for this chain beside the PPO steps' GEMM waves changes nothing: config 3 45.2 ms either way)
                    self._disc_stream = torch.cuda.Stream(device=dev)
                self._recording_disc = True
                try:
                    # r6: `unroll` steps per recording (a divisor of n_steps): two consecutive replays are ~8.5 us apart on the device, 0.7 ms of an
                    # 80-step chain whose steps are ~0.26 ms; the steps read their row block through the device-side counter, so a replay is k steps
                    unroll = max((k for k in (8, 5, 4, 2, 1) if n_steps % k == 0), default=1) if os.environ.get("QA_STEP_UNROLL", "1") != "0" else 1
                    with _no_gc(), torch.cuda.graph(g, stream=self._disc_stream):
                        for _ in range(unroll):
                            one_step()
                finally:
                    self._recording_disc = False
                self._disc_graph, self._disc_unroll, self._disc_graph_steps = g, unroll, n_steps
            except Exception as e:      # never fatal
                print(f"[disc update graph] capture failed, staying eager: {e}")
                self._disc_graph = False
                torch.cuda.synchronize()
                acc = torch.zeros(11, device=dev)
                gens = zip(self.disc_storage.feed_forward_generator(n_steps, mb), self.motion_loader.feed_forward_generator_lb(n_steps, mb),
                           self.motion_loader.feed_forward_generator_ulb(n_steps, mb))
                for s_pi, s_lb, s_ulb in gens:
                    acc += _vec11(self.update_ss_info_gail(s_pi, s_lb, s_ulb))
                return acc
        self._n_samples_dev.fill_(float(self.disc_storage.num_samples))
        self._info_max_dev.fill_(float(self.info_max_coef_on))
        if self._task_w_dev is not None:        # eager rollouts keep the task-observation weight on the host: the recorded step reads this scalar
            self._task_w_dev.fill_(float(self.env.task_obs_weight))
        self._acc_d.zero_()
        ml = self.motion_loader
        t_pi, t_lb, t_ulb, t_lab = self._d_tables
        if tables is not None:          # checker hook (update()): the step samples what it is told to
            t_pi.copy_(tables["pi"]); t_lb.copy_(tables["lb"]); t_ulb.copy_(tables["ulb"])
        else:
            t_pi.copy_((torch.rand(t_pi.shape, device=dev) * self._n_samples_dev).long())
            torch.randint(0, ml.preloaded_s_lb.shape[0], t_lb.shape, device=dev, out=t_lb)
            torch.randint(0, ml.preloaded_s_ulb.shape[0], t_ulb.shape, device=dev, out=t_ulb)
        t_lab.copy_(ml.preloaded_label[t_lb.view(-1)].view(t_lb.shape))
        self._d_step.zero_()
        assert n_steps == self._disc_graph_steps
        for _ in range(n_steps // self._disc_unroll):
            self._disc_graph.replay()
        return self._acc_d          # persistent: the caller copies it on ITS stream after joining

    def _sync_grads(self, params):
        if self.grad_sync is not None:
            self.grad_sync(params)

    def _ac_forward_backward(self, sample):
        """Everything of one PPO minibatch step up to the gradients: returns (6 loss scalars, minibatch KL or None)."""
        (obs, critic_obs, actions, target_values, advantages, returns, old_logp, old_mu, old_sigma, _hid, _masks) = sample[:11]
        hist_latent = sample[11] if len(sample) > 11 else None       # recorded updates: evaluated once per update() for all samples
        ac = self.actor_critic
        fused = self._on_gpu and self.use_fused_loss
        if fused and not ac.fixed_std:
            return self._ac_forward_backward_direct(obs, critic_obs, actions, target_values, advantages, returns, old_logp, old_mu, old_sigma, hist_latent)
        if fused:
            # mean / value only; log-prob, entropy, KL, the four loss terms and their gradient are ONE kernel (fused.py)
            mu = ac._actor_mean(obs.detach(), False)
            value = ac.evaluate(critic_obs.detach())
            ppo, stats = ppo_loss(mu, ac.std, value, actions, old_logp, old_mu, old_sigma, advantages, returns, target_values,
                                  clip=self.clip_param, c_surr=self.surrogate_loss_coef, c_value=self.value_loss_coef,
                                  c_bound=self.bounds_loss_coef, c_entropy=self.entropy_coef, clipped_value=self.use_clipped_value_loss)
        else:
            ac.update_distribution(obs.detach(), False)      # the reference calls act() here and discards the sample (gail.py:333)
            logp = ac.get_actions_log_prob(actions)
            value = ac.evaluate(critic_obs.detach())
            mu, sigma, entropy = ac.action_mean, ac.action_std, ac.entropy

        a = self.num_prop; b = a + self.num_explicit; c = b + self.num_latent; d = c + self.num_hist * self.num_prop
        obs_prop, obs_explicit, obs_latent, obs_hist = obs[:, :a], obs[:, a:b], obs[:, b:c], obs[:, c:d]
        priv_latent = ac.infer_priv_latent(obs_latent)
        if hist_latent is None:
            with torch.no_grad():
                hist_latent = ac.infer_hist_latent(obs_hist)
        if fused:
            priv_reg_loss = fused_mod.pair_loss(priv_latent, hist_latent, fused_mod.PAIR_ROW_L2)        # value + gradient in one pass
        else:
            priv_reg_loss = (priv_latent - hist_latent).norm(p=2, dim=1).mean()
        # a device scalar while the step is being recorded (the ramp changes between iterations, replays must see it)
        priv_reg_coef = self._priv_coef_dev if self._recording_ac else self._priv_reg_coef_now()

        # estimator regression on the true privileged explicit state (gail.py:356-362).  Its parameters do not enter the
        # actor-critic objective, so its optimiser step can wait until both backward passes are done: data-parallel runs
        # then need ONE collective per minibatch (estimator grads + actor-critic grads + the KL scalar in one bucket)
        if fused:
            estimator_loss = fused_mod.pair_loss(self.estimator(obs_prop), obs_explicit, fused_mod.PAIR_MSE)
        else:
            estimator_loss = (self.estimator(obs_prop) - obs_explicit).pow(2).mean()
        self.optim_estimator.zero_grad()
        estimator_loss.backward()

        adaptive = self.desired_kl is not None and self.schedule == "adaptive"
        kl_mean = None
        if adaptive:
            with torch.no_grad():
                if fused:
                    kl_mean = stats[5]
                else:
                    kl = torch.sum(torch.log(sigma / old_sigma + 1.0e-5) +
                                   (torch.square(old_sigma) + torch.square(old_mu - mu)) / (2.0 * torch.square(sigma)) - 0.5, dim=-1)
                    kl_mean = kl.mean()

        if fused:
            surrogate_loss, value_loss, b_mean, ent_mean = stats[1], stats[2], stats[3], stats[4]
            loss = ppo + priv_reg_coef * priv_reg_loss
        else:
            adv = torch.squeeze(advantages)
            ratio = torch.exp(logp - torch.squeeze(old_logp))
            surrogate_loss = torch.max(-adv * ratio, -adv * torch.clamp(ratio, 1.0 - self.clip_param, 1.0 + self.clip_param)).mean()
            if self.use_clipped_value_loss:
                v_clip = target_values + (value - target_values).clamp(-self.clip_param, self.clip_param)
                value_loss = torch.max((value - returns).pow(2), (v_clip - returns).pow(2)).mean()
            else:
                value_loss = (returns - value).pow(2).mean()
            b_loss = (torch.clamp(mu + 1.0, max=0.0) ** 2 + torch.clamp(mu - 1.0, min=0.0) ** 2).sum(dim=-1)
            b_mean, ent_mean = b_loss.mean(), entropy.mean()
            loss = (self.surrogate_loss_coef * surrogate_loss + self.value_loss_coef * value_loss +
                    self.bounds_loss_coef * b_mean - self.entropy_coef * ent_mean + priv_reg_coef * priv_reg_loss)
        self.optim_ac.zero_grad()
        loss.backward()
        return (surrogate_loss.detach(), value_loss.detach(), b_mean.detach(), ent_mean.detach(),
                priv_reg_loss.detach(), estimator_loss.detach()), kl_mean

    def _ac_forward_backward_direct(self, obs, critic_obs, actions, target_values, advantages, returns, old_logp, old_mu, old_sigma, hist_latent):
        """The same step on the GPU without scalar loss nodes: the kernels return every term's value AND its gradient w.r.t. the
        network outputs, which go straight into autograd.backward() of those outputs (what `loss.backward()` computes, minus the
        ones-fills, scalings and accumulations of the scalar graph: ~10 launches per step).

        The critic, the actor and the two small networks (privileged encoder, estimator) share nothing until the objective and
        nothing again after it, so they run as three BRANCHES on three streams -- forward and, because autograd replays every node
        on the stream its forward ran on, backward too.  A branch is a chain of dependent launches, two thirds of them <= 8 us
        kernels that leave the GPU idle between them; side by side, one branch's small kernels fill the gaps of another's and its
        GEMMs overlap the others' launch latencies.  Recorded into the step's hipGraph the streams become parallel graph branches."""
        ac = self.actor_critic
        a = self.num_prop; b = a + self.num_explicit; c = b + self.num_latent; d = c + self.num_hist * self.num_prop
        priv_reg_coef = self._priv_coef_dev if self._recording_ac else self._priv_reg_coef_now()
        chain = self._train_chain(obs, critic_obs)
        self._chain_sides = None
        if chain is not None:
            return self._ac_forward_backward_chain(chain, obs, actions, target_values, advantages, returns, old_logp, old_mu, old_sigma, hist_latent, priv_reg_coef)
        cur = torch.cuda.current_stream()
        if self.branch_streams and self._branch is None:
            self._branch = (torch.cuda.Stream(device=self.device), torch.cuda.Stream(device=self.device))
        s_critic, s_small = self._branch if self.branch_streams else (cur, cur)
        if self.branch_streams:
            s_critic.wait_stream(cur); s_small.wait_stream(cur)
        self.optim_estimator.zero_grad()
        self.optim_ac.zero_grad()
        with torch.cuda.stream(s_critic):
            value = ac.evaluate(critic_obs.detach())
        with torch.cuda.stream(s_small):
            priv_latent = ac.infer_priv_latent(obs[:, b:c])
            if hist_latent is None:
                with torch.no_grad():
                    hist_latent = ac.infer_hist_latent(obs[:, c:d])
            est = self.estimator(obs[:, :a])
            if torch.is_tensor(priv_reg_coef) and priv_reg_coef.dtype == torch.float32 and priv_reg_coef.device == obs.device and os.environ.get("QA_PAIR_LOSSES", "1") != "0":
                # r6 (ABI 18): one launch, rows through LDS (qa_pair_loss walks 24,576 rows of 29 floats lane by lane: 29 us each)
                (priv_reg_loss, g_priv), (estimator_loss, g_est) = fused_mod.pair_losses_raw(
                    [(priv_latent, hist_latent, fused_mod.PAIR_ROW_L2, priv_reg_coef), (est, obs[:, a:b], fused_mod.PAIR_MSE, None)])
            else:
                priv_reg_loss, g_priv = fused_mod.pair_loss_raw(priv_latent, hist_latent, fused_mod.PAIR_ROW_L2)
                g_priv = g_priv * priv_reg_coef
                estimator_loss, g_est = fused_mod.pair_loss_raw(est, obs[:, a:b], fused_mod.PAIR_MSE)
        mu = ac._actor_mean(obs.detach(), False)
        if self.branch_streams:
            cur.wait_stream(s_critic)
        out, dmu, dstd, dvalue = fused_mod.ppo_loss_raw(mu, ac.std, value, actions, old_logp, old_mu, old_sigma, advantages, returns, target_values,
                                                        clip=self.clip_param, c_surr=self.surrogate_loss_coef, c_value=self.value_loss_coef,
                                                        c_bound=self.bounds_loss_coef, c_entropy=self.entropy_coef,
                                                        clipped_value=self.use_clipped_value_loss)
        if self.branch_streams:
            cur.wait_stream(s_small)
            if not self._recording_ac:
                # eager (warm-up) steps: these blocks were allocated on the side streams and are read by current-stream kernels from here on;
                # tell the caching allocator, or another allocation on a side stream could reuse them under those reads (a recording has
                # its own private pool)
                for t in (value, priv_latent, est, g_priv, g_est):
                    t.record_stream(cur)
        # (tried and dropped: the dW GEMMs of the Linear+ELU layers on a fourth stream, off the dX critical path -- 34.1 -> 37.8 ms:
        # big GEMMs running side by side slow each other more than the shorter dependency chain saves)
        if self._recording_ac:
            fused_mod.assert_recordable_graph([est, mu, value, priv_latent], "PPO step")
        # r5: the weight / bias gradients of the wide layers stay in parts (split-K slabs, row-block column sums) and are added by the first
        # pass of the optimiser steps that follow (_ac_apply -> ClipAdam -> qa_clip_adam_step_reduce): 19 finish launches fewer per step.
        # The data-parallel step needs finished gradients for its bucket: one launch adds them all (flush_pending_grads).
        with fused_mod.deferred_grad_finishes():
            torch.autograd.backward([est, mu, value, priv_latent], [g_est, dmu, dvalue.view_as(value), g_priv])
        if self.grad_sync is not None:
            fused_mod.flush_pending_grads()
        ac.std.grad = dstd.view_as(ac.std)          # std enters the objective through qa_ppo_loss only
        adaptive = self.desired_kl is not None and self.schedule == "adaptive"
        return (out[1], out[2], out[3], out[4], priv_reg_loss, estimator_loss), (out[5] if adaptive else None)

    def _disc_train_chain(self, rows, n_u):
        from quadrupedal_agility_amd.rsl_rl.algorithms import train_chain
        if not (train_chain.ENABLED and train_chain.DISC_ENABLED) or rows > train_chain.MAX_ROWS:
            return None
        cache = self.__dict__.setdefault("_disc_train_chains", {})
        if (rows, n_u) not in cache:
            cache[(rows, n_u)] = train_chain.DiscTrainChain.describe(self.disc, rows, n_u) or False
        return cache[(rows, n_u)] or None

    def _train_chain(self, obs, critic_obs):
        """train_chain.PpoTrainChain for this minibatch size, or None: few rows per step (the per-GPU share of the 8-GPU job), where the
        step is a serial chain of launch-latency-sized kernels; actor and critic reading the same observation rows (they do: the reference
        stores the row twice, legged_robot.py:321)."""
        if obs.data_ptr() != critic_obs.data_ptr() or obs.stride(1) != 1:
            return None
        return self._train_chain_rows(obs.shape[0])

    def _train_chain_rows(self, rows):
        from quadrupedal_agility_amd.rsl_rl.algorithms import train_chain
        if not (train_chain.ENABLED and self._on_gpu and self.use_fused_loss) or rows > train_chain.MAX_ROWS or self.actor_critic.fixed_std:
            return None
        cache = self.__dict__.setdefault("_train_chains", {})
        if rows not in cache:
            cache[rows] = train_chain.PpoTrainChain.describe(self.actor_critic, self.estimator, rows) or False
        return cache[rows] or None

    def _ac_forward_backward_chain(self, chain, obs, actions, target_values, advantages, returns, old_logp, old_mu, old_sigma, hist_latent, priv_reg_coef):
        """_ac_forward_backward_direct with the networks as two chain launches (train_chain.py): forward with saved activations -> the three
        objectives (value + gradient at the networks' outputs, as before) -> input-gradient chain -> 13 weight-gradient products in parts."""
        ac = self.actor_critic
        a = self.num_prop; b = a + self.num_explicit; c = b + self.num_latent; d = c + self.num_hist * self.num_prop
        self.optim_estimator.zero_grad()
        self.optim_ac.zero_grad()
        chain.pack()                                 # the last optimiser step changed the weights: two small launches
        est, mu, value, priv_latent = chain.forward(obs)
        if hist_latent is None:
            with torch.no_grad():
                hist_latent = ac.infer_hist_latent(obs[:, c:d])
        if torch.is_tensor(priv_reg_coef) and priv_reg_coef.dtype == torch.float32 and priv_reg_coef.device == obs.device and os.environ.get("QA_PAIR_LOSSES", "1") != "0":
            # r6 (ABI 18): both losses, their finishes and the coefficient's multiply in ONE launch (were five)
            (priv_reg_loss, g_priv), (estimator_loss, g_est) = fused_mod.pair_losses_raw(
                [(priv_latent, hist_latent, fused_mod.PAIR_ROW_L2, priv_reg_coef), (est, obs[:, a:b], fused_mod.PAIR_MSE, None)])
        else:
            priv_reg_loss, g_priv = fused_mod.pair_loss_raw(priv_latent, hist_latent, fused_mod.PAIR_ROW_L2)
            g_priv = g_priv * priv_reg_coef
            estimator_loss, g_est = fused_mod.pair_loss_raw(est, obs[:, a:b], fused_mod.PAIR_MSE)
        out, dmu, dstd, dvalue = fused_mod.ppo_loss_raw(mu, ac.std, value, actions, old_logp, old_mu, old_sigma, advantages, returns, target_values,
                                                        clip=self.clip_param, c_surr=self.surrogate_loss_coef, c_value=self.value_loss_coef,
                                                        c_bound=self.bounds_loss_coef, c_entropy=self.entropy_coef,
                                                        clipped_value=self.use_clipped_value_loss)
        chain.backward(g_est, dmu, dvalue, g_priv)
        self._chain_sides = chain.sides if self.grad_sync is None else None
        if self.grad_sync is not None:
            fused_mod.flush_pending_grads()
        ac.std.grad = dstd.view_as(ac.std)
        adaptive = self.desired_kl is not None and self.schedule == "adaptive"
        return (out[1], out[2], out[3], out[4], priv_reg_loss, estimator_loss), (out[5] if adaptive else None)

    def _ac_apply(self, kl_mean):
        """Second half of the step: clip + step the estimator, the KL-adaptive learning rate, clip + step the actor-critic."""
        sides = getattr(self, "_chain_sides", None)
        if sides is not None:                        # chain steps (few rows): the estimator's optimiser beside the actor-critic's, disjoint parameters
            with sides.fork(0):
                self._step_estimator.step()
        elif (self._step_pair is not None and self.use_fused_loss and (kl_mean is None or (kl_mean.dtype == torch.float32 and kl_mean.is_contiguous()))
              and self._step_pair.step(kl_mean, self.desired_kl)):
            # r6 (ABI 18): both clipped Adam steps and the KL rule between them as three launches instead of seven (fused.ClipAdamPair)
            if fused_mod.pending_grads():
                fused_mod.flush_pending_grads()
            return
        else:
            self._step_estimator.step()              # clip_grad_norm_(max_grad_norm) + Adam, three launches on the GPU
        if kl_mean is not None:
            self._apply_kl_schedule(kl_mean)
        self._step_ac.step()
        if sides is not None:
            sides.join()
        if fused_mod.pending_grads():                # a parameter outside both optimisers had its gradient left in parts: finish it now
            fused_mod.flush_pending_grads()

    def update_actor_critic(self, sample):
        stats, kl_mean = self._ac_forward_backward(sample)
        if self.grad_sync is not None:      # one bucket: estimator grads | actor-critic grads | KL (every rank takes the same LR branch)
            params = list(self.estimator.parameters()) + list(self.actor_critic.parameters())
            synced = self.grad_sync(params, extra=[kl_mean] if kl_mean is not None else None)
            if kl_mean is not None:
                kl_mean = synced[0].reshape(())
        self._ac_apply(kl_mean)
        return stats

    def _apply_kl_schedule(self, kl_mean):
        """lr /= 1.5 if KL > 2 target; lr *= 1.5 if 0 < KL < target/2; clamp [1e-5, 1e-2] (gail.py:367-379)."""
        tgt = self.desired_kl
        if self._on_gpu and self.use_fused_loss and kl_mean.dtype == torch.float32 and kl_mean.is_contiguous():
            fused_mod.kl_lr_rule(kl_mean, tgt, self._lr_ac)          # one launch on device scalars (was 11)
        elif self._on_gpu:
            lr = self._lr_ac
            down = torch.clamp(lr / 1.5, min=1e-5)
            up = torch.clamp(lr * 1.5, max=1e-2)
            new = torch.where(kl_mean > tgt * 2.0, down, torch.where((kl_mean < tgt / 2.0) & (kl_mean > 0.0), up, lr))
            lr.copy_(new)
        else:
            k = float(kl_mean)
            if k > tgt * 2.0:
                self._lr_ac = max(1e-5, self._lr_ac / 1.5)
            elif tgt / 2.0 > k > 0.0:
                self._lr_ac = min(1e-2, self._lr_ac * 1.5)
            for g in self.optim_ac.param_groups:
                g["lr"] = self._lr_ac

    def _fused_prep_ok(self, width):
        nm = self.disc_normalizer
        return bool(self._on_gpu and self.use_fused_loss and nm is not None and hasattr(nm, "count") and torch.is_tensor(getattr(nm, "mean", None))
                    and self.disc.disc_obs_len * self.disc.num_disc_obs == width)

    def _task_weight_dev(self):
        w = getattr(self.env, "task_obs_weight_dev", None)
        if w is None:
            if self._task_w_dev is None:
                self._task_w_dev = torch.ones((), device=self.device)
            if not torch.cuda.is_current_stream_capturing():
                self._task_w_dev.fill_(float(self.env.task_obs_weight))
            w = self._task_w_dev
        return w

    def update_ss_info_gail(self, sample_disc_policy, sample_disc_expert_lb, sample_disc_expert_ulb, acc=None, step=None, prepared=None):
        """returns the 11 logged values (a tuple of scalars, or ONE 11-vector from qa_disc_step_tail -- `_vec11` takes either); with `acc` / `step`
        (the recorded step) the fused tail also adds them to the accumulator and bumps the device-side step counter: `self._tail_folded`"""
        self._tail_folded = False
        policy_state, policy_eps, policy_c = sample_disc_policy
        expert_lb, label_lb = sample_disc_expert_lb
        expert_ulb = sample_disc_expert_ulb
        w = getattr(self.env, "task_obs_weight_dev", None)
        fused_prep = prepared is None and self._fused_prep_ok(policy_state.shape[-1])
        if prepared is not None:        # the recorded step's sampling front wrote the prepared matrix already (qa_disc_sample_prepare)
            x_all, nl, npi = prepared
            expert_lb, policy_state, expert_ulb = x_all[:nl], x_all[nl:nl + npi], x_all[nl + npi:]
        elif fused_prep:
            # task weighting, frame weighting, normalisation + clip of the three batches, written as ONE (3B, 98) matrix
            w = self._task_weight_dev()
            x_all = disc_prepare([expert_lb.reshape(len(expert_lb), -1), policy_state.reshape(len(policy_state), -1), expert_ulb.reshape(len(expert_ulb), -1)],
                                 self.disc._task_mask, self.disc._frame_mult.view(-1), w if self.env.task_obs_weight_decay else None, self.disc_normalizer)
            nl, npi = len(expert_lb), len(policy_state)
            expert_lb, policy_state, expert_ulb = x_all[:nl], x_all[nl:nl + npi], x_all[nl + npi:]
        else:
            w = self.env.task_obs_weight if w is None else w
            prep = lambda x: self.disc.prepare_input(x.view(len(x), self.disc_obs_len, -1), w)
            policy_state, expert_lb, expert_ulb = prep(policy_state), prep(expert_lb), prep(expert_ulb)
            if self.disc_normalizer is not None:
                with torch.no_grad():
                    policy_state = self.disc_normalizer.normalize_torch(policy_state, self.device)
                    expert_lb = self.disc_normalizer.normalize_torch(expert_lb, self.device)
                    expert_ulb = self.disc_normalizer.normalize_torch(expert_ulb, self.device)
            x_all = None

        # ONE trunk pass over [labelled expert | policy | unlabelled expert] instead of the reference's four
        # (three forwards + a separate forward for the gradient penalty, gail.py:452-492): same functions of the
        # same parameters, so losses and gradients are identical up to GEMM rounding; half the launches.
        b_lb, b_pi = expert_lb.shape[0], policy_state.shape[0]
        analytic_gp = self.disc._relu_trunk() is not None
        fused_heads = self._on_gpu and self.use_fused_loss and self.disc_loss_function == "MSELoss"
        dchain, normaliser_done = None, False
        if analytic_gp and fused_heads and x_all is not None:
            dchain = self._disc_train_chain(x_all.shape[0], expert_ulb.shape[0])
        if dchain is not None:
            # r6: trunk + heads, the gradient penalty's path there and back, and the heads' way back as THREE chain launches (train_chain.DiscTrainChain)
            gp_proxies = None
            dchain.pack()
            d_all, eps_all, logits_all = dchain.forward(x_all)
            g = dchain.penalty_gradient()                       # d logit / d x on the unlabelled rows: the penalty's argument
            # r6 (ABI 18): the class head's softmax and its backward run inside the objective's launch (qa_disc_loss_logits) -- three launches fewer per step
            logits_in_kernel = os.environ.get("QA_DISC_LOSS_LOGITS", "1") != "0"
            c_all = logits_all if logits_in_kernel else torch.softmax(logits_all, -1)               # (the objective kernel clamps)
            if self.disc_normalizer is not None and self.grad_sync is None:
                # the input normaliser folds in this step's batches (gail.py:524-528, at the end of the step there): nothing else in the step reads or
                # writes its moments, so the fold runs beside the step instead of behind it
                nb_ = [policy_state, expert_lb, expert_ulb]          # (slices of the prepared matrix, in the order the reference folds them)
                dchain.beside(lambda: self.disc_normalizer.update_torch(nb_))
                normaliser_done = True
        elif analytic_gp:      # d logit / d x on the unlabelled rows as a chain of small GEMMs (discriminator.py), no second-order graph
            gp_proxies = [] if os.environ.get("QA_DISC_GP_PROXIES", "1") != "0" else None
            (d_all, eps_all, c_all), g = self.disc.forward_with_input_gradient(x_all if x_all is not None else torch.cat([expert_lb, policy_state, expert_ulb], dim=0),
                                                                                slice(b_lb + b_pi, None), clamp=not fused_heads, proxies=gp_proxies)
        else:
            x_ulb = expert_ulb.clone().requires_grad_(True)
            d_all, eps_all, c_all = self.disc(torch.cat([expert_lb, policy_state, x_ulb], dim=0))
        pred_c_lb = c_all[:b_lb]
        logits_pi, eps, pred_c = d_all[b_lb:b_lb + b_pi], eps_all[b_lb:b_lb + b_pi], c_all[b_lb:b_lb + b_pi]
        logits_exp, pred_c_ulb = d_all[b_lb + b_pi:], c_all[b_lb + b_pi:]
        direct = fused_heads and analytic_gp      # no scalar loss nodes: the kernels' gradients go straight into autograd.backward below
        if fused_heads:
            # the four head losses, their gradient, the four logged accuracies and the prior mean: ONE kernel (fused.py)
            self._info_max_dev.fill_(float(self.info_max_coef_on)) if not torch.cuda.is_current_stream_capturing() else None
            kw = dict(c_ss=self.ss_coef, info_coef_dev=self._info_max_dev, c_disc=self.disc_coef, c_us=self.us_coef)
            if direct:
                hs, g_d, g_eps, g_c = fused_mod.disc_loss_raw(d_all, eps_all, c_all, label_lb, policy_eps, policy_c, b_lb, b_pi, expert_ulb.shape[0],
                                                              from_logits=dchain is not None and logits_in_kernel, **kw)
            else:
                heads, hs = disc_loss(d_all, eps_all, c_all, label_lb, policy_eps, policy_c, b_lb, b_pi, expert_ulb.shape[0], **kw)
            ss_loss, info_max_loss, disc_loss_v, us_loss = hs[1], hs[2], hs[3], hs[4]
            pred_mean = hs[9:14]
        else:
            ss_loss = F.cross_entropy(pred_c_lb, label_lb)          # CE on softmaxed probabilities, as the reference
            policy_c_idx = torch.argmax(policy_c, dim=-1)
            pred_mean = torch.mean(pred_c_ulb, dim=0).detach()
            info_max_loss = torch.mean(-torch.sum(pred_c_ulb * torch.log(pred_c_ulb + 1e-20), dim=-1))
            if self.disc_loss_function == "BCEWithLogitsLoss":
                l_exp = F.binary_cross_entropy_with_logits(logits_exp, torch.ones_like(logits_exp))
                l_pi = F.binary_cross_entropy_with_logits(logits_pi, torch.zeros_like(logits_pi))
            elif self.disc_loss_function == "MSELoss":
                l_exp = F.mse_loss(logits_exp, torch.ones_like(logits_exp))
                l_pi = F.mse_loss(logits_pi, -torch.ones_like(logits_pi))
            elif self.disc_loss_function == "WassersteinLoss":
                l_exp, l_pi = -logits_exp.mean(), logits_pi.mean()
            else:
                raise ValueError("Unexpected loss function specified")
            disc_loss_v = 0.5 * (l_pi + l_exp)
            us_loss = F.l1_loss(eps, policy_eps)
        reg_w = [m.weight for m in self.disc.trunk.modules() if isinstance(m, nn.Linear)] + [self.disc.linear.weight]
        fused_tail = direct and len(reg_w) <= 7 and all(w.is_cuda and w.is_contiguous() for w in reg_w)
        prior_in_tail = (fused_tail and self.grad_sync is None and torch.is_tensor(self.env.prior_parameters) and self.env.prior_parameters.is_cuda
                         and self.env.prior_parameters.dtype == torch.float32 and self.env.prior_parameters.is_contiguous() and self.env.prior_parameters.numel() <= 5)
        if self.grad_sync is None and not prior_in_tail:      # data-parallel: the vector rides in the gradient bucket below (one collective per step)
            prior = self.env.prior_parameters
            if torch.is_tensor(prior) and prior.is_cuda:    # the arena's own tensor: the EMA (gail.py:463-464) in place, 2 launches
                prior.mul_(1 - self.prior_soft_coef).add_(pred_mean, alpha=self.prior_soft_coef)
            else:
                self.env.prior_parameters = pred_mean * self.prior_soft_coef + prior * (1 - self.prior_soft_coef)
        # gradient penalty on the unlabelled expert samples (double backward through the shared pass)
        if not analytic_gp:
            g = torch.autograd.grad(logits_exp, x_ulb, grad_outputs=torch.ones_like(logits_exp), create_graph=True, retain_graph=True, only_inputs=True)[0]
        if dchain is not None:
            dchain.wait_penalty()
        if fused_tail:  # the three sums of squares, the 11-vector, the accumulator and the step counter: one launch at the end of the step
            gdet = g.detach()
            grad_pen_loss = None
        elif direct:    # value for the log; its gradient w.r.t. g, 2 c_gp g / rows, is fed to backward() directly
            with torch.no_grad():
                gdet = g.detach()
                grad_pen_loss = gdet.square().sum() / gdet.shape[0]
        else:
            grad_pen_loss = torch.mean(torch.sum(torch.square(g), dim=-1))
        # The two weight regularisers (gail.py:497-504) are functions of the weights alone: c_logit |W_out|^2 + c_wd (|W_1|^2 +
        # |W_2|^2 + |W_out|^2).  On the GPU their values come from one multi-tensor norm and their gradient 2 c W is added to
        # .grad after backward() with one multi-tensor launch, instead of ~15 autograd launches; same numbers.
        fold_reg = fused_heads and all(w.is_cuda for w in reg_w)
        if fused_tail:
            # issued HERE, before the optimiser step below changes the weights it reads
            with torch.no_grad():
                out11 = fused_mod.disc_step_tail(hs, gdet, reg_w, acc=acc, step=step, prior=self.env.prior_parameters if prior_in_tail else None,
                                                 prior_soft_coef=self.prior_soft_coef)
            self._tail_folded = acc is not None
            disc_logit_loss = disc_weight_decay = None
            rest = None
        elif fold_reg:
            with torch.no_grad():
                sq = torch.stack(torch._foreach_norm(reg_w)).square()
                disc_logit_loss, disc_weight_decay = sq[-1], sq.sum()
            rest = None if direct else self.disc_grad_penalty * grad_pen_loss
        else:
            disc_logit_loss = torch.sum(torch.square(self.disc.get_disc_logit_weights()))
            disc_weight_decay = torch.sum(torch.square(torch.cat(self.disc.get_disc_weights(), dim=-1)))
            rest = self.disc_grad_penalty * grad_pen_loss + self.disc_logit_reg * disc_logit_loss + self.disc_weight_decay * disc_weight_decay
        if direct:
            loss = None
        elif fused_heads:
            loss = heads + rest
        else:
            info_coef = self._info_max_dev if (self._info_max_dev is not None and torch.cuda.is_current_stream_capturing()) else self.info_max_coef_on
            loss = self.ss_coef * ss_loss + info_coef * info_max_loss + self.disc_coef * disc_loss_v + self.us_coef * us_loss + rest
        for o in (self.optim_d, self.optim_q_eps, self.optim_q_c):
            o.zero_grad()
        stack_src = None
        if dchain is not None:
            with torch.no_grad():       # softmax backward on (rows, 5): d loss / d logits = c * (g_c - <g_c, c>): inside the objective's launch, or torch's two
                g_logits = g_c if logits_in_kernel else torch._softmax_backward_data(g_c, c_all, -1, torch.float32)
            # r6 (ABI 18): the products stay in parts and ONE launch adds them (the penalty's with its factor, the regularisers' 2 c W), writes `.grad`
            # and applies the three optimisers' states in order (fused.StackedAdam) -- unless gradients travel between ranks first
            stack = self._disc_stack if (fold_reg and self.grad_sync is None and self._disc_stack is not None and self._disc_stack.ready()) else None
            stack_src = dchain.backward(g_d, g_eps, g_logits, self.disc_grad_penalty, finish=stack is None)
        elif direct:
            if self._recording_disc:
                fused_mod.assert_recordable_graph([d_all, eps_all, c_all, g], "discriminator step")
            torch.autograd.backward([d_all, eps_all, c_all, g], [g_d, g_eps, g_c, gdet * (2.0 * self.disc_grad_penalty / gdet.shape[0])])
        else:
            loss.backward()
        if analytic_gp and gp_proxies:
            with torch.no_grad():       # the penalty's share of the weight gradients (see forward_with_input_gradient)
                torch._foreach_add_([w.grad for w, _ in gp_proxies], [q.grad for _, q in gp_proxies])
        if fold_reg and stack_src is None:
            with torch.no_grad():
                torch._foreach_add_([w.grad for w in reg_w[:-1]], reg_w[:-1], alpha=2.0 * self.disc_weight_decay)
                reg_w[-1].grad.add_(reg_w[-1], alpha=2.0 * (self.disc_weight_decay + self.disc_logit_reg))
        norm_batches = [policy_state, expert_lb, expert_ulb]
        synced_moments = None
        if self.grad_sync is not None:
            # ONE collective per discriminator step: gradients | class mean of the prior EMA | the normaliser's batch moments
            extra = [pred_mean]
            nm = self.disc_normalizer
            cold = nm is not None and hasattr(nm, "batch_moments") and nm.cold()
            if nm is not None and hasattr(nm, "batch_moments") and not cold:
                extra.append(nm.batch_moments(norm_batches))          # fp32, taken about the running mean
            back = self.grad_sync(list(self.disc.parameters()), extra=extra)
            pred_mean = back[0]
            if len(back) > 1:
                synced_moments = back[1].view(len(norm_batches), 2, -1)
            elif cold:      # the first folds of a fresh normaliser: raw moments in fp64 through a collective of their own (4.7 KB)
                m64 = nm.batch_moments_exact(norm_batches)
                self.grad_sync.all_reduce_(m64)
                synced_moments = m64 / self.grad_sync.world
            self.env.prior_parameters = pred_mean * self.prior_soft_coef + self.env.prior_parameters * (1 - self.prior_soft_coef)
        if stack_src is not None:
            reg = {w: 2.0 * self.disc_weight_decay for w in reg_w[:-1]}
            reg[reg_w[-1]] = 2.0 * (self.disc_weight_decay + self.disc_logit_reg)
            stack.step(stack_src, reg)
        else:                           # one after the other: the trunk's parameters are in all three (gail.py:107-132)
            for o in self._step_disc:
                o.step()
        if not self._recording_disc:          # the recorded step leaves this to update(): once after the loop is the same thing
            self._clamp_std()
        if self.disc_normalizer is not None and not normaliser_done:
            if synced_moments is not None:      # data-parallel: the moments of the GLOBAL batches, identical on every rank
                world = self.grad_sync.world
                self.disc_normalizer.update_from_batch_moments(synced_moments, [b.shape[0] * world for b in norm_batches])
            else:
                self.disc_normalizer.update_torch(norm_batches)
        if fused_tail:
            return out11
        if fused_heads:
            acc_lb, acc_pi, acc_exp, acc_ulb = hs[5], hs[6], hs[7], hs[8]
        else:
            with torch.no_grad():
                acc_lb = (torch.argmax(pred_c_lb, dim=-1) == label_lb).float().mean()
                acc_pi = (logits_pi < 0).float().mean()
                acc_exp = (logits_exp > 0).float().mean()
                acc_ulb = (torch.argmax(pred_c, dim=-1) == policy_c_idx).float().mean()
        return (ss_loss.detach(), info_max_loss.detach(), disc_loss_v.detach(), us_loss.detach(), grad_pen_loss.detach(),
                disc_logit_loss.detach(), disc_weight_decay.detach(), acc_lb, acc_pi, acc_exp, acc_ulb)

    def _dagger_step(self, obs):
        ac = self.actor_critic
        a = self.num_prop + self.num_explicit; b = a + self.num_latent; c = b + self.num_hist * self.num_prop
        with torch.no_grad():
            target = ac.infer_priv_latent(obs[:, a:b])
        hist = ac.infer_hist_latent(obs[:, b:c])
        if self._on_gpu and fused_mod.ENABLED:      # value + gradient in one pass, fixed-order row sum (no torch reduction in the recorded step)
            loss = fused_mod.pair_loss(hist, target, fused_mod.PAIR_ROW_L2)
        else:
            loss = (target - hist).norm(p=2, dim=1).mean()
        self.optim_hist_encoder.zero_grad()
        if self._on_gpu and torch.cuda.is_current_stream_capturing():
            fused_mod.assert_recordable_graph([hist], "DAgger step")
        loss.backward()
        params = list(ac.history_encoder.parameters())
        self._sync_grads(params)
        self._step_hist_encoder.step()
        return loss.detach()

    def update_dagger(self, perm=None):
        """History-encoder regression onto the privileged latent, every dagger_update_freq iterations (gail.py:543-575).  `perm`: checker hook, the
        rollout permutation this update otherwise draws (see update())."""
        n = self.num_learning_epochs * self.num_mini_batches
        st = self.storage
        if (self._on_gpu and self.use_update_graph and self.grad_sync is None and self._dagger_warm >= 1 and self._dagger_graph is not False):
            # same recording scheme as the PPO step: gather from the device index buffer + one optimiser step per replay
            mb = st.num_envs * st.num_transitions_per_env // self.num_mini_batches
            if self._dagger_graph is None:
                try:
                    self._dg_idx = torch.zeros(mb, dtype=torch.int64, device=self.device)
                    self._dg_acc = torch.zeros((), device=self.device)
                    flat_obs = st.observations.flatten(0, 1)

                    def one_step():
                        self._dg_acc.add_(self._dagger_step(flat_obs[self._dg_idx]))
                    torch.cuda.synchronize()
                    self.optim_hist_encoder.zero_grad(set_to_none=True)
                    from quadrupedal_agility_amd.rsl_rl.runners.on_policy_runner import _no_gc
                    g = torch.cuda.CUDAGraph()
                    with _no_gc(), torch.cuda.graph(g):
                        one_step()
                    self._dagger_graph = g
                except Exception as e:
                    print(f"[dagger update graph] capture failed, staying eager: {e}")
                    self._dagger_graph = False
                    torch.cuda.synchronize()
            if self._dagger_graph:
                self._dg_acc.zero_()
                if perm is None:
                    perm = torch.randperm(self.num_mini_batches * mb, device=self.device)
                for _ in range(self.num_learning_epochs):
                    for i in range(self.num_mini_batches):
                        self._dg_idx.copy_(perm[i * mb:(i + 1) * mb])
                        self._dagger_graph.replay()
                self._dagger_calls += 1
                self._dagger_warm += 1
                st.clear()
                self.priv_reg_counter += 1
                return float(self._dg_acc) / n
        total = torch.zeros((), device=self.device)
        for sample in st.mini_batch_generator(self.num_mini_batches, self.num_learning_epochs, perm=perm):
            total += self._dagger_step(sample[0])
        self._dagger_calls += 1
        self._dagger_warm += 1
        st.clear()
        self.priv_reg_counter += 1
        return float(total) / n
