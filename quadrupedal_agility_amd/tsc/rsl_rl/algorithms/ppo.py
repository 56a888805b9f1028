"""Hybrid-action PPO of the task-level controller (tsc/rsl_rl/algorithms/ppo.py:8-313).

Same constructor, attributes and call protocol as the reference (`act` / `act_bbc` / `process_env_step` /
`compute_returns` / `update` / `update_dagger` / `update_depth_actor`); the depth-camera distillation heads are optional as in
the reference (`depth_encoder` None = teacher training only).  One policy step draws a gait index from a categorical head and a parameter
vector for every gait from a Gaussian head; the surrogate is the SUM of two clipped PPO terms, one per head, sharing
the advantage (:222-234).  The entropy bonus adds the categorical entropy to the MEAN (not sum) of the Gaussian one.

MI355X: networks run through the fused Linear+ELU backward; the three Adam steps (policy / estimator / history
encoder) are the 3-launch `qa_clip_adam_step`; GAE is `qa_gae`; the adaptive learning rate is decided on the device
(no `.item()` in the minibatch loop, the losses are accumulated on the device and read once per update)."""
import os

import torch
import torch.nn as nn
import torch.optim as optim

from quadrupedal_agility_amd.rsl_rl.algorithms import fused
from quadrupedal_agility_amd.rsl_rl.algorithms.fused import ClipAdam
from ..storage import RolloutStorage


class PPO:
    def __init__(self, actor_critic, actor_critic_bbc, estimator, estimator_paras, depth_encoder, depth_encoder_paras, depth_actor,
                 num_learning_epochs=1, num_mini_batches=1, clip_param=0.2, gamma=0.998, lam=0.95, value_loss_coef=1.0,
                 entropy_coef=0.0, learning_rate=1e-3, max_grad_norm=1.0, use_clipped_value_loss=True, schedule="fixed",
                 desired_kl=0.01, device="cpu", dagger_update_freq=20, priv_reg_coef_schedual=[0, 0, 0], **kwargs):
        self.device = device
        self.desired_kl, self.schedule, self._learning_rate = desired_kl, schedule, learning_rate
        self.actor_critic = actor_critic.to(device)
        self.actor_critic_bbc = actor_critic_bbc.to(device) if actor_critic_bbc is not None else None
        self.storage = None
        # ROCm: `step` counters on the device (capturable) -- what qa_clip_adam_step needs, and what lets a step be recorded
        adam = dict(fused=True, capturable=True) if torch.device(device).type == "cuda" else {}
        self.optimizer = optim.Adam(self.actor_critic.parameters(), lr=learning_rate, **adam)
        self.transition = RolloutStorage.Transition()
        self.clip_param, self.num_learning_epochs, self.num_mini_batches = clip_param, num_learning_epochs, num_mini_batches
        self.value_loss_coef, self.entropy_coef, self.gamma, self.lam = value_loss_coef, entropy_coef, gamma, lam
        self.max_grad_norm, self.use_clipped_value_loss = max_grad_norm, use_clipped_value_loss
        self.hist_encoder_optimizer = optim.Adam(self.actor_critic.actor.history_encoder.parameters(), lr=learning_rate, **adam)
        self.priv_reg_coef_schedual = priv_reg_coef_schedual
        self.counter = 0
        self.estimator = estimator
        self.priv_states_dim = estimator_paras["priv_states_dim"]
        self.num_prop = estimator_paras["num_prop"]
        self.num_auxiliary = estimator_paras["num_auxiliary"]
        self.num_scan = estimator_paras["num_scan"]
        self.estimator_optimizer = optim.Adam(self.estimator.parameters(), lr=estimator_paras["learning_rate"], **adam)
        self.train_with_estimated_states = estimator_paras["train_with_estimated_states"]
        # depth encoder + student actor (:82-93): the student optimiser steps BOTH nets, BYOL has its own over the shared backbone
        self.grad_sync = None          # data-parallel runs: GradSync of the BBC tree's runner (one flat all-reduce per optimiser step)
        # the minibatch step as recorded launches (hipGraph): see _update_recorded
        self.use_update_graph = os.environ.get("QA_TSC_UPDATE_GRAPH", "1") != "0"
        self.use_fused_loss = os.environ.get("QA_TSC_FUSED_LOSS", "1") != "0"
        self._graph, self._warm_updates, self._lr_dev, self._recordable = None, 0, None, None
        self.if_depth = depth_encoder is not None
        if self.if_depth:
            self.depth_encoder, self.depth_encoder_paras, self.depth_actor = depth_encoder, depth_encoder_paras, depth_actor
            self.depth_encoder_optimizer = optim.Adam(self.depth_encoder.parameters(), lr=depth_encoder_paras["learning_rate"])
            self.depth_actor_optimizer = optim.Adam([*self.depth_actor.parameters(), *self.depth_encoder.parameters()], lr=depth_encoder_paras["learning_rate"])
            self.byol_optimizer = optim.Adam(self.depth_encoder.byol_learner.parameters(), lr=depth_encoder_paras["learning_rate_byol"])
        self.CE_loss = nn.CrossEntropyLoss().to(device)
        self.num_actions_d = self.actor_critic.num_actions_d
        self._step_ac = ClipAdam(self.optimizer, max_grad_norm)
        self._step_estimator = ClipAdam(self.estimator_optimizer, max_grad_norm)
        self._step_hist = ClipAdam(self.hist_encoder_optimizer, max_grad_norm)

    # ---- the learning rate lives on the host until the update is recorded, then in a device scalar the recorded LR rule writes and
    # the recorded Adam step reads (no host value decides anything inside a replay)
    @property
    def learning_rate(self):
        return float(self._lr_dev) if self._lr_dev is not None else self._learning_rate

    @learning_rate.setter
    def learning_rate(self, value):
        self._learning_rate = float(value)
        if self._lr_dev is not None:
            self._lr_dev.fill_(float(value))

    def lr_to_host(self):
        """param_groups carry a plain float again (checkpoints: optimizer.state_dict() must not hold a device tensor)"""
        lr = self.learning_rate
        self._learning_rate = lr
        for g in self.optimizer.param_groups:
            g["lr"] = lr

    def lr_to_device(self):
        if self._lr_dev is not None:
            self._lr_dev.fill_(float(self.optimizer.param_groups[0]["lr"]) if not torch.is_tensor(self.optimizer.param_groups[0]["lr"]) else float(self._lr_dev))
            for g in self.optimizer.param_groups:
                g["lr"] = self._lr_dev

    def init_storage(self, num_envs, num_transitions_per_env, actor_obs_shape, critic_obs_shape, action_shape):
        self.storage = RolloutStorage(num_envs, num_transitions_per_env, actor_obs_shape, critic_obs_shape, action_shape, self.device)

    def test_mode(self):
        self.actor_critic.eval()

    def train_mode(self):
        self.actor_critic.train()

    # ------------------------------------------------------------------ rollout side
    def _priv_slice(self, with_auxiliary):
        s = self.num_prop + (self.num_auxiliary if with_auxiliary else 0) + self.num_scan
        return slice(s, s + self.priv_states_dim)

    def _with_estimated_states(self, obs, with_auxiliary):
        if not self.train_with_estimated_states:
            return obs
        est = obs.clone()
        est[:, self._priv_slice(with_auxiliary)] = self.estimator(est[:, :self.num_prop])
        return est

    def act(self, obs, critic_obs, info=None, hist_encoding=False, chain=None, action_history=None, rng=None):
        """:101-125 -- the policy sees ESTIMATED privileged states, the storage keeps the true ones.  `chain` (fused.PolicyChain.describe_task_level,
        privileged-encoder variant): estimator, encoders, trunk, heads and critic of the step as ONE launch; the distributions, the samples and
        the transition record are the same objects either way."""
        tr, ac = self.transition, self.actor_critic
        if chain is not None and not hist_encoding and rng is not None and obs.is_cuda and self.storage.step < self.storage.num_transitions_per_env:
            # networks = one launch (the chain); sampling, both log-probs, the storage rows of the step and the runner's action-history roll = one
            # more (qa_rollout_act_hybrid; ~25 eager launches before).  `rng` = (seed, device step counter, global id of env 0): the draws are
            # Philox-keyed by (seed; env, step) like every other draw of the engine, so recorded rollouts sample afresh on every replay.
            import ctypes as C
            from quadrupedal_agility_amd import _capi
            logits, mean, value = chain.forward(obs)
            st, t, lib = self.storage, self.storage.step, _capi.load_library()
            P = lambda x: C.c_void_p(x.data_ptr())
            n, nd, nc = obs.shape[0], logits.shape[1], mean.shape[1]
            if getattr(self, "_act_buf", None) is None or self._act_buf.shape[0] != n:
                self._act_buf = torch.zeros(n, 1 + nc, device=obs.device)
            seed, step_dev, env0 = rng
            rc = lib.qa_rollout_act_hybrid(P(logits), P(mean), P(ac.std), P(value), int(seed), P(step_dev), 0, n, int(env0), nd, nc, P(self._act_buf), P(st.actions[t]),
                                           P(st.mu[t]), P(st.sigma[t]), P(st.actions_log_prob_d[t]), P(st.actions_log_prob_c[t]), P(st.values[t]),
                                           P(action_history[0]) if action_history is not None else None, P(action_history[1]) if action_history is not None else None,
                                           int(action_history[1].shape[1]) if action_history is not None else 0,
                                           C.c_void_p(torch.cuda.current_stream(obs.device).cuda_stream))
            if rc != 0:
                raise RuntimeError(f"qa_rollout_act_hybrid failed with code {rc}: {lib.qa_last_error().decode()}")
            st.observations[t].copy_(obs)
            if st.privileged_observations is not None:
                st.privileged_observations[t].copy_(critic_obs)
            tr.actions, tr.rows_stored = self._act_buf, True
            tr.observations = tr.critic_observations = None
            return tr.actions
        if chain is not None and not hist_encoding:
            from torch.distributions import Categorical, Normal
            logits, mean, value = chain.forward(obs)
            ac.distribution_d = Categorical(probs=torch.softmax(logits, dim=-1), validate_args=False)
            ac.distribution_c = Normal(mean, mean * 0.0 + ac.std, validate_args=False)
            p = ac.distribution_d.probs                   # ActorCriticTSC.act's sampling (no host read: see there)
            a_d = torch.argmax(p / torch.empty_like(p).exponential_(1.0), dim=-1)
            a_c = mean + ac.distribution_c.stddev * torch.randn_like(mean)
            tr.actions = torch.cat([a_d.unsqueeze(-1), a_c], dim=-1).detach()
            tr.values = value
        else:
            tr.actions = ac.act(self._with_estimated_states(obs, True), hist_encoding).detach()
            tr.values = ac.evaluate(critic_obs).detach()
        tr.actions_log_prob_d = ac.get_actions_log_prob_d(tr.actions[:, 0]).detach()
        tr.actions_log_prob_c = ac.get_actions_log_prob_c(tr.actions[:, 1:]).detach()
        tr.action_mean, tr.action_sigma = ac.action_mean.detach(), ac.action_std.detach()
        # The env's observation rows are persistent buffers its kernels overwrite in place during step() (the reference's env builds a
        # new tensor every step, so holding a reference until process_env_step works there): the rows go into the rollout NOW.
        t = self.storage.step
        self.storage.observations[t].copy_(obs)
        if self.storage.privileged_observations is not None:
            self.storage.privileged_observations[t].copy_(critic_obs)
        tr.observations = tr.critic_observations = None
        return tr.actions

    def act_bbc(self, obs):
        """:127-137 -- joint targets of the frozen behaviour controller (history branch, mean action)."""
        return self.actor_critic_bbc.act_inference(self._with_estimated_states(obs, False), hist_encoding=True).detach()

    def store_transition_rows(self, dones=None):
        """the transition's rows other than reward / done go into the storage now (add_transitions' copies); the caller's kernel writes those
        two (qa_rollout_post_amp) -- returns the storage step it must write to.  `dones`: process_env_step's actor_critic.reset(dones)
        (a no-op for the feed-forward policy; kept so that the two paths cannot diverge if a recurrent policy is plugged in -- ADVICE r4)"""
        tr, st = self.transition, self.storage
        if dones is not None:
            self.actor_critic.reset(dones)
        if st.step >= st.num_transitions_per_env:
            raise AssertionError("Rollout buffer overflow")
        t = st.step
        if not getattr(tr, "rows_stored", False):           # qa_rollout_act_hybrid wrote them at act time
            st.actions[t].copy_(tr.actions); st.values[t].copy_(tr.values)
            st.actions_log_prob_d[t].copy_(tr.actions_log_prob_d.view(-1, 1)); st.actions_log_prob_c[t].copy_(tr.actions_log_prob_c.view(-1, 1))
            st.mu[t].copy_(tr.action_mean); st.sigma[t].copy_(tr.action_sigma)
        tr.rows_stored = False
        st.step += 1
        self.transition.clear()
        return t

    def process_env_step(self, rewards, dones, infos):
        total = rewards.clone()
        self.transition.rewards = total.clone()
        self.transition.dones = dones
        if "time_outs" in infos:        # bootstrap on time-outs
            self.transition.rewards += self.gamma * torch.squeeze(self.transition.values * infos["time_outs"].unsqueeze(1).to(self.device), 1)
        self.storage.add_transitions(self.transition)
        self.transition.clear()
        self.actor_critic.reset(dones)
        return total

    def compute_returns(self, last_critic_obs):
        self.storage.compute_returns(self.actor_critic.evaluate(last_critic_obs).detach(), self.gamma, self.lam)

    # ------------------------------------------------------------------ learner side
    def _priv_reg_coef_now(self):
        s = self.priv_reg_coef_schedual
        stage = min(max(self.counter - s[2], 0) / s[3], 1)
        return stage * (s[1] - s[0]) + s[0]

    def _clipped_surrogate(self, logp, old_logp, adv):
        ratio = torch.exp(logp - old_logp.squeeze(-1))
        return torch.max(-adv * ratio, -adv * torch.clamp(ratio, 1.0 - self.clip_param, 1.0 + self.clip_param)).mean()

    def _adapt_learning_rate(self, mu, sigma, old_mu, old_sigma):
        """KL(old || new) of the Gaussian head only, as the reference (:205-219); the decision needs one scalar."""
        with torch.no_grad():
            kl = torch.sum(torch.log(sigma / old_sigma + 1.e-5) + (old_sigma.square() + (old_mu - mu).square()) / (2.0 * sigma.square()) - 0.5,
                           dim=-1).mean()
        self._adapt_from_kl(kl)

    def _adapt_from_kl(self, kl):
        with torch.no_grad():
            if self.grad_sync is not None:      # every rank takes the same LR branch
                kl = self.grad_sync.mean_scalar(kl)
            kl = kl.item()
        lr = self.learning_rate
        if kl > self.desired_kl * 2.0:
            lr = max(1e-5, lr / 1.5)
        elif 0.0 < kl < self.desired_kl / 2.0:
            lr = min(1e-2, lr * 1.5)
        self.learning_rate = lr
        if self._lr_dev is None:
            for g in self.optimizer.param_groups:
                g["lr"] = lr

    def update(self):
        ac = self.actor_critic
        coef = self._priv_reg_coef_now()
        if (self.use_update_graph and self._graph is not False and self._warm_updates >= 1 and fused.ENABLED and torch.device(self.device).type == "cuda"
                and self.desired_kl is not None and self.schedule == "adaptive" and self.use_fused_loss and self._recordable_networks()):
            # (use_fused_loss: the eager objective reduces through torch's two-stage sum / mean, which go stale under hipGraph replay
            # at 24576+ rows -- tools/graph_reduction_audit.py; the recorded step only ever contains our own fixed-order reductions)
            sums = self._update_recorded(coef)
            if sums is not None:
                v, s, e, p = (sums / (self.num_learning_epochs * self.num_mini_batches)).tolist()
                self.storage.clear()
                self.update_counter()
                return v, s, e, 0.0, 0.0, p, coef
        self._warm_updates += 1
        sums = torch.zeros(4, device=self.device)          # value, surrogate, estimator, priv_reg
        priv = self._priv_slice(True)
        for (obs, cobs, actions, target_values, adv, returns, old_logp_d, old_logp_c, old_mu, old_sigma, _h, _m) in \
                self.storage.mini_batch_generator(self.num_mini_batches, self.num_learning_epochs):
            if self._train_chain(obs, cobs) is not None:
                # r6: steps of this size run as chain launches (train_chain.TscTrainChain) in the recorded update; the eager update takes the same
                # step -- same kernels, same order -- so that "recorded = eager" keeps meaning what it says (tests/test_tsc_learner.py)
                with torch.no_grad():
                    hist_latent = ac.actor.infer_hist_latent(obs)
                self.estimator_optimizer.zero_grad(); self.optimizer.zero_grad()
                kl, stats = self._minibatch_forward_backward((obs, cobs, actions, target_values, adv, returns, old_logp_d, old_logp_c, old_mu, old_sigma), hist_latent, coef)
                if self.grad_sync is not None:
                    self.grad_sync(list(self.estimator.parameters())); self.grad_sync(list(ac.parameters()))
                self._step_estimator.step()
                if self.desired_kl is not None and self.schedule == "adaptive":
                    self._adapt_from_kl(kl)
                self._step_ac.step()
                self._add_stats(sums, stats)
                continue
            ac.act(obs, hist_encoding=False)
            logp_d = ac.get_actions_log_prob_d(actions[:, 0])
            logp_c = ac.get_actions_log_prob_c(actions[:, 1:])
            value = ac.evaluate(cobs)
            mu, sigma = ac.action_mean, ac.action_std
            entropy = ac.entropy_c + ac.entropy_d

            priv_latent = ac.actor.infer_priv_latent(obs)
            with torch.no_grad():
                hist_latent = ac.actor.infer_hist_latent(obs)
            priv_reg_loss = (priv_latent - hist_latent).norm(p=2, dim=1).mean()

            est_loss = (self.estimator(obs[:, :self.num_prop]) - obs[:, priv]).pow(2).mean()
            self.estimator_optimizer.zero_grad()
            est_loss.backward()
            if self.grad_sync is not None:
                self.grad_sync(list(self.estimator.parameters()))
            self._step_estimator.step()

            if self.desired_kl is not None and self.schedule == "adaptive":
                self._adapt_learning_rate(mu, sigma, old_mu, old_sigma)

            a = adv.squeeze(-1)
            surrogate = self._clipped_surrogate(logp_d, old_logp_d, a) + self._clipped_surrogate(logp_c, old_logp_c, a)
            if self.use_clipped_value_loss:
                clipped = target_values + (value - target_values).clamp(-self.clip_param, self.clip_param)
                value_loss = torch.max((value - returns).pow(2), (clipped - returns).pow(2)).mean()
            else:
                value_loss = (returns - value).pow(2).mean()
            loss = surrogate + self.value_loss_coef * value_loss - self.entropy_coef * entropy.mean() + coef * priv_reg_loss

            self.optimizer.zero_grad()
            loss.backward()
            if self.grad_sync is not None:
                self.grad_sync(list(ac.parameters()))
            self._step_ac.step()
            sums += torch.stack([value_loss.detach(), surrogate.detach(), est_loss.detach(), priv_reg_loss.detach()])

        n = self.num_learning_epochs * self.num_mini_batches
        v, s, e, p = (sums / n).tolist()
        self.storage.clear()
        self.update_counter()
        return v, s, e, 0.0, 0.0, p, coef

    def _recordable_networks(self):
        """the structure gate of the recorded update (the behaviour-level learner has the same one, SSInfoGAIL._recordable_networks): every
        trainable parameter's gradient must come out of our kernels -- torch's batch reductions go stale under hipGraph replay here -- else
        the update stays eager"""
        if self._recordable is None:
            self._recordable = bool(fused.recordable(self.actor_critic, self.estimator))
            if not self._recordable:
                print("[tsc ppo] networks outside the recorded step's whitelist (fused.recordable): the update stays eager")
                self._graph = False
        return self._recordable

    def _minibatch_losses(self, batch, hist_latent, coef):
        """forward of one minibatch (the body of update(), :222-262) -> (est_loss, loss, kl, [value, surrogate, priv_reg])"""
        ac = self.actor_critic
        obs, cobs, actions, target_values, adv, returns, old_logp_d, old_logp_c, old_mu, old_sigma = batch
        ac._distributions(obs, False)          # the reference calls act() here and throws the sample away (:224); no draw is recorded
        logp_d = ac.get_actions_log_prob_d(actions[:, 0])
        logp_c = ac.get_actions_log_prob_c(actions[:, 1:])
        value = ac.evaluate(cobs)
        mu, sigma = ac.action_mean, ac.action_std
        entropy = ac.entropy_c + ac.entropy_d
        priv_reg_loss = (ac.actor.infer_priv_latent(obs) - hist_latent).norm(p=2, dim=1).mean()
        est_loss = (self.estimator(obs[:, :self.num_prop]) - obs[:, self._priv_slice(True)]).pow(2).mean()
        with torch.no_grad():
            kl = torch.sum(torch.log(sigma / old_sigma + 1.e-5) + (old_sigma.square() + (old_mu - mu).square()) / (2.0 * sigma.square()) - 0.5, dim=-1).mean()
        a = adv.squeeze(-1)
        surrogate = self._clipped_surrogate(logp_d, old_logp_d, a) + self._clipped_surrogate(logp_c, old_logp_c, a)
        if self.use_clipped_value_loss:
            clipped = target_values + (value - target_values).clamp(-self.clip_param, self.clip_param)
            value_loss = torch.max((value - returns).pow(2), (clipped - returns).pow(2)).mean()
        else:
            value_loss = (returns - value).pow(2).mean()
        loss = surrogate + self.value_loss_coef * value_loss - self.entropy_coef * entropy.mean() + coef * priv_reg_loss
        return est_loss, loss, kl, [value_loss.detach(), surrogate.detach(), est_loss.detach(), priv_reg_loss.detach()]

    def _minibatch_forward_backward(self, batch, hist_latent, coef):
        """forward AND both backward passes of one minibatch -> (kl, [value, surrogate, estimator, priv_reg]).  On ROCm tensors with
        the built head widths the objective and its gradient are ONE kernel (qa_hybrid_ppo_loss; the two small regressions are
        qa_pair_loss) whose gradients go straight into autograd.backward() of the network outputs -- in eager PyTorch the hybrid
        objective is ~100 elementwise launches forward and ~150 backward; otherwise the eager expression and loss.backward()."""
        ac = self.actor_critic
        obs, cobs, actions, target_values, adv, returns, old_logp_d, old_logp_c, old_mu, old_sigma = batch
        chain = self._train_chain(obs, cobs)
        if chain is not None:
            # r6: the 17 layers of the step as two chain launches + 16 weight-gradient products in <= 4 (train_chain.TscTrainChain)
            chain.pack()
            est, logits, mean, value, priv_latent = chain.forward(obs)
            res = fused.hybrid_ppo_loss_raw(logits, mean, ac.std, value, actions, old_logp_d, old_logp_c, old_mu, old_sigma, adv, returns, target_values,
                                            clip=self.clip_param, c_value=self.value_loss_coef, c_entropy=self.entropy_coef,
                                            clipped_value=self.use_clipped_value_loss)
            out, dlogits, dmean, dstd, dvalue = res
            if torch.is_tensor(coef) and coef.dtype == torch.float32 and coef.device == obs.device and os.environ.get("QA_PAIR_LOSSES", "1") != "0":
                # r6 (ABI 18): both losses, their finishes and the coefficient's multiply in ONE launch (were five); same arithmetic
                (priv_reg_loss, g_priv), (est_loss, g_est) = fused.pair_losses_raw(
                    [(priv_latent, hist_latent, fused.PAIR_ROW_L2, coef), (est, obs[:, self._priv_slice(True)], fused.PAIR_MSE, None)])
            else:
                priv_reg_loss, g_priv = fused.pair_loss_raw(priv_latent, hist_latent, fused.PAIR_ROW_L2)
                est_loss, g_est = fused.pair_loss_raw(est, obs[:, self._priv_slice(True)], fused.PAIR_MSE)
                g_priv = g_priv * coef
            chain.backward(g_est, dlogits, dmean, dvalue, g_priv)
            if self.grad_sync is not None:
                fused.flush_pending_grads()
            ac.std.grad = dstd.view_as(ac.std)
            return out[4], (out[2], out[1], est_loss, priv_reg_loss)       # (a tuple: `_add_stats` puts them on the accumulator in one launch)
        if self.use_fused_loss and obs.is_cuda and fused.ENABLED and isinstance(ac.std, nn.Parameter):
            from quadrupedal_agility_amd.rsl_rl.modules.actor_critic import _head
            emb = ac.actor(obs, False)
            logits, mean = _head(ac.actor.actor_d, emb), _head(ac.actor.actor_c, emb)
            value = ac.evaluate(cobs)
            res = fused.hybrid_ppo_loss_raw(logits, mean, ac.std, value, actions, old_logp_d, old_logp_c, old_mu, old_sigma, adv, returns, target_values,
                                            clip=self.clip_param, c_value=self.value_loss_coef, c_entropy=self.entropy_coef,
                                            clipped_value=self.use_clipped_value_loss)
            if res is not None:
                out, dlogits, dmean, dstd, dvalue = res
                priv_latent = ac.actor.infer_priv_latent(obs)
                priv_reg_loss, g_priv = fused.pair_loss_raw(priv_latent, hist_latent, fused.PAIR_ROW_L2)
                est = self.estimator(obs[:, :self.num_prop])
                est_loss, g_est = fused.pair_loss_raw(est, obs[:, self._priv_slice(True)], fused.PAIR_MSE)
                torch.autograd.backward([est, logits, mean, value, priv_latent], [g_est, dlogits, dmean, dvalue.view_as(value), g_priv * coef])
                ac.std.grad = dstd.view_as(ac.std)          # std enters the objective through the kernel only
                return out[4], torch.stack([out[2], out[1], est_loss, priv_reg_loss])
        est_loss, loss, kl, stats = self._minibatch_losses(batch, hist_latent, coef)
        est_loss.backward()
        loss.backward()
        return kl, torch.stack(stats)

    @staticmethod
    def _add_stats(acc, stats):
        """the step's four logged values onto the accumulator: a stacked tensor (autograd path), or four device scalars (chain path: one launch)"""
        if torch.is_tensor(stats):
            acc.add_(stats)
        else:
            fused.accumulate_scalars(acc, list(stats))

    def _train_chain(self, obs, cobs):
        """train_chain.TscTrainChain for this minibatch size, or None (many rows, the critic on another row than the actor, head widths the fused
        objective was not built for, networks that are not the reference's Linear / ELU / tanh stacks)"""
        from quadrupedal_agility_amd.rsl_rl.algorithms import train_chain
        ac = self.actor_critic
        if not (obs.is_cuda and self.use_fused_loss and fused.ENABLED and train_chain.ENABLED and isinstance(ac.std, nn.Parameter)):
            return None
        rows = obs.shape[0]
        if rows > train_chain.MAX_ROWS or cobs.data_ptr() != obs.data_ptr() or obs.stride(1) != 1 or (ac.actor.actor_d.out_features, ac.actor.actor_c.out_features) != (3, 18):
            return None
        cache = self.__dict__.setdefault("_train_chains", {})
        if rows not in cache:
            cache[rows] = train_chain.TscTrainChain.describe(ac, self.estimator, rows, self.num_prop) or False
        return cache[rows] or None

    def _update_recorded(self, coef):
        """The 20 minibatch steps of update() as replays of recorded launches (hipGraph).  One step -- nine indexed reads of the
        rollout, estimator / actor / critic forward, the two clipped surrogates, both backward passes, clipping, the KL-adaptive
        learning rate (on the device) and the two Adam steps -- is ~400 launches of mostly small kernels: at the 1024 envs per GPU of
        BASELINE's 8-GPU teacher job the host cannot issue them as fast as the GPU retires them.  Per step the host copies the next
        index slice and replays.  The privileged-latent regulariser's target (history encoder, not trained by these steps) is evaluated
        once per update for the whole rollout.  Data-parallel runs record the step as TWO graphs around the gradient collective:
        [gather .. both backward passes, gradients + KL packed into a persistent bucket] -> all-reduce(bucket) -> [unpack, LR rule,
        both Adam steps].  Returns the summed losses, or None when capture is not possible (the caller stays eager)."""
        from quadrupedal_agility_amd.rsl_rl.runners.on_policy_runner import _no_gc
        dev, st, ac = self.device, self.storage, self.actor_critic
        batch = st.num_envs * st.num_transitions_per_env
        mb = batch // self.num_mini_batches
        flat = [x.flatten(0, 1) for x in (st.observations, st.actions, st.values, st.advantages, st.returns, st.actions_log_prob_d,
                                          st.actions_log_prob_c, st.mu, st.sigma)]
        if self._graph is None:
            try:
                self._mb_idx = torch.zeros(mb, dtype=torch.int64, device=dev)
                self._acc = torch.zeros(4, device=dev)
                self._coef_dev = torch.zeros((), device=dev)
                self._hist_latent_all = torch.zeros(batch, ac.actor.infer_hist_latent(flat[0][:2]).shape[1], device=dev)
                self._lr_dev = torch.full((), float(self._learning_rate), dtype=torch.float32, device=dev)
                for g in self.optimizer.param_groups:
                    g["lr"] = self._lr_dev
                sync = self.grad_sync
                params = list(self.estimator.parameters()) + list(ac.parameters())

                cflat = st.privileged_observations.flatten(0, 1) if st.privileged_observations is not None else None

                def front():
                    srcs = flat + [self._hist_latent_all] + ([cflat] if cflat is not None else [])
                    got = fused.gather_rows(self._mb_idx, srcs)          # the indexed reads of a minibatch in ONE launch
                    rows, hl = list(got[:len(flat)]), got[len(flat)]
                    rows.insert(1, got[-1] if cflat is not None else rows[0])
                    return self._minibatch_forward_backward(rows, hl, self._coef_dev)

                # r6: on one GPU, with steps that run as chain launches (<= 8192 rows, ~0.65 ms each), the rollout is gathered into permuted order ONCE per
                # update (the reference draws one permutation for all epochs, :122-170) and minibatch i is rows [i mb, (i + 1) mb) of that copy; the slots of an
                # epoch are ONE recording -- 5 replays per update instead of 20 x (index copy, replay, gather).  The behaviour-level learner's scheme since r3.
                nmb = self.num_mini_batches
                self._epoch_graph = False
                if sync is None and cflat is None and os.environ.get("QA_STEP_UNROLL", "1") != "0":
                    srcs_all = flat + [self._hist_latent_all]
                    bufs = fused.gather_rows(torch.arange(nmb * mb, device=dev), srcs_all)          # (also fills the buffers for the warm-up pass below)
                    if self._train_chain(bufs[0][:mb], bufs[0][:mb]) is not None:
                        self._perm_bufs, self._gather_srcs, self._epoch_graph = bufs, srcs_all, True

                def front_slot(i):
                    rows = [b[i * mb:(i + 1) * mb] for b in self._perm_bufs[:len(flat)]]
                    hl = self._perm_bufs[len(flat)][i * mb:(i + 1) * mb]
                    rows.insert(1, rows[0])
                    return self._minibatch_forward_backward(rows, hl, self._coef_dev)

                # (an attribute: its merged device tables are read by every REPLAY -- as a local of this capture block they were freed when it
                # returned, and the replays faulted)
                pair = self._step_pair = fused.ClipAdamPair(self._step_estimator, self._step_ac) if os.environ.get("QA_ADAM_PAIR", "1") != "0" else None

                def apply(kl):
                    # r6 (ABI 18): the two clipped Adam steps and the KL rule between them as three launches (same kernels, same sums: bit-identical)
                    if pair is not None and kl.dtype == torch.float32 and pair.step(kl, self.desired_kl):
                        return
                    if pair is not None and os.environ.get("QA_DEBUG_GRAPH"):
                        print("[tsc ppo] the paired optimiser launch was not taken:", pair.why_not)
                    self._step_estimator.step()
                    fused.kl_lr_rule(kl, self.desired_kl, self._lr_dev)
                    self._step_ac.step()

                # autograd keeps one AccumulateGrad node per parameter, tied to the stream it was first used on and alive as long as
                # any graph references it: drop what the eager update left (the modules' distribution objects hold a graph), then run
                # one forward + backward on a side stream so that the nodes the capture meets were not made on the default stream
                ac.distribution_d = ac.distribution_c = None
                import gc
                gc.collect()
                torch.cuda.synchronize()
                side = torch.cuda.Stream()
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    self.optimizer.zero_grad(set_to_none=True); self.estimator_optimizer.zero_grad(set_to_none=True)
                    front_slot(0) if self._epoch_graph else front()
                torch.cuda.current_stream().wait_stream(side)
                torch.cuda.synchronize()
                if pair is not None:
                    pair.warm()                 # the merged optimiser tables, built OUTSIDE the capture (the gradients of the pass above exist)
                ac.distribution_d = ac.distribution_c = None
                gc.collect()
                self.optimizer.zero_grad(set_to_none=True); self.estimator_optimizer.zero_grad(set_to_none=True)
                if self._epoch_graph:
                    g = torch.cuda.CUDAGraph()
                    with _no_gc(), torch.cuda.graph(g):
                        for i in range(nmb):
                            kl, stats = front_slot(i)
                            apply(kl.reshape(()))
                            self._add_stats(self._acc, stats)
                            self.optimizer.zero_grad(set_to_none=True); self.estimator_optimizer.zero_grad(set_to_none=True)
                    self._graph = (g, None)
                elif sync is None:
                    g = torch.cuda.CUDAGraph()
                    with _no_gc(), torch.cuda.graph(g):
                        kl, stats = front()
                        apply(kl.reshape(()))
                        self._add_stats(self._acc, stats)
                    self._graph = (g, None)
                else:
                    ga, gb = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
                    pool = torch.cuda.graph_pool_handle()
                    with _no_gc(), torch.cuda.graph(ga, pool=pool):
                        kl, st_ = front()
                        self._stats_tmp = st_ if torch.is_tensor(st_) else torch.stack(list(st_))
                        grads = [p.grad for p in params if p.grad is not None]
                        packed = grads + [kl.detach().reshape(1)]
                        self._bucket = torch._utils._flatten_dense_tensors(packed)          # lives in the graphs' pool
                    with _no_gc(), torch.cuda.graph(gb, pool=pool):
                        self._bucket.div_(sync.world)
                        parts = torch._utils._unflatten_dense_tensors(self._bucket, packed)
                        torch._foreach_copy_(grads, list(parts[:len(grads)]))
                        apply(parts[-1].reshape(()))
                        self._acc.add_(self._stats_tmp)
                    self._graph = (ga, gb)
                # the capture ran no kernels: parameters, Adam moments and the LR are untouched; the replays below do this update
            except Exception as e:      # never fatal: the eager loop is the same arithmetic
                print(f"[tsc ppo update graph] capture failed, staying eager: {e}")
                if os.environ.get("QA_DEBUG_GRAPH"):
                    import traceback
                    traceback.print_exc()
                self._graph = False
                torch.cuda.synchronize()
                if self._lr_dev is not None:
                    lr = float(self._learning_rate)
                    self._lr_dev = None
                    for g in self.optimizer.param_groups:
                        g["lr"] = lr
                return None
        self._coef_dev.fill_(float(coef))
        self._acc.zero_()
        with torch.no_grad():
            self._hist_latent_all.copy_(ac.actor.infer_hist_latent(flat[0]))
        ga, gb = self._graph
        perm = torch.randperm(self.num_mini_batches * mb, device=dev)          # one permutation for all epochs (:122-170)
        if getattr(self, "_epoch_graph", False):
            fused.gather_rows(perm, self._gather_srcs, dsts=self._perm_bufs)
            for _ in range(self.num_learning_epochs):
                ga.replay()
            return self._acc.clone()
        for _ in range(self.num_learning_epochs):
            for i in range(self.num_mini_batches):
                self._mb_idx.copy_(perm[i * mb:(i + 1) * mb])
                ga.replay()
                if gb is not None:
                    self.grad_sync.all_reduce_(self._bucket)
                    gb.replay()
        return self._acc.clone()

    def _dagger_step(self, obs):
        """one minibatch of the history-encoder regression (:264-283): || priv_encoder(latent) - history_encoder(history) ||_2 row mean, one Adam step"""
        ac = self.actor_critic
        with torch.no_grad():
            priv_latent = ac.actor.infer_priv_latent(obs)
        hist = ac.actor.infer_hist_latent(obs)
        if obs.is_cuda and fused.ENABLED and self.use_fused_loss:      # value + gradient in one pass, fixed-order row sum (no torch reduction in a recorded step)
            loss = fused.pair_loss(hist, priv_latent, fused.PAIR_ROW_L2)
        else:
            loss = (priv_latent - hist).norm(p=2, dim=1).mean()
        self.hist_encoder_optimizer.zero_grad()
        if obs.is_cuda and torch.cuda.is_current_stream_capturing():
            fused.assert_recordable_graph([hist], "DAgger step")
        loss.backward()
        if self.grad_sync is not None:
            self.grad_sync(list(ac.actor.history_encoder.parameters()))
        self._step_hist.step()
        return loss.detach()

    def update_dagger(self):
        """History-encoder regression onto the privileged latent every dagger_update_freq iterations (:264-283).  r6: on one GPU the 20 steps are
        replays of ONE recorded step (the same scheme as the behaviour-level learner's, rsl_rl/algorithms/gail.py `update_dagger`): eager they
        are ~1.6 ms each of launch-bound small kernels, and the iteration that carries them took 57 ms against 24 (1024 envs)."""
        ac, st = self.actor_critic, self.storage
        n = self.num_learning_epochs * self.num_mini_batches
        mb = (st.num_envs * st.num_transitions_per_env) // self.num_mini_batches
        obs_all = st.observations.flatten(0, 1)
        perm = torch.randperm(self.num_mini_batches * mb, requires_grad=False, device=self.device)      # one permutation for all epochs (:122-170)
        graph = self.__dict__.setdefault("_dagger_graph", None)
        warm = self.__dict__.setdefault("_dagger_warm", 0)
        if (self.use_update_graph and graph is not False and warm >= 1 and fused.ENABLED and self.use_fused_loss and self.grad_sync is None
                and torch.device(self.device).type == "cuda"):
            if graph is None:
                try:
                    from quadrupedal_agility_amd.rsl_rl.runners.on_policy_runner import _no_gc
                    self._dg_idx = torch.zeros(mb, dtype=torch.int64, device=self.device)
                    self._dg_acc = torch.zeros((), device=self.device)
                    self._dg_rows = (obs_all.data_ptr(), mb)
                    torch.cuda.synchronize()
                    self.hist_encoder_optimizer.zero_grad(set_to_none=True)
                    g = torch.cuda.CUDAGraph()
                    with _no_gc(), torch.cuda.graph(g):
                        self._dg_acc.add_(self._dagger_step(obs_all[self._dg_idx]))
                    graph = self._dagger_graph = g
                except Exception as e:      # never fatal: the eager loop is the same arithmetic
                    print(f"[tsc dagger update graph] capture failed, staying eager: {e}")
                    graph = self._dagger_graph = False
                    torch.cuda.synchronize()
            if graph and self._dg_rows == (obs_all.data_ptr(), mb):
                self._dg_acc.zero_()
                for _ in range(self.num_learning_epochs):
                    for i in range(self.num_mini_batches):
                        self._dg_idx.copy_(perm[i * mb:(i + 1) * mb])
                        graph.replay()
                self._dagger_warm += 1
                st.clear()
                self.update_counter()
                return float(self._dg_acc) / n
        total = torch.zeros((), device=self.device)
        for _ in range(self.num_learning_epochs):
            for i in range(self.num_mini_batches):
                total += self._dagger_step(obs_all[perm[i * mb:(i + 1) * mb]])
        self._dagger_warm = warm + 1
        st.clear()
        self.update_counter()
        return (total / n).item()

    def update_depth_actor(self, actions_student_batch, actions_teacher_batch, yaw_student_batch, yaw_teacher_batch,
                           obst_type_buffer_student, obst_type_buffer_teacher, depth_batch):
        """DAgger step of the vision student (:327-358): cross-entropy on the gait head + L2 on the parameter head against the teacher's
        action, L2 on the predicted goal headings (weights 2, 0.5), cross-entropy on the obstacle class; ONE Adam step over student
        actor + depth encoder (only the actor's gradients are norm-clipped, as in the reference); then 6 BYOL minibatches over the
        rollout's depth images with the EMA target update after each.  The four losses are read from the device once."""
        if not self.if_depth:
            return None
        nd = self.num_actions_d
        d_loss = self.CE_loss(actions_student_batch[:, :nd], actions_teacher_batch[:, 0].detach().to(torch.int64))
        c_loss = (actions_teacher_batch[:, 1:].detach() - actions_student_batch[:, nd:]).norm(p=2, dim=1).mean()
        depth_actor_loss = d_loss + c_loss
        scale = torch.tensor([2.0, 0.5], device=yaw_teacher_batch.device)
        yaw_loss = ((yaw_teacher_batch.detach() - yaw_student_batch) * scale).norm(p=2, dim=1).mean()
        obst_type_loss = self.CE_loss(obst_type_buffer_student, torch.argmax(obst_type_buffer_teacher, dim=-1))
        loss = depth_actor_loss + yaw_loss + obst_type_loss
        self.depth_actor_optimizer.zero_grad()
        loss.backward()
        if self.grad_sync is not None:
            self.grad_sync([*self.depth_actor.parameters(), *self.depth_encoder.parameters()])
        nn.utils.clip_grad_norm_(self.depth_actor.parameters(), self.max_grad_norm)
        self.depth_actor_optimizer.step()

        num_samples = depth_batch.size(0)
        batch_size = num_samples // 6
        depth_batch = depth_batch[torch.randperm(num_samples, device=depth_batch.device)]
        byol_sum = torch.zeros((), device=depth_batch.device)
        for i in range(0, num_samples, batch_size):
            byol_loss = self.depth_encoder.byol_learner(depth_batch[i:i + batch_size])
            self.byol_optimizer.zero_grad()
            byol_loss.backward()
            if self.grad_sync is not None:
                self.grad_sync(list(self.depth_encoder.byol_learner.parameters()))
            self.byol_optimizer.step()
            byol_sum += byol_loss.detach()
            self.depth_encoder.byol_learner.update_moving_average()
        out = torch.stack([depth_actor_loss.detach(), yaw_loss.detach(), obst_type_loss.detach(), byol_sum / (num_samples // batch_size)])
        return tuple(out.tolist())

    def update_counter(self):
        self.counter += 1
