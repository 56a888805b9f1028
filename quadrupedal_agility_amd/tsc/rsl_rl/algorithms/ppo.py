"""Hybrid-action PPO of the task-level controller (tsc/rsl_rl/algorithms/ppo.py:8-313).

Same constructor, attributes and call protocol as the reference (`act` / `act_bbc` / `process_env_step` /
`compute_returns` / `update` / `update_dagger` / `update_depth_actor`); the depth-camera distillation heads are optional as in
the reference (`depth_encoder` None = teacher training only).  One policy step draws a gait index from a categorical head and a parameter
vector for every gait from a Gaussian head; the surrogate is the SUM of two clipped PPO terms, one per head, sharing
the advantage (:222-234).  The entropy bonus adds the categorical entropy to the MEAN (not sum) of the Gaussian one.

MI355X: networks run through the fused Linear+ELU backward; the three Adam steps (policy / estimator / history
encoder) are the 3-launch `qa_clip_adam_step`; GAE is `qa_gae`; the adaptive learning rate is decided on the device
(no `.item()` in the minibatch loop, the losses are accumulated on the device and read once per update)."""
import torch
import torch.nn as nn
import torch.optim as optim

from quadrupedal_agility_amd.rsl_rl.algorithms.fused import ClipAdam
from ..storage import RolloutStorage


class PPO:
    def __init__(self, actor_critic, actor_critic_bbc, estimator, estimator_paras, depth_encoder, depth_encoder_paras, depth_actor,
                 num_learning_epochs=1, num_mini_batches=1, clip_param=0.2, gamma=0.998, lam=0.95, value_loss_coef=1.0,
                 entropy_coef=0.0, learning_rate=1e-3, max_grad_norm=1.0, use_clipped_value_loss=True, schedule="fixed",
                 desired_kl=0.01, device="cpu", dagger_update_freq=20, priv_reg_coef_schedual=[0, 0, 0], **kwargs):
        self.device = device
        self.desired_kl, self.schedule, self.learning_rate = desired_kl, schedule, learning_rate
        self.actor_critic = actor_critic.to(device)
        self.actor_critic_bbc = actor_critic_bbc.to(device) if actor_critic_bbc is not None else None
        self.storage = None
        self.optimizer = optim.Adam(self.actor_critic.parameters(), lr=learning_rate)
        self.transition = RolloutStorage.Transition()
        self.clip_param, self.num_learning_epochs, self.num_mini_batches = clip_param, num_learning_epochs, num_mini_batches
        self.value_loss_coef, self.entropy_coef, self.gamma, self.lam = value_loss_coef, entropy_coef, gamma, lam
        self.max_grad_norm, self.use_clipped_value_loss = max_grad_norm, use_clipped_value_loss
        self.hist_encoder_optimizer = optim.Adam(self.actor_critic.actor.history_encoder.parameters(), lr=learning_rate)
        self.priv_reg_coef_schedual = priv_reg_coef_schedual
        self.counter = 0
        self.estimator = estimator
        self.priv_states_dim = estimator_paras["priv_states_dim"]
        self.num_prop = estimator_paras["num_prop"]
        self.num_auxiliary = estimator_paras["num_auxiliary"]
        self.num_scan = estimator_paras["num_scan"]
        self.estimator_optimizer = optim.Adam(self.estimator.parameters(), lr=estimator_paras["learning_rate"])
        self.train_with_estimated_states = estimator_paras["train_with_estimated_states"]
        # depth encoder + student actor (:82-93): the student optimiser steps BOTH nets, BYOL has its own over the shared backbone
        self.grad_sync = None          # data-parallel runs: GradSync of the BBC tree's runner (one flat all-reduce per optimiser step)
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

    def act(self, obs, critic_obs, info=None, hist_encoding=False):
        """:101-125 -- the policy sees ESTIMATED privileged states, the storage keeps the true ones."""
        tr, ac = self.transition, self.actor_critic
        tr.actions = ac.act(self._with_estimated_states(obs, True), hist_encoding).detach()
        tr.values = ac.evaluate(critic_obs).detach()
        tr.actions_log_prob_d = ac.get_actions_log_prob_d(tr.actions[:, 0]).detach()
        tr.actions_log_prob_c = ac.get_actions_log_prob_c(tr.actions[:, 1:]).detach()
        tr.action_mean, tr.action_sigma = ac.action_mean.detach(), ac.action_std.detach()
        tr.observations, tr.critic_observations = obs, critic_obs
        return tr.actions

    def act_bbc(self, obs):
        """:127-137 -- joint targets of the frozen behaviour controller (history branch, mean action)."""
        return self.actor_critic_bbc.act_inference(self._with_estimated_states(obs, False), hist_encoding=True).detach()

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
            if self.grad_sync is not None:      # every rank takes the same LR branch
                kl = self.grad_sync.mean_scalar(kl)
            kl = kl.item()
        if kl > self.desired_kl * 2.0:
            self.learning_rate = max(1e-5, self.learning_rate / 1.5)
        elif 0.0 < kl < self.desired_kl / 2.0:
            self.learning_rate = min(1e-2, self.learning_rate * 1.5)
        for g in self.optimizer.param_groups:
            g["lr"] = self.learning_rate

    def update(self):
        ac = self.actor_critic
        sums = torch.zeros(4, device=self.device)          # value, surrogate, estimator, priv_reg
        coef = self._priv_reg_coef_now()
        priv = self._priv_slice(True)
        for (obs, cobs, actions, target_values, adv, returns, old_logp_d, old_logp_c, old_mu, old_sigma, _h, _m) in \
                self.storage.mini_batch_generator(self.num_mini_batches, self.num_learning_epochs):
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

    def update_dagger(self):
        ac = self.actor_critic
        total = torch.zeros((), device=self.device)
        for batch in self.storage.mini_batch_generator(self.num_mini_batches, self.num_learning_epochs):
            obs = batch[0]
            with torch.no_grad():
                priv_latent = ac.actor.infer_priv_latent(obs)
            loss = (priv_latent - ac.actor.infer_hist_latent(obs)).norm(p=2, dim=1).mean()
            self.hist_encoder_optimizer.zero_grad()
            loss.backward()
            if self.grad_sync is not None:
                self.grad_sync(list(ac.actor.history_encoder.parameters()))
            self._step_hist.step()
            total += loss.detach()
        self.storage.clear()
        self.update_counter()
        return (total / (self.num_learning_epochs * self.num_mini_batches)).item()

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
