"""OnPolicyRunner: rollout (24 env steps) -> GAE -> PPO / discriminator update -> log / checkpoint.

Public surface and model.pt layout of bbc/rsl_rl/runners/on_policy_runner.py (ctor :20-118,
learn :120-236, log :238-304, save/load :306-339).  The rollout loop is free of host syncs: resets
are handled with masks on the device (the reference calls nonzero()/tolist() every step, :183-206)
and finished-episode statistics are gathered with one device->host copy per iteration.  With
torch.distributed initialised (one process per GPU, RCCL) every rank owns num_envs envs and the
gradient buckets / normalisation statistics are all-reduced.
"""
import json
import contextlib
import gc
import os
import statistics
import time
from collections import deque

import numpy as np
import torch
import torch.distributed as dist

from quadrupedal_agility_amd.legged_gym.utils.torch_jit_utils import compute_flat_key_pos
from quadrupedal_agility_amd.rsl_rl.algorithms import SSInfoGAIL  # noqa: F401  (resolved by name from the cfg)
from quadrupedal_agility_amd.rsl_rl.algorithms.discriminator import Discriminator
from quadrupedal_agility_amd.rsl_rl.datasets.motion_loader import MotionLoader
from quadrupedal_agility_amd.rsl_rl.modules import ActorCritic, Estimator  # noqa: F401
from quadrupedal_agility_amd.rsl_rl.utils.utils import Normalizer, TorchNormalizer


class _ScalarLog:
    """SummaryWriter stand-in when tensorboard is not installed: JSON lines with the same tags."""

    def __init__(self, log_dir, flush_secs=10):
        os.makedirs(log_dir, exist_ok=True)
        self._fh = open(os.path.join(log_dir, "scalars.jsonl"), "a")

    def add_scalar(self, tag, value, step):
        self._fh.write(json.dumps({"tag": tag, "value": float(value), "step": int(step)}) + "\n")

    def flush(self):
        self._fh.flush()


def _make_writer(log_dir):
    try:
        from torch.utils.tensorboard import SummaryWriter
        return SummaryWriter(log_dir=log_dir, flush_secs=10)
    except Exception:
        return _ScalarLog(log_dir)


@contextlib.contextmanager
def _no_gc():
    """No cyclic garbage collection while a stream is capturing: a collection that happens to run inside the capture can
    destroy old device objects (graphs, tensors with recorded events), whose destructors call HIP APIs that are illegal
    during capture and abort the process.  (torch.cuda.graph collects once on entry; this keeps it from running again.)"""
    was = gc.isenabled()
    gc.collect()
    gc.disable()
    try:
        yield
    finally:
        if was:
            gc.enable()


class GradSync:
    """One flat all-reduce per optimiser step over RCCL/xGMI (gloo in the CPU tests).  The buckets are small
    (actor-critic 2.9 MB, discriminator 0.7 MB, estimator 64 KB), i.e. latency-bound: a single flat buffer keeps it
    to one collective launch per step."""

    def __init__(self):
        self.world = dist.get_world_size()

    def __call__(self, params, extra=None):
        """Average the gradients of `params` over the ranks in place; `extra` (a list of small tensors, e.g. the
        minibatch KL) rides in the same bucket and is returned averaged.  One cat, one all-reduce, one scale, one
        multi-tensor copy back."""
        grads = [p.grad for p in params if p.grad is not None]
        extra = [] if extra is None else [e.detach().reshape(-1).to(torch.float32) for e in extra]
        if not grads and not extra:
            return []
        flat = torch._utils._flatten_dense_tensors(grads + extra)
        dist.all_reduce(flat)
        flat.div_(self.world)
        parts = torch._utils._unflatten_dense_tensors(flat, grads + extra)
        if grads:
            torch._foreach_copy_(grads, list(parts[:len(grads)]))
        return [x.clone() for x in parts[len(grads):]]

    def all_reduce_(self, flat):
        """sum-all-reduce of a persistent bucket in place (the recorded update scales and unpacks it inside its second graph)"""
        dist.all_reduce(flat)

    def mean_scalar(self, x):
        x = x.clone()
        dist.all_reduce(x)
        return x / self.world

    mean_vector = mean_scalar


_TUNED_GEMMS = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tunableop_gfx950.csv")


def enable_tuned_gemms():
    """Load the hipBLASLt/rocBLAS solution picks for the learner's fp32 GEMM shapes (PyTorch TunableOp results,
    tuned once on MI355X: 113 -> 84 ms of learner time per iteration).  Read-only: no tuning at run time; shapes that
    are not in the file use the library default.  QA_TUNABLEOP=0 disables it."""
    if os.environ.get("QA_TUNABLEOP", "1") == "0" or not os.path.exists(_TUNED_GEMMS):
        return False
    try:
        import torch.cuda.tunable as tunable
        tunable.enable(True)
        tunable.tuning_enable(False)
        tunable.record_untuned_enable(False) if hasattr(tunable, "record_untuned_enable") else None
        tunable.read_file(_TUNED_GEMMS)
        return True
    except Exception as e:          # never fatal: the default GEMM solutions are correct, only slower
        print(f"[tunableop] not enabled: {e}")
        return False


class OnPolicyRunner:
    def __init__(self, env, train_cfg, log_dir=None, device="cpu"):
        self.device, self.env = device, env
        self.tuned_gemms = enable_tuned_gemms() if torch.device(device).type == "cuda" else False
        self.cfg, self.alg_cfg = train_cfg["runner"], dict(train_cfg["algorithm"])
        self.policy_cfg, self.estimator_cfg = train_cfg["policy"], train_cfg["estimator"]
        self.disc_loss_function = self.alg_cfg["disc_loss_function"]
        r = self.cfg
        self.reward_i_coef, self.reward_us_coef = r["reward_i_coef"], r["reward_us_coef"]
        self.reward_ss_coef, self.reward_t_coef = r["reward_ss_coef"], r["reward_t_coef"]
        ecfg = env.cfg.env
        self.disc_history_len, self.disc_obs_len = ecfg.disc_history_len, ecfg.disc_obs_len
        self.obs_disc_weight_step = ecfg.obs_disc_weight_step
        self.amp_enabled = bool(r.get("amp_enabled", True))
        # QA_FORCE_DATA_PARALLEL=1 takes the data-parallel code path with a process group of any size (a 1-rank group on one
        # GPU exercises the collectives and the two-graph update without a second device)
        self.distributed = dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or os.environ.get("QA_FORCE_DATA_PARALLEL") == "1")
        self.rank = dist.get_rank() if self.distributed else 0

        num_prop, num_hist = ecfg.num_prop, ecfg.history_len
        num_actor_obs = env.num_obs
        num_critic_obs = env.num_obs + num_hist * num_prop
        num_disc_obs = env.num_obs_disc
        actor_critic = eval(r["policy_class_name"])(num_actor_obs, num_critic_obs, env.num_actions, num_prop, num_hist,
                                                    ecfg.num_explicit, ecfg.num_latent, ecfg.num_command,
                                                    **self.policy_cfg).to(device)
        estimator = Estimator(input_dim=num_prop, output_dim=ecfg.num_explicit, hidden_dims=self.estimator_cfg["hidden_dims"]).to(device)
        motion_loader = None
        if self.amp_enabled:
            motion_loader = MotionLoader(device, time_between_frames=env.dt, motion_files_lb=r["motion_files_lb"],
                                         motion_files_ulb=r["motion_files_ulb"], mocap_category=env.mocap_category,
                                         num_preload_transitions=r["num_preload_transitions"],
                                         compute_flat_key_pos=compute_flat_key_pos, default_dof_pos=env.default_dof_pos,
                                         obs_scales=env.obs_scales, num_disc_obs=num_disc_obs, disc_obs_len=self.disc_obs_len,
                                         obs_disc_weight_step=self.obs_disc_weight_step,
                                         frame_duration_scale=ecfg.frame_duration_scale)
        disc_normalizer = TorchNormalizer(num_disc_obs * self.disc_obs_len, device)
        reward_i_normalizer = Normalizer(1) if self.disc_loss_function == "WassersteinLoss" else None
        discriminator = Discriminator(env, num_disc_obs * self.disc_obs_len, num_disc_obs, len(env.mocap_category), env.dt,
                                      self.disc_loss_function, reward_i_normalizer, self.reward_i_coef, self.reward_us_coef,
                                      self.reward_ss_coef, self.reward_t_coef, self.disc_history_len, self.disc_obs_len,
                                      self.obs_disc_weight_step, r["disc_hidden_units"], device).to(device)
        min_std = (torch.tensor(r["min_normalized_std"], device=device) *
                   torch.abs(env.dof_pos_limits[:, 1] - env.dof_pos_limits[:, 0]).to(device))
        alg_class = eval(r["algorithm_class_name"])
        self.alg = alg_class(env, actor_critic, discriminator, estimator, self.estimator_cfg, motion_loader, disc_normalizer,
                             self.disc_history_len, self.disc_obs_len, num_disc_obs, self.obs_disc_weight_step,
                             device=device, min_std=min_std, amp_enabled=self.amp_enabled, **self.alg_cfg)
        self.num_steps_per_env, self.save_interval = r["num_steps_per_env"], r["save_interval"]
        self.dagger_update_freq = r["dagger_update_freq"]
        self.alg.init_storage(env.num_envs, self.num_steps_per_env, [num_actor_obs + num_hist * num_prop],
                              [env.num_privileged_obs + num_hist * num_prop], [env.num_actions], gae_fn=self._make_gae())
        if self.distributed:
            self.alg.grad_sync = GradSync()
            for m in (actor_critic, estimator, discriminator):      # every rank starts from rank 0's weights
                for p in m.parameters():
                    dist.broadcast(p.data, src=0)
        self.log_dir, self.writer = log_dir, None
        self.tot_timesteps, self.tot_time, self.current_learning_iteration = 0, 0, 0
        self.last_perf = {}
        env.sync_reset_ids = False
        env.reset()

    # ------------------------------------------------------------------ GAE
    def _make_gae(self):
        sim = getattr(self.env, "sim", None)
        if sim is None or not hasattr(sim, "gae"):
            return None

        def gae(rewards, values, dones, last_values, returns, advantages, gamma, lam):
            T, N = rewards.shape[0], rewards.shape[1]
            v = lambda x: x.view(T, N)
            if not self.distributed:
                sim.gae(v(rewards), v(values), v(dones), last_values, v(returns), v(advantages), gamma, lam, normalize=True)
                return
            # data parallel: the reference normalises over ALL T*N samples, so the moments are all-reduced
            sim.gae(v(rewards), v(values), v(dones), last_values, v(returns), v(advantages), gamma, lam, normalize=False)
            a64 = advantages.to(torch.float64)
            s = torch.stack([a64.sum(), (a64 * a64).sum(), torch.tensor(float(a64.numel()), dtype=torch.float64, device=a64.device)])
            dist.all_reduce(s)
            mean = s[0] / s[2]
            std = torch.sqrt(torch.clamp((s[1] - s[2] * mean * mean) / (s[2] - 1), min=0.0))
            advantages.sub_(mean.float()).div_(std.float() + 1e-8)
        return gae

    # ------------------------------------------------------------------ rollout
    def _alloc_rollout_state(self):
        env, dev, N, T = self.env, self.device, self.env.num_envs, self.num_steps_per_env
        self._obs_cur = env.get_observations().to(dev).clone()
        d = env.get_disc_observations().to(dev)
        self._disc_hist = torch.stack([d] * self.disc_obs_len, dim=1).clone()
        self._cur = torch.zeros(6, N, device=dev)                # running sums: total, i, us, ss, t, length
        self._fin_vals = torch.zeros(T, 6, N, device=dev)
        self._fin_mask = torch.zeros(T, N, dtype=torch.bool, device=dev)
        self._zeros_n = torch.zeros(N, device=dev)
        self._disc_stage = None
        if self.amp_enabled:
            self._disc_stage = (torch.zeros(T, N, env.num_obs_disc * self.disc_obs_len, device=dev), torch.zeros(T, N, 1, device=dev),
                                torch.zeros(T, N, env.dim_c, device=dev))
        self._graph, self._graphs = None, {}          # hist_encoding -> (graph, action delay it was recorded with, ep_infos)
        self._act_buf = None
        self._graph_failed = False

    use_fused_rollout = True     # GPU, discriminator off: per-step bookkeeping as qa_rollout_act / qa_rollout_post
    phase_timing = os.environ.get("QA_PHASE_TIMING", "0") == "1" or bool(os.environ.get("QA_BENCH_TRACE"))
    use_fused_policy = os.environ.get("QA_FUSED_POLICY", "1") != "0"      # ... and the policy networks as one qa_mlp_forward launch per step
    _chain = None
    _dchain = None
    _task_w_dev = None

    def _rollout_steps(self, hist_encoding, logging, recorded):
        """The 24 env steps of one iteration (on_policy_runner.py:155-206).  Reads/writes only persistent tensors, so the
        same code runs eagerly or is recorded once into a hipGraph and replayed."""
        env, alg, T = self.env, self.alg, self.num_steps_per_env
        if self.use_fused_rollout and self._obs_cur.is_cuda and (not self.amp_enabled or self._disc_chain() is not None):
            return self._rollout_steps_fused(hist_encoding, logging, recorded)
        obs, hist, cur = self._obs_cur, self._disc_hist, self._cur
        ep_infos = []
        chain = self._policy_chain(hist_encoding) if obs.is_cuda else None
        if chain is not None:
            chain.pack()
        for i in range(T):
            actions = alg.act(obs, obs, hist_encoding, chain=chain)
            next_obs, _, rewards, dones, infos, _, _ = env.step(actions)
            done_mask = dones > 0
            if self.amp_enabled:
                # the frame pair seen by the discriminator ends with the TERMINAL frame for envs that reset (:168-172)
                hist = torch.cat([hist[:, 1:], env.obs_disc_term_buf.unsqueeze(1)], dim=1)
                rewards, r_i, r_us, r_ss, r_t = alg.disc.predict_disc_reward(rewards.unsqueeze(1), obs, hist, normalizer=alg.disc_normalizer)
            else:
                r_t = rewards
                rewards = self.reward_t_coef * rewards
                r_i = r_us = r_ss = self._zeros_n
            alg.process_env_step(rewards, dones.to(torch.uint8), infos, hist, disc_stage=self._disc_stage if recorded else None)
            obs = next_obs.clone()
            if self.amp_enabled:
                fresh = torch.stack([env.get_disc_observations()] * self.disc_obs_len, dim=1)
                hist = torch.where(done_mask[:, None, None], fresh, hist)
            if logging:
                if "episode" in infos:
                    ep_infos.append(dict(infos["episode"]))
                cur = cur + torch.stack([rewards, r_i, r_us, r_ss, r_t, torch.ones_like(rewards)])
                self._fin_vals[i].copy_(cur)
                self._fin_mask[i].copy_(done_mask)
                cur = cur * (~done_mask)
        self._obs_cur.copy_(obs)
        self._disc_hist.copy_(hist)
        self._cur.copy_(cur)
        return ep_infos

    def _disc_chain(self):
        """qa_mlp_forward description of Discriminator.forward for the rollout's reward (MSELoss mapping, device-resident
        normaliser), built once; None keeps predict_disc_reward()."""
        if not (self.use_fused_policy and self.amp_enabled):
            return None
        if self._dchain is None:
            from quadrupedal_agility_amd.rsl_rl.algorithms.fused import PolicyChain
            a = self.alg
            ok = (a.disc.disc_loss_function == "MSELoss" and a.disc_normalizer is not None and torch.is_tensor(getattr(a.disc_normalizer, "mean", None))
                  and a.disc.disc_obs_len == self.disc_obs_len)
            self._dchain = (PolicyChain.describe_discriminator(a.disc) if ok else None) or False
        return self._dchain or None

    def _rollout_steps_fused(self, hist_encoding, logging, recorded=False):
        """GPU: the same 24 steps with the per-step bookkeeping in a few kernels.  Per step: qa_mlp_forward (estimator,
        encoder, actor, critic) -> qa_rollout_act (sample, log-prob, storage rows) -> observation row copy -> qa_env_step ->
        qa_rollout_post (reward scaling, time-out bootstrap, dones, episode sums).  With the discriminator (config 3) the last
        one is qa_rollout_post_amp, fed by qa_disc_prepare (frame pair: weighting, normalisation) and qa_mlp_forward on the
        discriminator (trunk + three heads): Discriminator.predict_disc_reward + process_env_step in three launches."""
        import ctypes as C
        from quadrupedal_agility_amd import _capi
        env, alg, T, st = self.env, self.alg, self.num_steps_per_env, self.alg.storage
        lib = _capi.load_library()
        P = lambda t: C.c_void_p(t.data_ptr())
        N = env.num_envs
        if self._act_buf is None:
            self._act_buf = torch.zeros(N, env.num_actions, device=self._obs_cur.device)
        obs = self._obs_cur
        std = alg.actor_critic.std
        seed = int(env.sim.cfg.seed)
        ep_infos = []
        chain = self._policy_chain(hist_encoding)
        if chain is not None:
            chain.pack()                # the weights changed in the last update(); one small launch per rollout
        dchain = self._disc_chain() if self.amp_enabled else None
        if dchain is not None:
            from quadrupedal_agility_amd.rsl_rl.algorithms import fused
            dchain.pack()
            disc, hist = alg.disc, self._disc_hist
            task_w = None
            if env.task_obs_weight_decay:
                task_w = getattr(env, "task_obs_weight_dev", None)
                if task_w is None:          # eager rollouts (no recording): a device scalar refreshed per rollout
                    if self._task_w_dev is None:
                        self._task_w_dev = torch.ones((), device=self._obs_cur.device)
                    self._task_w_dev.fill_(float(env.task_obs_weight))
                    task_w = self._task_w_dev
        for i in range(T):
            t = st.step
            if t >= T:
                raise AssertionError("Rollout buffer overflow")
            mean, value = chain.forward(obs) if chain is not None else alg.act_mean_value(obs, obs, hist_encoding)
            stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
            ctr = env._step_ctr
            so = st.observations[t]
            if (obs.dim() == 2 and obs.dtype == torch.float32 and obs.stride(1) == 1 and so.stride(1) == 1 and so.shape == obs.shape
                    and os.environ.get("QA_ACT_STORE", "0") == "1"):
                # r6 (ABI 18), opt-in: the observation rows go to the storage in the sampling launch instead of a copy launch of their own.  Measured: rollout
                # 4.04 -> 3.97 ms at 4096 envs, 2.81 -> 2.85 ms at 512 (the merged launch is longer than the two 5 us ones it replaces): off
                rc = lib.qa_rollout_act_store(P(mean), P(std), P(value), None, seed, P(ctr) if ctr is not None else None, int(env.common_step_counter), N, int(env.qcfg.env_id_offset),
                                              P(self._act_buf), P(st.actions[t]), P(st.mu[t]), P(st.sigma[t]), P(st.actions_log_prob[t]), P(st.values[t]),
                                              P(obs), int(obs.stride(0)), int(obs.shape[1]), P(so), int(so.stride(0)), stream)
                if rc != 0:
                    raise RuntimeError(f"qa_rollout_act_store failed with code {rc}: {lib.qa_last_error().decode()}")
            else:
                rc = lib.qa_rollout_act(P(mean), P(std), P(value), None, seed, P(ctr) if ctr is not None else None, int(env.common_step_counter), N, int(env.qcfg.env_id_offset),
                                        P(self._act_buf), P(st.actions[t]), P(st.mu[t]), P(st.sigma[t]), P(st.actions_log_prob[t]), P(st.values[t]), stream)
                if rc != 0:
                    raise RuntimeError(f"qa_rollout_act failed with code {rc}: {lib.qa_last_error().decode()}")
                so.copy_(obs)
            next_obs, _, _rew, _dones, infos, _, _ = env.step(self._act_buf)
            log_ptrs = (P(self._cur) if logging else None, P(self._fin_vals[i]) if logging else None, P(self._fin_mask[i]) if logging else None)
            if dchain is None:
                rc = lib.qa_rollout_post(P(env.rew_buf), P(env.reset_buf), P(env.time_out_buf), P(st.values[t]), float(self.reward_t_coef), float(alg.gamma), N,
                                         P(st.rewards[t]), P(st.dones[t]), *log_ptrs, stream)
            else:
                # the frame pair seen by the discriminator ends with the TERMINAL frame for envs that reset (:168-172)
                hist = torch.cat([hist[:, 1:], env.obs_disc_term_buf.unsqueeze(1)], dim=1)
                flat = hist.view(N, -1)
                x = fused.disc_prepare([flat], disc._task_mask, disc._frame_mult.view(-1), task_w, alg.disc_normalizer)
                d, eps, logits = dchain.forward(x)
                rc = lib.qa_rollout_post_amp(P(env.rew_buf), P(env.reset_buf), P(env.time_out_buf), P(st.values[t]), P(d), P(eps), P(logits), int(disc.dim_c),
                                             P(st.observations[t]), int(st.observations.stride(1)), int(st.observations.shape[2]),
                                             float(disc.reward_i_coef), float(disc.reward_us_coef), float(disc.reward_ss_coef), float(disc.reward_t_coef),
                                             float(disc.dt), float(alg.gamma), N, P(st.rewards[t]), P(st.dones[t]), *log_ptrs, stream)
                if recorded:        # the ring position is a host variable: stage, insert after the replay
                    self._disc_stage[0][t].copy_(flat); self._disc_stage[1][t].copy_(env.latent_eps); self._disc_stage[2][t].copy_(env.latent_c)
                else:
                    alg.disc_storage.insert(flat, env.latent_eps, env.latent_c)
                fresh = torch.stack([env.get_disc_observations()] * self.disc_obs_len, dim=1)
                hist = torch.where((env.reset_buf > 0)[:, None, None], fresh, hist)
            if rc != 0:
                raise RuntimeError(f"qa_rollout_post failed with code {rc}: {lib.qa_last_error().decode()}")
            st.step += 1
            obs = next_obs          # the arena's observation rows themselves: everything that reads them (GEMMs, the storage copy)
                                    # is enqueued before the next env step overwrites them, so no private copy is needed
            if logging and "episode" in infos:
                ep_infos.append(dict(infos["episode"]))
        self._obs_cur.copy_(obs)
        if dchain is not None:
            self._disc_hist.copy_(hist)
        return ep_infos

    def _policy_chain(self, hist_encoding=False):
        """qa_mlp_forward description of the policy (estimator + latent encoder + actor + critic), one per actor variant
        (privileged encoder / history encoder), built once; None when the modules do not fit the kernel (then the rollout
        keeps the GEMM path)."""
        if not self.use_fused_policy:
            return None
        if self._chain is None:
            self._chain = {}
        key = bool(hist_encoding)
        if key not in self._chain:
            from quadrupedal_agility_amd.rsl_rl.algorithms.fused import PolicyChain
            alg = self.alg
            self._chain[key] = PolicyChain.describe(alg.actor_critic, alg.estimator, alg.train_with_estimated_explicit, hist_encoding=key)
        return self._chain[key]

    def _collect(self, hist_encoding, logging):
        """One rollout.  On the GPU the non-DAgger variant is recorded into a hipGraph the second time it runs and replayed
        from then on: ~2,400 launches per rollout become one graph launch (the rollout is launch-bound: the fused env step
        is 0.1 ms, the ~100 small inference/bookkeeping kernels around it 0.6 ms of host time per step)."""
        env, alg, T = self.env, self.alg, self.num_steps_per_env
        can_graph = (self.use_rollout_graph and not self._graph_failed and self._eager_rollouts >= 1
                     and env.steps_until_delay_change() >= T)
        # one recording per variant: the DAgger rollouts (every dagger_update_freq-th iteration) act through the history
        # encoder instead of the privileged encoder
        key = bool(hist_encoding)
        have = self._graphs.get(key)
        if can_graph and have is not None and have[1] == env.delay:
            have[0].replay()
            env.advance_host_counters(T)
            alg.storage.step = T
            ep_infos = have[2]
        elif can_graph:
            try:
                with torch.inference_mode():
                    alg.act(self._obs_cur, self._obs_cur, hist_encoding)          # touch every GEMM shape once outside the capture
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                alg.storage.step = 0
                with _no_gc(), torch.cuda.graph(g):
                    with torch.inference_mode():
                        ep_infos = self._rollout_steps(hist_encoding, logging, recorded=True)    # host side effects run now, GPU work on replay
                self._graphs[key] = (g, env.delay, ep_infos)
                if not key:
                    self._graph = g
                g.replay()
            except Exception as e:      # never fatal: fall back to eager launches
                print(f"[rollout graph] capture failed, staying eager: {e}")
                self._graph, self._graph_failed = None, True
                self._graphs = {}
                torch.cuda.synchronize()
                return self._collect(hist_encoding, logging)
        else:
            with torch.inference_mode():
                ep_infos = self._rollout_steps(hist_encoding, logging, recorded=False)
            self._eager_rollouts += 1
            return ep_infos
        if self.amp_enabled:            # ring insert of the staged (T*N) discriminator samples, in step order
            st = self._disc_stage
            alg.disc_storage.insert(st[0].flatten(0, 1), st[1].flatten(0, 1), st[2].flatten(0, 1))
        return ep_infos

    # ------------------------------------------------------------------ learn
    def learn(self, num_learning_iterations, init_at_random_ep_len=False):
        env, alg, dev = self.env, self.alg, self.device
        if self.log_dir is not None and self.writer is None and self.rank == 0:
            self.writer = _make_writer(self.log_dir)
        if init_at_random_ep_len:
            env.episode_length_buf = torch.randint_like(env.episode_length_buf, high=int(env.max_episode_length))
        if not hasattr(self, "_obs_cur"):
            self._alloc_rollout_state()
            self._eager_rollouts = 0
            on_gpu = torch.device(dev).type == "cuda"
            self.use_rollout_graph = on_gpu and bool(self.cfg.get("rollout_graph", True)) and os.environ.get("QA_ROLLOUT_GRAPH", "1") != "0"
            if self.use_rollout_graph:
                env.use_device_step_counter()
                if self.amp_enabled:
                    env.task_obs_weight_dev = torch.tensor(float(env.task_obs_weight), device=dev)
        # training never reads the per-step exports of the env (seam 1 / play / logging tensors): the fused step stops writing them WHILE
        # learn() runs (r4: measured traffic 1.37x -> see profiles/env_step_traffic.json); QA_LEAN_EXPORTS=0 keeps the reference's behaviour.
        # r5: the lean kernel keeps a two-slot action ring, so it is only chosen when neither the current action delay nor any
        # delay still in the schedule exceeds one step, and the full exports come back when learn() returns (eval / play / seam-1 readers in
        # the same process must not see frozen tensors).  The rollout recordings bake the kernel variant in: they are dropped if the mask changes.
        lean = self._lean_mask_for(env)
        if lean != getattr(self, "_lean_recorded", lean):
            self._graph, self._graphs = None, {}
        self._lean_recorded = lean
        if hasattr(env, "set_lean_exports") and torch.device(dev).type == "cuda":
            env.set_lean_exports(lean)
        try:
            self._learn_loop(num_learning_iterations)
        finally:
            if lean and hasattr(env, "set_lean_exports"):
                env.set_lean_exports(0)

    def _lean_mask_for(self, env):
        if (torch.device(self.device).type != "cuda" or not hasattr(env, "set_lean_exports") or os.environ.get("QA_LEAN_EXPORTS", "1") == "0"
                or getattr(env, "sync_reset_ids", False)):
            return 0
        delays = [int(getattr(env, "delay", 0))] + [int(d) for d in getattr(env, "_delay_schedule", [])]
        if env.cfg.domain_rand.action_delay and max(delays) > 1:
            return 0
        return 1 if self.amp_enabled else 3

    def _learn_loop(self, num_learning_iterations):
        env, alg, dev = self.env, self.alg, self.device
        alg.actor_critic.train()
        alg.disc.train()
        N, T = env.num_envs, self.num_steps_per_env
        logging = self.log_dir is not None
        if not hasattr(self, "_buffers"):
            self._buffers = {k: deque(maxlen=100) for k in ("rew", "rew_i", "rew_us", "rew_ss", "rew_t", "len")}
        buffers = self._buffers
        mean_hist_latent_loss = 0.0
        tot_iter = self.current_learning_iteration + num_learning_iterations
        for it in range(self.current_learning_iteration, tot_iter):
            self._sync()
            start = time.time()
            hist_encoding = it % self.dagger_update_freq == 0
            ep_infos = self._collect(hist_encoding, logging)
            self._sync()
            stop = time.time()
            collection_time = stop - start
            start = stop
            with torch.inference_mode():
                alg.compute_returns(self._obs_cur)
            losses = alg.update()
            if hist_encoding:
                mean_hist_latent_loss = alg.update_dagger()
            if env.task_obs_weight_decay_steps:
                env.task_obs_weight = max(0, env.task_obs_weight - 1.0 / env.task_obs_weight_decay_steps)
                if getattr(env, "task_obs_weight_dev", None) is not None:
                    env.task_obs_weight_dev.fill_(float(env.task_obs_weight))
            self._sync()
            stop = time.time()
            learn_time = stop - start
            self.last_perf = {"collection_time": collection_time, "learn_time": learn_time,
                              "fps": T * N / (collection_time + learn_time)}
            if logging:
                vals = self._fin_vals.permute(0, 2, 1)[self._fin_mask].cpu().numpy()       # (n_finished, 6) in (step, env) order
                for col, k in enumerate(("rew", "rew_i", "rew_us", "rew_ss", "rew_t", "len")):
                    buffers[k].extend(vals[:, col].tolist())
                if self.rank == 0:
                    self.log(dict(it=it, collection_time=collection_time, learn_time=learn_time, ep_infos=ep_infos,
                                  losses=losses, mean_hist_latent_loss=mean_hist_latent_loss, buffers=buffers))
                if (it + 1) % self.save_interval == 0 and self.rank == 0:
                    self.save(os.path.join(self.log_dir, "model.pt"))
        self.current_learning_iteration += num_learning_iterations
        if self.log_dir is not None and self.rank == 0:
            self.save(os.path.join(self.log_dir, "model.pt"))

    def _sync(self):
        """Phase timing needs the GPU drained at the phase boundaries.  Without a log directory nobody reads those times, so
        the three waits per iteration are skipped (QA_PHASE_TIMING=1 forces them) and the host runs ahead of the GPU: the
        next rollout's launch latency hides behind the update still executing."""
        if torch.device(self.device).type == "cuda" and (self.log_dir is not None or self.phase_timing):
            torch.cuda.synchronize()

    _LOSS_TAGS = ["surrogate_loss", "value_loss", "b_loss", "entropy_batch", "priv_reg_loss", "estimator_loss", "ss_loss",
                  "info_max_loss", "disc_loss", "us_loss", "grad_pen_loss", "disc_logit_loss", "disc_weight_decay"]
    _ACC_TAGS = ["acc_lb", "acc_pi", "acc_exp", "acc_ulb"]

    def log(self, locs, pbar=None):
        it = locs["it"]
        self.tot_timesteps += self.num_steps_per_env * self.env.num_envs
        self.tot_time += locs["collection_time"] + locs["learn_time"]
        w = self.writer
        if locs["ep_infos"]:
            for key in locs["ep_infos"][0]:
                vals = torch.stack([torch.as_tensor(e[key], device=self.device).reshape(()) for e in locs["ep_infos"]])
                w.add_scalar("Episode/" + key, (vals.mean() / self.env.reward_scales[key[4:]]).item(), it)
        for tag, v in zip(self._LOSS_TAGS, locs["losses"][:13]):
            w.add_scalar("Loss/" + tag, v, it)
        w.add_scalar("Loss/hist_latent_loss", locs["mean_hist_latent_loss"], it)
        w.add_scalar("Loss/mean_noise_std", self.alg.actor_critic.std.mean().item(), it)
        for tag, v in zip(self._ACC_TAGS, locs["losses"][13:]):
            w.add_scalar("Acc/" + tag, v, it)
        w.add_scalar("LR/lr_ac", self.alg.lr_ac, it)
        w.add_scalar("LR/lr_disc", self.alg.lr_disc, it)
        w.add_scalar("LR/lr_q", self.alg.lr_q, it)
        fps = int(self.num_steps_per_env * self.env.num_envs / (locs["collection_time"] + locs["learn_time"]))
        w.add_scalar("Perf/total_fps", fps, it)
        w.add_scalar("Perf/collection time", locs["collection_time"], it)
        w.add_scalar("Perf/learning_time", locs["learn_time"], it)
        b = locs["buffers"]
        if len(b["rew"]) > 0:
            for tag, k in (("mean_reward", "rew"), ("mean_reward_i", "rew_i"), ("mean_reward_us", "rew_us"),
                           ("mean_reward_ss", "rew_ss"), ("mean_reward_t", "rew_t"), ("mean_episode_length", "len")):
                w.add_scalar("Train/" + tag, statistics.mean(b[k]), it)
        if hasattr(w, "flush"):
            w.flush()

    # ------------------------------------------------------------------ checkpoints (same keys as the reference)
    def save(self, path, infos=None):
        a = self.alg
        norm = a.disc_normalizer.to_reference() if isinstance(a.disc_normalizer, TorchNormalizer) else a.disc_normalizer
        torch.save({
            "actor_critic": a.actor_critic.state_dict(), "estimator": a.estimator.state_dict(), "disc": a.disc.state_dict(),
            "optim_ac": a.optim_ac.state_dict(), "optim_hist_encoder": a.optim_hist_encoder.state_dict(),
            "optim_estimator": a.optim_estimator.state_dict(), "optim_d": a.optim_d.state_dict(),
            "optim_q_eps": a.optim_q_eps.state_dict(), "optim_q_c": a.optim_q_c.state_dict(),
            "disc_normalizer": norm, "reward_i_normalizer": a.disc.reward_i_normalizer,
            "iter": self.current_learning_iteration, "infos": infos,
        }, path)

    def load(self, path, load_optimizer=True):
        import quadrupedal_agility_amd
        quadrupedal_agility_amd.install_reference_aliases()     # so a pickled rsl_rl.utils.utils.Normalizer resolves
        d = torch.load(path, map_location=self.device, weights_only=False)
        a = self.alg
        a.actor_critic.load_state_dict(d["actor_critic"])
        a.estimator.load_state_dict(d["estimator"])
        a.disc.load_state_dict(d["disc"])
        a.disc_normalizer = TorchNormalizer.from_reference(d["disc_normalizer"], self.device)
        if d["reward_i_normalizer"]:
            a.disc.reward_i_normalizer = d["reward_i_normalizer"]
        if load_optimizer:
            for key in ("optim_ac", "optim_hist_encoder", "optim_estimator", "optim_d", "optim_q_eps", "optim_q_c"):
                try:
                    getattr(a, key).load_state_dict(d[key])
                except Exception as e:      # e.g. a reference checkpoint whose Adam step counters live on the host
                    print(f"[load] optimizer state {key} not restored: {e}")
        # Optimizer.load_state_dict replaces every group's 'lr' by a copy from the checkpoint (a float in a reference
        # checkpoint): re-link the adaptive-KL learning rate tensor that qa_kl_lr_rule writes and ClipAdam / the recorded
        # steps read, and restore the flags the loaded group dict overwrote
        if a._on_gpu:
            g0 = a.optim_ac.param_groups[0]
            a._lr_ac.fill_(float(g0["lr"]))
            for g in a.optim_ac.param_groups:
                g["lr"] = a._lr_ac
            for opt in (a.optim_ac, a.optim_hist_encoder, a.optim_estimator, a.optim_d, a.optim_q_eps, a.optim_q_c):
                if isinstance(opt, torch.optim.Adam):
                    for g in opt.param_groups:
                        g["capturable"], g["fused"] = True, True
        else:
            a.lr_ac = float(a.optim_ac.param_groups[0]["lr"])
        self.current_learning_iteration = d["iter"]
        # recorded launches hold the addresses of the optimizer state they were captured with: start over
        a._ac_graph = a._dagger_graph = a._disc_graph = None
        a._warm_updates = a._dagger_warm = 0
        self._graphs, self._graph = {}, None
        return d["infos"]

    def get_inference_policy(self, device=None):
        """obs (B, 671) -> action means, `ActorCritic.act_inference` (history-encoder variant by default), as the reference's.
        On the GPU the returned callable evaluates the history encoder and the actor in ONE launch (qa_mlp_forward) from a
        snapshot of the weights taken now; `policy.refresh()` re-reads them."""
        ac = self.alg.actor_critic
        ac.eval()
        if device is not None:
            ac.to(device)
        chain = None
        if self.use_fused_policy and next(ac.parameters()).is_cuda:
            from quadrupedal_agility_amd.rsl_rl.algorithms.fused import PolicyChain
            chain = PolicyChain.describe(ac, self.alg.estimator, False, hist_encoding=True, with_critic=False)
        if chain is None:
            return ac.act_inference

        def refresh():
            with torch.inference_mode():
                chain.pack()

        def policy(observations, hist_encoding=True):
            if not hist_encoding or not observations.is_cuda or observations.dim() != 2 or observations.dtype != torch.float32:
                return ac.act_inference(observations, hist_encoding)
            return chain.forward(observations if observations.stride(1) == 1 else observations.contiguous())[0].clone()
        refresh()
        policy.refresh = refresh
        return policy
