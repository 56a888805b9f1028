"""`OnPolicyRunner` of the task-level tree: the two-level rollout `learn_RL` (tsc/rsl_rl/runners/on_policy_runner.py:19-276) and
the checkpoint layout (:443-520).  Per env step: task policy (hybrid action) -> `env.set_commands` -> command block of the
behaviour row -> FROZEN behaviour policy (history-encoder variant, mean action) -> `env.step` -> frozen style discriminator's
reward -> `alg.process_env_step`.  Same constructor and attributes as the reference (`alg`, `actor_critic_bbc`,
`discriminator`, `learn`, `save` / `load` / `load_bbc`).  With `depth_encoder.if_depth` the runner builds the vision student
(depth encoder + a copy of the teacher's actor) and `learn` is `learn_vision` (:278-441): DAgger on the student's own rollouts,
depth images from `env.extras["depth"]` (csrc/qa_depth.hip).

MI355X: no host wait inside a rollout (`env.sync_reset_ids = False`: terminal discriminator rows come as a masked tensor,
episode statistics stay on the device until the logger reads them); the frozen behaviour policy is ONE `qa_mlp_forward` launch
per step (fused.PolicyChain, weights packed once -- they never change); GAE / Adam / Linear+ELU backward are the fused kernels
the task-level PPO mirror already uses."""
import os
import statistics
import time
from collections import deque
from copy import deepcopy

import torch
import torch.distributed as dist

from quadrupedal_agility_amd.rsl_rl.utils.utils import Normalizer, TorchNormalizer
from quadrupedal_agility_amd.tsc.rsl_rl.algorithms import PPO, Discriminator
from quadrupedal_agility_amd.tsc.rsl_rl.modules import ActorCriticBBC, ActorCriticTSC, Estimator
from quadrupedal_agility_amd.tsc.rsl_rl.modules.depth_backbone import DepthOnlyFCBackbone58x87, RecurrentDepthBackbone


class OnPolicyRunner:
    def __init__(self, env, train_cfg, log_dir=None, device="cpu"):
        self.cfg, self.alg_cfg = train_cfg["runner"], dict(train_cfg["algorithm"])
        self.policy_cfg, self.estimator_cfg = train_cfg["policy"], train_cfg["estimator"]
        self.depth_encoder_cfg = train_cfg["depth_encoder"]
        self.device, self.env = device, env
        if torch.device(device).type == "cuda":          # the committed hipBLASLt / rocBLAS picks for this learner's fp32 GEMM shapes
            from quadrupedal_agility_amd.rsl_rl.runners.on_policy_runner import enable_tuned_gemms
            self.tuned_gemms = enable_tuned_gemms()
        e = env.cfg.env
        self.num_obs, self.n_proprio, self.n_auxiliary, self.n_scan = env.num_obs, e.n_proprio, e.n_auxiliary, e.n_scan
        self.n_priv, self.n_priv_latent, self.history_len = e.n_priv, e.n_priv_latent, e.history_len
        self.num_actions_d, self.num_actions_c = env.num_actions_d, env.num_actions_c
        self.num_actions = 1 + self.num_actions_d * self.num_actions_c
        self.num_command, self.num_obs_bbc = e.num_command, e.num_observations_bbc
        self.num_critic_obs = e.num_observations_bbc + e.history_len * (e.n_proprio - e.n_auxiliary)
        self.num_actions_bbc = e.num_actions_bbc
        self.num_disc_obs, self.disc_obs_len = e.num_obs_disc, e.disc_obs_len
        r = self.cfg
        self.disc_loss_function = r["disc_loss_function"]
        self.if_depth = bool(self.depth_encoder_cfg["if_depth"])
        self.n_depth_latent, self.n_delta_yaw = self.policy_cfg["scan_encoder_dims"][-1], e.n_delta_yaw

        self.actor_critic = ActorCriticTSC(self.n_proprio, self.n_auxiliary, self.n_scan, self.num_obs, self.n_priv_latent, self.n_priv,
                                           self.history_len, self.num_actions_d, self.num_actions_c, device=device, **self.policy_cfg).to(device)
        self.actor_critic_bbc = ActorCriticBBC(self.num_obs_bbc, self.num_critic_obs, self.num_actions_bbc, self.n_proprio, self.n_auxiliary,
                                               self.history_len, self.n_priv, self.n_priv_latent, self.num_command, **self.policy_cfg).to(device)
        self.estimator = Estimator(input_dim=self.n_proprio - self.n_auxiliary, output_dim=self.n_priv,
                                   hidden_dims=self.estimator_cfg["hidden_dims"]).to(device)
        self.depth_encoder = self.depth_actor = None
        if self.if_depth:                        # :88-101
            self.depth_backbone = DepthOnlyFCBackbone58x87(self.n_proprio, self.n_depth_latent, self.depth_encoder_cfg["hidden_dims"])
            self.depth_encoder = RecurrentDepthBackbone(self.depth_backbone, self.n_depth_latent, env.cfg).to(device)
            if (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1 and dist.get_backend() == "nccl"
                    and torch.device(device).type == "cuda"):          # batch statistics over ALL ranks' samples (byol.py:41-43)
                self.depth_encoder = torch.nn.SyncBatchNorm.convert_sync_batchnorm(self.depth_encoder)
                self.depth_backbone = self.depth_encoder.base_backbone
            self.depth_actor = deepcopy(self.actor_critic.actor)
            self.depth_backbone.augment = self.depth_encoder.byol_learner.augment1
        self.alg = PPO(self.actor_critic, self.actor_critic_bbc, self.estimator, self.estimator_cfg, self.depth_encoder, self.depth_encoder_cfg,
                       self.depth_actor, device=device, **self.alg_cfg)
        self.num_steps_per_env, self.save_interval = r["num_steps_per_env"], r["save_interval"]
        self.dagger_update_freq = self.alg_cfg["dagger_update_freq"]
        self.alg.init_storage(env.num_envs, self.num_steps_per_env, [env.num_obs], [env.num_privileged_obs], [self.num_actions])
        on_gpu = torch.device(device).type == "cuda"
        norm = TorchNormalizer(self.num_disc_obs * self.disc_obs_len, device) if on_gpu else Normalizer(self.num_disc_obs * self.disc_obs_len)
        self.discriminator = Discriminator(self.num_disc_obs * self.disc_obs_len, self.num_disc_obs, env.dim_c, env.dt, self.disc_loss_function,
                                           Normalizer(1) if self.disc_loss_function == "WassersteinLoss" else None, r["reward_i_coef"],
                                           r["reward_us_coef"], r["reward_ss_coef"], r["reward_t_coef"], self.disc_obs_len, r["disc_hidden_units"],
                                           norm, device).to(device)
        # data parallel (BASELINE configs 3 and 4 are 8-GPU jobs): one process per GPU owns num_envs envs of the job, every
        # optimiser step all-reduces ONE flat gradient bucket, the KL mean and the advantage moments are global, all ranks start from
        # rank 0's weights -> the replicas stay bit-identical.  The frozen nets (behaviour policy, discriminator) are loaded, not trained.
        self.distributed = dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or os.environ.get("QA_FORCE_DATA_PARALLEL") == "1")
        self.rank = dist.get_rank() if self.distributed else 0
        if self.distributed:
            from quadrupedal_agility_amd.rsl_rl.runners.on_policy_runner import GradSync
            self.alg.grad_sync = GradSync()
            self.alg.storage.global_moments = True
            for m in (self.actor_critic, self.actor_critic_bbc, self.estimator, self.discriminator, self.depth_encoder, self.depth_actor):
                if m is not None:
                    for t in list(m.parameters()) + list(m.buffers()):
                        dist.broadcast(t.data, src=0)
        if self.rank != 0:
            log_dir = None                      # rank 0 logs and saves
        self.learn = self.learn_vision if self.if_depth else self.learn_RL
        self.log_dir, self.writer = log_dir, None
        self.tot_timesteps, self.tot_time, self.current_learning_iteration = 0, 0, 0
        self.last_perf = {}
        self._bbc_chain = None
        self._teacher_chain_obj = self._style_chain_obj = None
        self._rs, self._rollout_graphs, self._rollout_warm = None, {}, 0
        self.use_rollout_graph = os.environ.get("QA_TSC_ROLLOUT_GRAPH", "1") != "0"
        self.use_fused_policy = on_gpu and os.environ.get("QA_FUSED_POLICY", "1") != "0"
        self.use_hybrid_act = os.environ.get("QA_TSC_HYBRID_ACT", "1") != "0"      # sampling + log-probs + storage rows + action history: qa_rollout_act_hybrid
        env.sync_reset_ids = False              # rollouts never wait for the GPU

    # ------------------------------------------------------------------ the frozen behaviour policy: one launch per env step
    def _behaviour_policy(self):
        bbc = self.actor_critic_bbc
        if not self.use_fused_policy:
            return lambda obs: bbc.act_inference(obs, hist_encoding=True).detach()
        if self._bbc_chain is None:
            from quadrupedal_agility_amd.rsl_rl.algorithms.fused import PolicyChain
            chain = PolicyChain.describe(bbc, None, False, hist_encoding=True, with_critic=False)
            if chain is None:
                self.use_fused_policy = False
                return self._behaviour_policy()
            with torch.inference_mode():
                chain.pack()
            self._bbc_chain = chain
        return lambda obs: self._bbc_chain.forward(obs)[0]

    # ------------------------------------------------------------------ r4: the teacher's own networks and the style reward on qa_mlp_forward
    def _teacher_chain(self):
        """estimator + scan / privileged encoders + actor trunk + both heads + critic of one env step as ONE launch (fused.PolicyChain.
        describe_task_level; was ~55 launches: 13 GEMMs with their ELUs, concatenations and the clone of the 800-wide row).  None keeps the
        module path (history-encoder rollouts, other activations)."""
        if not self.use_fused_policy:
            return None
        if self._teacher_chain_obj is None:
            from quadrupedal_agility_amd.rsl_rl.algorithms.fused import PolicyChain, SplitTeacherChain
            # QA_TSC_SPLIT_CHAIN=1 (opt-in), few row tiles (<= 2048 envs = 128 workgroups on 256 CUs): actor side and critic as two launches side by
            # side.  Measured at 1024 envs (r4 GPU call 10): critic 55.5 us beside actor 48.6 us instead of 98 us in one launch, but the fork / join
            # waits put 16 us of gaps around them: 9 us per env step, 31.97 vs 31.82 ms per iteration = noise.  Correct (bit-identical), not worth a default.
            split = self.env.num_envs <= 2048 and os.environ.get("QA_TSC_SPLIT_CHAIN", "0") == "1"
            ch = SplitTeacherChain.describe(self.actor_critic, self.estimator, self.alg.train_with_estimated_states) if split else None
            self._teacher_chain_obj = ch or PolicyChain.describe_task_level(self.actor_critic, self.estimator, self.alg.train_with_estimated_states) or False
        return self._teacher_chain_obj or None

    def _style_chain(self):
        """the frozen discriminator's trunk + three heads as one launch (packed once: it is not trained here), fed by qa_disc_prepare and read
        by qa_rollout_post_amp -- Discriminator.predict_disc_reward + process_env_step's reward / done rows in three launches (was ~40)"""
        if not self.use_fused_policy:
            return None
        if self._style_chain_obj is None:
            from quadrupedal_agility_amd.rsl_rl.algorithms.fused import PolicyChain
            d = self.discriminator
            ok = d.disc_loss_function == "MSELoss" and torch.is_tensor(getattr(d.normalizer, "mean", None)) and d.normalizer.mean.is_cuda
            ch = PolicyChain.describe_discriminator(d) if ok else None
            if ch is not None:
                with torch.inference_mode():
                    ch.pack()
            self._style_chain_obj = ch or False
        return self._style_chain_obj or None

    # ------------------------------------------------------------------ the teacher's rollout: eager or as recorded launches
    KEYS = ("rew", "rew_i", "rew_us", "rew_ss", "rew_t", "len")

    def _alloc_rollout_state(self):
        """Everything a rollout carries from one env step to the next lives in persistent tensors updated in place, so that the 24
        steps can be recorded once and replayed (the env's own rows -- obs_buf, obs_bbc_buf, ... -- already are)."""
        env, dev, N, T = self.env, self.device, self.env.num_envs, self.num_steps_per_env
        d = env.get_observations_disc()
        self._rs = dict(obs_bbc=env.get_observations_bbc().clone(), hist=torch.stack([d] * self.disc_obs_len, dim=1).clone(),
                        ahist=torch.zeros(N, env.cfg.domain_rand.action_buf_len, self.num_actions, device=dev),
                        cur=torch.zeros(6, N, device=dev), fin_vals=torch.zeros(T, 6, N, device=dev),
                        fin_masks=torch.zeros(T, N, dtype=torch.bool, device=dev), fin_reach=torch.zeros(T, N, dtype=torch.bool, device=dev))
        self._rollout_graphs = {}          # hist_encoding -> (graph, ep_infos) | False

    def _rollout_steps(self, hist_encoding, logging):
        """num_steps_per_env x (task policy -> set_commands -> frozen behaviour policy -> env.step -> discriminator reward -> storage)
        (:167-246); returns the per-step `episode` dicts (device tensors)"""
        env, alg, rs = self.env, self.alg, self._rs
        bbc = self._behaviour_policy()
        n_cmd = 6 + env.dim_c
        obs = env.get_observations()
        infos, ep_infos = {"depth": None}, []
        on_gpu = obs.is_cuda
        tchain = self._teacher_chain() if (on_gpu and not hist_encoding) else None
        dchain = self._style_chain() if on_gpu else None
        if tchain is not None:
            tchain.pack()                    # the weights changed in the last update(); one small launch per rollout
        if dchain is not None:
            import ctypes as C
            from quadrupedal_agility_amd import _capi
            from quadrupedal_agility_amd.rsl_rl.algorithms import fused
            lib, disc, st, N = _capi.load_library(), self.discriminator, alg.storage, env.num_envs
            P = lambda x: C.c_void_p(x.data_ptr())
        for t in range(self.num_steps_per_env):
            #  the two history buffers swap ROLES on the host every step and a recording bakes both addresses in: with an odd number of
            # steps the newest history would end in the other buffer than the one the next replay starts reading -- the hybrid launch needs an even T
            hybrid = (tchain is not None and dchain is not None and self.use_hybrid_act and getattr(env, "_step_dev", None) is not None and env._step_dev.is_cuda
                      and self.num_steps_per_env % 2 == 0)
            if hybrid:
                if "ahist2" not in rs:
                    rs["ahist2"] = torch.zeros_like(rs["ahist"])
                # the history is rolled OUT of place into the other of two buffers (coalesced copy); `ahist` names the current one
                actions = alg.act(obs, obs, infos, hist_encoding=hist_encoding, chain=tchain, action_history=(rs["ahist"], rs["ahist2"]),
                                  rng=(int(env.sim.cfg.seed) + 7919, env._step_dev, int(getattr(env.sim.cfg, "env_id_offset", 0))))
                rs["ahist"], rs["ahist2"] = rs["ahist2"], rs["ahist"]
            else:
                actions = alg.act(obs, obs, infos, hist_encoding=hist_encoding, chain=tchain)
                rs["ahist"].copy_(torch.cat([rs["ahist"][:, 1:], actions[:, None, :]], dim=1))
            rs["obs_bbc"][:, -n_cmd:] = env.set_commands(actions)
            obs, privileged_obs, rewards, dones, infos, _ids, _term = env.step(bbc(rs["obs_bbc"]), rs["ahist"])
            disc_obs = env.get_observations_disc()
            # history of discriminator observations: the terminal row for envs that reset, then restart (:219-234)
            hist = torch.cat([rs["hist"][:, 1:], env.obs_disc_term_buf.unsqueeze(1)], dim=1)
            done = dones.view(torch.bool) if dones.dtype == torch.uint8 else dones != 0      # 0 / 1 bytes from the goal-step kernel
            if dchain is not None:
                # Discriminator.predict_disc_reward (:71-118) + PPO.process_env_step (:139-147): prepare (normalise + clip) -> trunk + heads ->
                # reward mapping, mixing, time-out bootstrap, reward / done rows of the storage and the episode sums, three launches
                x = fused.disc_prepare([hist.view(N, -1)], disc._task_mask, disc._frame_mult.view(-1), None, disc.normalizer)
                d, eps, logits = dchain.forward(x)
                ts = alg.store_transition_rows(dones)
                log_ptrs = (P(rs["cur"]), P(rs["fin_vals"][ts]), P(rs["fin_masks"][ts])) if logging else (None, None, None)
                #  the reference bootstraps time-outs only when the env sends them (`'time_outs' in infos` <=> cfg.env.send_timeouts)
                tout = env.bk.time_out_buf if "time_outs" in infos else self._no_timeouts(env)
                rc = lib.qa_rollout_post_amp(P(rewards), P(dones.to(torch.int64)), P(tout), P(st.values[ts]), P(d), P(eps), P(logits), int(disc.dim_c),
                                             P(rs["obs_bbc"]), int(rs["obs_bbc"].stride(0)), int(rs["obs_bbc"].shape[1]),
                                             float(disc.reward_i_coef), float(disc.reward_us_coef), float(disc.reward_ss_coef), float(disc.reward_t_coef),
                                             float(disc.dt), float(alg.gamma), N, P(st.rewards[ts]), P(st.dones[ts]), *log_ptrs,
                                             C.c_void_p(torch.cuda.current_stream().cuda_stream))
                if rc != 0:
                    raise RuntimeError(f"qa_rollout_post_amp failed with code {rc}: {lib.qa_last_error().decode()}")
                if logging:
                    if "episode" in infos:
                        ep_infos.append(infos["episode"])
                    rs["fin_reach"][t].copy_(infos["reach_goal"])
            else:
                rewards, r_i, r_us, r_ss, r_t = self.discriminator.predict_disc_reward(rewards.unsqueeze(1), rs["obs_bbc"], hist)
                total_rew = alg.process_env_step(rewards, dones, infos)
                if logging:
                    if "episode" in infos:
                        ep_infos.append(infos["episode"])
                    rs["cur"] += torch.stack([total_rew, r_i, r_us, r_ss, r_t, torch.ones_like(r_t)])
                    rs["fin_vals"][t].copy_(rs["cur"]); rs["fin_masks"][t].copy_(done); rs["fin_reach"][t].copy_(infos["reach_goal"])
                    rs["cur"] *= (~done).to(rs["cur"].dtype)
            rs["obs_bbc"].copy_(env.get_observations_bbc())
            # one launch: the restart row broadcast over the history slots, written straight into the persistent buffer (`hist` is a fresh tensor)
            torch.where(done.view(-1, 1, 1), disc_obs.unsqueeze(1), hist, out=rs["hist"])
        return ep_infos

    def _no_timeouts(self, env):
        z = getattr(self, "_zero_timeouts", None)
        if z is None or z.shape != env.bk.time_out_buf.shape:
            z = self._zero_timeouts = torch.zeros_like(env.bk.time_out_buf)
        return z

    def _collect(self, hist_encoding, logging):
        """`_collect_rollout` with extras['delta_yaw_ok'] switched off for its duration only: the flag is the student's (learn_vision), the
        teacher's rollout never reads it (two launches per env step), and every other user of the same env -- play / eval scripts, a student
        runner built later -- keeps getting the tensor the reference always provides."""
        env = self.env
        prev = getattr(env, "want_delta_yaw_ok", True)
        env.want_delta_yaw_ok = False
        try:
            return self._collect_rollout(hist_encoding, logging)
        finally:
            env.want_delta_yaw_ok = prev

    def _collect_rollout(self, hist_encoding, logging):
        """One rollout.  On the GPU the 24 steps are recorded into ONE hipGraph per actor variant (privileged / history encoder) the
        first time they run and replayed afterwards: an eager step is ~180 launches and 1.8 ms of host time against 0.3 ms of kernels.
        What makes the replays differ from each other lives on the device: the step counter that keys the env's draws and gates the
        push (`env._step_dev`), torch's generator (graph-safe Philox offsets), the policy's weights."""
        env, alg, T = self.env, self.alg, self.num_steps_per_env
        key = (bool(hist_encoding), bool(logging))
        # (a depth camera that does not update every step decides `update_yaw` / the depth pass from the HOST's step counter: a replay would
        # repeat the phase seen at capture, so such a rollout is not recorded)
        phase_on_host = bool(env.cfg.depth.use_camera) and int(env.cfg.depth.update_interval) != 1
        if not (self.use_rollout_graph and torch.device(self.device).type == "cuda") or phase_on_host or self._rollout_graphs.get(key) is False:
            with torch.inference_mode():
                return self._rollout_steps(hist_encoding, logging)
        if key not in self._rollout_graphs:
            # one eager rollout of THIS variant first (it used to be one per runner, so the first privileged-encoder rollout built its
            # chain description, its action buffers and the second history buffer INSIDE the capture): lazy initialisations happen eagerly
            warm = getattr(self, "_rollout_warm_keys", None)
            if warm is None:
                warm = self._rollout_warm_keys = set()
            if key not in warm:
                warm.add(key)
                self._rollout_warm += 1
                with torch.inference_mode():
                    return self._rollout_steps(hist_encoding, logging)
            from quadrupedal_agility_amd.rsl_rl.runners.on_policy_runner import _no_gc
            try:
                self._behaviour_policy()
                torch.cuda.synchronize()
                counters = (env.global_counter, env.total_env_steps_counter, env.common_step_counter)
                g = torch.cuda.CUDAGraph()
                alg.storage.step = 0
                with _no_gc(), torch.cuda.graph(g):
                    with torch.inference_mode():
                        ep_infos = self._rollout_steps(hist_encoding, logging)      # host side effects run now, GPU work on replay
                env.global_counter, env.total_env_steps_counter, env.common_step_counter = counters
                self._rollout_graphs[key] = (g, ep_infos)
            except Exception as e:      # never fatal: the eager loop is the same code
                print(f"[tsc rollout graph] capture failed, staying eager: {e}")
                if os.environ.get("QA_DEBUG_GRAPH"):
                    import traceback
                    traceback.print_exc()
                self._rollout_graphs[key] = False
                torch.cuda.synchronize()
                alg.storage.step = 0
                with torch.inference_mode():
                    return self._rollout_steps(hist_encoding, logging)
        g, ep_infos = self._rollout_graphs[key]
        g.replay()
        alg.storage.step = T
        env.global_counter += T; env.total_env_steps_counter += T; env.common_step_counter += T
        return ep_infos

    def learn_RL(self, num_learning_iterations, init_at_random_ep_len=False):
        env, alg, dev = self.env, self.alg, self.device
        if self.log_dir is not None and self.writer is None:
            from quadrupedal_agility_amd.rsl_rl.runners.on_policy_runner import _make_writer
            self.writer = _make_writer(self.log_dir)
        if init_at_random_ep_len:
            env.episode_length_buf = torch.randint_like(env.episode_length_buf, high=int(env.max_episode_length))
        if getattr(self, "_rs", None) is None:
            self._alloc_rollout_state()
        alg.actor_critic.train()
        keys = self.KEYS
        buffers = {k: deque(maxlen=1000) for k in keys}
        reach_goal_buffer = deque(maxlen=1000)
        logging = self.log_dir is not None
        tot_iter = self.current_learning_iteration + num_learning_iterations
        for it in range(self.current_learning_iteration, tot_iter):
            start = time.time()
            hist_encoding = it % self.dagger_update_freq == 0
            ep_infos = self._collect(hist_encoding, logging)
            collection_time = time.time() - start
            start = time.time()
            with torch.inference_mode():
                alg.compute_returns(env.get_observations())
            losses = alg.update()
            mean_hist_latent_loss = alg.update_dagger() if hist_encoding else 0.0
            learn_time = time.time() - start
            if logging:            # ONE host read per iteration for the episode statistics
                rs = self._rs
                masks = rs["fin_masks"]
                sel = rs["fin_vals"].permute(0, 2, 1)[masks].cpu()
                for i, k in enumerate(keys):
                    buffers[k].extend(sel[:, i].tolist())
                reach_goal_buffer.extend(rs["fin_reach"][masks].float().cpu().tolist())
                if len(reach_goal_buffer) > 0:
                    env.success_rate = statistics.mean(reach_goal_buffer)
                self._log(it, losses, mean_hist_latent_loss, collection_time, learn_time, buffers, ep_infos)
                if it % self.save_interval == 0:
                    self.save(os.path.join(self.log_dir, "model.pt"))
            self.last_perf = {"collection_time": collection_time, "learn_time": learn_time,
                              "fps": self.num_steps_per_env * env.num_envs / (collection_time + learn_time)}
        self.current_learning_iteration = tot_iter
        if logging:
            self.save(os.path.join(self.log_dir, "model.pt"))

    # ------------------------------------------------------------------ the vision student's env step: the part without autograd as recorded launches
    def _vision_env_step(self, vs, n_cmd):
        """One env step of `learn_vision` behind the student's action (:392-411): set_commands -> frozen behaviour policy -> env.step -> next behaviour
        observation, action-history restart of the envs that reset.  No autograd reaches in here (the student's action enters detached), and every
        value that crosses a step lives in a persistent tensor (`vs`, the env's own rows), so on the GPU the segment is recorded ONCE per camera phase
        -- the depth camera fires every `depth.update_interval`-th step, decided from the host's step counter -- and replayed: ~100 eager launches
        and their host time per env step become one graph launch.  The student's networks, which carry autograd through the 24 steps, stay eager
        (their batch reductions must not be recorded: profiles/r2_hipgraph_stale_reductions.md).  Returns (obs, rewards, infos)."""
        env = self.env
        upd = bool(env.cfg.depth.use_camera) and (env.global_counter + 1) % int(env.cfg.depth.update_interval) == 0

        def body():
            with torch.no_grad():
                vs["obs_bbc"][:, -n_cmd:] = env.set_commands(vs["ahist"][:, -1])
                obs, _priv, rewards, dones, infos, _ids, _term = env.step(vs["bbc"](vs["obs_bbc"]))
                vs["obs_bbc"].copy_(env.get_observations_bbc())
                done = dones.view(torch.bool) if dones.dtype == torch.uint8 else dones != 0
                vs["done"].copy_(done)
                vs["ahist"].mul_((~done).view(-1, 1, 1).to(vs["ahist"].dtype))
                vs["yaw_ok"].copy_(infos["delta_yaw_ok"])
                return obs, rewards, infos.get("episode"), infos["reach_goal"]

        graphs = vs["graphs"]
        recordable = self.use_rollout_graph and torch.device(self.device).type == "cuda" and graphs.get(upd) is not False
        if recordable and upd not in graphs:
            if upd not in vs["warm"]:                     # one eager step of this phase first: lazy allocations happen outside the recording
                vs["warm"].add(upd)
                recordable = False
            else:
                from quadrupedal_agility_amd.rsl_rl.runners.on_policy_runner import _no_gc
                counters = (env.global_counter, env.total_env_steps_counter, env.common_step_counter)
                try:
                    torch.cuda.synchronize()
                    g = torch.cuda.CUDAGraph()
                    with _no_gc(), torch.cuda.graph(g):
                        out = body()                      # host side effects run now, GPU work on replay
                    graphs[upd] = (g, out)
                except Exception as e:                    # never fatal: the eager step is the same code
                    print(f"[tsc vision env-step graph] capture failed, staying eager: {e}")
                    if os.environ.get("QA_DEBUG_GRAPH"):
                        import traceback
                        traceback.print_exc()
                    graphs[upd] = False
                    torch.cuda.synchronize()
                    recordable = False
                env.global_counter, env.total_env_steps_counter, env.common_step_counter = counters
        if recordable:
            g, (obs, rewards, episode, reach) = graphs[upd]
            g.replay()
            env.global_counter += 1; env.total_env_steps_counter += 1; env.common_step_counter += 1
        else:
            obs, rewards, episode, reach = body()
        infos = {"depth": env.depth_buffer[:, -2] if upd else None, "delta_yaw_ok": vs["yaw_ok"], "reach_goal": reach}
        if episode is not None:
            infos["episode"] = episode
        return obs, rewards, infos

    def learn_vision(self, num_learning_iterations, init_at_random_ep_len=False):
        """Distillation of the teacher into the depth student (:278-441).  Per env step: the depth encoder turns the previous depth
        image + masked proprioception into [scan latent 32 | goal headings 2 | obstacle class 6]; the student actor acts on the
        teacher's observation with its heading / class entries replaced by the encoder's predictions (headings only where the teacher's
        |delta_yaw| < 0.6) and the scan latent replaced by the depth latent; the STUDENT's action drives the env (DAgger); the teacher's
        action on the same observation is the label.  One `update_depth_actor` per 24-step rollout, linear LR decay over 20 k iterations."""
        env, alg, dev = self.env, self.alg, self.device
        if self.log_dir is not None and self.writer is None:
            from quadrupedal_agility_amd.rsl_rl.runners.on_policy_runner import _make_writer
            self.writer = _make_writer(self.log_dir)
        env.cfg.noise.add_noise = False
        env.cfg.obstacle.curriculum = False
        env.cfg.env.next_goal_threshold = 0.45
        if hasattr(env, "bk"):
            env.bk._cfg.next_goal_threshold = 0.45
        tot_iter = self.current_learning_iteration + num_learning_iterations
        n_aux, n_pro, nd = self.n_auxiliary, self.n_proprio, self.num_actions_d
        yaw_sl, type_sl = slice(n_pro - n_aux, n_pro - n_aux + self.n_delta_yaw), slice(n_pro - n_aux + self.n_delta_yaw, n_pro)
        obs = env.get_observations()
        # r5: what the env half of a step carries from one step to the next, in place (`_vision_env_step`).  The recorded env steps hold the
        # ADDRESSES of these tensors, so they are allocated once per runner and re-initialised in place when learn_vision() is entered again
        # (rebinding them to fresh tensors left the replays reading the first call's action history -- the student's commands
        # never reached the env from the second call on -- and writing into freed blocks)
        vs = getattr(self, "_vs", None)
        if vs is None:
            vs = self._vs = dict(graphs={}, warm=set(), yaw_ok=torch.ones(env.num_envs, dtype=torch.bool, device=dev),
                                 done=torch.zeros(env.num_envs, dtype=torch.bool, device=dev),
                                 ahist=torch.zeros(env.num_envs, env.cfg.domain_rand.action_buf_len, self.num_actions, device=dev),
                                 obs_bbc=env.get_observations_bbc().clone())
        else:
            with torch.no_grad():
                vs["ahist"].zero_()
                vs["obs_bbc"].copy_(env.get_observations_bbc())
                vs["yaw_ok"].fill_(True); vs["done"].fill_(False)
        infos = {"depth": env.depth_buffer[:, -1].clone(), "delta_yaw_ok": torch.ones(env.num_envs, dtype=torch.bool, device=dev)}
        alg.depth_encoder.train(); alg.depth_actor.train()
        vs["bbc"] = self._behaviour_policy()
        bbc_key = self._bbc_chain if self._bbc_chain is not None else self.actor_critic_bbc
        if vs.get("bbc_key") is not bbc_key:         # another behaviour policy object (load_bbc): the recorded steps call the old one
            vs["bbc_key"], vs["graphs"], vs["warm"] = bbc_key, {}, set()
        n_cmd = 6 + env.dim_c
        keys = ("rew", "len")
        buffers = {k: deque(maxlen=1000) for k in keys}
        reach_goal_buffer = deque(maxlen=1000)
        cur = torch.zeros(2, env.num_envs, device=dev)
        logging = self.log_dir is not None
        ep_infos = []
        cfg_d = self.depth_encoder_cfg
        depth_latent = delta_yaw = obst_type = None
        for it in range(self.current_learning_iteration, tot_iter):
            start = time.time()
            depth_buffer, actions_teacher_buffer, actions_student_buffer = [], [], []
            yaw_student, yaw_teacher, type_student, type_teacher, yaw_ok = [], [], [], [], []
            fin_vals, fin_masks, fin_reach = [], [], []
            for _ in range(cfg_d["num_steps_per_env"]):
                obs = obs.clone()
                if infos["depth"] is not None:
                    with torch.no_grad():
                        obs[:, alg._priv_slice(True)] = alg.estimator(obs[:, :alg.num_prop])
                    obs_prop_depth = obs[:, :n_pro].clone()
                    obs_prop_depth[:, n_pro - n_aux:n_pro] = 0                       # the student does not see headings / obstacle class
                    out = alg.depth_encoder(infos["depth"].clone(), obs_prop_depth)
                    depth_latent = out[:, :self.n_depth_latent]
                    delta_yaw = 1.5 * out[:, self.n_depth_latent:self.n_depth_latent + self.n_delta_yaw]
                    obst_type = out[:, self.n_depth_latent + self.n_delta_yaw:]
                    depth_buffer.append(infos["depth"].clone())
                    yaw_student.append(delta_yaw); yaw_teacher.append(obs[:, yaw_sl])
                    type_student.append(obst_type); type_teacher.append(obs[:, type_sl])
                with torch.no_grad():
                    actions_teacher_buffer.append(alg.actor_critic.act_inference(obs, hist_encoding=True, scandots_latent=None))
                obs_student = obs.clone()
                ok = infos["delta_yaw_ok"].view(-1, 1)
                obs_student[:, yaw_sl] = torch.where(ok, delta_yaw.detach(), obs_student[:, yaw_sl])
                obs_student[:, type_sl] = torch.nn.functional.one_hot(torch.argmax(obst_type.detach(), dim=-1), num_classes=obst_type.shape[-1]).to(obs.dtype)
                yaw_ok.append(ok.float().mean())
                embedding = alg.depth_actor(obs_student, hist_encoding=True, scandots_latent=depth_latent)
                prob, mean = self.depth_actor.actor_d(embedding), self.depth_actor.actor_c(embedding)
                actions_student = torch.cat([torch.argmax(prob, dim=-1, keepdim=True).to(mean.dtype), mean], dim=-1)
                actions_student_buffer.append(torch.cat([prob, mean], dim=-1))
                with torch.no_grad():
                    vs["ahist"].copy_(torch.cat([vs["ahist"][:, 1:], actions_student[:, None, :].detach()], dim=1))
                    obs, rewards, infos = self._vision_env_step(vs, n_cmd)
                    done = vs["done"]
                    if logging:
                        if "episode" in infos:          # (a replayed step rewrites the same tensors: keep this step's values)
                            ep_infos.append({k: (v.clone() if torch.is_tensor(v) else v) for k, v in infos["episode"].items()})
                        cur += torch.stack([rewards, torch.ones_like(rewards)])
                        fin_vals.append(cur.clone()); fin_masks.append(done.clone()); fin_reach.append(infos["reach_goal"].clone())
                        cur *= (~done).to(cur.dtype)
            collection_time = time.time() - start
            start = time.time()
            losses = alg.update_depth_actor(torch.cat(actions_student_buffer), torch.cat(actions_teacher_buffer), torch.cat(yaw_student),
                                            torch.cat(yaw_teacher), torch.cat(type_student), torch.cat(type_teacher), torch.cat(depth_buffer))
            learn_time = time.time() - start
            alg.depth_encoder.detach_hidden_states()
            decay = lambda lr0: max(lr0 - (lr0 - cfg_d["learning_rate_min"]) * it / 20000, cfg_d["learning_rate_min"])     # noqa: E731
            for opt, lr0 in ((alg.depth_encoder_optimizer, cfg_d["learning_rate"]), (alg.depth_actor_optimizer, cfg_d["learning_rate"]),
                             (alg.byol_optimizer, cfg_d["learning_rate_byol"])):
                for group in opt.param_groups:
                    group["lr"] = decay(lr0)
            self.last_vision = dict(zip(("depth_actor_loss", "yaw_loss", "obst_type_loss", "byol_loss"), losses))
            self.last_vision["delta_yaw_ok_percentage"] = torch.stack(yaw_ok).mean().item()
            if logging:
                vals, masks, reach = torch.stack(fin_vals), torch.stack(fin_masks), torch.stack(fin_reach)
                sel = vals.permute(0, 2, 1)[masks].cpu()
                for i, k in enumerate(keys):
                    buffers[k].extend(sel[:, i].tolist())
                reach_goal_buffer.extend(reach[masks].float().cpu().tolist())
                self._log_vision(it, collection_time, learn_time, buffers, reach_goal_buffer, ep_infos)
                if it % self.save_interval == 0:
                    self.save(os.path.join(self.log_dir, "model.pt"))
            ep_infos.clear()
            self.last_perf = {"collection_time": collection_time, "learn_time": learn_time,
                              "fps": cfg_d["num_steps_per_env"] * env.num_envs / (collection_time + learn_time)}
        self.current_learning_iteration = tot_iter
        if logging:
            self.save(os.path.join(self.log_dir, "model.pt"))

    def _log_vision(self, it, collection_time, learn_time, buffers, reach_goal_buffer, ep_infos):
        """log_vision (:443-511): same tags"""
        w = self.writer
        steps = self.depth_encoder_cfg["num_steps_per_env"] * self.env.num_envs
        self.tot_timesteps += steps
        self.tot_time += collection_time + learn_time
        if ep_infos:
            for key in ep_infos[0]:
                vals = torch.stack([torch.as_tensor(e[key], device=self.device).reshape(()) for e in ep_infos])
                w.add_scalar("Episode/" + key, vals.mean().item(), it)
        v = self.last_vision
        w.add_scalar("Loss_depth/delta_yaw_ok_percent", v["delta_yaw_ok_percentage"], it)
        w.add_scalar("Loss_depth/depth_actor", v["depth_actor_loss"], it)
        w.add_scalar("Loss_depth/yaw", v["yaw_loss"], it)
        w.add_scalar("Loss_depth/obst_type", v["obst_type_loss"], it)
        w.add_scalar("Loss_depth/byol", v["byol_loss"], it)
        w.add_scalar("Perf/total_fps", int(steps / (collection_time + learn_time)), it)
        w.add_scalar("Perf/collection time", collection_time, it)
        w.add_scalar("Perf/learning_time", learn_time, it)
        if len(buffers["rew"]) > 0:
            w.add_scalar("Train/mean_reward", statistics.mean(buffers["rew"]), it)
            w.add_scalar("Train/mean_episode_length", statistics.mean(buffers["len"]), it)
            w.add_scalar("Train/success_rate", statistics.mean(reach_goal_buffer), it)
        if hasattr(w, "flush"):
            w.flush()

    def _log(self, it, losses, hist_loss, collection_time, learn_time, buffers, ep_infos):
        w = self.writer
        self.tot_timesteps += self.num_steps_per_env * self.env.num_envs
        self.tot_time += collection_time + learn_time
        if ep_infos:
            for key in ep_infos[0]:
                vals = torch.stack([torch.as_tensor(e[key], device=self.device).reshape(()) for e in ep_infos])
                w.add_scalar("Episode/" + key, vals.mean().item(), it)
        for tag, v in zip(("value_function", "surrogate", "estimator", "disc", "disc_acc", "priv_reg", "priv_reg_coef"), losses):
            w.add_scalar("Loss/" + tag, v, it)
        w.add_scalar("Loss/hist_latent_loss", hist_loss, it)
        w.add_scalar("Loss/learning_rate", self.alg.learning_rate, it)
        w.add_scalar("Perf/total_fps", int(self.num_steps_per_env * self.env.num_envs / (collection_time + learn_time)), it)
        w.add_scalar("Perf/collection time", collection_time, it)
        w.add_scalar("Perf/learning_time", learn_time, it)
        if len(buffers["rew"]) > 0:
            for tag, k in (("mean_reward", "rew"), ("mean_reward_i", "rew_i"), ("mean_reward_t", "rew_t"), ("mean_episode_length", "len")):
                w.add_scalar("Train/" + tag, statistics.mean(buffers[k]), it)
            w.add_scalar("Train/success_rate", float(self.env.success_rate), it)
        if hasattr(w, "flush"):
            w.flush()

    # ------------------------------------------------------------------ checkpoints (:443-520: same keys)
    def save(self, path, infos=None):
        self.alg.lr_to_host()                    # the optimiser's state dict carries a plain float learning rate, as the reference's
        d = {"model_state_dict": self.alg.actor_critic.state_dict(), "estimator_state_dict": self.alg.estimator.state_dict(),
             "optimizer_state_dict": self.alg.optimizer.state_dict(), "iter": self.current_learning_iteration, "infos": infos}
        if self.if_depth:                        # :618-620
            d["depth_encoder_state_dict"] = self.alg.depth_encoder.state_dict()
            d["depth_actor_state_dict"] = self.alg.depth_actor.state_dict()
        torch.save(d, path)
        self.alg.lr_to_device()

    def load(self, path, load_optimizer=True):
        d = torch.load(path, map_location=self.device, weights_only=False)
        self.alg.actor_critic.load_state_dict(d["model_state_dict"])
        self.alg.estimator.load_state_dict(d["estimator_state_dict"])
        if self.if_depth:                        # :629-640: a teacher checkpoint has neither key -> the student starts as the teacher's actor
            if "depth_encoder_state_dict" in d:
                self.alg.depth_encoder.load_state_dict(d["depth_encoder_state_dict"])
            if "depth_actor_state_dict" in d:
                self.alg.depth_actor.load_state_dict(d["depth_actor_state_dict"])
            else:
                self.alg.depth_actor.load_state_dict(self.alg.actor_critic.actor.state_dict())
        if load_optimizer:
            self.alg.optimizer.load_state_dict(d["optimizer_state_dict"])
            self.alg.learning_rate = float(self.alg.optimizer.param_groups[0]["lr"])      # the loaded rate, on the host and (if recorded) the device
            self.alg.lr_to_device()
            if torch.device(self.device).type == "cuda":      # the loaded group dict overwrote the flags; recorded launches hold the old state's addresses
                for g in self.alg.optimizer.param_groups:
                    g["capturable"], g["fused"] = True, True
                self.alg._graph, self.alg._warm_updates = None, 0
        self.current_learning_iteration = d["iter"]
        return d["infos"]

    def load_bbc(self, path):
        """the frozen behaviour controller, its estimator (estimator.load_estimator_bbc) and the style discriminator with its
        input normaliser from a behaviour-level model.pt (keys `actor_critic`, `estimator`, `disc`, `disc_normalizer`)"""
        import quadrupedal_agility_amd
        quadrupedal_agility_amd.install_reference_aliases()
        d = torch.load(path, map_location=self.device, weights_only=False)
        self.actor_critic_bbc.load_state_dict(d["actor_critic"])
        if self.estimator_cfg.get("load_estimator_bbc", False):
            self.alg.estimator.load_state_dict(d["estimator"])
        self.discriminator.load_state_dict(d["disc"])
        n = d["disc_normalizer"]
        self.discriminator.normalizer = TorchNormalizer.from_reference(n, self.device) if torch.device(self.device).type == "cuda" else n
        self._bbc_chain = None
        self._rollout_graphs = {}                # recorded rollouts hold the old packed weights and normaliser
        if getattr(self, "_vs", None) is not None:   # and so do the student's recorded env steps
            self._vs["graphs"], self._vs["warm"] = {}, set()

    def get_inference_policy_bbc(self, device=None):
        self.alg.actor_critic_bbc.eval()
        if device is not None:
            self.alg.actor_critic_bbc.to(device)
        return self.alg.actor_critic_bbc.act_inference

    def get_estimator_inference_policy(self, device=None):
        self.alg.estimator.eval()
        if device is not None:
            self.alg.estimator.to(device)
        return self.alg.estimator

    def get_depth_actor_inference_policy(self, device=None):
        self.alg.depth_actor.eval()
        if device is not None:
            self.alg.depth_actor.to(device)
        return self.alg.depth_actor

    def get_depth_encoder_inference_policy(self, device=None):
        self.alg.depth_encoder.eval()
        if device is not None:
            self.alg.depth_encoder.to(device)
        return self.alg.depth_encoder

    def get_inference_policy(self, device=None):
        self.alg.actor_critic.eval()
        if device is not None:
            self.alg.actor_critic.to(device)
        return self.alg.actor_critic.act_inference
