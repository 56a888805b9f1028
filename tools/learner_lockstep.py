#!/usr/bin/env python3
"""Learner lockstep: is the PRODUCT's GPU learner the same arithmetic as the torch-CPU learner?

The return-curve comparison "hybrid vs cpu" (same oracle physics, GPU learner against torch-CPU learner, 24 v 7 seeds) read -13 %, and seven
2-hour CPU seeds of a bimodal quantity can neither confirm nor clear that.  A learner is a deterministic function of (state, rollout, sample
tables), so the question has a sharper form: give BOTH learners the same state, the same rollout and the same tables, step both, and compare
what comes out -- every parameter group, every optimiser moment, the discriminator's input normaliser, the prior, the learning rate, the
returns / advantages and the logged scalars -- after every iteration of a real training run.

One trajectory is driven by the product: its rollouts (the oracle's physics on the host cores through tools/hybrid_backend.py, or the HIP
env with --physics hip), its GAE kernel, its recorded PPO / discriminator / DAgger steps.  Beside it run two torch-CPU learners (the same
`SSInfoGAIL` class on device "cpu": the plain PyTorch expressions, i.e. the learner of the all-CPU arm), all three fed the SAME tables per
iteration (`SSInfoGAIL.update(tables=...)`: rollout permutation, the 80 x minibatch row indices into the replay ring, the labelled and the
unlabelled expert set; drawn here from one CPU generator) and the same rollout storage / ring inserts:

  forced  -- re-synchronised to the GPU learner's state before every iteration: its difference after the iteration is the arithmetic
             difference of ONE iteration (PPO steps + discriminator steps + DAgger step) at that point of training;
  free    -- synchronised once, at iteration 0, never again: its difference is what per-iteration differences ACCUMULATE to when both
             learners see the same data -- smooth growth at rounding level says "same learner", a jump names the iteration and the tensor.

The physics does not enter the comparison (one arm produces every rollout), so chaos in the contact dynamics cannot mask or fake a learner
difference.  Checker side only: loads the oracle, never imported by the product.

  python tools/learner_lockstep.py --amp --num_envs 1024 --iters 200 --out profiles/r6_learner_lockstep_cfg3.json
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def build(device, num_envs, seed, amp, physics, ring):
    import torch
    from quadrupedal_agility_amd.legged_gym.envs import task_registry
    from quadrupedal_agility_amd.legged_gym.envs.go2.go2_locomotion_config import Go2LocomotionCfg, Go2LocomotionCfgAlgo
    from quadrupedal_agility_amd.legged_gym.utils import get_args
    from quadrupedal_agility_amd.legged_gym.utils.cfg_to_c import make_qa_config
    cfg = Go2LocomotionCfg(); cfg.env.num_envs = num_envs; cfg.terrain.mesh_type = "plane"; cfg.env.mocap_state_init = bool(amp); cfg.seed = seed
    t = Go2LocomotionCfgAlgo(); t.runner.amp_enabled = bool(amp); t.seed = seed; t.runner.save_interval = 10 ** 9
    if ring:
        t.algorithm.disc_replay_buffer_size = int(ring)
    torch.manual_seed(seed)
    if device == "cpu":
        from tests.oracle_backend import OracleBackend
        args = get_args(["--device", "cpu"])
        env, _ = task_registry.make_env("go2_locomotion", args=args, env_cfg=cfg, backend=(OracleBackend if amp else OracleBackend(make_qa_config(cfg, seed=seed))))
    elif physics == "oracle":
        os.environ["QA_ROLLOUT_GRAPH"] = "0"
        from tools.hybrid_backend import HybridBackend
        args = get_args(["--device", "gpu"])
        env, _ = task_registry.make_env("go2_locomotion", args=args, env_cfg=cfg, backend=(HybridBackend if amp else HybridBackend(make_qa_config(cfg, seed=seed))))
    else:
        args = get_args(["--device", "gpu"])
        env, _ = task_registry.make_env("go2_locomotion", args=args, env_cfg=cfg)
    runner, _ = task_registry.make_alg_runner(env, name="go2_locomotion", args=args, train_cfg=t, log_root=None)
    return runner, env


# ---------------------------------------------------------------------------------------------------- state: read, copy, compare
def _optimisers(alg):
    return {"adam_ac": alg.optim_ac, "adam_estimator": alg.optim_estimator, "adam_hist_encoder": alg.optim_hist_encoder,
            **({"adam_disc": alg.optim_d, "adam_q_eps": alg.optim_q_eps, "adam_q_c": alg.optim_q_c} if alg.amp_enabled else {})}


def _opt_params(opt):
    return [p for g in opt.param_groups for p in g["params"]]


def state_groups(alg, env):
    """{group: flat float64 CPU tensor}: parameters by module, Adam moments by optimiser, normaliser, prior, learning rate"""
    import torch
    cat = lambda ts: torch.cat([t.detach().reshape(-1).double().cpu() for t in ts]) if ts else torch.zeros(0, dtype=torch.float64)
    ac = alg.actor_critic
    g = {"actor_trunk": cat(list(ac.actor_trunk.parameters())), "actor_head": cat(list(ac.actor_head.parameters())),
         "critic_trunk": cat(list(ac.critic_trunk.parameters())), "critic_head": cat(list(ac.critic_head.parameters())),
         "priv_encoder": cat(list(ac.priv_encoder.parameters())), "history_encoder": cat(list(ac.history_encoder.parameters())),
         "estimator": cat(list(alg.estimator.parameters())), "lr_ac": torch.tensor([alg.lr_ac], dtype=torch.float64)}
    if not ac.fixed_std:
        g["std"] = cat([ac.std])
    if alg.amp_enabled:
        d = alg.disc
        g.update({"disc_trunk": cat(list(d.trunk.parameters())), "disc_head": cat(list(d.linear.parameters())),
                  "disc_encoder_eps": cat(list(d.encoder_eps.parameters())), "disc_classifier": cat(list(d.classifier.parameters())),
                  "normaliser_mean": cat([alg.disc_normalizer.mean]), "normaliser_var": cat([alg.disc_normalizer.var]),
                  "normaliser_count": cat([alg.disc_normalizer.count]), "prior": cat([torch.as_tensor(env.prior_parameters)])})
    for name, opt in _optimisers(alg).items():
        ps = [p for p in _opt_params(opt) if p in opt.state and "exp_avg" in opt.state[p]]
        g[name + "_m"] = cat([opt.state[p]["exp_avg"] for p in ps])
        g[name + "_v"] = cat([opt.state[p]["exp_avg_sq"] for p in ps])
        g[name + "_step"] = cat([torch.as_tensor(opt.state[p]["step"]) for p in ps])
    return g


def rel_l2(a, b):
    """|a - b| / |b| per group (b = the torch-CPU learner); groups that do not exist yet on one side are skipped"""
    out = {}
    for k in a:
        if k in b and a[k].numel() == b[k].numel() and a[k].numel():
            nb = float(b[k].norm())
            out[k] = float((a[k] - b[k]).norm()) / (nb if nb > 0 else 1.0)
    return out


def sync_state(src_alg, src_env, dst_alg, dst_env):
    """dst learner := src learner: networks, optimiser moments and step counts, learning rate, normaliser, prior, schedule counters"""
    import torch
    with torch.no_grad():
        for name in ("actor_critic", "estimator", "disc"):
            sm, dm = getattr(src_alg, name), getattr(dst_alg, name)
            for (ks, ps), (kd, pd) in zip(sm.state_dict().items(), dm.state_dict().items()):
                assert ks == kd
                pd.copy_(ps.to(pd.device))
        if not src_alg.actor_critic.fixed_std:
            dst_alg.actor_critic.std.data.copy_(src_alg.actor_critic.std.data.to(dst_alg.actor_critic.std.device))
        for (n1, so), (n2, do) in zip(_optimisers(src_alg).items(), _optimisers(dst_alg).items()):
            for ps, pd in zip(_opt_params(so), _opt_params(do)):
                st = so.state.get(ps)
                if not st or "exp_avg" not in st:
                    do.state.pop(pd, None)
                    continue
                dd = do.state.setdefault(pd, {})
                for k in ("exp_avg", "exp_avg_sq"):
                    if k not in dd:
                        dd[k] = torch.zeros_like(pd, memory_format=torch.preserve_format)
                    dd[k].copy_(st[k].to(pd.device))
                step = float(torch.as_tensor(st["step"]).item())
                if "step" in dd and torch.is_tensor(dd["step"]):
                    dd["step"].fill_(step)
                else:      # the form torch's Adam creates for this device: a float32 scalar tensor (on the device when capturable)
                    dd["step"] = torch.tensor(step, dtype=torch.float32, device=pd.device if do.param_groups[0].get("capturable") else "cpu")
        dst_alg.lr_ac = src_alg.lr_ac
        if src_alg.amp_enabled:
            sn, dn = src_alg.disc_normalizer, dst_alg.disc_normalizer
            dn.mean.copy_(sn.mean.to(dn.mean.device)); dn.var.copy_(sn.var.to(dn.var.device)); dn.count.copy_(sn.count.to(dn.count.device))
            dst_env.prior_parameters = torch.as_tensor(src_env.prior_parameters).detach().to(torch.as_tensor(dst_env.prior_parameters).device)
    for k in ("learning_steps", "priv_reg_counter", "info_max_coef_on"):
        setattr(dst_alg, k, getattr(src_alg, k))
    dst_env.task_obs_weight = src_env.task_obs_weight


def copy_rollout(src_alg, dst_alg):
    s, d = src_alg.storage, dst_alg.storage
    for k in ("_obs_padded", "actions", "rewards", "dones", "values", "actions_log_prob", "mu", "sigma"):
        getattr(d, k).copy_(getattr(s, k).to(getattr(d, k).device))
    d.step = s.step


def draw_tables(gen, alg):
    import torch
    st = alg.storage
    batch = st.num_envs * st.num_transitions_per_env
    nmb = alg.num_mini_batches
    t = {"perm": torch.randperm(batch // nmb * nmb, generator=gen)}
    if alg.amp_enabled:
        n_d = alg.num_learning_epochs * nmb * 4
        mb = batch // n_d
        ml, rb = alg.motion_loader, alg.disc_storage
        t["pi"] = torch.randint(0, int(rb.num_samples), (n_d, mb), generator=gen)
        t["lb"] = torch.randint(0, ml.preloaded_s_lb.shape[0], (n_d, mb), generator=gen)
        t["ulb"] = torch.randint(0, ml.preloaded_s_ulb.shape[0], (n_d, mb), generator=gen)
    return t


def run(a):
    import torch
    threads = int(os.environ.get("QA_CPU_THREADS", "0")) or min(64, len(os.sched_getaffinity(0)))
    prev_threads = torch.get_num_threads()
    torch.set_num_threads(threads)
    try:
        return _run(a, threads)
    finally:
        torch.set_num_threads(prev_threads)          # (a caller in the same process, e.g. the test suite, keeps its own setting)


def _run(a, threads):
    import torch
    gpu, genv = build(a.driver, a.num_envs, a.seed, a.amp, a.physics, a.ring)
    sync = torch.cuda.synchronize if a.driver == "gpu" else (lambda: None)
    arms = {}
    for name in ["forced"] + (["free"] if a.free else []) + (["control"] if a.control > 0 else []):
        arms[name] = build("cpu", a.num_envs, a.seed, a.amp, a.physics, a.ring)
    noise_gen = torch.Generator().manual_seed(77 + a.seed)
    ga = gpu.alg
    dev = ga.device
    for r, e in arms.values():
        ca = r.alg
        if a.amp:      # the expert sets are drawn at construction: every learner gets the GPU learner's
            for k in ("preloaded_s_lb", "preloaded_s_ulb", "preloaded_label"):
                setattr(ca.motion_loader, k, getattr(ga.motion_loader, k).cpu().clone())
        sync_state(ga, genv, ca, e)
        ca.actor_critic.train(); ca.disc.train()
    if a.amp:          # ring inserts of the rollout reach every learner's ring, in the same order
        orig_insert = ga.disc_storage.insert

        def insert_all(states, eps, c):
            orig_insert(states, eps, c)
            hs, he, hc = states.detach().cpu(), eps.detach().cpu(), c.detach().cpu()
            for r, _ in arms.values():
                r.alg.disc_storage.insert(hs, he, hc)
        ga.disc_storage.insert = insert_all
    gpu.learn(0, init_at_random_ep_len=True)          # rollout state, lean exports, train() modes; zero iterations
    if hasattr(genv, "set_lean_exports") and a.driver == "gpu":
        genv.set_lean_exports(gpu._lean_mask_for(genv))
    gen = torch.Generator().manual_seed(1000 + a.seed)
    rows, t_start = [], time.time()
    for it in range(a.iters):
        hist_encoding = it % gpu.dagger_update_freq == 0
        gpu._collect(hist_encoding, False)
        sync()
        if "forced" in arms:
            sync_state(ga, genv, arms["forced"][0].alg, arms["forced"][1])
        if "control" in arms:
            # the CONTROL: a torch-CPU learner started from the same state with every parameter moved by ~one fp32 ulp (relative N(0, control^2)).  Its
            # difference from the `forced` arm after the iteration is what ONE iteration does to a rounding-sized difference in its input: the
            # yardstick for reading the GPU-vs-CPU difference (a learner with another arithmetic cannot agree better than this)
            ca, ce = arms["control"][0].alg, arms["control"][1]
            sync_state(ga, genv, ca, ce)
            with torch.no_grad():
                for m in (ca.actor_critic, ca.estimator, ca.disc):
                    for p in m.parameters():
                        p.mul_(1.0 + a.control * torch.randn(p.shape, generator=noise_gen))
        for r, e in arms.values():
            copy_rollout(ga, r.alg)
            e.task_obs_weight = genv.task_obs_weight
        tables = draw_tables(gen, ga)
        tables_dev = {k: v.to(dev) for k, v in tables.items()}
        perm_dagger = torch.randperm(tables["perm"].shape[0], generator=gen) if hist_encoding else None
        last_obs = gpu._obs_cur
        with torch.inference_mode():
            ga.compute_returns(last_obs)
        ret_g, adv_g = ga.storage.returns.double().cpu().reshape(-1), ga.storage.advantages.double().cpu().reshape(-1)
        t0 = time.time()
        losses_g = [float(v) for v in ga.update(tables_dev)]
        hl_g = ga.update_dagger(perm_dagger.to(dev)) if hist_encoding else None
        sync()
        t_gpu = time.time() - t0
        sg = state_groups(ga, genv)
        row = {"iteration": it + 1, "hist_encoding": bool(hist_encoding), "gpu_update_s": round(t_gpu, 3), "gpu_path": {
            "ppo_steps_recorded": bool(ga._ac_graph), "disc_steps_recorded": bool(ga._disc_graph) if a.amp else None}}
        for name, (r, e) in arms.items():
            ca = r.alg
            t0 = time.time()
            with torch.inference_mode():
                ca.compute_returns(last_obs.cpu())
            ret_c, adv_c = ca.storage.returns.double().reshape(-1), ca.storage.advantages.double().reshape(-1)
            losses_c = [float(v) for v in ca.update(tables)]
            hl_c = ca.update_dagger(perm_dagger) if hist_encoding else None
            sc = state_groups(ca, e)
            d = rel_l2(sg, sc)
            d["returns"] = float((ret_g - ret_c).norm() / ret_c.norm())
            d["advantages"] = float((adv_g - adv_c).norm() / adv_c.norm())
            if name == "control":        # against the UNPERTURBED torch-CPU learner, not the GPU learner
                d = rel_l2(sc, forced_state)
                d["returns"] = float((ret_c - forced_ra[0]).norm() / forced_ra[0].norm()); d["advantages"] = float((adv_c - forced_ra[1]).norm() / forced_ra[1].norm())
            if name == "forced":
                forced_state, forced_ra = sc, (ret_c, adv_c)
            row[name] = {"state_rel_l2": d, "cpu_update_s": round(time.time() - t0, 2),
                         "losses_gpu": losses_g, "losses_cpu": losses_c,
                         "losses_max_abs_diff": max(abs(x - y) for x, y in zip(losses_g, losses_c)),
                         "hist_latent_loss": [hl_g, hl_c] if hist_encoding else None}
        if genv.task_obs_weight_decay_steps:
            genv.task_obs_weight = max(0, genv.task_obs_weight - 1.0 / genv.task_obs_weight_decay_steps)
            if getattr(genv, "task_obs_weight_dev", None) is not None:
                genv.task_obs_weight_dev.fill_(float(genv.task_obs_weight))
        rows.append(row)
        if (it + 1) in (1, 2, 3, 5, 10, 20, 50, 100, 150, 200, 300, 500, a.iters) or a.verbose:
            msg = {k: {g: f"{v:.1e}" for g, v in row[k]["state_rel_l2"].items() if g in ("actor_trunk", "critic_trunk", "disc_trunk", "adam_ac_m", "adam_ac_v", "normaliser_mean", "advantages", "std")}
                   for k in arms}
            print(f"it {it + 1} ({time.time() - t_start:.0f} s)", json.dumps(msg), flush=True)
        if a.out and ((it + 1) % 25 == 0 or it + 1 == a.iters):
            json.dump(summarise(a, rows, threads), open(a.out, "w"), indent=1)
    return summarise(a, rows, threads)


STATE_GROUPS_PARAMS = ("actor_trunk", "actor_head", "critic_trunk", "critic_head", "priv_encoder", "history_encoder", "estimator", "std",
                       "disc_trunk", "disc_head", "disc_encoder_eps", "disc_classifier")


def summarise(a, rows, threads):
    out = {"what": "learner lockstep: the product's GPU learner (recorded PPO / discriminator / DAgger steps, fused kernels) against the torch-CPU learner "
                   "(same class on device cpu) on the SAME state, rollout and sample tables; relative L2 difference |gpu - cpu| / |cpu| per group after every iteration",
           "config": {"amp": bool(a.amp), "num_envs": a.num_envs, "iters": len(rows), "seed": a.seed, "physics": a.physics, "driver": a.driver, "cpu_threads": threads,
                      "replay_ring": a.ring or "config default"},
           "arms": {"control": "torch-CPU learner vs torch-CPU learner: both from the GPU learner's state, one with every parameter moved by ~1 fp32 ulp (relative N(0, "
                               f"{a.control:g}^2)) -- what one iteration does to a rounding-sized input difference (the floor for any two arithmetics)",
                    "forced": "CPU learner re-synchronised to the GPU learner's state before every iteration: the difference ONE iteration makes",
                    "free": "CPU learner synchronised at iteration 0 only, same rollouts and tables: what the differences accumulate to"}}
    for name in ("forced", "free", "control"):
        if not rows or name not in rows[0]:
            continue
        groups = sorted({g for r in rows for g in r[name]["state_rel_l2"]})
        worst = {g: max((r[name]["state_rel_l2"].get(g, 0.0), r["iteration"]) for r in rows) for g in groups}
        med = {g: sorted(r[name]["state_rel_l2"].get(g, 0.0) for r in rows)[len(rows) // 2] for g in groups}
        at = {str(r["iteration"]): {g: r[name]["state_rel_l2"].get(g) for g in groups} for r in rows
              if r["iteration"] in (1, 2, 5, 10, 20, 50, 100, 150, 200, 300, 500, 1000, len(rows))}
        params = [g for g in groups if g in STATE_GROUPS_PARAMS]
        first_over = None
        for r in rows:
            bad = [g for g in params if r[name]["state_rel_l2"].get(g, 0.0) > a.bound]
            if bad:
                first_over = {"iteration": r["iteration"], "groups": bad, "values": {g: r[name]["state_rel_l2"][g] for g in bad}}
                break
        out[name] = {"worst_over_iterations": {g: {"rel_l2": v, "iteration": i} for g, (v, i) in worst.items()}, "median_over_iterations": med,
                     "at_iteration": at, "parameter_bound": a.bound, "first_iteration_with_a_parameter_group_over_the_bound": first_over,
                     "parameters_within_bound_for_all_iterations": first_over is None,
                     "logged_scalars_max_abs_diff": max(r[name]["losses_max_abs_diff"] for r in rows),
                     "cpu_update_s_median": sorted(r[name]["cpu_update_s"] for r in rows)[len(rows) // 2]}
    if rows and "control" in rows[0]:
        # per group and iteration: (GPU vs CPU) / (CPU vs perturbed CPU).  ~1 = the GPU learner differs from the CPU learner by what a one-ulp change of
        # the weights would do; >> 1 in a group = something other than rounding
        ratio = {}
        for g in sorted(rows[0]["forced"]["state_rel_l2"]):
            rs = sorted(r["forced"]["state_rel_l2"].get(g, 0.0) / r["control"]["state_rel_l2"][g] for r in rows if r["control"]["state_rel_l2"].get(g, 0.0) > 0)
            if rs:
                ratio[g] = {"median": rs[len(rs) // 2], "p90": rs[int(0.9 * (len(rs) - 1))], "max": rs[-1], "n": len(rs)}
        out["gpu_vs_cpu_over_control"] = ratio
        out["lr_ac_differs_at_iterations"] = [r["iteration"] for r in rows if r["forced"]["state_rel_l2"].get("lr_ac", 0.0) > 1e-6]
        out["control_lr_ac_differs_at_iterations"] = [r["iteration"] for r in rows if r["control"]["state_rel_l2"].get("lr_ac", 0.0) > 1e-6]
    out["gpu_update_s_median"] = sorted(r["gpu_update_s"] for r in rows)[len(rows) // 2] if rows else None
    out["gpu_path_last"] = rows[-1]["gpu_path"] if rows else None
    if a.keep_rows:
        out["rows"] = rows
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--num_envs", type=int, default=1024)
    ap.add_argument("--iters", type=int, default=200)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--amp", action="store_true", help="BASELINE config 3 (discriminator on, mocap resets); without: config 2")
    ap.add_argument("--physics", choices=["oracle", "hip"], default="oracle", help="who produces the rollouts: the oracle's physics under the GPU learner (hybrid arm) or the HIP env")
    ap.add_argument("--driver", choices=["gpu", "cpu"], default="gpu", help="cpu: the driving learner is a torch-CPU learner too (self-test of this tool: every difference must be exactly 0)")
    ap.add_argument("--free", type=int, default=1, help="also run the never-resynchronised CPU learner")
    ap.add_argument("--control", type=float, default=0.0, help="> 0: the sensitivity control arm, relative size of its parameter perturbation (1e-7 ~ one fp32 ulp)")
    ap.add_argument("--ring", type=int, default=0, help="replay ring size (0: the config's)")
    ap.add_argument("--bound", type=float, default=1e-5, help="the state agreement asked for (relative L2 of a parameter group)")
    ap.add_argument("--keep_rows", type=int, default=0)
    ap.add_argument("--verbose", type=int, default=0)
    ap.add_argument("--out", type=str, default=None)
    a = ap.parse_args()
    res = run(a)
    if a.out:
        json.dump(res, open(a.out, "w"), indent=1)
    brief = {k: {"within_bound": res[k]["parameters_within_bound_for_all_iterations"], "first_over": res[k]["first_iteration_with_a_parameter_group_over_the_bound"],
                 "worst_param": max(((v["rel_l2"], g) for g, v in res[k]["worst_over_iterations"].items() if g in STATE_GROUPS_PARAMS), default=None)}
             for k in ("forced", "free", "control") if k in res}
    print(json.dumps(brief))


if __name__ == "__main__":
    main()
