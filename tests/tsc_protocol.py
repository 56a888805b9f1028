"""Deterministic driver shared by tools/gen_golden_tsc.py (runs it on the REFERENCE's tsc/rsl_rl classes) and
tests/test_tsc_learner.py (runs it on quadrupedal_agility_amd.tsc.rsl_rl): weights and inputs come from an integer
hash, so no state_dict has to be stored; the sampled actions come from torch's CPU generator under a fixed seed, which
both runs consume in the same order."""
import numpy as np
import torch

DIMS = dict(n_proprio=65, n_auxiliary=8, n_scan=132, n_priv=4, n_priv_latent=29, history_len=10, num_obs=800, num_actions_d=3,
            num_actions_c=6, num_obs_bbc=57 + 29 + 4 + 6 + 5, num_actions_bbc=12, num_command=9)
POLICY = dict(init_noise_std=1.0, scan_encoder_dims=[128, 64, 32], actor_hidden_dims=[512, 256, 128], critic_hidden_dims=[512, 256, 128],
              priv_encoder_dims=[64], activation="elu", tanh_encoder_output=False)
ESTIMATOR = dict(train_with_estimated_states=True, learning_rate=1.e-4, hidden_dims=[128, 64], priv_states_dim=4, num_prop=57,
                 num_auxiliary=8, num_scan=132)
ALGO = dict(value_loss_coef=1.0, use_clipped_value_loss=True, clip_param=0.2, entropy_coef=0.01, num_learning_epochs=2,
            num_mini_batches=2, learning_rate=5.e-4, schedule="adaptive", gamma=0.99, lam=0.95, desired_kl=0.01, max_grad_norm=1.0,
            dagger_update_freq=20, priv_reg_coef_schedual=[0, 0.1, 0, 2])
T, N = 6, 48


def det(shape, key, scale=1.0):
    """uniform(-scale, scale) from a 64-bit integer hash of (key, index): exact on every machine"""
    n = int(np.prod(shape))
    with np.errstate(over="ignore"):
        h = np.arange(n, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15) + np.uint64(key) * np.uint64(0xBF58476D1CE4E5B9)
        h ^= h >> np.uint64(30); h *= np.uint64(0xBF58476D1CE4E5B9)
        h ^= h >> np.uint64(27); h *= np.uint64(0x94D049BB133111EB)
        h ^= h >> np.uint64(31)
    u = (h >> np.uint64(40)).astype(np.float64) / float(1 << 24)
    return torch.tensor(((2.0 * u - 1.0) * scale).reshape(shape), dtype=torch.float32)


def fill(module, key):
    """every parameter := hash-uniform with the fan-in scaling of nn.Linear's default init (names sorted)"""
    with torch.no_grad():
        for i, (name, p) in enumerate(sorted(module.named_parameters())):
            if name.endswith("std"):
                p.fill_(0.8)
                continue
            fan_in = p[0].numel() if p.dim() > 1 else p.numel()
            p.copy_(det(tuple(p.shape), key * 1000 + i, 1.0 / np.sqrt(max(fan_in, 1))))


def build(mods, algs):
    """mods: namespace with ActorCriticTSC / ActorCriticBBC / Estimator; algs: namespace with PPO"""
    d = DIMS
    ac = mods.ActorCriticTSC(d["n_proprio"], d["n_auxiliary"], d["n_scan"], d["num_obs"], d["n_priv_latent"], d["n_priv"], d["history_len"],
                             d["num_actions_d"], d["num_actions_c"], device="cpu", **POLICY)
    n_prop = d["n_proprio"] - d["n_auxiliary"]
    bbc = mods.ActorCriticBBC(d["num_obs_bbc"], d["num_obs_bbc"] + d["history_len"] * n_prop, d["num_actions_bbc"], d["n_proprio"],
                              d["n_auxiliary"], d["history_len"], d["n_priv"], d["n_priv_latent"], d["num_command"], **POLICY)
    est = mods.Estimator(input_dim=n_prop, output_dim=d["n_priv"], hidden_dims=ESTIMATOR["hidden_dims"])
    fill(ac, 1); fill(bbc, 2); fill(est, 3)
    alg = algs.PPO(ac, bbc, est, ESTIMATOR, None, None, None, device="cpu", **ALGO)
    alg.init_storage(N, T, [d["num_obs"]], [d["num_obs"]], [1 + d["num_actions_d"] * d["num_actions_c"]])
    return ac, bbc, est, alg


def param_probe(module):
    """(num_tensors, 6): a few entries + sum + |sum| of every parameter tensor (names sorted) -- pins a whole update cheaply"""
    rows = []
    for _, p in sorted(module.named_parameters()):
        f = p.detach().double().flatten()
        rows.append([f[0].item(), f[f.numel() // 2].item(), f[-1].item(), f.sum().item(), f.abs().sum().item(), float(f.numel())])
    return np.asarray(rows)


def run(mods, algs):
    out = {}
    ac, bbc, est, alg = build(mods, algs)
    d = DIMS
    obs = det((N, d["num_obs"]), 11)
    with torch.no_grad():
        out["act_inference_priv"] = ac.act_inference(obs, hist_encoding=False)
        out["act_inference_hist"] = ac.act_inference(obs, hist_encoding=True)
        out["evaluate"] = ac.evaluate(obs)
        out["priv_latent"] = ac.actor.infer_priv_latent(obs)
        out["hist_latent"] = ac.actor.infer_hist_latent(obs)
        out["scan_latent"] = ac.actor.infer_scandots_latent(obs)
        out["estimator"] = est(obs[:, :57])
        obs_bbc = det((N, d["num_obs_bbc"] + 570), 12)      # prop | explicit | latent | history | command
        out["bbc_act_inference"] = bbc.act_inference(obs_bbc, hist_encoding=True)
        out["bbc_act_inference_priv"] = bbc.act_inference(obs_bbc, hist_encoding=False)
        out["bbc_evaluate"] = bbc.evaluate(obs_bbc)
        out["act_bbc"] = alg.act_bbc(obs_bbc)

    for phase, hist in (("rl", False), ("dagger", True)):
        torch.manual_seed(1234 + hist)
        acts, vals, lpd, lpc = [], [], [], []
        for t in range(T):
            o = det((N, d["num_obs"]), 100 + 50 * hist + t)
            a = alg.act(o, o, None, hist_encoding=hist)
            tr = alg.transition
            acts.append(a.clone()); vals.append(tr.values.clone()); lpd.append(tr.actions_log_prob_d.clone()); lpc.append(tr.actions_log_prob_c.clone())
            rew = det((N,), 200 + 50 * hist + t)
            dones = det((N,), 300 + 50 * hist + t) > 0.8
            touts = (det((N,), 400 + 50 * hist + t) > 0.9) & dones
            alg.process_env_step(rew, dones, {"time_outs": touts})
        out[f"{phase}_actions"], out[f"{phase}_values"] = torch.stack(acts), torch.stack(vals)
        out[f"{phase}_logp_d"], out[f"{phase}_logp_c"] = torch.stack(lpd), torch.stack(lpc)
        alg.compute_returns(det((N, d["num_obs"]), 500 + hist))
        out[f"{phase}_rewards"] = alg.storage.rewards.clone()
        out[f"{phase}_returns"], out[f"{phase}_advantages"] = alg.storage.returns.clone(), alg.storage.advantages.clone()
        if not hist:
            out["update"] = np.asarray(alg.update(), dtype=np.float64)
            out["lr_after_update"] = np.float64(alg.learning_rate)
            out["probe_ac_after_update"], out["probe_est_after_update"] = param_probe(ac), param_probe(est)
        else:
            out["update_dagger"] = np.float64(alg.update_dagger())
            out["probe_ac_after_dagger"] = param_probe(ac)
    out["counter"] = np.int64(alg.counter)
    return {k: (v.detach().numpy() if torch.is_tensor(v) else v) for k, v in out.items()}
