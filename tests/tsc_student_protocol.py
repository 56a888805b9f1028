"""Deterministic driver for the vision student's learner pieces, shared by tools/gen_golden_tsc_student.py (runs it on the REFERENCE's
tsc/rsl_rl classes: DepthOnlyFCBackbone58x87, RecurrentDepthBackbone, BYOL, PPO.update_depth_actor) and tests/test_tsc_student.py
(runs it on quadrupedal_agility_amd.tsc.rsl_rl).  Weights and inputs come from the integer hash of tests/tsc_protocol.py, the BYOL
augmentations (Python `random` + torchvision in the reference) are replaced by the identity on both sides, and the one random draw left
in update_depth_actor (the permutation of the depth images) comes from torch's CPU generator under a fixed seed.

`ns` is a namespace with DepthOnlyFCBackbone58x87, RecurrentDepthBackbone, Actor / ActorCriticTSC / ActorCriticBBC / Estimator and PPO."""
import copy
from types import SimpleNamespace

import numpy as np
import torch

from tests import tsc_protocol as P

B = 12                      # images per step (BYOL minibatches of B * STEPS // 6 need > 1 sample for the projector's BatchNorm)
STEPS = 3
N_LATENT, N_YAW, N_TYPE = 32, 2, 6
DEPTH = dict(learning_rate=1.e-3, learning_rate_byol=3.e-4)


def env_cfg():
    return SimpleNamespace(env=SimpleNamespace(n_delta_yaw=N_YAW, n_obst_type=N_TYPE, n_proprio=P.DIMS["n_proprio"]), depth=SimpleNamespace(buffer_len=2))


def images(key, n=B):
    """depth images in the env's range (-0.5, 0.5) with structure (a ramp + hash noise)"""
    ramp = torch.linspace(-0.4, 0.4, 58).view(1, 58, 1) * torch.linspace(0.5, 1.0, 87).view(1, 1, 87)
    return (ramp + P.det((n, 58, 87), key, 0.1)).clamp(-0.5, 0.5)


def build(ns, device="cpu"):
    torch.manual_seed(0)
    backbone = ns.DepthOnlyFCBackbone58x87(P.DIMS["n_proprio"], N_LATENT, 512)
    enc = ns.RecurrentDepthBackbone(backbone, N_LATENT, env_cfg())          # its BYOL head creates projector + target encoder on a mock batch
    ident = torch.nn.Identity()
    enc.byol_learner.augment1 = enc.byol_learner.augment2 = ident
    P.fill(enc, 7)                                                          # every parameter (online, target, GRU, heads), names sorted
    ac, bbc, est, _ = P.build(ns, SimpleNamespace(PPO=lambda *a, **k: SimpleNamespace(init_storage=lambda *a, **k: None)))
    depth_actor = copy.deepcopy(ac.actor)
    for m in (enc, depth_actor, ac, bbc, est):          # created and filled on the host (same order on both sides), then moved
        m.to(device)
    alg = ns.PPO(ac, bbc, est, P.ESTIMATOR, enc, dict(DEPTH), depth_actor, device=device, **P.ALGO)
    return enc, depth_actor, alg


def run(ns, device="cpu"):
    out = {}
    enc, depth_actor, alg = build(ns, device)
    det = lambda *a, **k: P.det(*a, **k).to(device)
    img = lambda *a, **k: images(*a, **k).to(device)
    d = P.DIMS
    enc.eval()                                                               # BatchNorm of the BYOL heads: running statistics in the probes below
    with torch.no_grad():
        out["backbone_latent"] = enc.base_backbone(img(21))
        enc.hidden_states = None
        prop = det((B, d["n_proprio"]), 22)
        out["encoder_step0"] = enc(img(21), prop)
        out["encoder_step1"] = enc(img(23), det((B, d["n_proprio"]), 24))      # the GRU state carries over
        out["hidden_after_2"] = enc.hidden_states.clone()
        out["byol_embedding"] = enc.byol_learner(img(25), return_embedding=True, return_projection=False)
    enc.train()
    out["byol_loss"] = enc.byol_learner(img(26)).detach()
    # two BYOL optimiser steps on fixed minibatches with the EMA target update after each, straight from the filled weights: identical
    # inputs on both sides, so this pins the optimiser wiring (byol_optimizer: Adam over byol_learner.parameters() at learning_rate_byol)
    # and the EMA decay tightly.  (Inside update_depth_actor below the same steps come after the DAgger step, whose fp32 rounding differs
    # between two equivalent evaluations of the student actor; the projector's pre-BatchNorm biases have an analytically ZERO gradient,
    # Adam normalises their rounding noise to +-lr, and the BYOL-trained tensors then differ in the third digit -- noise, not arithmetic.)
    for k in range(2):
        loss = enc.byol_learner(img(27 + k))
        alg.byol_optimizer.zero_grad()
        loss.backward()
        alg.byol_optimizer.step()
        enc.byol_learner.update_moving_average()
    out["probe_byol_after_2_steps"] = P.param_probe(enc.byol_learner)
    # one learn_vision-style update: STEPS encoder steps with gradients, the student actor on the teacher's observation row with the
    # depth latent as scan latent, then PPO.update_depth_actor (one Adam step over actor + encoder, 6 BYOL minibatches, EMA updates)
    enc.hidden_states = None
    ims, act_s, act_t, yaw_s, yaw_t, typ_s, typ_t = [], [], [], [], [], [], []
    nd = d["num_actions_d"]
    for t in range(STEPS):
        im = img(30 + t)
        obs = det((B, d["num_obs"]), 40 + t)
        o = enc(im, obs[:, :d["n_proprio"]])
        latent, yaw, typ = o[:, :N_LATENT], 1.5 * o[:, N_LATENT:N_LATENT + N_YAW], o[:, N_LATENT + N_YAW:]
        emb = depth_actor(obs, hist_encoding=True, scandots_latent=latent)
        act_s.append(torch.cat([depth_actor.actor_d(emb), depth_actor.actor_c(emb)], dim=-1))
        with torch.no_grad():
            act_t.append(alg.actor_critic.act_inference(obs, hist_encoding=True))
        ims.append(im); yaw_s.append(yaw); typ_s.append(typ)
        yaw_t.append(det((B, N_YAW), 50 + t, 0.6))
        typ_t.append(torch.nn.functional.one_hot(torch.arange(B) % N_TYPE, N_TYPE).float().to(device))
    torch.manual_seed(4321)                                                  # the permutation of the depth images
    losses = alg.update_depth_actor(torch.cat(act_s), torch.cat(act_t), torch.cat(yaw_s), torch.cat(yaw_t), torch.cat(typ_s), torch.cat(typ_t), torch.cat(ims))
    out["update_depth_actor"] = np.asarray(losses, dtype=np.float64)
    out["student_actions"] = torch.cat(act_s).detach()
    out["probe_encoder_after"] = P.param_probe(enc)
    out["probe_actor_after"] = P.param_probe(depth_actor)
    return {k: (v.detach().cpu().numpy() if torch.is_tensor(v) else v) for k, v in out.items()}
