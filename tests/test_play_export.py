"""the deployment export (TorchScript) and the play loop."""
import os

import numpy as np
import pytest
import torch

from quadrupedal_agility_amd.legged_gym.utils import Logger, export_policy_as_jit
from quadrupedal_agility_amd.rsl_rl.modules import ActorCritic


def _ac(seed=0):
    torch.manual_seed(seed)
    return ActorCritic(num_actor_obs=101, num_critic_obs=671, num_actions=12, num_prop=57, num_hist=10, num_explicit=4,
                       num_latent=29, num_command=11, actor_hidden_dims=[512, 256, 128], critic_hidden_dims=[512, 256, 128],
                       priv_encoder_dims=[64], activation="elu", train_with_estimated_latent=True)


def test_exported_policy_matches_act_inference(tmp_path):
    ac = _ac().eval()
    path = export_policy_as_jit(ac, str(tmp_path / "exported" / "policies"))
    assert os.path.basename(path) == "policy_1.pt"
    pol = torch.jit.load(path)
    obs = torch.randn(33, 671)
    with torch.no_grad():
        want = ac.act_inference(obs, hist_encoding=True)
        got = pol(obs)
        one = pol(obs[:1])                                 # batch 1, the deployed case
    assert got.shape == (33, 12)
    assert torch.allclose(got, want, atol=2e-6, rtol=1e-5)
    assert torch.allclose(one, want[:1], atol=2e-6, rtol=1e-5)
    # the export is a copy: training the live model afterwards does not change the file's behaviour
    with torch.no_grad():
        for p in ac.parameters():
            p.add_(0.1)
        assert torch.allclose(pol(obs), want, atol=2e-6, rtol=1e-5)


def test_logger_reward_bookkeeping(tmp_path, capsys):
    lg = Logger(0.02)
    lg.log_rewards({"rew_tracking_lin_vel": torch.tensor(0.5), "rew_torques": torch.tensor(-0.1), "other": torch.tensor(9.0)}, 2)
    lg.log_rewards({"rew_tracking_lin_vel": torch.tensor(1.0), "rew_torques": torch.tensor(-0.3)}, 1)
    m = lg.mean_rewards()
    assert lg.num_episodes == 3 and set(m) == {"rew_tracking_lin_vel", "rew_torques"}
    assert m["rew_tracking_lin_vel"] == pytest.approx((0.5 * 2 + 1.0) / 3) and m["rew_torques"] == pytest.approx((-0.2 - 0.3) / 3)
    for i in range(5):
        lg.log_states({"dof_pos": np.full(12, i, np.float32), "base_vel_x": 0.1 * i})
    s = lg.series()
    assert s["dof_pos"].shape == (5, 12) and s["base_vel_x"].shape == (5,)
    lg.print_rewards()
    assert "Total number of episodes: 3" in capsys.readouterr().out
    lg.reset()
    assert not lg.series()


@pytest.mark.gpu
def test_play_resumes_checkpoint_and_walks(tmp_path):
    """train 2 iterations -> model.pt -> play.py path: overrides, resume, inference policy, fixed command, export"""
    from quadrupedal_agility_amd.legged_gym.envs import task_registry
    from quadrupedal_agility_amd.legged_gym.envs.go2.go2_locomotion_config import Go2LocomotionCfg, Go2LocomotionCfgAlgo
    from quadrupedal_agility_amd.legged_gym.scripts.play import play
    from quadrupedal_agility_amd.legged_gym.utils import get_args
    cfg = Go2LocomotionCfg(); cfg.env.num_envs = 256; cfg.terrain.mesh_type = "plane"; cfg.env.mocap_state_init = False
    t = Go2LocomotionCfgAlgo(); t.runner.amp_enabled = False
    args = get_args(["--device", "gpu", "--terrain", "plane"])
    env, _ = task_registry.make_env("go2_locomotion", args=args, env_cfg=cfg)
    runner, _ = task_registry.make_alg_runner(env, name="go2_locomotion", args=args, train_cfg=t, log_root=str(tmp_path))
    runner.learn(2, init_at_random_ep_len=True)
    trained = {k: v.clone() for k, v in runner.alg.actor_critic.state_dict().items()}
    del env, runner
    env, runner, logger, exported = play(args, num_steps=60, export_policy=True, log_root=str(tmp_path), stop_state_log=40)
    assert env.num_envs == 16 and env.cfg.env.episode_length_s == 500.0 and env.sim.cfg.push_robots == 0
    for k, v in runner.alg.actor_critic.state_dict().items():
        assert torch.equal(v, trained[k]), k
    assert (env.latent_c.argmax(1) == 2).all() and torch.allclose(env.commands[:, 0], torch.full((16,), 2.0, device="cuda"))
    assert logger.series()["dof_pos"].shape == (40, 12)
    pol = torch.jit.load(exported)
    obs = env.get_observations()
    with torch.no_grad():
        assert torch.allclose(pol(obs.cpu()), runner.alg.actor_critic.act_inference(obs, hist_encoding=True).cpu(), atol=1e-4, rtol=1e-4)
