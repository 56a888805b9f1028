"""Command line / registry layer of the task-level tree (tsc/legged_gym/utils/{helpers,task_registry}.py, scripts/train.py): the
reference's flags reach the configs, `--use_camera` switches env and runner to the depth student, a run resumes from its own
checkpoint directory by `--resumeid`.  CPU: the oracle twins under the same host code."""
import os

import pytest
import torch

from quadrupedal_agility_amd.tsc.legged_gym.envs import task_registry
from quadrupedal_agility_amd.tsc.legged_gym.envs.go2.go2_agility_config import Go2AgilityCfg, Go2AgilityCfgPPO
from quadrupedal_agility_amd.tsc.legged_gym.utils import helpers, task_registry as tr_mod
from quadrupedal_agility_amd.tsc.legged_gym.utils.helpers import get_args, update_cfg_from_args


def test_flags_reach_the_configs():
    a = get_args(["--task", "go2", "--headless", "--use_camera", "--randomize_base_mass", "--push_robots", "--randomize_start", "--seed", "7",
                  "--tracking_yaw", "3.5", "--target_lin_vel", "1.2", "--max_iterations", "11", "--reward_i_coef", "0.3", "--device", "cpu"])
    env_cfg, train_cfg = update_cfg_from_args(Go2AgilityCfg(), Go2AgilityCfgPPO(), a)
    assert env_cfg.depth.use_camera and env_cfg.env.num_envs == env_cfg.depth.camera_num_envs == 256 and train_cfg.depth_encoder.if_depth
    d = env_cfg.domain_rand
    assert d.randomize_base_mass and d.push_robots and not d.randomize_base_com and env_cfg.obstacle.randomize_start and not env_cfg.obstacle.curriculum
    assert env_cfg.seed == train_cfg.seed == 7 and env_cfg.rewards.scales.tracking_yaw == 3.5 and env_cfg.rewards.target_lin_vel == 1.2
    assert train_cfg.runner.max_iterations == 11 and train_cfg.runner.reward_i_coef == 0.3 and a.rl_device == a.sim_device == "cpu"
    b = get_args(["--resume", "--num_envs", "48", "--device", "cpu"])
    env_cfg, train_cfg = update_cfg_from_args(Go2AgilityCfg(), Go2AgilityCfgPPO(), b)
    assert env_cfg.env.num_envs == 48 and train_cfg.runner.resume and not train_cfg.estimator.load_estimator_bbc
    assert train_cfg.algorithm.priv_reg_coef_schedual == train_cfg.algorithm.priv_reg_coef_schedual_resume
    assert "go2" in task_registry.task_classes


def test_make_env_runner_train_and_resume(tmp_path, monkeypatch):
    from tests.oracle_backend import OracleBackend
    from tests.oracle_lib import load_oracle
    from quadrupedal_agility_amd.tsc.legged_gym.envs.base import legged_robot as lr
    from quadrupedal_agility_amd.tsc.legged_gym.utils.obstacle import Obstacle
    monkeypatch.setattr(tr_mod, "LEGGED_GYM_ROOT_DIR", str(tmp_path))
    argv = ["--task", "go2", "--headless", "--device", "cpu", "--num_envs", "6", "--seed", "2", "--max_iterations", "1", "--exptid", "abc123-first"]

    def build(argv):
        args = get_args(argv)
        cfg = Go2AgilityCfg(); cfg.env.episode_length_s = 0.5
        cfg, _ = update_cfg_from_args(cfg, None, args)
        ob = Obstacle(cfg.obstacle, cfg.env.num_envs, seed=cfg.seed)
        env, cfg = task_registry.make_env("go2", args=args, env_cfg=cfg, backend=OracleBackend(lr.make_qa_config(cfg, ob, seed=cfg.seed)),
                                          bookkeeping_lib=(load_oracle(), "qo_"))
        tc = Go2AgilityCfgPPO(); tc.runner.num_steps_per_env = 4
        runner, tc = task_registry.make_alg_runner(env, name=None, args=args, train_cfg=tc)
        return runner, tc
    runner, tc = build(argv)
    assert runner.log_dir == os.path.join(str(tmp_path), "logs", "agility", "abc123-first")
    runner.learn(tc.runner.max_iterations, init_at_random_ep_len=True)
    assert os.path.exists(os.path.join(runner.log_dir, "model.pt"))
    w = {k: v.clone() for k, v in runner.alg.actor_critic.state_dict().items()}
    # a second run resumes from the first by its 6-character prefix (:70-96)
    runner2, _ = build(argv[:-1] + ["xyz789-second", "--resumeid", "abc123"])
    assert all(torch.equal(v, runner2.alg.actor_critic.state_dict()[k]) for k, v in w.items())
    assert runner2.current_learning_iteration == 1


def test_play_resumes_a_run_and_drives_the_course(tmp_path, monkeypatch):
    """scripts/play.py headless: the reference's test-time overrides, the runner resumed from the run directory, teacher and student
    deployment loops; returns the success rate of the episodes that ended"""
    from tests.oracle_backend import OracleBackend
    from tests.oracle_lib import load_oracle
    from quadrupedal_agility_amd.tsc.legged_gym.scripts.play import play
    from quadrupedal_agility_amd.tsc.legged_gym.scripts.train import train
    monkeypatch.setattr(tr_mod, "LEGGED_GYM_ROOT_DIR", str(tmp_path))
    kw = dict(backend=OracleBackend, bookkeeping_lib=(load_oracle(), "qo_"))
    real_make_env = task_registry.make_env
    monkeypatch.setattr(task_registry, "make_env", lambda name, args=None, env_cfg=None, **k: real_make_env(name, args=args, env_cfg=env_cfg, **kw))
    for cam in ([], ["--use_camera"]):
        exptid = "cam001-x" if cam else "tea001-x"
        targs = get_args(["--task", "go2", "--device", "cpu", "--num_envs", "6", "--max_iterations", "1", "--exptid", exptid] + cam)
        r = train(targs)
        assert r.if_depth == bool(cam)
        pargs = get_args(["--task", "go2", "--device", "cpu", "--num_envs", "6", "--exptid", exptid, "--resumeid", exptid] + cam)
        sr = play(pargs, num_steps=6, quiet=True)
        assert sr is None or 0.0 <= sr <= 1.0
