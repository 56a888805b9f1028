"""The task-level course generator (quadrupedal_agility_amd/tsc/legged_gym/utils/obstacle.py) against arrays produced by the
REFERENCE'S OWN `Obstacle` class (tools/gen_golden_tsc_obstacle.py; tsc/legged_gym/utils/obstacle.py:75-203, 235-517): same
seeds -> the same int16 height map, edge mask, goals, obstacle types / origins / yaws / joint positions, bit for bit."""
import os

import numpy as np
import pytest

from quadrupedal_agility_amd.tsc.legged_gym.utils.obstacle import Obstacle, fill_polygon

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "tsc_obstacle.npz")


def _cfg(curriculum):
    from quadrupedal_agility_amd.tsc.legged_gym.envs.base.legged_robot_config import LeggedRobotCfg
    c = LeggedRobotCfg.obstacle()
    c.curriculum = bool(curriculum)
    return c


@pytest.mark.parametrize("case", [0, 1])
def test_course_matches_reference(case):
    g = np.load(GOLD)
    n, seed = int(g[f"c{case}_n"]), int(g[f"c{case}_seed"])
    ob = Obstacle(_cfg(g[f"c{case}_curriculum"]), n, seed=seed)
    for name in ("obstacle_types", "bar_jump_mask", "tire_jump_mask", "x_edge_mask", "height_field_raw"):
        assert np.array_equal(getattr(ob, name), g[f"c{case}_{name}"]), name
    for name in ("env_goals", "obstacle_origins", "obstacle_yaws", "obstacle_joint_pos", "env_origins"):
        assert np.allclose(getattr(ob, name), g[f"c{case}_{name}"], rtol=0, atol=1e-12), name
    # the course is what SURVEY 8a/8f describe: six different obstacles per env, 2 m walls around the tile, movable parts at their height
    assert all(sorted(row) == list(range(6)) for row in ob.obstacle_types)
    assert ob.height_field_raw.max() == int(2.0 / 0.005) and ob.height_field_raw.min() >= -80
    assert ob.flat_goals().shape == (n, 6 * 4 + 2, 3)


def test_module_level_generators_like_the_reference():
    """seed=None draws from `random` / `numpy.random`, as the reference class does"""
    import random
    random.seed(3); np.random.seed(3)
    a = Obstacle(_cfg(False), 2)
    b = Obstacle(_cfg(False), 2, seed=3)
    assert np.array_equal(a.height_field_raw, b.height_field_raw) and np.array_equal(a.env_goals, b.env_goals)


def test_fill_polygon_contract():
    rr, cc = fill_polygon([1, 1, 4, 4], [2, 6, 6, 2], (10, 10))                     # axis-aligned: closed rectangle
    img = np.zeros((10, 10), int); img[rr, cc] = 1
    assert img.sum() == 4 * 5 and img[1:5, 2:7].all()
    rr, cc = fill_polygon([0, 0, 20, 20], [-5, 3, 3, -5], (8, 8))                   # clipped to the image
    assert rr.max() == 7 and cc.min() == 0 and cc.max() == 3
    rr, cc = fill_polygon([2.5, 2.5, 2.9, 2.9], [2.2, 2.8, 2.8, 2.2], (8, 8))       # no pixel centre inside
    assert len(rr) == 0
    th = np.radians(30.0)                                                          # rotated square: area ~ side^2
    R = np.array([[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]])
    p = (R @ (np.array([[-10, -10], [10, -10], [10, 10], [-10, 10]]).T)).T + 30
    rr, cc = fill_polygon(p[:, 0], p[:, 1], (64, 64))
    assert abs(len(rr) - 400) < 25


@pytest.mark.parametrize("curriculum", [False, True])
def test_shards_build_the_same_course_as_the_one_process_job(curriculum):
    """SURVEY 8e for the task-level tree: rank r of a W-rank job builds envs [r N/W, (r+1) N/W) of the job's ONE course -- the reference's
    sequential stream (tsc/legged_gym/utils/obstacle.py:75-203), with the draws of the envs before the shard consumed, not re-keyed -- so
    an 8-env job has the same obstacles on 1, 2 and 4 ranks (r2: every rank seeded its own stream, seed + 7919 rank)"""
    cfg = _cfg(curriculum)
    whole = Obstacle(cfg, 8, seed=5)
    L, Wd = whole.length_per_env_pixels, whole.width_per_env_pixels

    def tile(ob, i, field):
        sx = int(ob.border + ob.env_origins[i, 0] / ob.horizontal_scale); sy = int(ob.border + ob.env_origins[i, 1] / ob.horizontal_scale)
        return getattr(ob, field)[sx:sx + L, sy:sy + Wd]

    for world in (2, 4):
        n = 8 // world
        for r in range(world):
            part = Obstacle(cfg, n, seed=5, skip_envs=r * n)
            for i in range(n):
                gi = r * n + i
                assert np.array_equal(part.obstacle_types[i], whole.obstacle_types[gi])
                assert np.array_equal(part.obstacle_yaws[i], whole.obstacle_yaws[gi]) and np.array_equal(part.obstacle_joint_pos[i], whole.obstacle_joint_pos[gi])
                for f in ("height_field_raw", "x_edge_mask", "ceiling_raw", "bar_jump_mask", "tire_jump_mask"):
                    assert np.array_equal(tile(part, i, f), tile(whole, gi, f)), (world, r, i, f)
                # goals and obstacle origins relative to the env's own origin (the tiles sit elsewhere in a smaller grid)
                np.testing.assert_allclose(part.env_goals[i] - part.env_origins[i], whole.env_goals[gi] - whole.env_origins[gi], rtol=0, atol=1e-9)
                np.testing.assert_allclose(part.obstacle_origins[i] - part.env_origins[i], whole.obstacle_origins[gi] - whole.env_origins[gi], rtol=0, atol=1e-9)
    assert not np.array_equal(Obstacle(cfg, 4, seed=5, skip_envs=4).obstacle_types, Obstacle(cfg, 4, seed=5).obstacle_types)      # the shards do differ
