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


def _skimage_polygon_rule(rows, cols, shape):
    """skimage.draw.polygon as scikit-image 0.21.0 (the reference's pin, requirements.txt:5; NOT installed here) publishes it, restated:
    `_polygon` (skimage/draw/_draw.pyx) visits the pixels of the box [max(0, min), ceil(max)] clipped to `shape` and keeps those for
    which `point_in_polygon` (skimage/_shared/geometry.pxd) is non-zero.  That test is O'Rourke's InPoly1 (Computational Geometry in C,
    1998, 7.4): translate the polygon so that the query is the origin; a vertex within 1e-12 -> VERTEX; count the edges that cross the
    positive x half-axis (strict straddle on y > 0) and those that cross the negative one (strict straddle on y < 0); different parities
    -> EDGE, odd -> INSIDE, else OUTSIDE.  Vertex, edge and inside pixels are all drawn.  Scalar loops: an independent restatement, not
    the vectorised crossing-number + on-segment test `fill_polygon` uses."""
    xp, yp = [float(v) for v in cols], [float(v) for v in rows]
    r0, r1 = int(max(0, min(yp))), int(np.ceil(max(yp)))
    c0, c1 = int(max(0, min(xp))), int(np.ceil(max(xp)))
    r1, c1 = min(r1, shape[0] - 1), min(c1, shape[1] - 1)
    out = []
    n = len(xp)
    for r in range(r0, r1 + 1):
        for c in range(c0, c1 + 1):
            x1, y1 = xp[n - 1] - c, yp[n - 1] - r
            rc = lc = 0
            code = None
            for i in range(n):
                x0, y0 = xp[i] - c, yp[i] - r
                if -1e-12 < x0 < 1e-12 and -1e-12 < y0 < 1e-12:
                    code = 2
                    break
                if (y0 > 0) != (y1 > 0) and (x0 * y1 - x1 * y0) / (y1 - y0) > 0:
                    rc += 1
                if (y0 < 0) != (y1 < 0) and (x0 * y1 - x1 * y0) / (y1 - y0) < 0:
                    lc += 1
                x1, y1 = x0, y0
            if code is None:
                code = 3 if (rc & 1) != (lc & 1) else (1 if rc & 1 else 0)
            if code:
                out.append((r, c))
    return set(out)


def test_fill_polygon_is_the_published_skimage_rule():
    """the rasterisation rule behind the course (tsc/legged_gym/utils/obstacle.py:151,166 call skimage.draw.polygon): the build's scan
    fill against the restated library rule on (a) integer rectangles and triangles whose boundary passes through pixel centres,
    (b) random simple (convex and star-shaped concave) polygons with integer and fractional vertices, (c) EVERY polygon the course generator draws"""
    rng = np.random.default_rng(0)
    cases = [([1, 2, 8], [1, 7, 4]), ([1, 1, 4, 4], [2, 6, 6, 2]), ([0, 0, 20, 20], [-5, 3, 3, -5]), ([3, 3, 9, 9, 6], [2, 9, 9, 2, 5]),
             ([2.5, 2.5, 2.9, 2.9], [2.2, 2.8, 2.8, 2.2]), ([0, 5, 10, 5], [5, 0, 5, 10])]
    for _ in range(80):                         # SIMPLE polygons (what the generator draws): vertices ordered by angle about an interior point
        k = int(rng.integers(3, 8))
        ang = np.sort(rng.random(k) * 2 * np.pi)
        if np.max(np.diff(np.concatenate([ang, ang[:1] + 2 * np.pi]))) >= np.pi:
            continue                            # the centre must be inside
        rad = rng.random(k) * 11 + 2
        pr, pc = 12 + rad * np.sin(ang), 13 + rad * np.cos(ang)
        if rng.random() < 0.5:
            pr, pc = np.round(pr), np.round(pc)     # integer vertices: boundaries through pixel centres
        cases.append((pr.tolist(), pc.tolist()))
    for rows, cols in cases:
        rr, cc = fill_polygon(rows, cols, (26, 28))
        assert set(zip(rr.tolist(), cc.tolist())) == _skimage_polygon_rule(rows, cols, (26, 28)), (rows, cols)
    # (c) the generator's own polygons (rotated bars, planks, poles ...)
    import quadrupedal_agility_amd.tsc.legged_gym.utils.obstacle as ob
    seen = []
    real = ob.fill_polygon

    def spy(rows, cols, shape):
        seen.append((np.asarray(rows, dtype=np.float64).copy(), np.asarray(cols, dtype=np.float64).copy(), tuple(shape)))
        return real(rows, cols, shape)

    ob.fill_polygon = spy
    try:
        Obstacle(_cfg(False), 6, seed=11)
    finally:
        ob.fill_polygon = real
    assert len(seen) >= 6
    for rows, cols, shape in seen:
        rr, cc = real(rows, cols, shape)
        assert set(zip(rr.tolist(), cc.tolist())) == _skimage_polygon_rule(rows, cols, shape), (rows, cols)


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
