"""Height-field terrain on the CPU oracle: the scan-height sample against the reference's own
_get_heights (golden), plane equivalence of a constant-height field, a robot at rest on a slope, tile generator sanity."""
import ctypes as C
import os

import numpy as np
import pytest

from tests.oracle_lib import OracleSim, go2_cfg

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
MASS, G = 15.019, 9.81


def hf_cfg(n, rows, cols, border=3.0, **over):
    q = go2_cfg(n, plane=True, **over)
    q.terrain_type = 1
    q.hf_rows, q.hf_cols, q.hf_hscale, q.hf_vscale, q.hf_border = rows, cols, 0.1, 0.005, border
    return q


def test_scan_height_matches_reference_get_heights():
    g = np.load(os.path.join(GOLD, "heights.npz"))
    hs, root = g["height_samples"], g["root_states"]
    n = root.shape[0]
    o = OracleSim(hf_cfg(n, hs.shape[0], hs.shape[1], border=float(g["border"]), push_robots=0, add_noise=0))
    o.t["HEIGHT_SAMPLES"][...] = hs
    o.reset_all()
    o.t["ROOT_STATES"][...] = root
    o.t["EPISODE_LENGTH"][:] = 5
    o.lib.qo_debug_post_physics.argtypes = [C.c_void_p, C.c_int64]
    assert o.lib.qo_debug_post_physics(o.h, 7) == 0
    assert np.allclose(o.t["SCAN_HEIGHT"], g["heights"][:, 94], atol=1e-7)     # point 94 = measured_heights.shape[1] // 2 + 1
    assert len(np.unique(g["heights"][:, 94])) > 20


def test_constant_height_field_equals_plane():
    """A height field that is 0.25 m everywhere must reproduce the plane run shifted by 0.25 m."""
    n = 16
    kw = dict(randomize_base_mass=0, randomize_base_com=0, push_robots=0, add_noise=0, seed=4)
    p = OracleSim(go2_cfg(n, **kw))
    q = hf_cfg(n, 400, 400, border=3.0, **kw); q.reset_xy_jitter = 0.0
    h = OracleSim(q)
    h.t["HEIGHT_SAMPLES"][...] = 50                     # 50 * 0.005 = 0.25 m
    h.t["ENV_ORIGINS"][:, 2] = 0.25
    p.reset_all(); h.reset_all()
    rng = np.random.default_rng(0)
    for _ in range(30):
        a = rng.normal(0, 0.5, (n, 12)).astype(np.float32)
        p.step(a); h.step(a)
    zp, zh = p.t["ROOT_STATES"].copy(), h.t["ROOT_STATES"].copy()
    zh[:, 2] -= 0.25
    same = p.t["EPISODE_LENGTH"] == h.t["EPISODE_LENGTH"]
    assert same.mean() > 0.9
    assert np.allclose(zp[same], zh[same], atol=2e-4)
    assert np.allclose(p.t["REW"][same], h.t["REW"][same], atol=1e-5)
    assert np.allclose(p.t["OBS"][same], h.t["OBS"][same], atol=2e-3)       # root_h = z - scan height, so the rows agree
    assert np.allclose(h.t["SCAN_HEIGHT"], 0.25)


def test_rest_on_slope_carries_weight_along_gravity():
    """10 % slope in x: the robot settles, friction holds it, the contact forces sum to the weight."""
    n = 4
    q = hf_cfg(n, 300, 300, border=3.0, randomize_base_mass=0, randomize_base_com=0, randomize_motor=0, randomize_friction=0,
               push_robots=0, add_noise=0)
    q.reset_xy_jitter = 0.0; q.env_spacing = 3.0
    o = OracleSim(q)
    rows = np.arange(300)[:, None]
    o.t["HEIGHT_SAMPLES"][...] = np.rint(rows * 0.1 * 0.10 / 0.005).astype(np.int16)     # z = 0.1 * (x + border)
    o.t["ENV_ORIGINS"][:, 0] = 5.0 + np.arange(n); o.t["ENV_ORIGINS"][:, 1] = 5.0
    o.t["ENV_ORIGINS"][:, 2] = 0.1 * (o.t["ENV_ORIGINS"][:, 0] + 3.0)
    o.reset_all()
    o.t["ROOT_STATES"][:, 7:13] = 0
    for k in range(200):
        o.step(np.zeros((n, 12), np.float32))
        assert not o.t["RESET"].any()
        if k == 99:
            feet0 = o.t["RIGID_BODY_POS"][:, [6, 10, 14, 18]].copy()
    f = o.t["CONTACT_FORCES"].sum(1)
    assert np.allclose(f[:, 2], MASS * G, rtol=0.03) and np.all(np.abs(f[:, :2]) < 0.03 * MASS * G)   # net force = -weight
    # the soft PD stance (Kp = 40) leans back on the slope, but the feet do not slide (mu = 1 >> 0.1): no creep over 2 s
    assert np.abs(o.t["RIGID_BODY_POS"][:, [6, 10, 14, 18]] - feet0).max() < 2e-3
    assert np.all(np.abs(o.t["ROOT_STATES"][:, 7:13]) < 0.2)
    exp_h = 0.1 * (o.t["ROOT_STATES"][:, 0] + 3.0)
    assert np.allclose(o.t["SCAN_HEIGHT"], exp_h, atol=0.02)


def test_tile_generator_statistics():
    from quadrupedal_agility_amd.legged_gym.envs.go2.go2_locomotion_config import Go2LocomotionCfg
    from quadrupedal_agility_amd.legged_gym.utils.terrain import SubTerrain, Terrain, pyramid_sloped_terrain, random_uniform_terrain
    np.random.seed(3)
    cfg = Go2LocomotionCfg()
    t = Terrain(cfg.terrain, 64)
    assert t.height_field_raw.shape == (1600, 1600) and t.height_field_raw.dtype == np.int16
    assert np.all(t.height_field_raw[:300] == 0) and np.all(t.height_field_raw[:, -300:] == 0)      # 30 m flat border
    org = t.env_origins
    assert np.allclose(org[3, 7, :2], [35.0, 75.0]) and org[..., 2].max() <= 0.4 * 0.4 * 5 + 0.06
    s = SubTerrain(width=100, length=100, vertical_scale=0.005, horizontal_scale=0.1)
    pyramid_sloped_terrain(s, slope=0.16, platform_size=3.0)
    h = s.height_field_raw * 0.005
    assert h.max() == pytest.approx(0.16 * 5.0 * (35 / 50) ** 2, abs=0.03)                     # platform clips the peak
    assert np.mean(np.diff(h[:24, 50])) / 0.1 == pytest.approx(0.16, abs=0.01)                  # the stated slope along the mid line (5 mm quantised)
    r = SubTerrain(width=100, length=100, vertical_scale=0.005, horizontal_scale=0.1)
    random_uniform_terrain(r, -0.05, 0.05, 0.005, downsampled_scale=0.2)
    hr = r.height_field_raw * 0.005
    assert -0.0501 <= hr.min() and hr.max() <= 0.0501 and hr.std() > 0.015


def test_slide_down_a_slope_follows_coulomb():
    """tan(theta) = 0.4 > mu = 0.2: the standing robot slides as a rigid block, a = g (sin theta - mu cos theta)."""
    n, slope, mu = 2, 0.4, 0.2
    q = hf_cfg(n, 600, 120, border=3.0, randomize_base_mass=0, randomize_base_com=0, randomize_motor=0, randomize_friction=0,
               push_robots=0, add_noise=0)
    q.reset_xy_jitter = 0.0; q.env_spacing = 3.0; q.ground_friction = mu
    o = OracleSim(q)
    o.t["FRICTION"][:] = mu                                   # contact friction = mean of robot and ground friction
    z = 30.0 - slope * (np.arange(600)[:, None] * 0.1)        # downhill towards +x
    o.t["HEIGHT_SAMPLES"][...] = np.rint(z / 0.005).astype(np.int16) * np.ones((1, 120), np.int16)
    o.t["ENV_ORIGINS"][:, 0] = 5.0; o.t["ENV_ORIGINS"][:, 1] = 3.0 + np.arange(n)
    o.t["ENV_ORIGINS"][:, 2] = 30.0 - slope * (5.0 + 3.0)
    o.reset_all()
    o.t["ROOT_STATES"][:, 7:13] = 0
    speed = []
    for k in range(60):
        o.step(np.zeros((n, 12), np.float32))
        assert not o.t["RESET"].any()
        speed.append(np.linalg.norm(o.t["ROOT_STATES"][:, 7:10], axis=1))
    speed = np.array(speed)
    th = np.arctan(slope)
    a = G * (np.sin(th) - mu * np.cos(th))                    # 1.82 m/s^2
    t = np.arange(60) * 0.02
    fit = np.polyfit(t[25:], speed[25:], 1)                   # after the landing transient
    assert fit[0][0] == pytest.approx(a, rel=0.04) and fit[0][1] == pytest.approx(a, rel=0.04)
    v = o.t["ROOT_STATES"][:, 7:10]
    assert np.all(np.abs(v[:, 1]) < 0.1) and np.allclose(v[:, 2] / v[:, 0], -slope, atol=0.05)       # along the fall line
