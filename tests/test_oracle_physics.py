"""Analytic known-answer tests that pin the physics part of the CPU oracle.

The reference's physics is the closed Isaac Gym binary (legged_robot.py:103-106), so there is
no golden vector to compare with ("parity unpinned").  These tests instead pin
the build's own rigid-body model by conservation laws and closed-form answers:
  * kinetic-energy identity  1/2 u^T M(q) u == sum over bodies (CRBA vs independent body sums)
  * free flight: CoM accelerates at exactly g, linear/angular momentum about the CoM conserved
  * free flight: total mechanical energy conserved up to O(dt) integrator drift
  * static stance: sum of foot normal forces == total weight
  * a dropped robot never penetrates the ground by more than a few mm and comes to rest
"""
import ctypes as C

import numpy as np
import pytest

from tests.oracle_lib import OracleSim, go2_cfg, load_oracle

G = 9.81
MASS = 15.019


def quat_to_mat(q):
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def bodies(lib, q, qd, ub):
    com = np.zeros((13, 3)); vel = np.zeros((13, 3)); omg = np.zeros((13, 3)); mass = np.zeros(13); I = np.zeros((13, 3, 3))
    q = np.ascontiguousarray(q, np.float32); qd = np.ascontiguousarray(qd, np.float32); ub = np.ascontiguousarray(ub, np.float32)
    lib.qo_debug_bodies(q.ctypes.data, qd.ctypes.data, ub.ctypes.data, com.ctypes.data, vel.ctypes.data, omg.ctypes.data,
                        mass.ctypes.data, I.ctypes.data)
    return com, vel, omg, mass, I


def dynamics(lib, q, qd, ub, quat=(0, 0, 0, 1)):
    M = np.zeros((18, 18)); h = np.zeros(18)
    q = np.ascontiguousarray(q, np.float32); qd = np.ascontiguousarray(qd, np.float32); ub = np.ascontiguousarray(ub, np.float32)
    qu = np.ascontiguousarray(quat, np.float32)
    lib.qo_debug_dynamics(q.ctypes.data, qd.ctypes.data, ub.ctypes.data, qu.ctypes.data, M.ctypes.data, h.ctypes.data)
    return M, h


def rand_state(rng):
    q = np.array([0, 0.9, -1.8] * 4) + rng.uniform(-0.4, 0.4, 12)
    qd = rng.uniform(-5, 5, 12)
    ub = rng.uniform(-2, 2, 6)
    return q.astype(np.float32), qd.astype(np.float32), ub.astype(np.float32)


def test_total_mass_and_symmetry():
    lib = load_oracle()
    rng = np.random.default_rng(0)
    q, qd, ub = rand_state(rng)
    M, _ = dynamics(lib, q, qd, ub)
    assert np.allclose(M, M.T, atol=1e-12)
    assert np.allclose(np.diag(M)[3:6], MASS, atol=1e-4)
    assert np.all(np.linalg.eigvalsh(M) > 0)


def test_kinetic_energy_identity():
    """CRBA mass matrix against an independent per-body sum of 1/2 m v^2 + 1/2 w^T I w."""
    lib = load_oracle()
    rng = np.random.default_rng(1)
    for _ in range(20):
        q, qd, ub = rand_state(rng)
        M, _ = dynamics(lib, q, qd, ub)
        u = np.concatenate([ub, qd]).astype(np.float64)
        T_crba = 0.5 * u @ M @ u
        com, vel, omg, mass, I = bodies(lib, q, qd, ub)
        T_sum = sum(0.5 * mass[b] * vel[b] @ vel[b] + 0.5 * omg[b] @ I[b] @ omg[b] for b in range(13))
        assert T_crba == pytest.approx(T_sum, rel=1e-9)


def test_gravity_bias_at_rest():
    """With zero velocity the base rows of the bias are the wrench that holds the robot: force = -m g_B,
    moment = -(m c) x g_B, and joint rows equal the static gravity torques dV/dq (finite differences)."""
    lib = load_oracle()
    rng = np.random.default_rng(2)
    q, _, _ = rand_state(rng)
    z = np.zeros(12, np.float32); ub0 = np.zeros(6, np.float32)
    quat = np.array([0.1, -0.2, 0.05, 0.97], np.float32); quat /= np.linalg.norm(quat)
    R = quat_to_mat(quat.astype(np.float64))
    gB = R.T @ np.array([0, 0, -G])
    M, h = dynamics(lib, q, z, ub0, quat)
    com, _, _, mass, _ = bodies(lib, q, z, ub0)
    mc = (mass[:, None] * com).sum(0)
    assert np.allclose(h[3:6], -MASS * gB, atol=1e-4)
    assert np.allclose(h[0:3], -np.cross(mc, gB), atol=1e-5)

    def V(qq):
        c, _, _, m, _ = bodies(lib, qq, z, ub0)
        return -(m[:, None] * c).sum(0) @ gB   # potential energy in the base frame up to a constant
    eps = 1e-3
    for j in range(12):
        qp = q.copy(); qm = q.copy(); qp[j] += eps; qm[j] -= eps
        assert h[6 + j] == pytest.approx((V(qp) - V(qm)) / (2 * eps), abs=2e-4)


def _free_flight_sim(n_steps, seed):
    qc = go2_cfg(1, randomize_base_mass=0, randomize_base_com=0, randomize_motor=0, randomize_friction=0)
    s = OracleSim(qc)
    rng = np.random.default_rng(seed)
    q, qd, ub = rand_state(rng)
    quat = rng.normal(size=4); quat /= np.linalg.norm(quat)
    s.t["ROOT_STATES"][0] = np.concatenate([[0, 0, 50.0], quat, rng.uniform(-1, 1, 3), rng.uniform(-2, 2, 3)])
    s.t["DOF_STATE"][0, :, 0] = q
    s.t["DOF_STATE"][0, :, 1] = qd * 0.3
    out = []
    lib = s.lib
    for _ in range(n_steps + 1):
        root = s.t["ROOT_STATES"][0].astype(np.float64)
        R = quat_to_mat(root[3:7])
        ubody = np.concatenate([R.T @ root[10:13], R.T @ root[7:10]])
        qq = s.t["DOF_STATE"][0, :, 0].copy(); qqd = s.t["DOF_STATE"][0, :, 1].copy()
        com, vel, omg, mass, I = bodies(lib, qq, qqd, ubody)
        cw = root[:3] + com @ R.T
        vw = vel @ R.T
        c = (mass[:, None] * cw).sum(0) / mass.sum()
        p = (mass[:, None] * vw).sum(0)
        Lc = sum(mass[b] * np.cross(cw[b] - c, vw[b]) + R @ (I[b] @ omg[b]) for b in range(13))
        E = sum(0.5 * mass[b] * vw[b] @ vw[b] + 0.5 * omg[b] @ I[b] @ omg[b] + mass[b] * G * cw[b][2] for b in range(13))
        out.append((c, p, Lc, E))
        s.simulate(np.zeros((1, 12), np.float32))
    return out, qc.sim_dt


def test_free_flight_momentum():
    out, dt = _free_flight_sim(100, 3)
    p0, L0 = out[0][1], out[0][2]
    for k, (c, p, Lc, E) in enumerate(out):
        # semi-implicit Euler integrates constant gravity exactly in momentum
        assert np.allclose(p[:2], p0[:2], atol=2e-3)
        assert p[2] == pytest.approx(p0[2] - MASS * G * dt * k, abs=3e-3)
        assert np.allclose(Lc, L0, atol=5e-3)        # O(dt) integrator drift + fp32 state storage


def test_free_flight_energy_drift():
    out, dt = _free_flight_sim(200, 4)
    E = np.array([o[3] for o in out])
    assert abs(E[-1] - E[0]) / abs(E[0]) < 2e-3      # 1 s of tumbling with swinging legs


def test_static_stance_weight():
    qc = go2_cfg(2, randomize_base_mass=0, randomize_base_com=0, randomize_motor=0, push_robots=0, add_noise=0)
    s = OracleSim(qc)
    s.reset_all()
    s.t["ROOT_STATES"][:, 7:13] = 0
    for _ in range(150):
        s.step(np.zeros((2, 12), np.float32))
        assert not s.t["RESET"].any()
    fz = s.t["CONTACT_FORCES"][:, :, 2].sum(1)
    assert np.allclose(fz, MASS * G, rtol=0.03)
    assert np.all(np.abs(s.t["ROOT_STATES"][:, 7:13]) < 0.15)
    feet_z = s.t["RIGID_BODY_POS"][:, [6, 10, 14, 18], 2]
    assert np.all(feet_z > 0.022 - 0.004) and np.all(feet_z < 0.022 + 0.012)


def test_drop_does_not_tunnel_or_explode():
    qc = go2_cfg(4, randomize_base_mass=0, randomize_base_com=0, randomize_motor=0, push_robots=0, add_noise=0)
    s = OracleSim(qc)
    s.reset_all()
    s.t["ROOT_STATES"][:, 2] = 1.0
    zmin = 1e9
    for _ in range(100):
        s.step(np.zeros((4, 12), np.float32))
        assert np.isfinite(s.t["ROOT_STATES"]).all() and np.isfinite(s.t["OBS"]).all()
        zmin = min(zmin, s.t["RIGID_BODY_POS"][:, :, 2].min())
    assert zmin > -0.02


def test_zero_gravity_kinetic_energy():
    """No gravity, no contact, no torque: kinetic energy is conserved (checks the velocity-product
    (Coriolis/centrifugal) terms of the bias); stop before any joint reaches a limit stop."""
    qc = go2_cfg(1, randomize_base_mass=0, randomize_base_com=0, randomize_motor=0, randomize_friction=0, gravity_z=0.0)
    s = OracleSim(qc)
    rng = np.random.default_rng(5)
    q, qd, ub = rand_state(rng)
    quat = rng.normal(size=4); quat /= np.linalg.norm(quat)
    s.t["ROOT_STATES"][0] = np.concatenate([[0, 0, 50.0], quat, rng.uniform(-1, 1, 3), rng.uniform(-2, 2, 3)])
    s.t["DOF_STATE"][0, :, 0] = q
    s.t["DOF_STATE"][0, :, 1] = qd * 0.3
    E = []
    for _ in range(41):
        root = s.t["ROOT_STATES"][0].astype(np.float64)
        R = quat_to_mat(root[3:7])
        ubody = np.concatenate([R.T @ root[10:13], R.T @ root[7:10]])
        com, vel, omg, mass, I = bodies(s.lib, s.t["DOF_STATE"][0, :, 0].copy(), s.t["DOF_STATE"][0, :, 1].copy(), ubody)
        E.append(sum(0.5 * mass[b] * vel[b] @ vel[b] + 0.5 * omg[b] @ I[b] @ omg[b] for b in range(13)))
        s.simulate(np.zeros((1, 12), np.float32))
    E = np.array(E)
    assert np.abs(E / E[0] - 1).max() < 2e-3
