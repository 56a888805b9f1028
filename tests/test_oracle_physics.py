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


# ------------------------------------------------------------------ round-2 KATs: friction cone, joint stops, actuation, resting depth, body velocities
def _ramp_sim(n, slope, ground_friction):
    """robots standing (PD on the default pose) on a plane of slope `slope` = tan(theta) along +x, built as a height field"""
    rows, cols = 400, 120
    qc = go2_cfg(n, randomize_base_mass=0, randomize_base_com=0, randomize_motor=0, randomize_friction=0, push_robots=0, add_noise=0,
                 ground_friction=ground_friction)
    qc.terrain_type = 1
    qc.hf_rows, qc.hf_cols, qc.hf_hscale, qc.hf_vscale, qc.hf_border = rows, cols, 0.1, 0.0005, 2.0
    s = OracleSim(qc)
    x = np.arange(rows)[:, None] * 0.1 - 2.0
    s.t["HEIGHT_SAMPLES"][...] = np.rint(np.repeat(-slope * x, cols, axis=1) / 0.0005).astype(np.int16)       # downhill towards +x
    s.t["ENV_ORIGINS"][:, 0] = 3.0; s.t["ENV_ORIGINS"][:, 1] = 3.0 + 0.8 * np.arange(n); s.t["ENV_ORIGINS"][:, 2] = -slope * 3.0
    s.reset_all()
    s.t["ROOT_STATES"][:, 7:13] = 0
    s.t["DOF_STATE"][:, :, 0] = np.array([0, 0.9, -1.8] * 4, np.float32)
    return s


@pytest.mark.parametrize("slope,mu,slides", [(0.30, 0.8, False), (0.45, 0.35, True)])
def test_friction_cone_on_an_incline(slope, mu, slides):
    """stick below tan(theta) = mu, slide above it with a = g (sin(theta) - mu cos(theta)): mu = 1/2 (robot + ground friction)"""
    s = _ramp_sim(2, slope, 2 * mu - 1.0)                       # robot-shape friction is 1 without randomisation
    vx, xs = [], []
    for k in range(60 if slides else 200):
        s.step(np.zeros((2, 12), np.float32))
        vx.append(s.t["ROOT_STATES"][:, 7].copy()); xs.append(s.t["ROOT_STATES"][:, 0].copy())
    vx, xs = np.array(vx), np.array(xs)
    th = np.arctan(slope)
    if not slides:                                                # the landing wobble dies out and the robot stays where it is
        assert np.abs(vx[-40:]).max() < 0.03 and np.abs(xs[-1] - xs[-80]).max() < 0.01
    else:
        a = np.polyfit(np.arange(25, 60) * 0.02, vx[25:, 0] / np.cos(th), 1)[0]          # along-slope acceleration
        assert a == pytest.approx(G * (np.sin(th) - mu * np.cos(th)), rel=0.12)
        assert (s.t["RESET"] == 0).all()


@pytest.mark.parametrize("joint,torque,stop", [(0, -20.0, -1.0472), (2, 40.0, -0.83776), (1, -20.0, -1.5708)])
def test_joint_stop_holds_against_saturated_torque(joint, torque, stop):
    """one joint driven into its stop by the full motor torque in free flight: it arrives with the speculative approach speed
    gap/dt, overshoots by less than 1 degree and ends at rest on the stop"""
    qc = go2_cfg(1, randomize_base_mass=0, randomize_base_com=0, randomize_motor=0, randomize_friction=0)
    s = OracleSim(qc)
    s.t["ROOT_STATES"][0] = [0, 0, 30.0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0]
    s.t["DOF_STATE"][0, :, 0] = [0, 0.9, -1.8] * 4
    tau = np.zeros((1, 12), np.float32); tau[0, joint] = torque
    sgn = -1.0 if stop < s.t["DOF_STATE"][0, joint, 0] else 1.0
    worst = 0.0
    for k in range(120):
        s.simulate(tau)
        worst = max(worst, sgn * (s.t["DOF_STATE"][0, joint, 0] - stop))
    assert worst < 0.0175
    assert abs(s.t["DOF_STATE"][0, joint, 0] - stop) < 0.0175 and abs(s.t["DOF_STATE"][0, joint, 1]) < 0.05


def test_coupled_joint_stops_converge_with_solver_iterations():
    """two saturated motors of one leg pressing two stops at once is the hard case for a 4-sweep projected Gauss-Seidel (PhysX's
    own iteration count, legged_robot_config.py:176): the leak past the stop is bounded at 4 sweeps and gone at 50"""
    leak = {}
    for iters in (4, 50):
        qc = go2_cfg(1, randomize_base_mass=0, randomize_base_com=0, randomize_motor=0, randomize_friction=0, solver_iterations=iters)
        s = OracleSim(qc)
        s.t["ROOT_STATES"][0] = [0, 0, 30.0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0]
        s.t["DOF_STATE"][0, :, 0] = [0, 0.9, -1.8] * 4
        tau = np.zeros((1, 12), np.float32); tau[0, 2] = 40.0; tau[0, 0] = -20.0
        w = 0.0
        for k in range(60):
            s.simulate(tau)
            q = s.t["DOF_STATE"][0, :, 0]
            w = max(w, q[2] - (-0.83776), -1.0472 - q[0])
        leak[iters] = w
    assert leak[50] < 0.003 and leak[4] < 0.15, leak


def test_pd_actuation_is_internal_and_converges():
    """zero gravity, no contact: the PD torques are internal forces -- total linear and angular momentum stay exactly where they
    were -- and every joint settles on its target q0 + action_scale * a (hips: * hip_scale_reduction)"""
    qc = go2_cfg(1, randomize_base_mass=0, randomize_base_com=0, randomize_motor=0, randomize_friction=0, gravity_z=0.0, push_robots=0, add_noise=0)
    s = OracleSim(qc)
    s.reset_all()
    s.t["ROOT_STATES"][0] = [0, 0, 40.0, 0, 0, 0, 1, 0.3, -0.2, 0.1, 0.2, 0.1, -0.3]
    s.t["DOF_STATE"][0, :, 0] = [0, 0.9, -1.8] * 4; s.t["DOF_STATE"][0, :, 1] = 0
    s.t["EPISODE_LENGTH"][:] = 5

    def momentum():
        root = s.t["ROOT_STATES"][0].astype(np.float64); R = quat_to_mat(root[3:7])
        ub = np.concatenate([R.T @ root[10:13], R.T @ root[7:10]])
        com, vel, omg, mass, I = bodies(s.lib, s.t["DOF_STATE"][0, :, 0].copy(), s.t["DOF_STATE"][0, :, 1].copy(), ub)
        p = sum(mass[b] * (R @ vel[b]) for b in range(13))
        c = sum(mass[b] * (R @ com[b]) for b in range(13)) / mass.sum()
        L = sum(mass[b] * np.cross(R @ com[b] - c, R @ vel[b]) + R @ (I[b] @ omg[b]) for b in range(13))
        return p, L
    p0, L0 = momentum()
    act = np.zeros((1, 12), np.float32); act[0] = [1.0, 0.8, -0.6, -1.0, 0.4, 0.5, 0.5, -0.5, 0.3, -0.5, 0.2, -0.4]
    for k in range(60):
        s.step(act)
    p1, L1 = momentum()
    assert np.allclose(p1, p0, atol=5e-3) and np.allclose(L1, L0, atol=8e-3)
    target = np.array([0, 0.9, -1.8] * 4) + 0.25 * act[0] * np.array([0.5, 1, 1] * 4)
    assert np.allclose(s.t["DOF_STATE"][0, :, 0], target, atol=0.02) and np.abs(s.t["DOF_STATE"][0, :, 1]).max() < 0.2


def test_resting_feet_sit_inside_the_contact_offset():
    """standing at rest: every foot sphere floats between 0 and contact_offset above the plane (speculative contacts hold it at
    the surface; it neither sinks nor hovers above the offset)"""
    qc = go2_cfg(4, randomize_base_mass=0, randomize_base_com=0, randomize_motor=0, push_robots=0, add_noise=0)
    s = OracleSim(qc)
    s.reset_all(); s.t["ROOT_STATES"][:, 7:13] = 0
    for _ in range(120):
        s.step(np.zeros((4, 12), np.float32))
    gap = s.t["RIGID_BODY_POS"][:, [6, 10, 14, 18], 2] - 0.022
    assert (gap > -0.002).all() and (gap < qc.contact_offset).all(), gap


def test_rigid_body_state_velocities_are_the_time_derivative_of_the_positions():
    """QA_T_RIGID_BODY_STATE (seam 1): over one substep in free flight, (x_new - x_old) / dt of every body origin equals the reported
    origin velocity to O(dt), the reported quaternion rotates the body like the link chain does, and base rows equal the root state"""
    qc = go2_cfg(1, randomize_base_mass=0, randomize_base_com=0, randomize_motor=0, randomize_friction=0, gravity_z=0.0, export_body_state=1)
    s = OracleSim(qc)
    rng = np.random.default_rng(2)
    q, qd, ub = rand_state(rng)
    quat = rng.normal(size=4); quat /= np.linalg.norm(quat)
    s.t["ROOT_STATES"][0] = np.concatenate([[0, 0, 20.0], quat, rng.uniform(-1, 1, 3), rng.uniform(-2, 2, 3)])
    s.t["DOF_STATE"][0, :, 0] = q; s.t["DOF_STATE"][0, :, 1] = qd * 0.5
    z = np.zeros((1, 12), np.float32)
    s.simulate(z)
    a = s.t["RIGID_BODY_STATE"][0].astype(np.float64).copy()
    s.simulate(z)
    b = s.t["RIGID_BODY_STATE"][0].astype(np.float64).copy()
    fd = (b[:, :3] - a[:, :3]) / qc.sim_dt
    vmid = 0.5 * (a[:, 7:10] + b[:, 7:10])
    assert np.abs(fd - vmid).max() < 0.03 * max(1.0, np.abs(vmid).max())
    assert np.allclose(np.linalg.norm(b[:, 3:7], axis=1), 1.0, atol=1e-6)
    root = s.t["ROOT_STATES"][0]
    assert np.allclose(b[0, :3], root[:3]) and np.allclose(b[0, 3:7], root[3:7]) and np.allclose(b[0, 7:13], root[7:13], atol=1e-6)
    # orientation: the calf frame's x axis in the world = R_base Rx(q_hip) Ry(q_thigh + q_calf) e_x
    qn = s.t["DOF_STATE"][0, :, 0].astype(np.float64)
    Rb = quat_to_mat(root[3:7].astype(np.float64))
    for l in range(4):
        h, t2 = qn[3 * l], qn[3 * l + 1] + qn[3 * l + 2]
        Rx = np.array([[1, 0, 0], [0, np.cos(h), -np.sin(h)], [0, np.sin(h), np.cos(h)]])
        Ry = np.array([[np.cos(t2), 0, np.sin(t2)], [0, 1, 0], [-np.sin(t2), 0, np.cos(t2)]])
        assert np.allclose(quat_to_mat(b[3 + 4 * l + 2, 3:7]), Rb @ Rx @ Ry, atol=2e-6)
        assert np.allclose(b[3 + 4 * l + 3, 3:7], b[3 + 4 * l + 2, 3:7])            # the foot is fixed to the calf
    # angular velocity of the calf: base + the three joint rates about their axes
    for l in range(4):
        w = Rb @ (Rb.T @ root[10:13] + np.array([1, 0, 0]) * s.t["DOF_STATE"][0, 3 * l, 1]
                  + np.array([0, np.cos(qn[3 * l]), np.sin(qn[3 * l])]) * (s.t["DOF_STATE"][0, 3 * l + 1, 1] + s.t["DOF_STATE"][0, 3 * l + 2, 1]))
        assert np.allclose(b[3 + 4 * l + 2, 10:13], w, atol=1e-5)


def test_every_body_of_a_leg_can_report_contact():
    """r2 contact model: one contact per BODY of a leg (hip link / base share | thigh | calf, besides the foot).  A robot dropped
    on its belly with the legs swept back rests on hips AND calves at once: two non-foot bodies of the SAME leg carry force in
    the same substep -- with the single shared non-foot slot per leg of round 1 only one of them could (check_termination reads
    the hips, _reward_collision the thighs and calves: legged_robot.py:168-176, 1275-1278)"""
    qc = go2_cfg(1, randomize_base_mass=0, randomize_base_com=0, randomize_motor=0, randomize_friction=0, push_robots=0, add_noise=0)
    s = OracleSim(qc)
    s.reset_all()
    s.t["ROOT_STATES"][0] = [0, 0, 0.15, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0]
    s.t["DOF_STATE"][0, :, 0] = [0, -1.2, -0.85] * 4; s.t["DOF_STATE"][0, :, 1] = 0
    tau = np.zeros((1, 12), np.float32)
    most = 0
    for _ in range(200):
        s.simulate(tau)
        f = np.linalg.norm(s.t["CONTACT_FORCES"][0], axis=1)
        most = max(most, max(int((f[3 + 4 * l:3 + 4 * l + 3] > 0.1).sum()) for l in range(4)))
    assert most >= 2
    assert np.isfinite(s.t["ROOT_STATES"]).all() and s.t["ROOT_STATES"][0, 2] > 0.02
    assert s.t["CONTACT_FORCES"][0, :, 2].sum() == pytest.approx(MASS * G, rel=0.15)      # at rest the contacts carry the weight


def _ceiling_sim(make=OracleSim, height=0.45):
    """flat height-field floor; env 0 stands under an overhang `height` m above it, env 1 under open sky"""
    rows, cols = 160, 160
    qc = go2_cfg(2, randomize_base_mass=0, randomize_base_com=0, randomize_motor=0, randomize_friction=0, push_robots=0, add_noise=0)
    qc.terrain_type = 1; qc.hf_ceiling = 1
    qc.hf_rows, qc.hf_cols, qc.hf_hscale, qc.hf_vscale, qc.hf_border = rows, cols, 0.05, 0.005, 1.0
    s = make(qc)
    ceil = np.full((rows, cols), 32767, np.int16); ceil[:80] = int(round(height / 0.005))        # x < 3 m (map) is roofed
    return s, ceil


def test_a_ceiling_stops_a_robot_thrown_upwards():
    """overhang contacts (QA_T_CEILING_SAMPLES; the course's tunnel roof and the tyre's upper arc): a robot launched upwards at
    2.5 m/s flies to ~0.6 m under open sky, and is stopped by a 0.45 m roof -- the base reports a DOWNWARD contact force"""
    s, ceil = _ceiling_sim()
    assert (s.t["CEILING_SAMPLES"] == 32767).all()                  # qa_create's default: no overhang anywhere
    s.t["CEILING_SAMPLES"][...] = ceil
    s.t["ENV_ORIGINS"][:, 0] = [1.0, 5.0]; s.t["ENV_ORIGINS"][:, 1] = 3.0
    s.reset_all()
    s.t["ROOT_STATES"][:, 7:13] = 0; s.t["ROOT_STATES"][:, 9] = 2.5
    tau = np.zeros((2, 12), np.float32)
    top, fz = np.zeros(2), np.zeros(2)
    for _ in range(120):
        s.simulate(tau)
        top = np.maximum(top, s.t["ROOT_STATES"][:, 2]); fz = np.minimum(fz, s.t["CONTACT_FORCES"][:, 0, 2])
    assert top[1] > 0.55 and fz[1] == 0.0
    assert top[0] < 0.43 and fz[0] < -20.0                            # the trunk (5.7 cm half height) stays under the roof
    assert np.isfinite(s.t["ROOT_STATES"]).all()


def test_a_point_well_above_an_overhang_does_not_feel_it():
    """thin-shell rule (QA_CEILING_SHELL): the underside of an overhang does not act on a body more than 5 cm above it -- robots
    standing on a 1 m platform whose ceiling field (wrongly) says 0.45 m step exactly as they do with an empty field"""
    out = []
    for with_field in (True, False):
        s, ceil = _ceiling_sim()
        s.t["HEIGHT_SAMPLES"][...] = 200                               # 1 m platform everywhere
        s.t["ENV_ORIGINS"][:, 0] = [1.0, 5.0]; s.t["ENV_ORIGINS"][:, 1] = 3.0; s.t["ENV_ORIGINS"][:, 2] = 1.0
        if with_field:
            s.t["CEILING_SAMPLES"][...] = ceil
        s.reset_all()
        for _ in range(25):
            s.step(np.zeros((2, 12), np.float32))
        out.append(s.t["ROOT_STATES"].copy())
    assert np.array_equal(out[0], out[1]) and out[0][0, 2] > 1.2
