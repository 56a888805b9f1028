"""The per-env-step tolerance tables of the HIP-vs-oracle parity tests, and the helpers that apply them.

No pytest and no GPU in here: `__graft_entry__.smoke()` imports this module as well as tests/test_hip_parity.py (the entry point
must not need the test module, and with it pytest, to know its tolerances)."""
import numpy as np

# (atol, rtol) per tensor for one env step (4 substeps) from identical state.  DERIVED, not asserted: atol = 3 x the measured
# p99.9 of the per-env-step error |hip - oracle| (max over the env's elements) of this very protocol -- 40 steps x 1000 envs, 40,000 env-steps --
# in profiles/r4_step_error_distribution.json (tools/step_error_distribution.py), rounded up to one digit; the p99.9 is quoted beside each entry.
# r6: re-measured on HEAD's kernel (packed rows, env-local coordinates, three-wavefront data flow) --
# profiles/r6_step_error_distribution.json: every p50 equals r4's to three digits, the p99.9 are within 2-10 % of r4's (same arithmetic per value: the
# r5 changes re-arranged WHERE the work runs, not the order of the sums), so the tables stand; one entry moved: PROJECTED_GRAVITY p99.9 1.04e-6 -> 1.13e-6.
# Medians are 1e-6 (world positions: one fp32 ulp at 96 m is 7.6e-6) to 2e-5 (joint state), i.e. ~1e-5 relative; the tail above p99 is envs in which
# a contact switches on or off in one of the four substeps on one side only (fp32 Schur / PGS vs the oracle's dense solve in double), which is
# why BASELINE.md's "<= 1e-5 relative" holds for the median env-step and not for all of them.  Env-steps outside these tolerances are counted
# against a budget of twice their measured share (FLIP_BUDGET).
TOL = {
    "ROOT_STATES": (5e-4, 1e-6), "LAST_ROOT_VEL": (5e-4, 0),                     # p99.9 1.6e-4 (p50 1.4e-6); rtol = 8 ulp of the world position (the course tests run 900 m from the origin)
    "DOF_STATE": (6e-3, 0), "LAST_DOF_VEL": (6e-3, 0),                            # p99.9 2.0e-3 (p50 2.0e-5; velocities up to 30 rad/s)
    "CONTACT_FORCES": (0.16, 5e-3), "FEET_FORCE": (0.03, 5e-3),                   # p99.9 5.1e-2 / 8.9e-3 N on forces up to 5,000 N
    "RIGID_BODY_POS": (3e-5, 1e-6),                                               # p99.9 7.6e-6 = one ulp at 64-128 m
    "TORQUES": (4e-3, 0), "TORQUES_ORG": (4e-3, 0), "LAST_TORQUES_ORG": (4e-3, 0),   # p99.9 1.1e-3 / 1.2e-3 Nm
    "ACTIONS": (0, 0), "LAST_ACTIONS": (0, 0), "ACTION_HISTORY": (0, 0),
    "OBS": (4e-4, 0), "OBS_DISC": (4e-4, 0), "OBS_DISC_TERM": (4e-4, 0),          # p99.9 1.1e-4
    "COMMANDS": (4e-7, 0), "LATENT_EPS": (1e-7, 0), "LATENT_C": (0, 0),           # p99.9 1.2e-7 (one ulp of a resampled command)
    "REW": (1e-6, 0), "EPISODE_SUMS": (6e-6, 0),                                  # p99.9 5.4e-8 / 1.9e-6
    "RESET": (0, 0), "TIME_OUT": (0, 0), "EPISODE_LENGTH": (0, 0), "LAST_CONTACTS": (0, 0), "CONTACT_FILT": (0, 0),
    "FOOT_IMPULSE": (1.3e-4, 0),                                                  # p99.9 4.2e-5 N s
    "BASE_LIN_VEL": (6e-5, 0), "BASE_ANG_VEL": (6e-4, 0), "PROJECTED_GRAVITY": (4e-6, 0), "RPY": (5e-6, 0),   # p99.9 1.9e-5 / 1.9e-4 / 1.1e-6 / 1.6e-6 (r6)
}
# The ACCURACY statement: TOL above is a fence around the kernel's own error TAIL -- a defect that was present when the
# distribution was taken sits inside it by construction.  What says "this env step is computed to ~1e-5 relative" is the MEDIAN env-step error,
# asserted here per tensor over all env-steps of the test: 3 x the p50 of profiles/r4_step_error_distribution.json = of profiles/r6_step_error_distribution.json (quoted beside each entry),
# floored at one fp32 ulp of the tensor's largest value.  A systematic error (a wrong term, a stale operand) moves the median of every env, not the tail.
MEDIAN_TOL = {
    "ROOT_STATES": 5e-6, "LAST_ROOT_VEL": 5e-6,            # p50 1.4e-6 on positions up to 96 m (one ulp 7.6e-6) and velocities up to 41 m/s
    "DOF_STATE": 6e-5, "LAST_DOF_VEL": 6e-5,               # p50 2.0e-5 on velocities up to 30 rad/s (7e-7 relative)
    "CONTACT_FORCES": 5e-4, "FEET_FORCE": 4e-4,            # p50 1.7e-4 / 1.2e-4 N on forces up to 5,000 N
    "RIGID_BODY_POS": 8e-6,                                # p50 6e-8; floor: one ulp at 64-128 m
    "TORQUES": 7e-5, "TORQUES_ORG": 8e-5, "LAST_TORQUES_ORG": 8e-5,   # p50 2.1e-5 / 2.4e-5 Nm
    "OBS": 4e-6, "OBS_DISC": 4e-6, "OBS_DISC_TERM": 4e-6,  # p50 1.2e-6
    "REW": 1e-8, "EPISODE_SUMS": 2e-6,                     # p50 0 / 1.5e-8 (floor: one ulp of a 32-point sum)
    "FOOT_IMPULSE": 3e-6,                                  # p50 7.2e-7 N s
    "BASE_LIN_VEL": 8e-7, "BASE_ANG_VEL": 5e-6, "PROJECTED_GRAVITY": 4e-7, "RPY": 3e-7,   # p50 2.4e-7 / 1.7e-6 / 1.2e-7 / 4.5e-8
}


def env_errors(name, a, b, n_envs):
    """per-env error of one tensor: max over the env's elements of |a - b|"""
    a = a.astype(np.float64); b = b.astype(np.float64)
    if name == "EPISODE_SUMS":
        a = a.T; b = b.T
    return np.abs(a - b).reshape(n_envs, -1).max(axis=1)


def check_medians(per_tensor_errors):
    """per_tensor_errors: {tensor: [per-env error arrays, one per step]}"""
    med = {k: float(np.median(np.concatenate(v))) for k, v in per_tensor_errors.items()}
    print("median env-step error per tensor:", {k: f"{v:.2e}" for k, v in med.items()})
    bad = {k: (v, MEDIAN_TOL[k]) for k, v in med.items() if v > MEDIAN_TOL[k]}
    assert not bad, f"median env-step error above 3 x the measured p50: {bad}"


# share of env-steps allowed outside TOL per protocol: ~2x the share measured with this TOL table on MI355X (profiles/r5_parity_flip_shares.txt,
# `QA_PARITY_MEASURE=1 pytest -m gpu -s -k parity`); filled in from that run
# measured (r5, MI355X, env-local coordinates inside a step in kernel and oracle): plane 0.08-0.10 %, height field 0.35 % (64 envs) / 0.13 % (600), ceiling 1.27 %,
# mocap 0.04 %, self-collision 0.10 %, articulated obstacles 0 of 1,024 (64 envs) and 0 of 49,152 (8192 envs), course 0 of 1,920 / 0 of 98,304.
# (r4, world coordinates in fp32: height field 1.1-1.5 %, ceiling 3.75 %, articulated at 8192 envs 3.7 % -- an ulp 900 m from the origin is 6e-5 m;
# budgets then: 0.03 / 0.075 / 0.075.)
BUDGET = {"plane": 0.002, "height_field": 0.008, "ceiling": 0.026, "mocap": 0.001, "self_collision": 0.002, "articulated": 0.002, "articulated_8192": 0.002, "course": 0.001}


def check_flips(name, flips, total, budget):
    """env-steps outside TOL against a budget that is ~2x the share MEASURED with the current TOL table (profiles/r5_parity_flip_shares.txt).
    QA_PARITY_MEASURE=1: print the share and do not judge it (how that profile is made)."""
    import os
    share = flips / max(total, 1)
    print(f"FLIPSHARE {name} {flips}/{total} = {share:.5f} (budget {budget})")
    if os.environ.get("QA_PARITY_MEASURE") != "1":
        assert share <= budget + 2.0 / max(total, 1), (name, share, budget)



def env_mismatch(name, a, b, n_envs):
    atol, rtol = TOL[name]
    a = a.astype(np.float64); b = b.astype(np.float64)
    if name == "EPISODE_SUMS":
        a = a.T; b = b.T
    bad = ~np.isclose(a, b, atol=atol, rtol=rtol)
    return bad.reshape(n_envs, -1).any(axis=1)
