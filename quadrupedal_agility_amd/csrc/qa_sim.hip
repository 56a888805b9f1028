// qa_sim.hip -- gfx950 kernels and the C ABI of include/qa_sim.h.
//
// Launch geometry: one quad (4 lanes) per env, 64-thread workgroups = 16 envs per wavefront, so
// 4096 envs give 256 workgroups -- one per CU of the MI355X, each wave alone on its SIMD with the
// full 512-VGPR budget (the substep keeps ~300 live values in registers).  The whole
// LeggedRobot.step() is ONE launch: action-history roll, 4 x (PD torque -> dynamics -> contact
// PGS -> integrate), termination, rewards, resets, observation assembly.  Per-env rows are
// (N,k) row-major exactly as the reference's gym tensors; a quad reads its 12 joint values as 4 x
// 12 B contiguous (a wave covers 768 contiguous bytes), so the AoS layout is already coalesced
// for this lane mapping.  The 671-float observation rows are staged in LDS and written by the
// whole wave, 256 B per instruction.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <stdio.h>
#include <string.h>

#include "../../include/qa_sim.h"
#include "qa_go2_model.h"
#include "qa_physics.h"

#define QA_BLOCK 64
#define ENVS_PER_BLOCK (QA_BLOCK / 4)

static_assert(QA_NUM_LEG_PTS == QA_LEG_PTS && QA_NUM_BASE_PTS == QA_BASE_PTS, "model/table mismatch");

__constant__ float c_tbl[QA_TBL_FLOATS];

// ------------------------------------------------------------------ arena layout (ABI: enum order, 256 B aligned)
struct Layout {
    int64_t off[QA_T_COUNT];
    int64_t shape[QA_T_COUNT][3];
    int32_t ndim[QA_T_COUNT];
    int32_t dtype[QA_T_COUNT];
    int64_t total;
};
static int64_t dtype_size(int d) { return d == QA_F32 ? 4 : (d == QA_I64 || d == QA_F64) ? 8 : d == QA_I32 ? 4 : d == QA_I16 ? 2 : 1; }
static void make_layout(const qa_config *cfg, Layout *L) {
    const int64_t N = cfg->num_envs, F = cfg->num_mocap_frames > 0 ? cfg->num_mocap_frames : 1;
    const int64_t HR = cfg->terrain_type == 1 ? cfg->hf_rows : 1, HC = cfg->terrain_type == 1 ? cfg->hf_cols : 1;
    struct Spec { int t, dt, nd; int64_t a, b, c; };
    const Spec specs[] = {
        {QA_T_ROOT_STATES, QA_F32, 2, N, 13, 1}, {QA_T_DOF_STATE, QA_F32, 3, N, 12, 2},
        {QA_T_CONTACT_FORCES, QA_F32, 3, N, 19, 3}, {QA_T_RIGID_BODY_POS, QA_F32, 3, N, 19, 3},
        {QA_T_TORQUES, QA_F32, 2, N, 12, 1}, {QA_T_TORQUES_ORG, QA_F32, 2, N, 12, 1},
        {QA_T_ACTIONS, QA_F32, 2, N, 12, 1}, {QA_T_LAST_ACTIONS, QA_F32, 2, N, 12, 1},
        {QA_T_LAST_DOF_VEL, QA_F32, 2, N, 12, 1}, {QA_T_LAST_TORQUES_ORG, QA_F32, 2, N, 12, 1},
        {QA_T_LAST_ROOT_VEL, QA_F32, 2, N, 6, 1}, {QA_T_ACTION_HISTORY, QA_F32, 3, N, QA_ACTION_BUF_LEN, 12},
        {QA_T_OBS, QA_F32, 2, N, QA_NUM_OBS, 1},
        {QA_T_OBS_DISC, QA_F32, 2, N, QA_NUM_OBS_DISC, 1}, {QA_T_OBS_DISC_TERM, QA_F32, 2, N, QA_NUM_OBS_DISC, 1},
        {QA_T_COMMANDS, QA_F32, 2, N, 5, 1}, {QA_T_LATENT_EPS, QA_F32, 2, N, 1, 1},
        {QA_T_LATENT_C, QA_F32, 2, N, QA_NUM_GAITS, 1}, {QA_T_REW, QA_F32, 1, N, 1, 1},
        {QA_T_RESET, QA_I64, 1, N, 1, 1}, {QA_T_TIME_OUT, QA_U8, 1, N, 1, 1},
        {QA_T_EPISODE_LENGTH, QA_I64, 1, N, 1, 1}, {QA_T_EPISODE_SUMS, QA_F32, 2, QA_NUM_REWARDS, N, 1},
        {QA_T_EPISODE_STATS, QA_F32, 2, 2, 16, 1}, {QA_T_LAST_CONTACTS, QA_U8, 2, N, 4, 1},
        {QA_T_CONTACT_FILT, QA_U8, 2, N, 4, 1}, {QA_T_FEET_FORCE, QA_F32, 2, N, 4, 1},
        {QA_T_BASE_LIN_VEL, QA_F32, 2, N, 3, 1}, {QA_T_BASE_ANG_VEL, QA_F32, 2, N, 3, 1},
        {QA_T_PROJECTED_GRAVITY, QA_F32, 2, N, 3, 1}, {QA_T_RPY, QA_F32, 2, N, 3, 1},
        {QA_T_MOTOR_STRENGTH, QA_F32, 3, 2, N, 12}, {QA_T_MASS_PARAMS, QA_F32, 2, N, 4, 1},
        {QA_T_FRICTION, QA_F32, 1, N, 1, 1}, {QA_T_ENV_ORIGINS, QA_F32, 2, N, 3, 1},
        {QA_T_BASE_INERTIA, QA_F32, 2, N, 10, 1}, {QA_T_PRIOR_PARAMETERS, QA_F32, 1, QA_NUM_GAITS, 1, 1},
        {QA_T_MOCAP_FRAMES, QA_F32, 2, F, QA_MOCAP_FRAME, 1}, {QA_T_HEIGHT_SAMPLES, QA_I16, 2, HR, HC, 1},
        {QA_T_SCAN_HEIGHT, QA_F32, 1, N, 1, 1}, {QA_T_FOOT_IMPULSE, QA_F32, 3, N, 4, 3},
        {QA_T_MOCAP_CLIPS, QA_F64, 2, QA_MAX_MOCAP_CLIPS, QA_MOCAP_CLIP, 1},
        {QA_T_RIGID_BODY_STATE, QA_F32, 3, cfg->export_body_state ? N : 1, QA_NUM_BODIES_ABI, 13},
        {QA_T_STEP_TICKET, QA_I32, 1, 4, 1, 1},
        {QA_T_CEILING_SAMPLES, QA_I16, 2, (cfg->terrain_type == 1 && cfg->hf_ceiling) ? HR : 1, (cfg->terrain_type == 1 && cfg->hf_ceiling) ? HC : 1, 1},
        {QA_T_OBST_DESC, QA_F32, 3, (cfg->terrain_type == 1 && cfg->articulated_obstacles) ? N : 1, QA_OBST_PER_ENV, QA_OBST_DESC},
        {QA_T_OBST_STATE, QA_F32, 3, (cfg->terrain_type == 1 && cfg->articulated_obstacles) ? N : 1, QA_OBST_PER_ENV, QA_OBST_STATE},
    };
    static_assert(sizeof(specs) / sizeof(specs[0]) == QA_T_COUNT, "every tensor needs a spec");
    memset(L, 0, sizeof(*L));
    for (const Spec &s : specs) { L->dtype[s.t] = s.dt; L->ndim[s.t] = s.nd; L->shape[s.t][0] = s.a; L->shape[s.t][1] = s.b; L->shape[s.t][2] = s.c; }
    int64_t off = 0;
    for (int t = 0; t < QA_T_COUNT; ++t) {
        off = (off + 255) & ~(int64_t)255;
        L->off[t] = off;
        off += L->shape[t][0] * L->shape[t][1] * L->shape[t][2] * dtype_size(L->dtype[t]);
    }
    L->total = (off + 255) & ~(int64_t)255;
}

// device pointers into the arena
struct Ptrs {
    float *root, *dof, *cforce, *rbpos, *torques, *torques_org, *actions, *last_actions, *last_dof_vel,
        *last_torques_org, *last_root_vel, *action_hist, *obs, *obs_disc, *obs_disc_term, *commands,
        *latent_eps, *latent_c, *rew, *episode_sums, *episode_stats, *feet_force, *base_lin_vel, *base_ang_vel,
        *proj_grav, *rpy, *motor_strength, *mass_params, *friction, *env_origins, *base_inertia, *prior, *mocap, *foot_impulse, *scan_height;
    float *rbstate;
    double *mocap_clips;
    int32_t *ticket;
    int16_t *height_samples, *ceil_samples;
    float *obst_desc, *obst_state;
    int64_t *reset, *episode_length;
    uint8_t *time_out, *last_contacts, *contact_filt;
};

struct qa_sim {
    long long *prof = nullptr;
    int lanes = 4;                 // lanes per env of the step / simulate kernels (one quad: lane & 3 = leg)
    int lean = 0;                  // qa_set_lean_exports
    int num_cus = 256;             // compute units of the device the arena lives on (launch_env_step: helper wavefronts by launch size)
    qa_config cfg;
    Layout L;
    char *arena;
    Ptrs p;
    int32_t mocap_first[QA_NUM_GAITS + 1];
    double clip_stage[QA_MAX_MOCAP_CLIPS * QA_MOCAP_CLIP];     // host staging of the clip table (must outlive the async copy)
};

struct MocapIdx { int32_t on; };      // clips uploaded; the per-gait clip ranges live in column 5 of QA_T_MOCAP_CLIPS

thread_local char qa_err_buf[512] = "";      // shared with qa_learner.hip
#define g_err qa_err_buf
static int fail_hip(hipError_t e, const char *what) {
    snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
    return QA_E_DEVICE;
}
#define HIP_TRY(x) do { hipError_t _e = (x); if (_e != hipSuccess) return fail_hip(_e, #x); } while (0)

// ------------------------------------------------------------------ shared device pieces
// Workgroups of the env kernels are ONE wavefront (QA_BLOCK == 64), and a wavefront's LDS operations execute in issue
// order, so lanes exchange data through LDS without s_barrier.  What is needed is (a) that the LDS queue has drained and
// (b) that the compiler does not move LDS accesses across the point.  __syncthreads() would additionally wait for
// vmcnt(0), i.e. drain every outstanding global load and store at each exchange -- 8 full memory drains in the
// observation phase alone.
#ifndef QA_ENV_HELPERS_DEFAULT
#define QA_ENV_HELPERS_DEFAULT (-1)      // -1: by launch size (launch_env_step), 0: never, 1: always
#endif
static_assert(QA_BLOCK == 64, "wave_lds_sync() assumes single-wavefront workgroups");
__device__ __forceinline__ void wave_lds_sync() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
constexpr int QA_TBL_PER = (QA_TBL_FLOATS + QA_BLOCK - 1) / QA_BLOCK;
// the table's way into LDS in two halves, so that a kernel can put its other loads between them (one round trip instead of two)
__device__ __forceinline__ void stage_table_load(float (&v)[QA_TBL_PER]) {
#pragma unroll
    for (int r = 0; r < QA_TBL_PER; ++r) { int i = (threadIdx.x & (QA_BLOCK - 1)) + QA_BLOCK * r; v[r] = (i < QA_TBL_FLOATS) ? c_tbl[i] : 0.f; }   // loads in flight together
}
__device__ __forceinline__ void stage_table_store(float *s_tbl, const float (&v)[QA_TBL_PER]) {
#pragma unroll
    for (int r = 0; r < QA_TBL_PER; ++r) { int i = (threadIdx.x & (QA_BLOCK - 1)) + QA_BLOCK * r; if (i < QA_TBL_FLOATS) s_tbl[i] = v[r]; }
    wave_lds_sync();
}
__device__ __forceinline__ void stage_table(float *s_tbl) {
    float v[QA_TBL_PER];
    stage_table_load(v);
    stage_table_store(s_tbl, v);
}

// legged_robot.py:532-540 + :474-530, executed redundantly by the 4 lanes of the quad
__device__ __forceinline__ void resample_commands(const qa_config &c, const float *prior, int env, int64_t step, int stream,
                                                  float cmd[5], float &eps, int &gait) {
    F4 u0 = rng4(c.seed, env + c.env_id_offset, step, stream, 0), u1 = rng4(c.seed, env + c.env_id_offset, step, stream, 1);
    eps = u0.v[1] * 2.0f - 1.0f;
    float z[5], zmax = -1e30f, sum = 0.f;
#pragma unroll
    for (int g = 0; g < 5; ++g) { z[g] = prior[g] / c.latent_temperature; zmax = fmaxf(zmax, z[g]); }
#pragma unroll
    for (int g = 0; g < 5; ++g) { z[g] = expf(z[g] - zmax); sum += z[g]; }
    gait = 4; float acc = 0.f; bool found = false;
#pragma unroll
    for (int g = 0; g < 5; ++g) { acc += z[g] / sum; if (!found && u0.v[0] < acc) { gait = g; found = true; } }
    // per-gait ranges blended arithmetically: a compare/select chain over the config arrays is folded into a select of
    // ADDRESSES, and a dynamically addressed kernel-argument record is kept as a 1 KB private (scratch) copy
    float vxl = 0.f, vxh = 0.f, vyl = 0.f, vyh = 0.f, wl = 0.f, wh = 0.f;
#pragma unroll
    for (int g = 0; g < 5; ++g) {
        const float m = gait == g ? 1.0f : 0.0f;
        vxl = fmaf(m, c.lin_vel_x[g][0], vxl); vxh = fmaf(m, c.lin_vel_x[g][1], vxh); vyl = fmaf(m, c.lin_vel_y[g][0], vyl);
        vyh = fmaf(m, c.lin_vel_y[g][1], vyh); wl = fmaf(m, c.ang_vel_yaw[g][0], wl); wh = fmaf(m, c.ang_vel_yaw[g][1], wh);
    }
    float vx = (vxh - vxl) * u0.v[2] + vxl, vy = (vyh - vyl) * u0.v[3] + vyl, wz = (wh - wl) * u1.v[0] + wl;
    bool jump = gait == QA_NUM_GAITS - 1;
    float hj = ((c.jump_height[1] - c.jump_height[0]) * u1.v[1] + c.jump_height[0]) * (jump ? 1.0f : 0.0f);
    float hl = ((c.locomotion_height[1] - c.locomotion_height[0]) * u1.v[2] + c.locomotion_height[0]) * (jump ? 0.0f : 1.0f);
    cmd[0] = vx * (fabsf(vx) > c.lin_vel_x_clip ? 1.0f : 0.0f);
    cmd[1] = vy * (fabsf(vy) > c.lin_vel_y_clip ? 1.0f : 0.0f);
    cmd[2] = wz * (fabsf(wz) > c.ang_vel_yaw_clip ? 1.0f : 0.0f);
    cmd[3] = hj; cmd[4] = hl;
}

// isaacgym torch_utils.quat_rotate / quat_rotate_inverse (sign = +1 / -1), xyzw
__device__ __forceinline__ V3 quat_rot(float qx, float qy, float qz, float qw, V3 v, float sign) {
    float s = 2.0f * qw * qw - 1.0f;
    V3 qv = v3(qx, qy, qz);
    V3 a = s * v, b = (2.0f * qw) * cross(qv, v), cc = (2.0f * dot(qv, v)) * qv;
    return v3(a.x + sign * b.x + cc.x, a.y + sign * b.y + cc.y, a.z + sign * b.z + cc.z);
}

// quaternion_slerp of bbc/rsl_rl/utils/utils.py:126-159 for one pair (fp32 like the torch tensors it runs on), masks in the
// order the reference applies them: identical / zero-angle pairs -> q0; fraction ~ 1 -> q1; fraction ~ 0 -> q0; otherwise
// q0 sin((1-f) a) / a + (+-q1) sin(f a) / a with a = acos|q0.q1| -- the reference divides by the ANGLE, not by its sine, so
// the result is not exactly unit length; it is written to the root state as is, like the reference does.
__device__ __forceinline__ void mocap_slerp(const float *q0, const float *q1, float f, float out[4]) {
    float d = q0[0] * q1[0] + q0[1] * q1[1] + q0[2] * q1[2] + q0[3] * q1[3];
    const bool at_zero = fabsf(f) <= 1e-8f, at_one = fabsf(f - 1.0f) <= (1e-8f + 1e-5f);
    const bool same = fabsf(fabsf(d) - 1.0f) < 8.8817842e-16f;
    const float sg = d < 0.f ? -1.0f : 1.0f;
    d = clampf(d * sg, -1.0f, 1.0f);
    const float ang = acosf(d);
    const bool tiny = fabsf(ang) < 8.8817842e-16f;
    const float isin = 1.0f / ang;
    const float w0 = sinf((1.0f - f) * ang) * isin, w1 = sinf(f * ang) * isin * sg;
#pragma unroll
    for (int i = 0; i < 4; ++i) out[i] = (same || tiny) ? q0[i] : (at_one ? q1[i] : (at_zero ? q0[i] : q0[i] * w0 + q1[i] * w1));
}

// reset of one env by its quad (legged_robot.py:178-240).  Updates st and the command/latent registers.
__device__ __forceinline__ void reset_env(const qa_config &c, const Ptrs &p, const MocapIdx &mi, int env, int leg, int64_t step,
                                          EnvState &st, float cmd[5], float &eps, int &gait) {
    resample_commands(c, p.prior, env, step, RS_CMD_RESET, cmd, eps, gait);
    const float ox = p.env_origins[3 * env], oy = p.env_origins[3 * env + 1], oz = p.env_origins[3 * env + 2];
    if (c.reset_mode == 1 && mi.on) {
        // MotionLoader.get_full_frame_batch (motion_loader.py:461-474): clip ~ MotionWeight inside the gait, time ~ U over the
        // clip's sampling range, the two bracketing frames blended (get_full_frame_at_time_batch :410-447); index arithmetic in
        // float64 as numpy does it, the blend itself in fp32 as torch does it
        F4 u = rng4(c.seed, env + c.env_id_offset, step, RS_RESET, 0);
        const int c0 = (int)p.mocap_clips[QA_MOCAP_CLIP * gait + 5], c1 = (int)p.mocap_clips[QA_MOCAP_CLIP * (gait + 1) + 5];   // first_clip[] rides in column 5
        int clip = c1 - 1;
        for (int i = c1 - 2; i >= c0; --i) if ((double)u.v[0] < p.mocap_clips[QA_MOCAP_CLIP * i + 4]) clip = i;     // first clip with u < cdf
        const double *ct = p.mocap_clips + QA_MOCAP_CLIP * clip;
        const double nf = ct[1];
        const double t = fmax(1e-7, ct[3] * (double)u.v[1]);
        const double pn = t / ct[2] * nf;
        const double lo = floor(pn), hi = ceil(pn);
        const float b = (float)(pn - lo);
        const float *f0 = p.mocap + ((int64_t)ct[0] + (int64_t)lo) * QA_MOCAP_FRAME, *f1 = p.mocap + ((int64_t)ct[0] + (int64_t)hi) * QA_MOCAP_FRAME;
        auto mix = [&](int i) { return (1.0f - b) * f0[i] + b * f1[i]; };
        st.pos = v3(mix(0) + ox, mix(1) + oy, mix(2) + oz);
        float q[4];
        mocap_slerp(f0 + 3, f1 + 3, b, q);
        st.qx = q[0]; st.qy = q[1]; st.qz = q[2]; st.qw = q[3];
        st.vw = quat_rot(q[0], q[1], q[2], q[3], v3(mix(19), mix(20), mix(21)), 1.0f);
        st.ww = quat_rot(q[0], q[1], q[2], q[3], v3(mix(22), mix(23), mix(24)), 1.0f);
#pragma unroll
        for (int k = 0; k < 3; ++k) { st.q[k] = mix(7 + 3 * leg + k); st.qd[k] = mix(25 + 3 * leg + k); }
    } else {
        float u[20];
#pragma unroll
        for (int b = 0; b < 5; ++b) { F4 t = rng4(c.seed, env + c.env_id_offset, step, RS_RESET, b); u[4 * b] = t.v[0]; u[4 * b + 1] = t.v[1]; u[4 * b + 2] = t.v[2]; u[4 * b + 3] = t.v[3]; }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            float uj = leg == 0 ? u[k] : (leg == 1 ? u[3 + k] : (leg == 2 ? u[6 + k] : u[9 + k]));
            st.q[k] = c.default_dof_pos[k] * ((1.5f - 0.5f) * uj + 0.5f);    // default pose is the same for every leg
            st.qd[k] = 0.f;
        }
        st.pos = v3(c.init_pos[0] + ox, c.init_pos[1] + oy, c.init_pos[2] + oz);
        if (c.reset_xy_jitter > 0.0f) { st.pos.x += (2.0f * u[18] - 1.0f) * c.reset_xy_jitter; st.pos.y += (2.0f * u[19] - 1.0f) * c.reset_xy_jitter; }   // :622-625
        st.qx = 0; st.qy = 0; st.qz = 0; st.qw = 1;
        st.vw = v3(u[12] - 0.5f, u[13] - 0.5f, u[14] - 0.5f);
        st.ww = v3(u[15] - 0.5f, u[16] - 0.5f, u[17] - 0.5f);
    }
}

__device__ __forceinline__ TerrainView terrain_view(const qa_config &c, const Ptrs &p, const float *patch) {
    TerrainView T;
    T.patch = patch; T.samples = p.height_samples; T.ceil = c.hf_ceiling ? p.ceil_samples : nullptr; T.ix0 = 0; T.iy0 = 0; T.rows = c.hf_rows; T.cols = c.hf_cols;
    T.border = c.hf_border; T.hscale = c.hf_hscale; T.inv_hscale = 1.0f / c.hf_hscale; T.vscale = c.hf_vscale;
    T.ob = nullptr; T.ob_acc = nullptr;
    T.cax = 0; T.cay = 0; T.fax = 0.f; T.fay = 0.f; T.ancx = 0.f; T.ancy = 0.f;
    return T;
}

// articulated obstacles of an env -> LDS: lane k < 3 of the quad loads slot k (8 descriptor floats + q, q_dot, -, damping) and clears the
// slot's force accumulator; QA_OB_LDS floats per env
#define QA_OB_LDS (12 * QA_OBST_PER_ENV + 4)
QA_DEV void stage_obstacles(const Ptrs &p, int env, int leg, float *rec, float ancx, float ancy) {
    if (leg < QA_OBST_PER_ENV) {
        const float *d = p.obst_desc + ((int64_t)env * QA_OBST_PER_ENV + leg) * QA_OBST_DESC;
        const float *st = p.obst_state + ((int64_t)env * QA_OBST_PER_ENV + leg) * QA_OBST_STATE;
#pragma unroll
        for (int i = 0; i < QA_OBST_DESC; ++i) rec[12 * leg + i] = d[i];
        // the obstacle's centre relative to the step's anchor (the contact queries work in env-local coordinates, qa_physics.h)
        rec[12 * leg] = (float)((double)d[0] - (double)ancx); rec[12 * leg + 1] = (float)((double)d[1] - (double)ancy);
        rec[12 * leg + 8] = st[0]; rec[12 * leg + 9] = st[1]; rec[12 * leg + 10] = 0.f; rec[12 * leg + 11] = st[3];
        rec[12 * QA_OBST_PER_ENV + leg] = 0.f;
    }
}

// ------------------------------------------------------------------ the fused env step
struct StepArgs { qa_config c; Ptrs p; MocapIdx mi; const float *actions; int delay; int64_t step; int64_t *step_ptr; long long *prof; };
// LEAN (qa_set_lean_exports, MODE 0 only): bit 0 = do not write the tensors nothing between two steps of a training run reads (seam-1 / logging
// exports: CONTACT_FORCES, RIGID_BODY_POS, TORQUES, TORQUES_ORG, ACTIONS, BASE_LIN_VEL, BASE_ANG_VEL, PROJECTED_GRAVITY, RPY, FEET_FORCE,
// CONTACT_FILT, SCAN_HEIGHT) and keep only the two newest slots of the action ring (all a delay <= 1 can reach); bit 1 = no discriminator
// observations either (OBS_DISC, OBS_DISC_TERM: read by the AMP runner only)
#ifdef QA_SUBPROF
#define QA_STAMP(k) do { } while (0)      // the substep stamps own the buffer in this build
#else
#define QA_STAMP(k) do { if (qa_prof && tix == 0) qa_prof[bix * 16 + (k)] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
#endif

#define S_PROP 0        // 57  proprioception (noise-free)
#define S_HEAD 57       // 90  obs[0:90] with noise
#define S_TAIL 147      // 11  obs[660:671]
#define S_DISC 158      // 49
#define S_DISCT 207     // 49
#define S_FLAGS 256     // [0] = refill history (episode length <= 1)
#define S_ENV 260       // floats of LDS staging per env
#define S_ROW 676       // assembled observation row per env (+ up to 3 floats of phase padding in front)
#define OBS_GROUP 4     // rows assembled and streamed out per pass

// The 671-float observation row of env `ge` starts at a 4-byte-aligned address whose 16-byte phase is
// head = floats until the next 16-B boundary (0..3).  The LDS copy of the row is stored with the SAME phase
// (row base shifted by pad = (4 - head) & 3), so the body moves as ds_read_b128 -> global_store_dwordx4:
// 167 vectors = 2 full wave iterations + 39 lanes, plus `head` leading and 3 - head trailing scalars.
__device__ __forceinline__ int obs_row_head(const float *dst) { return (int)(((16u - ((uintptr_t)dst & 15u)) & 15u) >> 2); }

__device__ __forceinline__ void store_obs_row(float *dst, const float *row, int head) {
    const int lane = threadIdx.x & (QA_BLOCK - 1);
    const float4 *s4 = reinterpret_cast<const float4 *>(row + head);      // 16-B aligned by construction
    float4 *d4 = reinterpret_cast<float4 *>(dst + head);
    float4 a = s4[lane], b = s4[lane + 64];
    d4[lane] = a; d4[lane + 64] = b;
    if (lane < 167 - 128) d4[lane + 128] = s4[lane + 128];
    if (lane < head) dst[lane] = row[lane];
    if (lane < 3 - head) dst[head + 668 + lane] = row[head + 668 + lane];
}

// QA_T_RIGID_BODY_STATE rows of this lane's leg (hip, thigh, calf, foot) and -- lane 0 -- of base / Head_upper / Head_lower:
// body-origin position, orientation (xyzw), linear velocity of the origin and angular velocity, world frame, from the NEW
// state.  Link k is moved by joints 0..k, its origin (= joint k) by joints 0..k-1; the foot is fixed to the calf.
__device__ __forceinline__ void quat_mul(const float a[4], const float b[4], float o[4]) {     // xyzw, a (x) b
    o[0] = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
    o[1] = a[3] * b[1] - a[0] * b[2] + a[1] * b[3] + a[2] * b[0];
    o[2] = a[3] * b[2] + a[0] * b[1] - a[1] * b[0] + a[2] * b[3];
    o[3] = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
}
__device__ __forceinline__ void write_body_state(float *rows, const EnvState &st, const M3 &R, const V3 org[4], int leg) {
    const float qb[4] = {st.qx, st.qy, st.qz, st.qw};
    float s1, c1, s2, c2, s3, c3;
    sincosf(0.5f * st.q[0], &s1, &c1); sincosf(0.5f * st.q[1], &s2, &c2); sincosf(0.5f * (st.q[1] + st.q[2]), &s3, &c3);
    const float qh[4] = {s1, 0.f, 0.f, c1}, qy2[4] = {0.f, s2, 0.f, c2}, qy3[4] = {0.f, s3, 0.f, c3};
    float ql[3][4], t[4];
    quat_mul(qb, qh, ql[0]); quat_mul(qh, qy2, t); quat_mul(qb, t, ql[1]); quat_mul(qh, qy3, t); quat_mul(qb, t, ql[2]);
    float sq, cq; sincosf(st.q[0], &sq, &cq);
    const V3 ax[3] = {v3(1, 0, 0), v3(0, cq, sq), v3(0, cq, sq)};
    const V3 wB = mulT(R, st.ww), vB = mulT(R, st.vw);        // base twist in base axes
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        const int kq = b < 3 ? b : 2;                          // link whose frame the body shares
        V3 w = wB, v = vB + cross(wB, org[b]);
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            if (j <= kq) w = w + st.qd[j] * ax[j];
            if (j < b) v = v + st.qd[j] * cross(ax[j], org[b] - org[j]);
        }
        const V3 pw = mul(R, org[b]) + st.pos, vw = mul(R, v), ww = mul(R, w);
        float *r = rows + 13 * (3 + 4 * leg + b);
        r[0] = pw.x; r[1] = pw.y; r[2] = pw.z; r[3] = ql[kq][0]; r[4] = ql[kq][1]; r[5] = ql[kq][2]; r[6] = ql[kq][3];
        r[7] = vw.x; r[8] = vw.y; r[9] = vw.z; r[10] = ww.x; r[11] = ww.y; r[12] = ww.z;
    }
    if (leg == 0) {
        const V3 offs[3] = {v3(0, 0, 0), v3(0.285f, 0.f, 0.01f), v3(0.293f, 0.f, -0.06f)};
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            const V3 pw = mul(R, offs[b]) + st.pos, vw = st.vw + cross(st.ww, mul(R, offs[b]));
            float *r = rows + 13 * b;
            r[0] = pw.x; r[1] = pw.y; r[2] = pw.z; r[3] = qb[0]; r[4] = qb[1]; r[5] = qb[2]; r[6] = qb[3];
            r[7] = vw.x; r[8] = vw.y; r[9] = vw.z; r[10] = st.ww.x; r[11] = st.ww.y; r[12] = st.ww.z;
        }
    }
}

// The per-env scalars and small rows a step ends with, as 4-byte words (integers as bit patterns): one record, so that the wavefront that
// computed them and the wavefront that stores them (qa_env_step_kernel with helper wavefronts) need not be the same
enum { SO_ST = 0, SO_Q = 13, SO_QD = 16, SO_TAU = 19, SO_TORG = 22, SO_ACT = 25, SO_FIMP = 28, SO_SCANH = 31, SO_REW = 32, SO_RESET = 33, SO_TIMEOUT = 34, SO_EPL = 35,
       SO_BLV = 37, SO_BAV = 40, SO_PG = 43, SO_RPY = 46, SO_LAV = 49, SO_LAW = 52, SO_CMDDIRTY = 55, SO_GAIT = 56, SO_EPS = 57, SO_CMD = 58, SO_FFN = 63, SO_CONTACT = 64,
       SO_CFILT = 65, SO_ESUM = 66, SO_WORDS = (66 + QA_NUM_REWARDS + 3) / 4 * 4 };
static_assert(SO_WORDS == 66 + QA_NUM_REWARDS, "pad the record to whole 16-byte records (and zero the padding) when the reward count changes");
struct ScalarOut { float w[SO_WORDS]; };
static_assert(SO_WORDS / 4 <= QA_MAIL_F4, "the record travels in a lane's mail");
template <bool PLANE, int LEAN>
__device__ __forceinline__ void write_scalars(const Ptrs &p, const int N, const int env, const int leg, const bool valid, const ScalarOut &so) {
    const float *w = so.w;
    const int reset = __float_as_int(w[SO_RESET]);
    if (valid) {
        float *rt = p.root + (int64_t)env * 13;
        if (leg == 0) {
#pragma unroll
            for (int i = 0; i < 13; ++i) rt[i] = w[SO_ST + i];
            if (!PLANE && !(LEAN & 1)) p.scan_height[env] = w[SO_SCANH];
            p.rew[env] = w[SO_REW]; p.reset[env] = reset; p.time_out[env] = (uint8_t)__float_as_int(w[SO_TIMEOUT]);
            p.episode_length[env] = (int64_t)(((uint64_t)(uint32_t)__float_as_int(w[SO_EPL + 1]) << 32) | (uint64_t)(uint32_t)__float_as_int(w[SO_EPL]));
            if (!(LEAN & 1)) {
                float *o3;
                o3 = p.base_lin_vel + (int64_t)env * 3; o3[0] = w[SO_BLV]; o3[1] = w[SO_BLV + 1]; o3[2] = w[SO_BLV + 2];
                o3 = p.base_ang_vel + (int64_t)env * 3; o3[0] = w[SO_BAV]; o3[1] = w[SO_BAV + 1]; o3[2] = w[SO_BAV + 2];
                o3 = p.proj_grav + (int64_t)env * 3; o3[0] = w[SO_PG]; o3[1] = w[SO_PG + 1]; o3[2] = w[SO_PG + 2];
                o3 = p.rpy + (int64_t)env * 3; o3[0] = w[SO_RPY]; o3[1] = w[SO_RPY + 1]; o3[2] = w[SO_RPY + 2];
            }
            float *lr = p.last_root_vel + (int64_t)env * 6;
#pragma unroll
            for (int i = 0; i < 3; ++i) { lr[i] = w[SO_LAV + i]; lr[3 + i] = w[SO_LAW + i]; }
            if (__float_as_int(w[SO_CMDDIRTY])) {
                const int gait = __float_as_int(w[SO_GAIT]);
#pragma unroll
                for (int i = 0; i < 5; ++i) { p.commands[(int64_t)env * 5 + i] = w[SO_CMD + i]; p.latent_c[(int64_t)env * 5 + i] = (gait == i) ? 1.0f : 0.0f; }
                p.latent_eps[env] = w[SO_EPS];
            }
        }
        float *d = p.dof + (int64_t)env * 24 + 6 * leg;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int64_t j = (int64_t)env * 12 + 3 * leg + k;
            d[2 * k] = w[SO_Q + k]; d[2 * k + 1] = w[SO_QD + k];
            if (!(LEAN & 1)) { p.torques[j] = w[SO_TAU + k]; p.torques_org[j] = w[SO_TORG + k]; p.actions[j] = w[SO_ACT + k]; }
            p.last_actions[j] = w[SO_ACT + k]; p.last_dof_vel[j] = w[SO_QD + k]; p.last_torques_org[j] = w[SO_TORG + k];   // :158-161
        }
        if (!(LEAN & 1)) p.feet_force[(int64_t)env * 4 + leg] = w[SO_FFN];
#pragma unroll
        for (int k = 0; k < 3; ++k) p.foot_impulse[(int64_t)env * 12 + 3 * leg + k] = reset ? 0.f : w[SO_FIMP + k];
        p.last_contacts[(int64_t)env * 4 + leg] = (uint8_t)__float_as_int(w[SO_CONTACT]);
        if (!(LEAN & 1)) p.contact_filt[(int64_t)env * 4 + leg] = (uint8_t)__float_as_int(w[SO_CFILT]);
#pragma unroll
        for (int r = 0; r < QA_NUM_REWARDS; ++r) if ((r & 3) == leg) p.episode_sums[(int64_t)r * N + env] = w[SO_ESUM + r];
        if (reset) {   // action history is zeroed by reset_idx (:227)
            float *ah = p.action_hist + (int64_t)env * (QA_ACTION_BUF_LEN * 12) + 3 * leg;
#pragma unroll
            for (int r = (LEAN & 1) ? QA_ACTION_BUF_LEN - 2 : 0; r < QA_ACTION_BUF_LEN; ++r) { ah[12 * r] = 0.f; ah[12 * r + 1] = 0.f; ah[12 * r + 2] = 0.f; }
        }
    }
}

// What the physics phase of the fused step hands to post_physics_step, per lane (lane = leg): the new state, the torques of
// the LAST substep, the contact forces of the leg's bodies and of the base.  qa_post_physics_kernel fills the same record from
// the arena instead, which makes the post-physics device code testable on its own against the reference's fixtures.
struct PostIn {
    EnvState st;
    float act[3], raw_act[3], tau[3], tau_org[3], sp[3], sd[3], fimp[3];
    V3 foot_f, hip_f, thigh_f, calf_f, base_f;
    V3 foot_w;
    float fric;
    // r5: what the post-physics phase reads of the PREVIOUS step (last_* for three rate rewards, the running episode sums) is loaded with the
    // rest of the state in front of the physics, not in the middle of the reward phase where a wavefront alone on its SIMD has nothing to do
    // for the ~2 us of each round trip
    float last_act[3], last_torg[3], last_qd[3], esum[QA_NUM_REWARDS];
    float cmd[5], latc[5], eps; int64_t epl0; uint8_t last_contact;
};
// (the height-field build has no registers to spare for these ~40 values across the substeps -- they would spill to scratch -- and loads them
// right in front of the post-physics phase, as r4 did)
QA_DEV void post_in_preload(PostIn &in, const Ptrs &p, int env, int leg, int N) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int64_t j = (int64_t)env * 12 + 3 * leg + k;
        in.last_act[k] = p.last_actions[j]; in.last_torg[k] = p.last_torques_org[j]; in.last_qd[k] = p.last_dof_vel[j];
    }
#pragma unroll
    for (int r = 0; r < QA_NUM_REWARDS; ++r) in.esum[r] = p.episode_sums[(int64_t)r * N + env];
#pragma unroll
    for (int i = 0; i < 5; ++i) { in.cmd[i] = p.commands[(int64_t)env * 5 + i]; in.latc[i] = p.latent_c[(int64_t)env * 5 + i]; }
    in.eps = p.latent_eps[env]; in.epl0 = p.episode_length[env]; in.last_contact = p.last_contacts[(int64_t)env * 4 + leg];
}

template <bool PLANE, int LPE, int LEAN = 0, int HELP = 0>
__device__ __forceinline__ void post_physics_phase(const qa_config &c, const Ptrs &p, const MocapIdx &mi, long long *qa_prof, PostIn &in, const TerrainView &T,
                                                   const float *tbl, float *s_stage, float *s_rows, const int tix, const int bix, const int env,
                                                   const int leg, const bool valid, const bool owner, const int le, const int64_t step, f4 *mail = nullptr) {
    // (the kernel argument record is NOT passed as a whole: a reference to it makes the compiler keep a 1 KB private copy)
    constexpr int EPB = QA_BLOCK / LPE;                // envs per wavefront
    const int N = c.num_envs;
    EnvState &st = in.st;
    float (&act)[3] = in.act, (&raw_act)[3] = in.raw_act, (&tau)[3] = in.tau, (&tau_org)[3] = in.tau_org, (&sp)[3] = in.sp, (&sd)[3] = in.sd, (&fimp)[3] = in.fimp;
    const V3 foot_f = in.foot_f, hip_f = in.hip_f, thigh_f = in.thigh_f, calf_f = in.calf_f, base_f = in.base_f, foot_w = in.foot_w;
    const float fric = in.fric;
    const float q0[3] = {c.default_dof_pos[0], c.default_dof_pos[1], c.default_dof_pos[2]};
    QA_STAMP(4);
    // =========================== post_physics_step (legged_robot.py:124-166) ===========================
    int64_t epl = in.epl0 + 1;
    const int64_t common = step + 1;
    V3 blv = quat_rot(st.qx, st.qy, st.qz, st.qw, st.vw, -1.f), bav = quat_rot(st.qx, st.qy, st.qz, st.qw, st.ww, -1.f);
    V3 pg = quat_rot(st.qx, st.qy, st.qz, st.qw, v3(0, 0, -1), -1.f);
    float roll = atan2f(2.0f * (st.qw * st.qx + st.qy * st.qz), 1.0f - 2.0f * (st.qx * st.qx + st.qy * st.qy));
    float pitch = asinf(clampf(2.0f * (st.qw * st.qy - st.qz * st.qx), -1.0f, 1.0f));
    float yaw = atan2f(2.0f * (st.qw * st.qz + st.qx * st.qy), 1.0f - 2.0f * (st.qy * st.qy + st.qz * st.qz));
    const float ffn = sqrtf(dot(foot_f, foot_f));
    const uint8_t contact = ffn > 2.0f;
    const uint8_t cfilt = contact | in.last_contact;

    // commands / latents in registers (replicated)
    float cmd[5], eps; int gait = 0;
#pragma unroll
    for (int i = 0; i < 5; ++i) cmd[i] = in.cmd[i];
    eps = in.eps;
    {
        float best = in.latc[0];
#pragma unroll
        for (int g = 1; g < 5; ++g) { float v = in.latc[g]; if (v > best) { best = v; gait = g; } }
    }
    bool cmd_dirty = false;
    if (__any(epl % c.resampling_steps == 0)) {
        float c2[5], e2; int g2;
        resample_commands(c, p.prior, env, step, RS_CMD, c2, e2, g2);
        if (epl % c.resampling_steps == 0) {
#pragma unroll
            for (int i = 0; i < 5; ++i) cmd[i] = c2[i];
            eps = e2; gait = g2; cmd_dirty = true;
        }
    }
    if (c.push_robots && (common % c.push_interval == 0)) {   // uniform over the grid
        F4 u = rng4(c.seed, env + c.env_id_offset, step, RS_PUSH, 0);
        st.vw.x = (c.max_push_vel_xy - -c.max_push_vel_xy) * u.v[0] + -c.max_push_vel_xy;
        st.vw.y = (c.max_push_vel_xy - -c.max_push_vel_xy) * u.v[1] + -c.max_push_vel_xy;
    }
    // measured_heights (:469-470): before any reset, like the reference's callback
    const float scan_h = PLANE ? 0.0f : scan_center_height(T, st.pos, st.qz, st.qw);
    // ---- check_termination :168-176
    int term_c = (sqrtf(dot(hip_f, hip_f)) > 1.0f) ? 1 : 0;
    term_c = xor_<LPE>(term_c) | (sqrtf(dot(base_f, base_f)) > 1.0f ? 1 : 0);
    int timeout = (epl > c.max_episode_length) || (st.pos.z < -6.0f);
    {
        float chk = st.pos.x + st.pos.y + st.pos.z + st.qx + st.qy + st.qz + st.qw + st.vw.x + st.vw.y + st.vw.z + st.ww.x + st.ww.y + st.ww.z;
        if (!isfinite(chk)) timeout = 1;          // build-added failure detection
    }
    const int reset = term_c | timeout;

    QA_STAMP(5);
    // ---- rewards :242-259, alphabetical order
    float term[QA_NUM_REWARDS];
    {
        const float dtp = c.sim_dt * (float)c.decimation;
        float s_ar = 0, s_dt = 0, s_acc = 0, s_err = 0, s_hip = 0, s_pl = 0, s_vl = 0, s_tl = 0, s_tq = 0;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int64_t j = (int64_t)env * 12 + 3 * leg + k;
            float d;
            d = in.last_act[k] - act[k]; s_ar += d * d;
            d = tau_org[k] - in.last_torg[k]; s_dt += d * d;
            d = (in.last_qd[k] - st.qd[k]) / dtp; s_acc += d * d;
            d = st.q[k] - q0[k]; s_err += d * d; if (k == 0) s_hip += d * d;
            float lo = tbl[T_LOWER + k], hi = tbl[T_UPPER + k], mid = (lo + hi) / 2, rng = hi - lo;
            float slo = mid - 0.5f * rng * c.soft_dof_pos_limit, shi = mid + 0.5f * rng * c.soft_dof_pos_limit;
            s_pl += -fminf(st.q[k] - slo, 0.f) + fmaxf(st.q[k] - shi, 0.f);
            s_vl += clampf(fabsf(st.qd[k]) - tbl[T_VELLIM + k] * c.soft_dof_vel_limit, 0.f, 1.f);
            s_tl += fmaxf(fabsf(tau_org[k]) - tbl[T_EFFORT + k] * c.soft_torque_limit, 0.f);
            s_tq += tau_org[k] * tau_org[k];
        }
        float ncol = (sqrtf(dot(thigh_f, thigh_f)) > 0.1f ? 1.f : 0.f) + (sqrtf(dot(calf_f, calf_f)) > 0.1f ? 1.f : 0.f);
        term[QA_R_ACTION_RATE] = xsum<LPE>(s_ar); term[QA_R_COLLISION] = xsum<LPE>(ncol); term[QA_R_DELTA_TORQUES] = xsum<LPE>(s_dt);
        term[QA_R_DOF_ACC] = xsum<LPE>(s_acc); term[QA_R_DOF_ERROR] = xsum<LPE>(s_err); term[QA_R_DOF_POS_LIMITS] = xsum<LPE>(s_pl);
        term[QA_R_DOF_VEL_LIMITS] = xsum<LPE>(s_vl); term[QA_R_HIP_POS] = xsum<LPE>(s_hip); term[QA_R_TORQUE_LIMITS] = xsum<LPE>(s_tl);
        term[QA_R_TORQUES] = xsum<LPE>(s_tq);
        const float root_h = st.pos.z - scan_h;
        float ej = sqrtf((cmd[3] - root_h) * (cmd[3] - root_h));
        term[QA_R_JUMP_UP_HEIGHT] = (ej < 0.05f && cmd[3] >= c.jump_height[0]) ? c.jump_goal : 0.f;
        float el = sqrtf((cmd[4] - root_h) * (cmd[4] - root_h));
        term[QA_R_LOCOMOTION_HEIGHT] = (cmd[3] > c.jump_height[0]) ? 0.f : expf(-10.0f * (el * el) / c.tracking_sigma);
        float ea = (cmd[2] - bav.z) * (cmd[2] - bav.z);
        term[QA_R_TRACKING_ANG_VEL] = expf(-ea / c.tracking_sigma);
        float elv = (cmd[0] - blv.x) * (cmd[0] - blv.x) + (cmd[1] - blv.y) * (cmd[1] - blv.y);
        term[QA_R_TRACKING_LIN_VEL] = expf(-elv / c.tracking_sigma);
    }
    float rew = 0.f;
    float esum[QA_NUM_REWARDS];
#pragma unroll
    for (int r = 0; r < QA_NUM_REWARDS; ++r) {
        esum[r] = in.esum[r];
        if (c.reward_scale_dt[r] != 0.0f) { float v = term[r] * c.reward_scale_dt[r]; rew += v; esum[r] += v; }
    }
    if (c.only_positive_rewards) rew = fmaxf(rew, 0.f);

    QA_STAMP(6);
    wave_lds_sync();          // physics scratch is dead from here on; the staging area takes its place
    // ---- terminal disc obs = previous OBS_DISC row; stage it
    float *sst = s_stage + le * S_ENV;
    if (!(LEAN & 2)) for (int i = leg; i < QA_NUM_OBS_DISC; i += 4) sst[S_DISCT + i] = p.obs_disc[(int64_t)env * QA_NUM_OBS_DISC + i];

    // ---- reset_idx :178-240
    V3 lav = st.vw, law = st.ww;      // last_root_vel is taken after the reset (:160)
    if (__any(reset)) {
        if (reset) {
#pragma unroll
            for (int r = 0; r < QA_NUM_REWARDS; ++r) {
                if ((r & 3) == leg && valid) atomicAdd(&p.episode_stats[16 * (step & 1) + r], esum[r]);
                esum[r] = 0.f;
            }
            if (leg == 0 && valid) atomicAdd(&p.episode_stats[16 * (step & 1) + 14], 1.0f);
            reset_env(c, p, mi, env, leg, step, st, cmd, eps, gait);
            cmd_dirty = true;
            epl = 0;
            lav = st.vw; law = st.ww;
        }
    }
    const bool refill = epl <= 1;

    QA_STAMP(7);
    // ---- observations :261-331
    // heading-inverse rotation of the (stale for reset envs) foot position, torch_jit_utils.py:23-76
    V3 key;
    {
        V3 rd = quat_rot(st.qx, st.qy, st.qz, st.qw, v3(1, 0, 0), 1.f);
        float heading = atan2f(rd.y, rd.x);
        float sh, ch; sincosf(-0.5f * heading, &sh, &ch);
        float hn = rsqrtf(sh * sh + ch * ch);
        V3 rel = foot_w - st.pos;
        key = quat_rot(0.f, 0.f, sh * hn, ch * hn, rel, 1.f);
    }
    const float root_h = st.pos.z - scan_h;      // post-reset z, pre-reset measured height (:261-273 after :178)
    // proprioception (57): lanes write their own joints, lane 0 the shared entries
    {
        float *pr = sst + S_PROP;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int j = 3 * leg + k;
            pr[5 + j] = (st.q[k] - q0[k]) * c.s_dof_pos;
            pr[17 + j] = st.qd[k] * c.s_dof_vel;
            pr[29 + j] = reset ? 0.0f : raw_act[k];      // reset_idx zeroes the action history first (:227)
            pr[45 + j] = 0.0f * (k == 0 ? key.x : (k == 1 ? key.y : key.z));
            sst[S_DISC + 9 + j] = (st.q[k] - q0[k]) * c.s_dof_pos;
            sst[S_DISC + 21 + j] = st.qd[k] * c.s_dof_vel;
            sst[S_DISC + 33 + j] = (k == 0 ? key.x : (k == 1 ? key.y : key.z)) * c.s_key_pos;
        }
        pr[41 + leg] = (cfilt ? 1.0f : 0.0f) - 0.5f;
        sst[S_DISC + 45 + leg] = (cfilt ? 1.0f : 0.0f) * c.s_foot_contact;
        if (leg == 0) {
            pr[0] = roll; pr[1] = pitch; pr[2] = bav.x * c.s_ang_vel; pr[3] = bav.y * c.s_ang_vel; pr[4] = bav.z * c.s_ang_vel;
            sst[S_DISC + 0] = roll; sst[S_DISC + 1] = pitch; sst[S_DISC + 2] = root_h;
            sst[S_DISC + 3] = blv.x * c.s_lin_vel_dist; sst[S_DISC + 4] = blv.y * c.s_lin_vel_dist; sst[S_DISC + 5] = blv.z * c.s_lin_vel_dist;
            sst[S_DISC + 6] = bav.x * c.s_ang_vel_dist; sst[S_DISC + 7] = bav.y * c.s_ang_vel_dist; sst[S_DISC + 8] = bav.z * c.s_ang_vel_dist;
            sst[S_FLAGS] = refill ? 1.0f : 0.0f;
            sst[S_FLAGS + 1] = reset ? 1.0f : 0.0f;
            // tail: commands, eps, one-hot gait
#pragma unroll
            for (int i = 0; i < 5; ++i) { sst[S_TAIL + i] = cmd[i]; sst[S_TAIL + 6 + i] = (gait == i) ? 1.0f : 0.0f; }
            sst[S_TAIL + 5] = eps;
        }
    }
    wave_lds_sync();
    // head of the obs row: prop + explicit + latent, with noise on the 32 noisy entries
    {
        float *hd = sst + S_HEAD;
        for (int i = leg; i < QA_NUM_PROP; i += 4) hd[i] = sst[S_PROP + i];
        if (leg == 1) { hd[57] = root_h; hd[58] = blv.x * c.s_lin_vel; hd[59] = blv.y * c.s_lin_vel; hd[60] = blv.z * c.s_lin_vel; }
        if (leg == 2) {
#pragma unroll
            for (int i = 0; i < 4; ++i) hd[61 + i] = p.mass_params[(int64_t)env * 4 + i];
            hd[65] = fric;
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) { hd[66 + 3 * leg + k] = sp[k] - 1.0f; hd[78 + 3 * leg + k] = sd[k] - 1.0f; }
    }
    wave_lds_sync();
    if (c.add_noise) {
        // draw i (0..31) -> obs index i (<29) or 58 + (i - 29); lane handles blocks 2*leg, 2*leg+1
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            F4 u = rng4(c.seed, env + c.env_id_offset, step, RS_NOISE, 2 * leg + b);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                int i = 4 * (2 * leg + b) + e;
                int idx = i < 29 ? i : 58 + (i - 29);
                // arithmetic blend, not a select chain over config fields (see resample_commands)
                const float sc = (idx < 2 ? 1.f : 0.f) * c.noise_roll_pitch + ((idx >= 2 && idx < 5) ? 1.f : 0.f) * c.noise_ang_vel + ((idx >= 5 && idx < 17) ? 1.f : 0.f) * c.noise_dof_pos +
                                 ((idx >= 17 && idx < 29) ? 1.f : 0.f) * c.noise_dof_vel + (idx >= 29 ? 1.f : 0.f) * c.noise_lin_vel;
                if (owner) sst[S_HEAD + idx] += (2.0f * u.v[e] - 1.0f) * sc;
            }
        }
    }
    QA_STAMP(8);
    // ---- per-env scalars and small rows, written by the quad (write_scalars).  With helper wavefronts the ~45 scattered dword stores -- 9.7 k of the
    // launch's 112 k cycles on a wavefront that has nothing else to issue -- go through the mail to the contact helper, which is idle by now
    {
        ScalarOut so;
        float *w = so.w;
        w[SO_ST + 0] = st.pos.x; w[SO_ST + 1] = st.pos.y; w[SO_ST + 2] = st.pos.z; w[SO_ST + 3] = st.qx; w[SO_ST + 4] = st.qy; w[SO_ST + 5] = st.qz; w[SO_ST + 6] = st.qw;
        w[SO_ST + 7] = st.vw.x; w[SO_ST + 8] = st.vw.y; w[SO_ST + 9] = st.vw.z; w[SO_ST + 10] = st.ww.x; w[SO_ST + 11] = st.ww.y; w[SO_ST + 12] = st.ww.z;
#pragma unroll
        for (int k = 0; k < 3; ++k) { w[SO_Q + k] = st.q[k]; w[SO_QD + k] = st.qd[k]; w[SO_TAU + k] = tau[k]; w[SO_TORG + k] = tau_org[k]; w[SO_ACT + k] = act[k]; w[SO_FIMP + k] = fimp[k]; }
        w[SO_SCANH] = scan_h; w[SO_REW] = rew; w[SO_RESET] = __int_as_float((int)reset); w[SO_TIMEOUT] = __int_as_float((int)timeout);
        w[SO_EPL] = __int_as_float((int)(uint32_t)((uint64_t)epl & 0xffffffffu)); w[SO_EPL + 1] = __int_as_float((int)(uint32_t)((uint64_t)epl >> 32));
        w[SO_BLV] = blv.x; w[SO_BLV + 1] = blv.y; w[SO_BLV + 2] = blv.z; w[SO_BAV] = bav.x; w[SO_BAV + 1] = bav.y; w[SO_BAV + 2] = bav.z;
        w[SO_PG] = pg.x; w[SO_PG + 1] = pg.y; w[SO_PG + 2] = pg.z; w[SO_RPY] = roll; w[SO_RPY + 1] = pitch; w[SO_RPY + 2] = yaw;
        w[SO_LAV] = lav.x; w[SO_LAV + 1] = lav.y; w[SO_LAV + 2] = lav.z; w[SO_LAW] = law.x; w[SO_LAW + 1] = law.y; w[SO_LAW + 2] = law.z;
        w[SO_CMDDIRTY] = __int_as_float(cmd_dirty ? 1 : 0); w[SO_GAIT] = __int_as_float(gait); w[SO_EPS] = eps;
#pragma unroll
        for (int i = 0; i < 5; ++i) w[SO_CMD + i] = cmd[i];
        w[SO_FFN] = ffn; w[SO_CONTACT] = __int_as_float((int)contact); w[SO_CFILT] = __int_as_float((int)cfilt);
#pragma unroll
        for (int r = 0; r < QA_NUM_REWARDS; ++r) w[SO_ESUM + r] = esum[r];
        if (HELP) {
#pragma unroll
            for (int i = 0; i < SO_WORDS / 4; ++i) mail[i] = f4{w[4 * i], w[4 * i + 1], w[4 * i + 2], w[4 * i + 3]};
        } else {
            write_scalars<PLANE, LEAN>(p, N, env, leg, valid, so);
        }
    }
    wave_lds_sync();

    // the workgroup's last barrier: behind it the bias helper's history-shift stores of these rows have been acknowledged, and the contact helper finds
    // this wavefront's scalars in the mail (qa_env_step_kernel)
    if (HELP) __syncthreads();
    QA_STAMP(9);
    // ---- wave-cooperative row writes.  The complete 671-float observation row of every env of the block is
    // assembled in LDS (s_rows), then streamed out with 16-byte stores (1 KiB per wave instruction); rows are only
    // 4-byte aligned (671 and 570 are not multiples of 4), so each copy has a <=3-float head and tail.
    const float clipo = c.clip_obs;
    const int lane = tix;
    // The history slots 0..8 of every row were written by shift_history_rows() at the start of the launch.  What is left: head
    // (prop + explicit + latent, 90 floats), the newest history frame (slot 9) and the command tail -- 158 floats, four wave stores
    // per env.  An env in the first step of an episode refills all ten slots with the current frame (wave-uniform, rare).
    // which envs of the block refill their whole history (first step of an episode): ONE LDS read per lane and a vote, instead of a
    // read + readfirstlane + branch in front of every env's stores (r5: the loop below then has no control dependence on LDS data and
    // its reads for several envs go out together)
    const unsigned long long refill_mask = __ballot(lane < EPB && s_stage[(lane < EPB ? lane : 0) * S_ENV + S_FLAGS] != 0.f);
#pragma unroll 4
    for (int e = 0; e < EPB; ++e) {
        const int ge = (int)(bix * EPB) + e;
        if (ge >= N) continue;
        const float *ss = s_stage + e * S_ENV;
        float *dst = p.obs + (int64_t)ge * QA_NUM_OBS;
        dst[lane] = clampf(ss[S_HEAD + lane], -clipo, clipo);
        if (lane < 26) dst[64 + lane] = clampf(ss[S_HEAD + 64 + lane], -clipo, clipo);
        if (lane < 57) dst[603 + lane] = clampf(ss[S_PROP + lane], -clipo, clipo);
        if (lane < 11) dst[660 + lane] = clampf(ss[S_TAIL + lane], -clipo, clipo);
        if (!(LEAN & 2) && lane < QA_NUM_OBS_DISC) {
            float dv = ss[S_DISC + lane];
            p.obs_disc[(int64_t)ge * QA_NUM_OBS_DISC + lane] = dv;
            p.obs_disc_term[(int64_t)ge * QA_NUM_OBS_DISC + lane] = (ss[S_FLAGS + 1] != 0.f) ? ss[S_DISCT + lane] : dv;
        }
    }
    for (unsigned long long m = refill_mask; m != 0; m &= m - 1) {          // rare: slots 0..8 <- the current frame (slot 9 was written above)
        const int e = __builtin_ctzll(m), ge = (int)(bix * EPB) + e;
        if (ge >= N) continue;
        const float *ss = s_stage + e * S_ENV;
        float *dst = p.obs + (int64_t)ge * QA_NUM_OBS;
        for (int i = lane; i < 513; i += QA_BLOCK) dst[90 + i] = clampf(ss[S_PROP + (i % 57)], -clipo, clipo);
    }
    QA_STAMP(10);
}

// LPE = lanes per env: 4 (lane & 3 = leg).  The template parameter is what is left of round 1's 16-lanes-per-env experiment (slower in wall
// time); the kernel asserts LPE == 4.
// History part of the NEXT observation row, written at the start of the step: slots 0..8 of the new row are slots 1..9 of the old
// one whatever this step does (unless the env resets: then the tail of the kernel refills all ten), so the 513-float shift of every
// env of the block -- 3/4 of the row's bytes -- is read and written here, in the shadow of the physics, and the tail of the kernel
// only adds the 158 floats this step produces.  In place: every lane holds its 9 values of a row before the row's first store issues.
// r5: the loads and stores are written as inline assembly with ACCUMULATOR registers as their data operands ("a" constraint; gfx950 loads /
// stores address the unified file directly).  The 144 values of a lane must stay out of the way of the physics for the whole step; left to the
// compiler, a few of them were loaded into VGPRs and moved to AGPRs behind an `s_waitcnt vmcnt(0)` in front of the substep loop -- the very wait
// the late stores were meant to remove.  The compiler does not count these loads in its own `s_waitcnt vmcnt` arithmetic: its waits only become
// stricter (the counter is in-order), and the stores below are preceded by an explicit vmcnt(0).
#define QA_HLOAD(dst, ptr, OFF) asm volatile("global_load_dword %0, %1, off offset:" #OFF : "=a"(dst) : "v"(ptr) : "memory")
#define QA_HSTORE(ptr, src, OFF) asm volatile("global_store_dword %0, %1, off offset:" #OFF :: "v"(ptr), "a"(src) : "memory")
template <int EPB>
QA_DEV void shift_history_load(const Ptrs &p, int bix, int lane, int N, float (&hv)[EPB][9]) {
#pragma unroll
    for (int g = 0; g < EPB; ++g) {                     // all rows of the block in flight: 144 registers nothing else needs yet
        const int ge = min(bix * EPB + g, N - 1);
        const float *hist = p.obs + (int64_t)ge * QA_NUM_OBS + 90 + 57 + lane;
        QA_HLOAD(hv[g][0], hist, 0); QA_HLOAD(hv[g][1], hist, 256); QA_HLOAD(hv[g][2], hist, 512); QA_HLOAD(hv[g][3], hist, 768);
        QA_HLOAD(hv[g][4], hist, 1024); QA_HLOAD(hv[g][5], hist, 1280); QA_HLOAD(hv[g][6], hist, 1536); QA_HLOAD(hv[g][7], hist, 1792);
        const float *last = lane == 0 ? hist + 512 : hist;          // only lane 0's ninth value is stored; the others re-read a valid address
        QA_HLOAD(hv[g][8], last, 0);
    }
}
template <int EPB>
QA_DEV void shift_history_store(const Ptrs &p, int bix, int lane, int N, float (&hv)[EPB][9]) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int g = 0; g < EPB; ++g) {
        const int ge = bix * EPB + g;
        if (ge < N) {
            float *row = p.obs + (int64_t)ge * QA_NUM_OBS + 90 + lane;
            QA_HSTORE(row, hv[g][0], 0); QA_HSTORE(row, hv[g][1], 256); QA_HSTORE(row, hv[g][2], 512); QA_HSTORE(row, hv[g][3], 768);
            QA_HSTORE(row, hv[g][4], 1024); QA_HSTORE(row, hv[g][5], 1280); QA_HSTORE(row, hv[g][6], 1536); QA_HSTORE(row, hv[g][7], 1792);
            if (lane == 0) { float *r8 = row + 512; QA_HSTORE(r8, hv[g][8], 0); }
        }
    }
}

// (r6: the variants that lost their A/B -- stores inside the substep loop through a scalar row base, 16-byte history accesses, the history stores in
// front of the physics, the constant table staged in two halves, sweeps without the idle-row early-outs -- were deleted with their switches; what each
// measured is in DESIGN.md 4.1c / 9 and profiles/r5_env_step_*.txt.)

// MODE 0: the whole LeggedRobot.step of the behaviour-level (BBC) tree.  MODE 1: the physics part only -- action-history
// roll, delay, clip, decimation x (PD torque -> substep), refresh of the simulator tensors -- for the task-level (TSC) env,
// whose own post_physics_step (goals, termination, rewards, reset, three observation rows) are separate kernels (qa_tsc_*).
// HELP = 1 (plane kernels): two helper wavefronts per workgroup take the bias forces and the contact candidates of every substep off the env's own
// wavefront (phys_substep ROLE 1 / 2 / 3, qa_physics.h) -- three of a CU's four SIMDs idle otherwise at one wavefront per CU.
template <bool PLANE, int LPE, int MODE = 0, int LEAN = 0, int HELP = 0>
__global__ void __launch_bounds__(QA_BLOCK * (HELP ? 3 : 1)) qa_env_step_kernel(StepArgs a) {
    static_assert(LEAN == 0 || MODE == 0, "the task-level tree reads every tensor the physics step exports");
    static_assert(HELP == 0 || (PLANE && MODE == 0), "helper wavefronts: the behaviour-level plane kernels");
    constexpr int WPB = 1;                             // env groups (wavefronts with envs of their own) per workgroup
    constexpr int EPB = QA_BLOCK / LPE;                // envs per wavefront
    const int tix = threadIdx.x & (QA_BLOCK - 1);      // lane
    const int role = HELP ? (int)(threadIdx.x >> 6) : 0;     // 0: the envs' own wavefront; 1, 2: its helpers
    const int bix = blockIdx.x;                        // index of this workgroup's env group
    __shared__ __attribute__((aligned(16))) f4 s_mail[HELP ? QA_BLOCK * QA_MAIL_F4 : 1];
    f4 *const mail = s_mail + (HELP ? tix * QA_MAIL_F4 : 0);
    static_assert(LPE == 4, "one quad per env (the 16-lanes-per-env experiment of round 1 was slower in wall time and is gone)");
    __shared__ float s_tbl[QA_TBL_FLOATS];
    // One LDS scratch region used by two disjoint phases (a workgroup's LDS footprint decides how many of them a CU
    // holds -- 160 KB per CU -- and with it the throughput once there are more workgroups than CUs):
    //   physics phase:      per-lane private slots | terrain windows (height field only)
    //   observation phase:  per-env staging | OBS_GROUP assembled observation rows
    constexpr int U_PHYS = QA_PRIV_FLOATS * QA_PRIV_STRIDE + (PLANE ? 0 : EPB * QA_PATCH * QA_PATCH + EPB * QA_OB_LDS);
    constexpr int U_OBS = EPB * S_ENV + OBS_GROUP * S_ROW;
    constexpr int U_ALL = ((U_PHYS > U_OBS ? U_PHYS : U_OBS) + 3) & ~3;
    __shared__ __attribute__((aligned(16))) float s_u_all[WPB * U_ALL];
    float *s_u = s_u_all;
    float *s_priv = s_u, *s_patch = s_u + QA_PRIV_FLOATS * QA_PRIV_STRIDE;
    float *s_stage = s_u, *s_rows = s_u + EPB * S_ENV;
    static_assert((EPB * S_ENV) % 4 == 0, "row buffer must stay 16-byte aligned");
    long long *const qa_prof = a.prof;
    QA_STAMP(0);
    const qa_config &c = a.c;
    const Ptrs &p = a.p;
    const int N = c.num_envs;
    // r5: the history shift's 144 loads per lane are issued here and its stores AFTER the physics (r2 had a
    // load -> stage table -> store prologue).  All 256 wavefronts of a 4096-env launch start together, so the prologue's 17 MB read + 17 MB
    // write was a burst nobody overlapped: 9.7 k of the kernel's 136 k ticks waiting for it (profiles/r5_env_step_phase_profile.txt).  The
    // values wait in registers the substeps do not use (the compiler parks them in AGPRs: 2 x 144 v_accvgpr moves, ~1.3 k ticks).
    if (HELP && role != 0) {
        // ---- a helper wavefront: per substep, the state the env's wavefront published -> bias forces (role 1) / contact candidates (role 2) -> mail
        const int leg = tix & 3, env = min((bix * QA_BLOCK + tix) / LPE, a.c.num_envs - 1);
        (void)env;
        const float *tbl = s_tbl + leg * QA_LEG_TBL, *btbl = s_tbl + 4 * QA_LEG_TBL;
        const qa_config &c = a.c;
        PhysParams P; P.dt = c.sim_dt; P.gz = c.gravity_z; P.contact_offset = c.contact_offset; P.max_depen = c.max_depenetration_velocity;
        P.ground_friction = c.ground_friction; P.iters = c.solver_iterations; P.slots = c.contact_slots == 1 ? 1 : 2; P.self_collision = c.self_collision;
        TerrainView T = terrain_view(c, a.p, s_u);
        float *priv = priv_of(s_u, tix);
        EnvState st; ContactOut co;
        float fimp[3] = {0.f, 0.f, 0.f}, tau[3] = {0.f, 0.f, 0.f};
        // The history shift of the observation rows (513 floats per env read and written, 3/4 of the launch's bytes) is pure data movement: the bias
        // helper (the one with time to spare in every substep) issues its loads now and its stores after the last substep, and the env's own
        // wavefront issues none of those 288 memory instructions and keeps none of the 144 values.  Ordering against that wavefront's own writes to
        // the same rows (the new frame in slot 9, the refill of a reset env's slots): the stores have been acknowledged (vmcnt 0) before the
        // workgroup's last barrier, which the env's wavefront passes before it writes a row.
        if (role == 2) stage_table(s_tbl);                     // (the env's wavefront meets it behind the first barrier)
        float hvh[EPB][9];
        if (role == 1) shift_history_load<EPB>(a.p, bix, tix, c.num_envs, hvh);
        for (int d = 0; d < c.decimation; ++d) {
            __syncthreads();                                   // the state of this substep is in the mail (first time: the table and the parked persistents too)
            asm volatile("" ::: "memory");
            mail_get_state(mail, st);
            float bi[10], pa[3], psp[3], psd[3];
            priv_unpark(priv, pa, psp, psd, bi);
            if (role == 1) phys_substep<PLANE, HELP ? 2 : 0>(st, tbl, btbl, bi, tau, 0.f, leg, P, co, priv, fimp, T, mail);
            else phys_substep<PLANE, HELP ? 3 : 0>(st, tbl, btbl, bi, tau, 0.f, leg, P, co, priv, fimp, T, mail);
            // the history stores go out behind the second-to-last substep's last barrier, while the env's wavefront runs that substep's sweeps
            // (~10 k cycles in which this helper has nothing to do): 17 MB of writes from 256 CUs need ~5 us to be acknowledged, and the env's
            // wavefront must not find itself waiting for that at the workgroup's last barrier
            if (role == 1 && d == (c.decimation > 1 ? c.decimation - 2 : 0)) {
                shift_history_store<EPB>(a.p, bix, tix, c.num_envs, hvh);
            }
        }
        if (role == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                       // the rows' history is in place: the env's wavefront may write them
        if (role == 2) {                    // ... and its scalars are in the mail: the idle contact helper stores them
            ScalarOut so;
#pragma unroll
            for (int i = 0; i < SO_WORDS / 4; ++i) { const f4 v = mail[i]; so.w[4 * i] = v.x; so.w[4 * i + 1] = v.y; so.w[4 * i + 2] = v.z; so.w[4 * i + 3] = v.w; }
            const int env_raw = (bix * QA_BLOCK + tix) / LPE;
            write_scalars<PLANE, LEAN>(a.p, c.num_envs, env_raw < c.num_envs ? env_raw : c.num_envs - 1, leg, env_raw < c.num_envs, so);
        }
        return;
    }
    float hv[EPB][9];
#define QA_SHIFT_LOAD() shift_history_load<EPB>(p, bix, tix, N, hv)
#define QA_SHIFT_STORE() shift_history_store<EPB>(p, bix, tix, N, hv)
    if (!HELP) stage_table(s_tbl);
    const int tid = bix * QA_BLOCK + tix;
    const int leg = LPE == 4 ? (tix & 3) : ((tix >> 2) & 3);
    const int sub = LPE == 4 ? 0 : (tix & 3);
    const bool owner = sub == 0;                     // always true with one quad per env
    const int env_raw = tid / LPE;
    const bool in_range = env_raw < N;
    const bool valid = in_range && owner;            // guards every global write
    const int env = in_range ? env_raw : N - 1;
    const int le = tix / LPE;                // env slot inside the block
    const float *tbl = s_tbl + leg * QA_LEG_TBL;
    const float *btbl = s_tbl + 4 * QA_LEG_TBL;
    const int64_t step = a.step_ptr ? *a.step_ptr : a.step;

    if (bix == 0 && tix < 16) p.episode_stats[16 * ((step + 1) & 1) + tix] = 0.f;   // next step's bin

    // ---- action history roll, delay, clip (legged_robot.py:84-98); lane handles its 3 joints
    float act[3], raw_act[3];
    {
        float *ah = p.action_hist + (int64_t)env * (QA_ACTION_BUF_LEN * 12) + 3 * leg;
        const float clipa = c.clip_actions / c.action_scale;
        if (LEAN & 1) {
            // the two newest slots only (the host refuses a delay > 1 in this mode): 48 B read + 96 B written per env instead of 336 + 384
            float prev[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) { prev[k] = ah[12 * (QA_ACTION_BUF_LEN - 1) + k]; raw_act[k] = a.actions[(int64_t)env * 12 + 3 * leg + k]; }
            if (valid) {
#pragma unroll
                for (int k = 0; k < 3; ++k) { ah[12 * (QA_ACTION_BUF_LEN - 2) + k] = prev[k]; ah[12 * (QA_ACTION_BUF_LEN - 1) + k] = raw_act[k]; }
            }
#pragma unroll
            for (int k = 0; k < 3; ++k) act[k] = clampf(a.delay == 0 ? raw_act[k] : prev[k], -clipa, clipa);
        } else {
        float h[QA_ACTION_BUF_LEN][3];
#pragma unroll
        for (int r = 1; r < QA_ACTION_BUF_LEN; ++r)
#pragma unroll
            for (int k = 0; k < 3; ++k) h[r - 1][k] = ah[12 * r + k];
#pragma unroll
        for (int k = 0; k < 3; ++k) { raw_act[k] = a.actions[(int64_t)env * 12 + 3 * leg + k]; h[QA_ACTION_BUF_LEN - 1][k] = raw_act[k]; }
        if (valid) {
#pragma unroll
            for (int r = 0; r < QA_ACTION_BUF_LEN; ++r)
#pragma unroll
                for (int k = 0; k < 3; ++k) ah[12 * r + k] = h[r][k];
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            float src = a.delay == 0 ? h[QA_ACTION_BUF_LEN - 1][k] : h[QA_ACTION_BUF_LEN - 2][k];
            if (a.delay > 1) {
#pragma unroll
                for (int r = 0; r < QA_ACTION_BUF_LEN - 2; ++r) if (QA_ACTION_BUF_LEN - 1 - r == a.delay) src = h[r][k];
            }
            act[k] = clampf(src, -clipa, clipa);
        }
        }
    }

    QA_STAMP(1);
    // ---- load state
    EnvState st;
    {
        const float *r = p.root + (int64_t)env * 13;
        st.pos = v3(r[0], r[1], r[2]); st.qx = r[3]; st.qy = r[4]; st.qz = r[5]; st.qw = r[6];
        st.vw = v3(r[7], r[8], r[9]); st.ww = v3(r[10], r[11], r[12]);
        const float *d = p.dof + (int64_t)env * 24 + 6 * leg;
#pragma unroll
        for (int k = 0; k < 3; ++k) { st.q[k] = d[2 * k]; st.qd[k] = d[2 * k + 1]; }
    }
    // MODE 1: last_* := the values this step starts from (tsc/.../legged_robot.py:275-278 sets them at the END of the previous
    // step, after its reset, i.e. to exactly what the arena holds now)
    float old_act[3], old_torg[3], old_qd[3]; V3 old_vw, old_ww;
    if (MODE == 1) {
#pragma unroll
        for (int k = 0; k < 3; ++k) { const int64_t j = (int64_t)env * 12 + 3 * leg + k; old_act[k] = p.actions[j]; old_torg[k] = p.torques_org[j]; old_qd[k] = st.qd[k]; }
        old_vw = st.vw; old_ww = st.ww;
    }
    PostIn in;
    if (MODE == 0 && PLANE) post_in_preload(in, p, env, leg, N);
    float binert[10];
#pragma unroll
    for (int i = 0; i < 10; ++i) binert[i] = p.base_inertia[(int64_t)env * 10 + i];
    float sp[3], sd[3], q0[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        sp[k] = c.randomize_motor ? p.motor_strength[((int64_t)0 * N + env) * 12 + 3 * leg + k] : 1.0f;
        sd[k] = c.randomize_motor ? p.motor_strength[((int64_t)1 * N + env) * 12 + 3 * leg + k] : 1.0f;
        q0[k] = c.default_dof_pos[k];
    }
    const float fric = p.friction[env];
    const float mu = 0.5f * (fric + c.ground_friction);
    PhysParams P; P.dt = c.sim_dt; P.gz = c.gravity_z; P.contact_offset = c.contact_offset; P.max_depen = c.max_depenetration_velocity;
    P.ground_friction = c.ground_friction; P.iters = c.solver_iterations; P.slots = c.contact_slots == 1 ? 1 : 2; P.self_collision = c.self_collision;

    // ---- terrain window: staged in the rows buffer, which is idle until the observation phase
    TerrainView T = terrain_view(c, p, s_patch + le * (QA_PATCH * QA_PATCH));
    float *ob_rec = s_patch + EPB * (QA_PATCH * QA_PATCH) + le * QA_OB_LDS;
    if (!PLANE) {
        patch_origin(T, st.pos.x, st.pos.y);
        stage_patch(T, s_patch + le * (QA_PATCH * QA_PATCH), leg);
        if (c.articulated_obstacles) { stage_obstacles(p, env, leg, ob_rec, T.ancx, T.ancy); T.ob = ob_rec; T.ob_acc = ob_rec + 12 * QA_OBST_PER_ENV; }
        wave_lds_sync();
    }
    // env-local horizontal coordinates for the substeps (TerrainView): offsets from the world position the step starts at
    const float anc_x = st.pos.x, anc_y = st.pos.y;
    st.pos.x = 0.f; st.pos.y = 0.f;

    QA_STAMP(2);
    // ---- decimation x (PD torque -> physics)   legged_robot.py:101-106, :547-579
    // per-step constants are parked in per-lane LDS slots between substeps so that they do not hold VGPRs
    // through the substep (the substep alone needs ~340 registers)
    float *priv = priv_of(s_priv, tix);
    priv_park(priv, act, sp, sd, binert);
    float tau[3], tau_org[3];
    ContactOut co;
    float fimp[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) fimp[k] = p.foot_impulse[(int64_t)env * 12 + 3 * leg + k];
    // the history loads go out LAST, behind every load the substeps wait for: the wait counter is in-order (and 6 bits wide), so loads issued
    // in front of the state's would have to land before the physics could start; nothing waits for these until the stores after the loop
    if (MODE == 0 && LPE == 4) {
        // every value loaded so far is "used" here, so that the compiler waits for it NOW (nothing is outstanding behind it yet) and not
        // at its first real use inside the loop, where the 6-bit counter could only say "all but the newest 62 loads" = 82 history loads
        asm volatile("" :: "v"(fimp[0]), "v"(fimp[1]), "v"(fimp[2]), "v"(mu), "v"(st.q[0]), "v"(st.qd[2]), "v"(st.pos.z), "v"(st.ww.z) : "memory");
        if (PLANE) {
        asm volatile("" :: "v"(in.last_act[0]), "v"(in.last_act[2]), "v"(in.last_torg[0]), "v"(in.last_torg[2]), "v"(in.last_qd[0]), "v"(in.last_qd[2]) : "memory");
        asm volatile("" :: "v"(in.esum[0]), "v"(in.esum[3]), "v"(in.esum[6]), "v"(in.esum[9]), "v"(in.esum[13]) : "memory");
        asm volatile("" :: "v"(in.cmd[0]), "v"(in.cmd[4]), "v"(in.latc[0]), "v"(in.latc[4]), "v"(in.eps), "v"((int)in.epl0), "v"((int)in.last_contact) : "memory");
        }
        if (!HELP) QA_SHIFT_LOAD();
    }
    for (int d = 0; d < c.decimation; ++d) {
        // compiler fence: without it LICM hoists the ~120 loop-invariant LDS table reads of the substep out of this
        // loop and keeps them in registers across it, which is what pushed the kernel into scratch
        asm volatile("" ::: "memory");
        // the helpers start on this substep's state; behind the FIRST of these barriers the constant table is in LDS too (with helpers the
        // contact helper stages it while this wavefront loads its state)
        if (HELP) { mail_put_state(mail, st); __syncthreads(); }
        float bi[10], pa[3], psp[3], psd[3];
        priv_unpark(priv, pa, psp, psd, bi);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            float as = pa[k] * c.action_scale;
            if (k == 0) as *= c.hip_scale_reduction;
            float t = c.randomize_motor ? psp[k] * c.kp * (as + q0[k] - st.q[k]) - psd[k] * c.kd * st.qd[k]
                                        : c.kp * (as + q0[k] - st.q[k]) - c.kd * st.qd[k];
            tau_org[k] = t;
            float lim = tbl[T_EFFORT + k];
            tau[k] = clampf(t, -lim, lim);
        }
        phys_substep<PLANE, HELP ? 1 : 0>(st, tbl, btbl, bi, tau, mu, leg, P, co, priv, fimp, T, mail);
    }
    { float bi_[10]; priv_unpark(priv, act, sp, sd, bi_); }
    st.pos.x += anc_x; st.pos.y += anc_y;               // back to world coordinates: ONE rounding per step, like the oracle's double -> float store
    // ---- the articulated obstacles' joints: one step of h = decimation x dt under the mean contact force of the substeps (lane k: slot k)
    if (!PLANE) {
        if (c.articulated_obstacles) {
            wave_lds_sync();
            if (leg < QA_OBST_PER_ENV && valid) {
                float q = ob_rec[12 * leg + 8], qd = ob_rec[12 * leg + 9];
                obstacle_joint_step(ob_rec[12 * leg + 7], ob_rec[12 * leg + 11], ob_rec[12 * QA_OBST_PER_ENV + leg] / (float)c.decimation,
                                    c.sim_dt * (float)c.decimation, q, qd);
                float *stt = p.obst_state + ((int64_t)env * QA_OBST_PER_ENV + leg) * QA_OBST_STATE;
                stt[0] = q; stt[1] = qd; stt[2] = ob_rec[12 * QA_OBST_PER_ENV + leg];
            }
        }
    }

    QA_STAMP(3);
    if (MODE == 0 && LPE == 4 && !HELP) QA_SHIFT_STORE();     // in place: every lane loaded its 9 values of a row long ago; the tail's writes to the same rows come after these in program order
    // ---- refresh_*: body positions of the new state, contact forces per body
    V3 org[4];
    leg_origins(st.q, tbl, org);
    M3 R = quat_to_mat(st.qx, st.qy, st.qz, st.qw);
    V3 foot_w = mul(R, org[3]) + st.pos;
    // route the lane's (up to two) extra contacts to their bodies: base / Head_upper / Head_lower are summed over the quad
    V3 base_f, hu_f, hl_f;
    const int myb = 3 + 4 * leg;
    auto on_body = [&](int b) {
        V3 r = v3(0, 0, 0);
#pragma unroll
        for (int j = 0; j < QA_EXTRA_SLOTS; ++j) if (co.extra_body[j] == b) r = r + co.extra_f[j];
        return r;
    };
    {
        V3 eb = on_body(0), e1 = on_body(1), e2 = on_body(2);
        base_f = v3(xsum<LPE>(eb.x), xsum<LPE>(eb.y), xsum<LPE>(eb.z));
        hu_f = v3(xsum<LPE>(e1.x), xsum<LPE>(e1.y), xsum<LPE>(e1.z));
        hl_f = v3(xsum<LPE>(e2.x), xsum<LPE>(e2.y), xsum<LPE>(e2.z));
    }
    V3 hip_f = on_body(myb), thigh_f = on_body(myb + 1), calf_f = on_body(myb + 2) + co.self_f;
    if (valid && !(LEAN & 1)) {
        float *cf = p.cforce + (int64_t)env * 57, *rb = p.rbpos + (int64_t)env * 57;
        float *m = cf + 3 * myb;
        m[0] = hip_f.x; m[1] = hip_f.y; m[2] = hip_f.z; m[3] = thigh_f.x; m[4] = thigh_f.y; m[5] = thigh_f.z;
        m[6] = calf_f.x; m[7] = calf_f.y; m[8] = calf_f.z; m[9] = co.foot_f.x; m[10] = co.foot_f.y; m[11] = co.foot_f.z;
#pragma unroll
        for (int k = 0; k < 3; ++k) { V3 w = mul(R, org[k]) + st.pos; rb[3 * (myb + k)] = w.x; rb[3 * (myb + k) + 1] = w.y; rb[3 * (myb + k) + 2] = w.z; }
        rb[3 * (myb + 3)] = foot_w.x; rb[3 * (myb + 3) + 1] = foot_w.y; rb[3 * (myb + 3) + 2] = foot_w.z;
        if (leg == 0) {
            cf[0] = base_f.x; cf[1] = base_f.y; cf[2] = base_f.z; cf[3] = hu_f.x; cf[4] = hu_f.y; cf[5] = hu_f.z; cf[6] = hl_f.x; cf[7] = hl_f.y; cf[8] = hl_f.z;
            V3 h1 = mul(R, v3(0.285f, 0.f, 0.01f)) + st.pos, h2 = mul(R, v3(0.293f, 0.f, -0.06f)) + st.pos;
            rb[0] = st.pos.x; rb[1] = st.pos.y; rb[2] = st.pos.z; rb[3] = h1.x; rb[4] = h1.y; rb[5] = h1.z; rb[6] = h2.x; rb[7] = h2.y; rb[8] = h2.z;
        }
    }

    if (c.export_body_state && valid) write_body_state(p.rbstate + (int64_t)env * (QA_NUM_BODIES_ABI * 13), st, R, org, leg);
    if (MODE == 1) {
        if (valid) {
            if (leg == 0) {
                float *rt = p.root + (int64_t)env * 13;
                rt[0] = st.pos.x; rt[1] = st.pos.y; rt[2] = st.pos.z; rt[3] = st.qx; rt[4] = st.qy; rt[5] = st.qz; rt[6] = st.qw;
                rt[7] = st.vw.x; rt[8] = st.vw.y; rt[9] = st.vw.z; rt[10] = st.ww.x; rt[11] = st.ww.y; rt[12] = st.ww.z;
                float *lr = p.last_root_vel + (int64_t)env * 6;
                lr[0] = old_vw.x; lr[1] = old_vw.y; lr[2] = old_vw.z; lr[3] = old_ww.x; lr[4] = old_ww.y; lr[5] = old_ww.z;
            }
            float *d = p.dof + (int64_t)env * 24 + 6 * leg;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const int64_t j = (int64_t)env * 12 + 3 * leg + k;
                d[2 * k] = st.q[k]; d[2 * k + 1] = st.qd[k];
                p.torques[j] = tau[k]; p.torques_org[j] = tau_org[k]; p.actions[j] = act[k]; p.foot_impulse[j] = fimp[k];
                p.last_actions[j] = old_act[k]; p.last_dof_vel[j] = old_qd[k]; p.last_torques_org[j] = old_torg[k];
            }
        }
        return;
    }
    QA_STAMP(4);
    if (!PLANE) post_in_preload(in, p, env, leg, N);
    in.st = st; in.foot_f = co.foot_f; in.hip_f = hip_f; in.thigh_f = thigh_f; in.calf_f = calf_f; in.base_f = base_f; in.foot_w = foot_w; in.fric = fric;
#pragma unroll
    for (int k = 0; k < 3; ++k) { in.act[k] = act[k]; in.raw_act[k] = raw_act[k]; in.tau[k] = tau[k]; in.tau_org[k] = tau_org[k]; in.sp[k] = sp[k]; in.sd[k] = sd[k]; in.fimp[k] = fimp[k]; }
    post_physics_phase<PLANE, LPE, LEAN, HELP>(c, p, a.mi, a.prof, in, T, tbl, s_stage, s_rows, tix, bix, env, leg, valid, owner, le, step, mail);
    // the device-side step counter advances once every wavefront of the launch is done with it (they all read it at their
    // start): the last one to arrive resets the arrival counter and bumps the step -- no separate 1-thread launch per env step
    if (a.step_ptr && tix == 0) {
        __threadfence();
        const int total = (int)gridDim.x * WPB;
        if (atomicAdd(p.ticket, 1) == total - 1) { *p.ticket = 0; *a.step_ptr = step + 1; __threadfence(); }
    }
}

// post_physics_step alone, from the arena (qa_debug_post_physics): the PostIn record is loaded instead of computed
template <bool PLANE>
__global__ void __launch_bounds__(QA_BLOCK) qa_post_physics_kernel(StepArgs a) {
    constexpr int LPE = 4, EPB = QA_BLOCK / LPE;
    const int tix = threadIdx.x & (QA_BLOCK - 1), bix = blockIdx.x;
    __shared__ float s_tbl[QA_TBL_FLOATS];
    constexpr int U_OBS = (EPB * S_ENV + OBS_GROUP * S_ROW + 3) & ~3;
    __shared__ __attribute__((aligned(16))) float s_u[U_OBS];
    const qa_config &c = a.c;
    const Ptrs &p = a.p;
    const int N = c.num_envs, tid = bix * QA_BLOCK + tix, leg = tix & 3, env_raw = tid / LPE;
    {
        float hv[EPB][9];
        shift_history_load<EPB>(p, bix, tix, N, hv);
        stage_table(s_tbl);
        shift_history_store<EPB>(p, bix, tix, N, hv);
    }
    const bool valid = env_raw < N;
    const int env = valid ? env_raw : N - 1, le = tix / LPE;
    const int64_t step = a.step;
    if (bix == 0 && tix < 16) p.episode_stats[16 * ((step + 1) & 1) + tix] = 0.f;
    PostIn in;
    post_in_preload(in, p, env, leg, N);
    {
        const float *r = p.root + (int64_t)env * 13;
        in.st.pos = v3(r[0], r[1], r[2]); in.st.qx = r[3]; in.st.qy = r[4]; in.st.qz = r[5]; in.st.qw = r[6];
        in.st.vw = v3(r[7], r[8], r[9]); in.st.ww = v3(r[10], r[11], r[12]);
        const float *d = p.dof + (int64_t)env * 24 + 6 * leg;
        const float *cf = p.cforce + (int64_t)env * 57, *m = cf + 3 * (3 + 4 * leg), *rb = p.rbpos + (int64_t)env * 57 + 3 * (3 + 4 * leg + 3);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int64_t j = (int64_t)env * 12 + 3 * leg + k;
            in.st.q[k] = d[2 * k]; in.st.qd[k] = d[2 * k + 1];
            in.act[k] = p.actions[j]; in.raw_act[k] = p.action_hist[(int64_t)env * (QA_ACTION_BUF_LEN * 12) + 12 * (QA_ACTION_BUF_LEN - 1) + 3 * leg + k];
            in.tau[k] = p.torques[j]; in.tau_org[k] = p.torques_org[j]; in.fimp[k] = p.foot_impulse[j];
            in.sp[k] = c.randomize_motor ? p.motor_strength[((int64_t)0 * N + env) * 12 + 3 * leg + k] : 1.0f;
            in.sd[k] = c.randomize_motor ? p.motor_strength[((int64_t)1 * N + env) * 12 + 3 * leg + k] : 1.0f;
        }
        in.hip_f = v3(m[0], m[1], m[2]); in.thigh_f = v3(m[3], m[4], m[5]); in.calf_f = v3(m[6], m[7], m[8]); in.foot_f = v3(m[9], m[10], m[11]);
        in.base_f = v3(cf[0], cf[1], cf[2]);
        in.foot_w = v3(rb[0], rb[1], rb[2]);
        in.fric = p.friction[env];
    }
    TerrainView T = terrain_view(c, p, s_u);
    post_physics_phase<PLANE, LPE>(c, p, a.mi, nullptr, in, T, s_tbl + leg * QA_LEG_TBL, s_u, s_u + EPB * S_ENV, tix, bix, env, leg, valid, true, le, step);
}

// ------------------------------------------------------------------ init / reset / simulate kernels
__device__ __forceinline__ float normal_from(float u1, float u2) {
    u1 = fmaxf(u1, 1e-7f);
    return sqrtf(-2.0f * logf(u1)) * cosf(6.28318530717958647692f * u2);
}

struct BaseConst { float m, com[3], I[6]; };

__global__ void qa_init_kernel(qa_config c, Ptrs p, BaseConst bc) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    const int N = c.num_envs;
    if (e >= N) return;
    const int ge = e + c.env_id_offset;                    // global env id: RNG key and spawn-grid slot
    const int ncols = (int)floor(sqrt((double)(c.num_envs_global > 0 ? c.num_envs_global : N)));
    p.env_origins[3 * e] = c.env_spacing * (float)(ge / ncols); p.env_origins[3 * e + 1] = c.env_spacing * (float)(ge % ncols); p.env_origins[3 * e + 2] = 0.f;
    float fr = 1.0f;
    if (c.randomize_friction) {
        F4 u = rng4(c.seed, ge, 0, RS_INIT_FRICTION, 0); int b = min((int)(u.v[0] * 64.0f), 63);
        F4 ub = rng4(c.seed, b, 0, RS_INIT_BUCKET, 0);
        fr = (c.friction_range[1] - c.friction_range[0]) * ub.v[0] + c.friction_range[0];
    }
    p.friction[e] = fr;
    F4 um = rng4(c.seed, ge, 0, RS_INIT_MASS, 0);
    float mp[4];
    mp[0] = c.randomize_base_mass ? (c.added_mass_range[1] - c.added_mass_range[0]) * um.v[0] + c.added_mass_range[0] : 0.f;
    for (int i = 0; i < 3; ++i) mp[1 + i] = c.randomize_base_com ? (c.added_com_range[1] - c.added_com_range[0]) * um.v[1 + i] + c.added_com_range[0] : 0.f;
    for (int i = 0; i < 4; ++i) p.mass_params[4 * e + i] = mp[i];
    {
        double m = (double)bc.m + mp[0], sc = m / (double)bc.m;
        double cx = (double)bc.com[0] + mp[1], cy = (double)bc.com[1] + mp[2], cz = (double)bc.com[2] + mp[3];
        double cc = cx * cx + cy * cy + cz * cz;
        float *bi = p.base_inertia + 10 * e;
        bi[0] = (float)m; bi[1] = (float)(m * cx); bi[2] = (float)(m * cy); bi[3] = (float)(m * cz);
        bi[4] = (float)(bc.I[0] * sc + m * (cc - cx * cx)); bi[5] = (float)(bc.I[1] * sc + m * (cc - cy * cy)); bi[6] = (float)(bc.I[2] * sc + m * (cc - cz * cz));
        bi[7] = (float)(bc.I[3] * sc - m * cx * cy); bi[8] = (float)(bc.I[4] * sc - m * cx * cz); bi[9] = (float)(bc.I[5] * sc - m * cy * cz);
    }
    float uu[48];
    for (int b = 0; b < 12; ++b) { F4 t = rng4(c.seed, ge, 0, RS_INIT_MOTOR, b); for (int i = 0; i < 4; ++i) uu[4 * b + i] = t.v[i]; }
    for (int j = 0; j < 12; ++j) {
        int pi = 2 * (j % 3); float s_p, s_d;
        if (!c.randomize_motor) { s_p = s_d = 1.0f; }
        else if (c.use_easi) { s_p = c.easi_mean[pi] + c.easi_var[pi] * normal_from(uu[j], uu[12 + j]); s_d = c.easi_mean[pi + 1] + c.easi_var[pi + 1] * normal_from(uu[24 + j], uu[36 + j]); }
        else { s_p = (c.motor_strength_range[1] - c.motor_strength_range[0]) * uu[j] + c.motor_strength_range[0]; s_d = (c.motor_strength_range[1] - c.motor_strength_range[0]) * uu[12 + j] + c.motor_strength_range[0]; }
        p.motor_strength[((int64_t)0 * N + e) * 12 + j] = s_p; p.motor_strength[((int64_t)1 * N + e) * 12 + j] = s_d;
    }
    p.reset[e] = 1;
    p.root[13 * e + 6] = 1.0f;
    if (e < QA_NUM_GAITS) p.prior[e] = 1.0f / QA_NUM_GAITS;
}

__global__ void __launch_bounds__(QA_BLOCK) qa_reset_all_kernel(qa_config c, Ptrs p, MocapIdx mi, int64_t step) {
    const int tid = blockIdx.x * QA_BLOCK + threadIdx.x, leg = threadIdx.x & 3, env = tid >> 2, N = c.num_envs;
    if (blockIdx.x == 0 && threadIdx.x < 16) { p.episode_stats[16 * (step & 1) + threadIdx.x] = 0.f; }
    if (env >= N) return;
    EnvState st; float cmd[5], eps; int gait;
    reset_env(c, p, mi, env, leg, step, st, cmd, eps, gait);
    if (leg == 0) {
        float *rt = p.root + (int64_t)env * 13;
        rt[0] = st.pos.x; rt[1] = st.pos.y; rt[2] = st.pos.z; rt[3] = st.qx; rt[4] = st.qy; rt[5] = st.qz; rt[6] = st.qw;
        rt[7] = st.vw.x; rt[8] = st.vw.y; rt[9] = st.vw.z; rt[10] = st.ww.x; rt[11] = st.ww.y; rt[12] = st.ww.z;
        for (int i = 0; i < 5; ++i) { p.commands[(int64_t)env * 5 + i] = cmd[i]; p.latent_c[(int64_t)env * 5 + i] = (gait == i) ? 1.0f : 0.0f; }
        p.latent_eps[env] = eps;
        p.episode_length[env] = 0; p.reset[env] = 1;
        for (int i = 0; i < 6; ++i) p.last_root_vel[(int64_t)env * 6 + i] = 0.f;
        for (int r = 0; r < QA_NUM_REWARDS; ++r) p.episode_sums[(int64_t)r * N + env] = 0.f;   // stats of a full reset are not reported
    }
    float *d = p.dof + (int64_t)env * 24 + 6 * leg;
    for (int k = 0; k < 3; ++k) {
        const int64_t j = (int64_t)env * 12 + 3 * leg + k;
        d[2 * k] = st.q[k]; d[2 * k + 1] = st.qd[k];
        p.last_actions[j] = 0.f; p.last_dof_vel[j] = 0.f; p.last_torques_org[j] = 0.f; p.foot_impulse[j] = 0.f;
        for (int r = 0; r < QA_ACTION_BUF_LEN; ++r) p.action_hist[(int64_t)env * 96 + 12 * r + 3 * leg + k] = 0.f;
    }
    for (int i = leg; i < 570; i += 4) p.obs[(int64_t)env * QA_NUM_OBS + 90 + i] = 0.f;
}

template <bool PLANE, int LPE>
__global__ void __launch_bounds__(QA_BLOCK) qa_simulate_kernel(qa_config c, Ptrs p, const float *torques, const uint8_t *cond) {
    if (cond && *cond == 0) return;               // qa_simulate_if: the whole launch is a no-op (uniform over the grid)
    __shared__ float s_tbl[QA_TBL_FLOATS];
    __shared__ __attribute__((aligned(16))) float s_priv[QA_PRIV_FLOATS * QA_PRIV_STRIDE];
    __shared__ float s_patch[PLANE ? 1 : ENVS_PER_BLOCK * QA_PATCH * QA_PATCH];
    __shared__ float s_ob[PLANE ? 1 : ENVS_PER_BLOCK * QA_OB_LDS];
    static_assert(LPE == 4, "one quad per env");
    stage_table(s_tbl);
    const int tid = blockIdx.x * QA_BLOCK + threadIdx.x, N = c.num_envs;
    const int leg = LPE == 4 ? (threadIdx.x & 3) : ((threadIdx.x >> 2) & 3), sub = LPE == 4 ? 0 : (threadIdx.x & 3);
    const bool in_range = (tid / LPE) < N, valid = in_range && sub == 0;
    const int env = in_range ? (tid / LPE) : N - 1;
    const float *tbl = s_tbl + leg * QA_LEG_TBL, *btbl = s_tbl + 4 * QA_LEG_TBL;
    EnvState st;
    const float *r = p.root + (int64_t)env * 13;
    st.pos = v3(r[0], r[1], r[2]); st.qx = r[3]; st.qy = r[4]; st.qz = r[5]; st.qw = r[6]; st.vw = v3(r[7], r[8], r[9]); st.ww = v3(r[10], r[11], r[12]);
    const float *d = p.dof + (int64_t)env * 24 + 6 * leg;
    float tau[3], binert[10];
    for (int k = 0; k < 3; ++k) { st.q[k] = d[2 * k]; st.qd[k] = d[2 * k + 1]; float lim = tbl[T_EFFORT + k]; tau[k] = clampf(torques[(int64_t)env * 12 + 3 * leg + k], -lim, lim); }
    for (int i = 0; i < 10; ++i) binert[i] = p.base_inertia[(int64_t)env * 10 + i];
    PhysParams P; P.dt = c.sim_dt; P.gz = c.gravity_z; P.contact_offset = c.contact_offset; P.max_depen = c.max_depenetration_velocity; P.ground_friction = c.ground_friction; P.iters = c.solver_iterations; P.slots = c.contact_slots == 1 ? 1 : 2; P.self_collision = c.self_collision;
    ContactOut co;
    float fimp[3];
    for (int k = 0; k < 3; ++k) fimp[k] = p.foot_impulse[(int64_t)env * 12 + 3 * leg + k];
    TerrainView T = terrain_view(c, p, s_patch);
    if (!PLANE) {
        float *mine = s_patch + (threadIdx.x >> 2) * (QA_PATCH * QA_PATCH);
        T.patch = mine;
        patch_origin(T, st.pos.x, st.pos.y);
        stage_patch(T, mine, leg);
        if (c.articulated_obstacles) {          // the obstacles' geometry and surface velocity; their joints only move in env steps
            float *rec = s_ob + (threadIdx.x >> 2) * QA_OB_LDS;
            stage_obstacles(p, env, leg, rec, T.ancx, T.ancy); T.ob = rec;
        }
        wave_lds_sync();
    }
    const float anc_x = st.pos.x, anc_y = st.pos.y;     // env-local horizontal coordinates inside the substep (TerrainView)
    st.pos.x = 0.f; st.pos.y = 0.f;
    phys_substep<PLANE>(st, tbl, btbl, binert, tau, 0.5f * (p.friction[env] + c.ground_friction), leg, P, co, priv_of(s_priv, threadIdx.x), fimp, T);
    st.pos.x += anc_x; st.pos.y += anc_y;
    V3 org[4]; leg_origins(st.q, tbl, org);
    M3 R = quat_to_mat(st.qx, st.qy, st.qz, st.qw);
    V3 z = v3(0, 0, 0);
    auto on_body = [&](int b) {
        V3 r = z;
        for (int j = 0; j < QA_EXTRA_SLOTS; ++j) if (co.extra_body[j] == b) r = r + co.extra_f[j];
        return r;
    };
    V3 eb = on_body(0), e1 = on_body(1), e2 = on_body(2);
    V3 base_f = v3(xsum<LPE>(eb.x), xsum<LPE>(eb.y), xsum<LPE>(eb.z)), hu_f = v3(xsum<LPE>(e1.x), xsum<LPE>(e1.y), xsum<LPE>(e1.z)), hl_f = v3(xsum<LPE>(e2.x), xsum<LPE>(e2.y), xsum<LPE>(e2.z));
    if (!valid) return;
    const int myb = 3 + 4 * leg;
    float *cf = p.cforce + (int64_t)env * 57, *rb = p.rbpos + (int64_t)env * 57;
    for (int k = 0; k < 3; ++k) {
        V3 f = on_body(myb + k);
        if (k == 2) f = f + co.self_f;
        cf[3 * (myb + k)] = f.x; cf[3 * (myb + k) + 1] = f.y; cf[3 * (myb + k) + 2] = f.z;
        V3 w = mul(R, org[k]) + st.pos; rb[3 * (myb + k)] = w.x; rb[3 * (myb + k) + 1] = w.y; rb[3 * (myb + k) + 2] = w.z;
    }
    cf[3 * (myb + 3)] = co.foot_f.x; cf[3 * (myb + 3) + 1] = co.foot_f.y; cf[3 * (myb + 3) + 2] = co.foot_f.z;
    { V3 w = mul(R, org[3]) + st.pos; rb[3 * (myb + 3)] = w.x; rb[3 * (myb + 3) + 1] = w.y; rb[3 * (myb + 3) + 2] = w.z; }
    if (c.export_body_state) write_body_state(p.rbstate + (int64_t)env * (QA_NUM_BODIES_ABI * 13), st, R, org, leg);
    float *dd = p.dof + (int64_t)env * 24 + 6 * leg;
    for (int k = 0; k < 3; ++k) { dd[2 * k] = st.q[k]; dd[2 * k + 1] = st.qd[k]; p.foot_impulse[(int64_t)env * 12 + 3 * leg + k] = fimp[k]; }
    if (leg == 0) {
        float *rt = p.root + (int64_t)env * 13;
        rt[0] = st.pos.x; rt[1] = st.pos.y; rt[2] = st.pos.z; rt[3] = st.qx; rt[4] = st.qy; rt[5] = st.qz; rt[6] = st.qw;
        rt[7] = st.vw.x; rt[8] = st.vw.y; rt[9] = st.vw.z; rt[10] = st.ww.x; rt[11] = st.ww.y; rt[12] = st.ww.z;
        cf[0] = base_f.x; cf[1] = base_f.y; cf[2] = base_f.z; cf[3] = hu_f.x; cf[4] = hu_f.y; cf[5] = hu_f.z; cf[6] = hl_f.x; cf[7] = hl_f.y; cf[8] = hl_f.z;
        V3 h1 = mul(R, v3(0.285f, 0.f, 0.01f)) + st.pos, h2 = mul(R, v3(0.293f, 0.f, -0.06f)) + st.pos;
        rb[0] = st.pos.x; rb[1] = st.pos.y; rb[2] = st.pos.z; rb[3] = h1.x; rb[4] = h1.y; rb[5] = h1.z; rb[6] = h2.x; rb[7] = h2.y; rb[8] = h2.z;
    }
}

// ------------------------------------------------------------------ fused GAE (rollout_storage.py:97-111)
// pass 1: one lane per env, reverse scan over T (coalesced across envs); per-block partial sums in double.
#define GAE_BLOCK 256
#define GAE_MAX_BLOCKS 128          // scratch = 128 x 2 doubles = 2 KB + 2 doubles of result
__global__ void __launch_bounds__(GAE_BLOCK) qa_gae_scan_kernel(const float *rewards, const float *values, const uint8_t *dones,
                                                                  const float *last_values, float *returns, float *advantages,
                                                                  int T, int N, float gamma, float lam, double *partial) {
    double sum = 0.0, sq = 0.0;
    for (int e = blockIdx.x * GAE_BLOCK + threadIdx.x; e < N; e += gridDim.x * GAE_BLOCK) {
        float adv = 0.f, nv = last_values[e];
        for (int t = T - 1; t >= 0; --t) {
            const int64_t i = (int64_t)t * N + e;
            const float v = values[i];
            const float nt = 1.0f - (float)dones[i];
            const float delta = rewards[i] + nt * gamma * nv - v;
            adv = delta + nt * gamma * lam * adv;
            const float ret = adv + v;
            returns[i] = ret;
            const float a = ret - v;
            advantages[i] = a;
            sum += (double)a; sq += (double)a * (double)a;
            nv = v;
        }
    }
    __shared__ double s_sum[GAE_BLOCK], s_sq[GAE_BLOCK];
    s_sum[threadIdx.x] = sum; s_sq[threadIdx.x] = sq;
    __syncthreads();
    for (int s = GAE_BLOCK / 2; s > 0; s >>= 1) {
        if (threadIdx.x < s) { s_sum[threadIdx.x] += s_sum[threadIdx.x + s]; s_sq[threadIdx.x] += s_sq[threadIdx.x + s]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { partial[2 * blockIdx.x] = s_sum[0]; partial[2 * blockIdx.x + 1] = s_sq[0]; }
}
// pass 2: every block re-reduces the <=128 partials in the same fixed order, then normalises its slice.
__global__ void __launch_bounds__(GAE_BLOCK) qa_gae_norm_kernel(float *advantages, int64_t n, const double *partial, int nblocks) {
    double sum = 0.0, sq = 0.0;
    for (int b = 0; b < nblocks; ++b) { sum += partial[2 * b]; sq += partial[2 * b + 1]; }
    const double mean = sum / (double)n;
    double var = (sq - (double)n * mean * mean) / (double)(n - 1);
    var = var > 0.0 ? var : 0.0;
    const float fm = (float)mean, inv = (float)(1.0 / (sqrt(var) + 1e-8));
    for (int64_t i = (int64_t)blockIdx.x * GAE_BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * GAE_BLOCK)
        advantages[i] = (advantages[i] - fm) * inv;
}

// ------------------------------------------------------------------ host side
static void fill_ptrs(qa_sim *s) {
    char *a = s->arena; const Layout &L = s->L; Ptrs &p = s->p;
#define FP(name, T_) p.name = (float *)(a + L.off[T_])
    FP(root, QA_T_ROOT_STATES); FP(dof, QA_T_DOF_STATE); FP(cforce, QA_T_CONTACT_FORCES); FP(rbpos, QA_T_RIGID_BODY_POS);
    FP(torques, QA_T_TORQUES); FP(torques_org, QA_T_TORQUES_ORG); FP(actions, QA_T_ACTIONS); FP(last_actions, QA_T_LAST_ACTIONS);
    FP(last_dof_vel, QA_T_LAST_DOF_VEL); FP(last_torques_org, QA_T_LAST_TORQUES_ORG); FP(last_root_vel, QA_T_LAST_ROOT_VEL);
    FP(action_hist, QA_T_ACTION_HISTORY); FP(obs, QA_T_OBS); FP(obs_disc, QA_T_OBS_DISC);
    FP(obs_disc_term, QA_T_OBS_DISC_TERM); FP(commands, QA_T_COMMANDS); FP(latent_eps, QA_T_LATENT_EPS); FP(latent_c, QA_T_LATENT_C);
    FP(rew, QA_T_REW); FP(episode_sums, QA_T_EPISODE_SUMS); FP(episode_stats, QA_T_EPISODE_STATS); FP(feet_force, QA_T_FEET_FORCE);
    FP(base_lin_vel, QA_T_BASE_LIN_VEL); FP(base_ang_vel, QA_T_BASE_ANG_VEL); FP(proj_grav, QA_T_PROJECTED_GRAVITY); FP(rpy, QA_T_RPY);
    FP(motor_strength, QA_T_MOTOR_STRENGTH); FP(mass_params, QA_T_MASS_PARAMS); FP(friction, QA_T_FRICTION); FP(env_origins, QA_T_ENV_ORIGINS);
    FP(base_inertia, QA_T_BASE_INERTIA); FP(prior, QA_T_PRIOR_PARAMETERS); FP(mocap, QA_T_MOCAP_FRAMES); FP(foot_impulse, QA_T_FOOT_IMPULSE); FP(scan_height, QA_T_SCAN_HEIGHT);
#undef FP
    p.reset = (int64_t *)(a + L.off[QA_T_RESET]); p.episode_length = (int64_t *)(a + L.off[QA_T_EPISODE_LENGTH]);
    p.time_out = (uint8_t *)(a + L.off[QA_T_TIME_OUT]); p.last_contacts = (uint8_t *)(a + L.off[QA_T_LAST_CONTACTS]);
    p.contact_filt = (uint8_t *)(a + L.off[QA_T_CONTACT_FILT]);
    p.height_samples = (int16_t *)(a + L.off[QA_T_HEIGHT_SAMPLES]);
    p.ceil_samples = (int16_t *)(a + L.off[QA_T_CEILING_SAMPLES]);
    p.obst_desc = (float *)(a + L.off[QA_T_OBST_DESC]); p.obst_state = (float *)(a + L.off[QA_T_OBST_STATE]);
    p.rbstate = (float *)(a + L.off[QA_T_RIGID_BODY_STATE]);
    p.mocap_clips = (double *)(a + L.off[QA_T_MOCAP_CLIPS]);
    p.ticket = (int32_t *)(a + L.off[QA_T_STEP_TICKET]);
}

static void build_table(float *t) {
    memset(t, 0, sizeof(float) * QA_TBL_FLOATS);
    for (int l = 0; l < 4; ++l) {
        float *g = t + l * QA_LEG_TBL;
        for (int i = 0; i < 3; ++i) {
            g[T_HIP_ORG + i] = QA_HIP_ORG[l][i]; g[T_THIGH_ORG + i] = QA_THIGH_ORG[l][i]; g[T_CALF_ORG + i] = QA_CALF_ORG[l][i]; g[T_FOOT_ORG + i] = QA_FOOT_ORG[l][i];
            g[T_MASS + i] = QA_LINK_MASS[l][i]; g[T_LOWER + i] = QA_DOF_LOWER[l][i]; g[T_UPPER + i] = QA_DOF_UPPER[l][i];
            g[T_EFFORT + i] = QA_DOF_EFFORT[l][i]; g[T_VELLIM + i] = QA_DOF_VELLIM[l][i];
            for (int j = 0; j < 3; ++j) g[T_COM + 3 * i + j] = QA_LINK_COM[l][i][j];
            for (int j = 0; j < 6; ++j) g[T_INERTIA + 6 * i + j] = QA_LINK_I[l][i][j];
        }
        for (int c = 0; c < QA_LEG_PTS; ++c) {
            for (int j = 0; j < 3; ++j) g[T_POINTS + 4 * c + j] = QA_LEG_PT_POS[l][c][j];
            g[T_POINTS + 4 * c + 3] = QA_LEG_PT_RAD[l][c];
        }
    }
    float *b = t + 4 * QA_LEG_TBL;
    for (int c = 0; c < QA_BASE_PTS; ++c) { for (int j = 0; j < 3; ++j) b[4 * c + j] = QA_BASE_PT_POS[c][j]; b[4 * c + 3] = QA_BASE_PT_RAD[c]; }
}

extern "C" {

int qa_abi_version(void) { return QA_ABI_VERSION; }
const char *qa_last_error(void) { return g_err; }

int64_t qa_arena_bytes(const qa_config *cfg) {
    if (!cfg || cfg->num_envs <= 0) return QA_E_ARG;
    Layout L; make_layout(cfg, &L); return L.total;
}

int qa_tensor_info(const qa_config *cfg, int which, int64_t *off, int64_t shape[3], int32_t *ndim, int32_t *dtype) {
    if (!cfg || which < 0 || which >= QA_T_COUNT || cfg->num_envs <= 0) return QA_E_ARG;
    Layout L; make_layout(cfg, &L);
    if (off) *off = L.off[which];
    if (shape) for (int i = 0; i < 3; ++i) shape[i] = L.shape[which][i];
    if (ndim) *ndim = L.ndim[which];
    if (dtype) *dtype = L.dtype[which];
    return QA_OK;
}

int qa_create(const qa_config *cfg, void *arena, int64_t arena_bytes, void *stream, qa_sim **out) {
    if (!cfg || !arena || !out || cfg->num_envs <= 0) { snprintf(g_err, sizeof(g_err), "qa_create: bad argument"); return QA_E_ARG; }
    if (cfg->abi_version != QA_ABI_VERSION) return QA_E_VERSION;
    if ((cfg->terrain_type != 0 && cfg->terrain_type != 1) || (cfg->terrain_type == 1 && (cfg->hf_rows < 2 || cfg->hf_cols < 2 || !(cfg->hf_hscale > 0.0f))) || cfg->decimation <= 0 || cfg->solver_iterations <= 0) { snprintf(g_err, sizeof(g_err), "qa_create: unsupported config"); return QA_E_ARG; }
    qa_sim *s = new qa_sim();
    s->cfg = *cfg; make_layout(cfg, &s->L); s->arena = (char *)arena;
    s->lanes = 4;                  // one quad per env (lane&3 = leg)
    memset(s->mocap_first, 0, sizeof(s->mocap_first));
    if (arena_bytes < s->L.total || ((uintptr_t)arena & 255)) { delete s; snprintf(g_err, sizeof(g_err), "qa_create: arena too small or misaligned"); return QA_E_ARENA; }
    fill_ptrs(s);
    hipStream_t st = (hipStream_t)stream;
    float tbl[QA_TBL_FLOATS]; build_table(tbl);
    { int dev = 0, cus = 0; if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0) s->num_cus = cus; }
    hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(c_tbl), tbl, sizeof(tbl));
    if (e != hipSuccess) { delete s; return fail_hip(e, "hipMemcpyToSymbol(c_tbl)"); }
    e = hipMemsetAsync(arena, 0, (size_t)s->L.total, st);
    if (e != hipSuccess) { delete s; return fail_hip(e, "hipMemsetAsync(arena)"); }
    if (cfg->terrain_type == 1 && cfg->hf_ceiling) {     // "no overhang anywhere" until the caller writes the ceiling field
        e = hipMemsetD16Async((hipDeviceptr_t)s->p.ceil_samples, (unsigned short)QA_NO_CEILING, (size_t)cfg->hf_rows * cfg->hf_cols, st);
        if (e != hipSuccess) { delete s; return fail_hip(e, "hipMemsetD16Async(ceiling)"); }
    }
    const int N = cfg->num_envs;
    BaseConst bc; bc.m = QA_BASE_MASS; for (int i = 0; i < 3; ++i) bc.com[i] = QA_BASE_COM[i]; for (int i = 0; i < 6; ++i) bc.I[i] = QA_BASE_I[i];
    hipLaunchKernelGGL(qa_init_kernel, dim3((N + 255) / 256), dim3(256), 0, st, s->cfg, s->p, bc);
    e = hipGetLastError();
    if (e != hipSuccess) { delete s; return fail_hip(e, "qa_init_kernel"); }
    *out = s;
    return QA_OK;
}

int qa_destroy(qa_sim *s) { delete s; return QA_OK; }

int qa_set_mocap(qa_sim *s, const float *frames, int32_t nf, const double *clips, int32_t nc, const int32_t first[QA_NUM_GAITS + 1], void *stream) {
    if (!s || !frames || !clips || !first || nf <= 0 || nf > s->cfg.num_mocap_frames || nc <= 0 || nc > QA_MAX_MOCAP_CLIPS) return QA_E_ARG;
    if (first[0] != 0 || first[QA_NUM_GAITS] != nc) return QA_E_ARG;
    for (int g = 0; g < QA_NUM_GAITS; ++g) if (first[g + 1] <= first[g]) { snprintf(g_err, sizeof(g_err), "qa_set_mocap: gait %d has no clip", g); return QA_E_ARG; }
    for (int i = 0; i < nc; ++i) {
        const double *r = clips + QA_MOCAP_CLIP * i;
        if (r[0] < 0 || r[1] < 2 || r[0] + r[1] > nf || !(r[2] > 0) || !(r[3] > 0) || r[3] > r[2]) { snprintf(g_err, sizeof(g_err), "qa_set_mocap: bad clip row %d", i); return QA_E_ARG; }
    }
    HIP_TRY(hipMemcpyAsync(s->p.mocap, frames, (size_t)nf * QA_MOCAP_FRAME * sizeof(float), hipMemcpyHostToDevice, (hipStream_t)stream));
    memset(s->clip_stage, 0, sizeof(s->clip_stage));
    for (int i = 0; i < nc; ++i) for (int k = 0; k < 5; ++k) s->clip_stage[QA_MOCAP_CLIP * i + k] = clips[QA_MOCAP_CLIP * i + k];
    for (int g = 0; g <= QA_NUM_GAITS; ++g) s->clip_stage[QA_MOCAP_CLIP * g + 5] = (double)first[g];
    HIP_TRY(hipMemcpyAsync(s->p.mocap_clips, s->clip_stage, sizeof(s->clip_stage), hipMemcpyHostToDevice, (hipStream_t)stream));
    memcpy(s->mocap_first, first, sizeof(s->mocap_first));
    return QA_OK;
}

static MocapIdx mocap_idx(const qa_sim *s) { MocapIdx m; m.on = s->mocap_first[QA_NUM_GAITS] > 0; return m; }

static void launch_env_step(qa_sim *s, const StepArgs &a, hipStream_t st) {
    const int epb = QA_BLOCK / s->lanes, blocks = (s->cfg.num_envs + epb - 1) / epb;
    const bool hf = s->cfg.terrain_type == 1;
    // helper wavefronts on the plane kernels while a launch has at most one workgroup per CU (<= 4096 envs: three of a CU's four SIMDs would idle;
    // measured 69.3 -> 63.1 us at 4096 envs, 135 -> 162 us at 16,384 where every SIMD has work of its own).  QA_ENV_HELPERS in the environment:
    // 0 = never (the one-wavefront kernels, A/B runs), 1 = always, unset = by launch size
    static const int helpers_switch = [] { const char *e = getenv("QA_ENV_HELPERS"); return e ? (atoi(e) != 0 ? 1 : 0) : QA_ENV_HELPERS_DEFAULT; }();
    const bool helpers = helpers_switch == 1 || (helpers_switch < 0 && blocks <= s->num_cus);
    if (!hf && helpers) {
        if (s->lean == 3) hipLaunchKernelGGL((qa_env_step_kernel<true, 4, 0, 3, 1>), dim3(blocks), dim3(3 * QA_BLOCK), 0, st, a);
        else if (s->lean == 1) hipLaunchKernelGGL((qa_env_step_kernel<true, 4, 0, 1, 1>), dim3(blocks), dim3(3 * QA_BLOCK), 0, st, a);
        else hipLaunchKernelGGL((qa_env_step_kernel<true, 4, 0, 0, 1>), dim3(blocks), dim3(3 * QA_BLOCK), 0, st, a);
        return;
    }
    if (s->lean == 3) {
        if (hf) hipLaunchKernelGGL((qa_env_step_kernel<false, 4, 0, 3>), dim3(blocks), dim3(QA_BLOCK), 0, st, a);
        else hipLaunchKernelGGL((qa_env_step_kernel<true, 4, 0, 3>), dim3(blocks), dim3(QA_BLOCK), 0, st, a);
    } else if (s->lean == 1) {
        if (hf) hipLaunchKernelGGL((qa_env_step_kernel<false, 4, 0, 1>), dim3(blocks), dim3(QA_BLOCK), 0, st, a);
        else hipLaunchKernelGGL((qa_env_step_kernel<true, 4, 0, 1>), dim3(blocks), dim3(QA_BLOCK), 0, st, a);
    } else {
        if (hf) hipLaunchKernelGGL((qa_env_step_kernel<false, 4>), dim3(blocks), dim3(QA_BLOCK), 0, st, a);
        else hipLaunchKernelGGL((qa_env_step_kernel<true, 4>), dim3(blocks), dim3(QA_BLOCK), 0, st, a);
    }
}

int qa_set_lean_exports(qa_sim *s, int32_t mask) {
    if (!s || (mask != 0 && mask != 1 && mask != 3)) { snprintf(g_err, sizeof(g_err), "qa_set_lean_exports: mask must be 0, 1 or 3"); return QA_E_ARG; }
    s->lean = mask;
    return QA_OK;
}

int qa_env_step(qa_sim *s, const float *actions, int32_t delay_steps, int64_t global_step, void *stream) {
    if (!s || !actions || delay_steps < 0 || delay_steps >= QA_ACTION_BUF_LEN) return QA_E_ARG;
    if (s->lean && delay_steps > 1) { snprintf(g_err, sizeof(g_err), "qa_env_step: lean exports keep two action slots (delay <= 1)"); return QA_E_ARG; }
    StepArgs a; a.c = s->cfg; a.p = s->p; a.mi = mocap_idx(s); a.actions = actions; a.delay = delay_steps; a.step = global_step; a.step_ptr = nullptr; a.prof = s->prof;
    launch_env_step(s, a, (hipStream_t)stream);
    HIP_TRY(hipGetLastError());
    return QA_OK;
}

int qa_env_step_dev(qa_sim *s, const float *actions, int32_t delay_steps, int64_t *step_counter_dev, void *stream) {
    if (!s || !actions || !step_counter_dev || delay_steps < 0 || delay_steps >= QA_ACTION_BUF_LEN) return QA_E_ARG;
    if (s->lean && delay_steps > 1) { snprintf(g_err, sizeof(g_err), "qa_env_step_dev: lean exports keep two action slots (delay <= 1)"); return QA_E_ARG; }
    StepArgs a; a.c = s->cfg; a.p = s->p; a.mi = mocap_idx(s); a.actions = actions; a.delay = delay_steps; a.step = 0;
    a.step_ptr = step_counter_dev; a.prof = s->prof;
    launch_env_step(s, a, (hipStream_t)stream);
    HIP_TRY(hipGetLastError());
    return QA_OK;
}

int qa_env_physics_step(qa_sim *s, const float *actions, int32_t delay_steps, void *stream) {
    if (!s || !actions || delay_steps < 0 || delay_steps >= QA_ACTION_BUF_LEN) return QA_E_ARG;
    StepArgs a; a.c = s->cfg; a.p = s->p; a.mi = mocap_idx(s); a.actions = actions; a.delay = delay_steps; a.step = 0; a.step_ptr = nullptr; a.prof = nullptr;
    const int blocks = (s->cfg.num_envs + ENVS_PER_BLOCK - 1) / ENVS_PER_BLOCK;
    if (s->cfg.terrain_type == 1) hipLaunchKernelGGL((qa_env_step_kernel<false, 4, 1>), dim3(blocks), dim3(QA_BLOCK), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL((qa_env_step_kernel<true, 4, 1>), dim3(blocks), dim3(QA_BLOCK), 0, (hipStream_t)stream, a);
    HIP_TRY(hipGetLastError());
    return QA_OK;
}

int qa_reset_all(qa_sim *s, int64_t global_step, void *stream) {
    if (!s) return QA_E_ARG;
    const int blocks = (s->cfg.num_envs + ENVS_PER_BLOCK - 1) / ENVS_PER_BLOCK;
    hipLaunchKernelGGL(qa_reset_all_kernel, dim3(blocks), dim3(QA_BLOCK), 0, (hipStream_t)stream, s->cfg, s->p, mocap_idx(s), global_step);
    HIP_TRY(hipGetLastError());
    return QA_OK;
}

static int launch_simulate(qa_sim *s, const float *torques, const uint8_t *cond, void *stream) {
    if (!s) return QA_E_ARG;
    if (!torques) torques = s->p.torques;          // the actuation forces set last (gym keeps applying them)
    const int epb = QA_BLOCK / s->lanes, blocks = (s->cfg.num_envs + epb - 1) / epb;
    if (s->cfg.terrain_type == 1) hipLaunchKernelGGL((qa_simulate_kernel<false, 4>), dim3(blocks), dim3(QA_BLOCK), 0, (hipStream_t)stream, s->cfg, s->p, torques, cond);
    else hipLaunchKernelGGL((qa_simulate_kernel<true, 4>), dim3(blocks), dim3(QA_BLOCK), 0, (hipStream_t)stream, s->cfg, s->p, torques, cond);
    HIP_TRY(hipGetLastError());
    return QA_OK;
}

int qa_simulate(qa_sim *s, const float *torques, void *stream) {
    if (!torques) return QA_E_ARG;
    return launch_simulate(s, torques, nullptr, stream);
}

int qa_simulate_if(qa_sim *s, const float *torques, const uint8_t *cond_dev, void *stream) {
    if (!cond_dev) return QA_E_ARG;
    return launch_simulate(s, torques, cond_dev, stream);
}

// reset_idx of the task-level env for the flagged envs (tsc/legged_gym/envs/base/legged_robot.py:348-410, 796-884): one thread per env
__global__ void qa_tsc_reset_kernel(qa_config c, Ptrs p, const uint8_t *flags, const float *start_xy, const float *start_yaw, float yaw_range,
                                    float x_range, float y_range, float pitch_range, int64_t step, const int64_t *step_ptr) {
    if (step_ptr) step = *step_ptr;
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= c.num_envs) return;
    float *lr = p.last_root_vel + (int64_t)e * 6;
#pragma unroll
    for (int i = 0; i < 6; ++i) lr[i] = 0.f;                       // self.last_root_vel[:] = 0.  (ALL envs, :389)
    if (!flags[e]) return;
    const F4 u = rng4(c.seed, e + c.env_id_offset, step, RS_RESET, 0);
    const float yaw = start_yaw[e] + yaw_range * (2.0f * u.v[0] - 1.0f);
    const float pitch = pitch_range * (2.0f * u.v[3] - 1.0f);
    float *rt = p.root + (int64_t)e * 13;
    rt[0] = c.init_pos[0] + start_xy[2 * e] + x_range * (u.v[1] - 1.0f);            // rand_x_range * U(-1, 0)
    rt[1] = c.init_pos[1] + start_xy[2 * e + 1] + y_range * (2.0f * u.v[2] - 1.0f);
    rt[2] = c.init_pos[2];
    {   // quat_from_euler_xyz(0, pitch, yaw), xyzw (isaacgym torch_utils)
        float sy, cy, sp, cp; sincosf(0.5f * yaw, &sy, &cy); sincosf(0.5f * pitch, &sp, &cp);
        rt[3] = -sy * sp; rt[4] = cy * sp; rt[5] = sy * cp; rt[6] = cy * cp;
    }
#pragma unroll
    for (int i = 7; i < 13; ++i) rt[i] = 0.f;
    float *d = p.dof + (int64_t)e * 24;
#pragma unroll
    for (int j = 0; j < 12; ++j) {
        d[2 * j] = c.default_dof_pos[j]; d[2 * j + 1] = 0.f;
        const int64_t k = (int64_t)e * 12 + j;
        p.last_actions[k] = 0.f; p.last_dof_vel[k] = 0.f; p.last_torques_org[k] = 0.f; p.foot_impulse[k] = 0.f;
    }
    float *ah = p.action_hist + (int64_t)e * (QA_ACTION_BUF_LEN * 12);
    for (int i = 0; i < QA_ACTION_BUF_LEN * 12; ++i) ah[i] = 0.f;
    p.reset[e] = 1;
}

int qa_tsc_reset(qa_sim *s, const uint8_t *reset_flags, const float *start_xy, const float *start_yaw, float rand_yaw_range, float rand_x_range,
                 float rand_y_range, float rand_pitch_range, int64_t global_step, void *stream) {
    if (!s || !reset_flags || !start_xy || !start_yaw) return QA_E_ARG;
    const int N = s->cfg.num_envs;
    hipLaunchKernelGGL(qa_tsc_reset_kernel, dim3((N + 63) / 64), dim3(64), 0, (hipStream_t)stream, s->cfg, s->p, reset_flags, start_xy, start_yaw,
                       rand_yaw_range, rand_x_range, rand_y_range, rand_pitch_range, global_step, (const int64_t *)nullptr);
    HIP_TRY(hipGetLastError());
    return QA_OK;
}

int qa_tsc_reset_dev(qa_sim *s, const uint8_t *reset_flags, const float *start_xy, const float *start_yaw, float rand_yaw_range, float rand_x_range,
                     float rand_y_range, float rand_pitch_range, const int64_t *global_step_dev, void *stream) {
    if (!s || !reset_flags || !start_xy || !start_yaw || !global_step_dev) return QA_E_ARG;
    const int N = s->cfg.num_envs;
    hipLaunchKernelGGL(qa_tsc_reset_kernel, dim3((N + 63) / 64), dim3(64), 0, (hipStream_t)stream, s->cfg, s->p, reset_flags, start_xy, start_yaw,
                       rand_yaw_range, rand_x_range, rand_y_range, rand_pitch_range, (int64_t)0, global_step_dev);
    HIP_TRY(hipGetLastError());
    return QA_OK;
}

int qa_debug_post_physics(qa_sim *s, int64_t global_step, void *stream) {
    if (!s) return QA_E_ARG;
    StepArgs a; a.c = s->cfg; a.p = s->p; a.mi = mocap_idx(s); a.actions = nullptr; a.delay = 0; a.step = global_step; a.step_ptr = nullptr; a.prof = nullptr;
    const int blocks = (s->cfg.num_envs + ENVS_PER_BLOCK - 1) / ENVS_PER_BLOCK;
    if (s->cfg.terrain_type == 1) hipLaunchKernelGGL((qa_post_physics_kernel<false>), dim3(blocks), dim3(QA_BLOCK), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL((qa_post_physics_kernel<true>), dim3(blocks), dim3(QA_BLOCK), 0, (hipStream_t)stream, a);
    HIP_TRY(hipGetLastError());
    return QA_OK;
}

/* development aid (not part of include/qa_sim.h): per-block s_memtime stamps of the env-step phases, 16 slots per block */
int qa_debug_set_profile_buffer(qa_sim *s, long long *dev_buf) {
    if (!s) return QA_E_ARG;
    s->prof = dev_buf;
#ifdef QA_SUBPROF
    hipMemcpyToSymbol(HIP_SYMBOL(g_subprof), &dev_buf, sizeof(dev_buf));
#endif
    return QA_OK;
}

int qa_gae(const float *rewards, const float *values, const uint8_t *dones, const float *last_values, float *returns,
           float *advantages, int32_t T, int32_t N, float gamma, float lam, int32_t normalize, void *scratch, void *stream) {
    if (!rewards || !values || !dones || !last_values || !returns || !advantages || !scratch || T <= 0 || N <= 0) return QA_E_ARG;
    int blocks = (N + GAE_BLOCK - 1) / GAE_BLOCK; if (blocks > GAE_MAX_BLOCKS) blocks = GAE_MAX_BLOCKS;
    hipLaunchKernelGGL(qa_gae_scan_kernel, dim3(blocks), dim3(GAE_BLOCK), 0, (hipStream_t)stream, rewards, values, dones, last_values,
                       returns, advantages, T, N, gamma, lam, (double *)scratch);
    HIP_TRY(hipGetLastError());
    if (normalize) {
        const int64_t n = (int64_t)T * N;
        int nb = (int)((n + GAE_BLOCK - 1) / GAE_BLOCK); if (nb > 1024) nb = 1024;
        hipLaunchKernelGGL(qa_gae_norm_kernel, dim3(nb), dim3(GAE_BLOCK), 0, (hipStream_t)stream, advantages, n, (const double *)scratch, blocks);
        HIP_TRY(hipGetLastError());
    }
    return QA_OK;
}

}  // extern "C"
