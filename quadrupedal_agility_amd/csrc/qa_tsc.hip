// Task-level (TSC) env-side math for gfx950: the command mapping and the goal / termination / reward bookkeeping of
// tsc/legged_gym/envs/base/legged_robot.py that need no obstacle physics (include/qa_sim.h has the
// contracts).  Both are one-thread-per-env streaming kernels: a few hundred bytes in and out per env, no reuse, so they are
// launch/latency-bound at 8192 envs (128 wavefronts) and exist to replace ~150 eager launches per step, not to move bytes.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/qa_sim.h"

extern thread_local char qa_err_buf[512];
#define g_terr qa_err_buf

namespace {

constexpr int TSC_BLOCK = 64;          // one wavefront per workgroup: 8192 envs = 128 workgroups spread over 128 CUs (256-thread blocks would use 32)
constexpr int TSC_MAX_DIM_C = 8;
constexpr int TSC_NUM_C = 6;

struct SetCmdArgs {
    const float *actions; const int64_t *episode_length; const float *noise;
    float *commands, *latent_eps, *latent_c, *next_commands;
    int64_t n;
    int num_d, dim_c, interval;
    int mocap_index[TSC_MAX_DIM_C];
    float vel[3][TSC_MAX_DIM_C][2];
    float jump[2], height[2];
};

__global__ void __launch_bounds__(TSC_BLOCK) qa_tsc_set_commands_kernel(SetCmdArgs a) {
    const int64_t e = (int64_t)blockIdx.x * TSC_BLOCK + threadIdx.x;
    if (e >= a.n) return;
    const int width = 1 + a.num_d * TSC_NUM_C;
    float cmd[5], eps, c[TSC_MAX_DIM_C];
    if (a.episode_length[e] % a.interval == 0) {
        const float *row = a.actions + e * width;
        int id = (int)row[0];                      // .to(torch.long): truncation
        id = id < 0 ? 0 : (id >= a.num_d ? a.num_d - 1 : id);
        const int g = a.mocap_index[id];
        float u[TSC_NUM_C];
#pragma unroll
        for (int k = 0; k < TSC_NUM_C; ++k) {
            const float p = fminf(fmaxf(row[1 + id * TSC_NUM_C + k], -1.0f), 1.0f);
            if (k == TSC_NUM_C - 1) eps = p;
            u[k] = (p + 1.0f) / 2.0f;
        }
        for (int k = 0; k < a.dim_c; ++k) c[k] = k == g ? 1.0f : 0.0f;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float lo = a.vel[k][g][0], hi = a.vel[k][g][1];
            cmd[k] = lo + (hi - lo) * u[k];
        }
        const bool jump = g == a.dim_c - 1;
        cmd[3] = (a.jump[0] + (a.jump[1] - a.jump[0]) * u[3]) * (jump ? 1.0f : 0.0f);
        cmd[4] = (a.height[0] + (a.height[1] - a.height[0]) * u[4]) * (jump ? 0.0f : 1.0f);
    } else {
#pragma unroll
        for (int k = 0; k < 5; ++k) cmd[k] = a.commands[e * 5 + k];
        eps = a.latent_eps[e];
        for (int k = 0; k < a.dim_c; ++k) c[k] = a.latent_c[e * a.dim_c + k];
    }
    if (a.noise) {
#pragma unroll
        for (int k = 0; k < 5; ++k) cmd[k] *= a.noise[e * 5 + k];
    }
    float *nx = a.next_commands + e * (6 + a.dim_c);
#pragma unroll
    for (int k = 0; k < 5; ++k) { a.commands[e * 5 + k] = cmd[k]; nx[k] = cmd[k]; }
    a.latent_eps[e] = eps; nx[5] = eps;
    for (int k = 0; k < a.dim_c; ++k) { a.latent_c[e * a.dim_c + k] = c[k]; nx[6 + k] = c[k]; }
}

// v rotated into the frame of q (xyzw), the isaacgym.torch_utils.quat_rotate_inverse formula:  v (2w^2 - 1) - 2w (q x v) + 2 q (q . v)
__device__ inline void rotate_inverse(const float q[4], const float v[3], float out[3]) {
    const float w = q[3];
    const float s = 2.0f * w * w - 1.0f;
    const float cx = q[1] * v[2] - q[2] * v[1], cy = q[2] * v[0] - q[0] * v[2], cz = q[0] * v[1] - q[1] * v[0];
    const float d = q[0] * v[0] + q[1] * v[1] + q[2] * v[2];
    out[0] = v[0] * s - cx * w * 2.0f + q[0] * d * 2.0f;
    out[1] = v[1] * s - cy * w * 2.0f + q[1] * d * 2.0f;
    out[2] = v[2] * s - cz * w * 2.0f + q[2] * d * 2.0f;
}

__device__ inline float norm3(const float *f) { return sqrtf(f[0] * f[0] + f[1] * f[1] + f[2] * f[2]); }

// python's float modulo (result has the sign of b)
__device__ inline float floor_mod(float a, float b) {
    float r = fmodf(a, b);
    if (r != 0.0f && ((r < 0.0f) != (b < 0.0f))) r += b;
    return r;
}

// (Staging the 64 envs' array-of-structure records -- root state, contact forces, task-action history -- through LDS with
// coalesced loads was measured and is SLOWER here: 17.9 us against 12.5 us at 8192 envs.  With one wavefront per CU nothing
// hides the extra load -> LDS -> barrier round trips, while the direct per-thread loads are all issued up front.)
// term_mask / pen_mask: bit b set = body b is in the termination / penalised list.  Every body's contact-force norm is taken
// once in a branch-free loop (independent loads, one exposed round trip) and the lists become bit tests; a loop over each
// list with `||` short-circuits costs one dependent round trip per listed body (11 + 15 of them for the Go2).
__global__ void __launch_bounds__(TSC_BLOCK) qa_tsc_goal_step_kernel(qa_tsc_goal_cfg c, qa_tsc_goal_io io, uint32_t term_mask, uint32_t pen_mask) {
    const int64_t e = (int64_t)blockIdx.x * TSC_BLOCK + threadIdx.x;
    if (e >= c.num_envs) return;
    const int64_t N = c.num_envs;
    const float *rs = io.root_states + e * 13;
    const float *cf = io.contact_forces + e * c.num_bodies * 3;
    const int64_t ep_len = io.episode_length[e] + 1;
    io.episode_length[e] = ep_len;

    // body-frame quantities and euler angles
    const float q[4] = {rs[3], rs[4], rs[5], rs[6]};
    const float grav[3] = {0.0f, 0.0f, -1.0f};
    float lin[3], ang[3], pg[3];
    rotate_inverse(q, rs + 7, lin);
    rotate_inverse(q, rs + 10, ang);
    rotate_inverse(q, grav, pg);
    const float x = q[0], y = q[1], z = q[2], w = q[3];
    const float roll = atan2f(2.0f * (w * x + y * z), 1.0f - 2.0f * (x * x + y * y));
    const float pitch = asinf(fminf(fmaxf(2.0f * (w * y - z * x), -1.0f), 1.0f));
    const float yaw = atan2f(2.0f * (w * z + x * y), 1.0f - 2.0f * (y * y + z * z));
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        io.base_lin_vel[e * 3 + k] = lin[k]; io.base_ang_vel[e * 3 + k] = ang[k]; io.projected_gravity[e * 3 + k] = pg[k];
    }
    io.rpy[e * 3 + 0] = roll; io.rpy[e * 3 + 1] = pitch; io.rpy[e * 3 + 2] = yaw;

    // contact-force norms of all bodies, as three threshold bit sets
    uint32_t over01 = 0, over1 = 0, over2 = 0;
    for (int b = 0; b < c.num_bodies; ++b) {
        const float f = norm3(cf + b * 3);
        over01 |= (uint32_t)(f > 0.1f) << b; over1 |= (uint32_t)(f > 1.0f) << b; over2 |= (uint32_t)(f > 2.0f) << b;
    }
    // filtered foot contacts
    bool filt[4];
#pragma unroll
    for (int f = 0; f < 4; ++f) {
        const bool now = (over2 >> c.feet_bodies[f]) & 1u;
        filt[f] = now || io.last_contacts[e * 4 + f] != 0;
        io.last_contacts[e * 4 + f] = now;
        io.contact_filt[e * 4 + f] = filt[f];
    }

    // goals: advance after the dwell, distances to the goal gathered at the end of the previous step
    int64_t gi = io.cur_goal_idx[e];
    float timer = io.reach_goal_timer[e];
    if (timer > c.reach_goal_delay_steps) { gi += 1; timer = 0.0f; }
    const float gx = io.cur_goals[e * 3 + 0], gy = io.cur_goals[e * 3 + 1];
    const float nx = io.next_goals[e * 3 + 0], ny = io.next_goals[e * 3 + 1];
    const float dxr = rs[0] - gx, dyr = rs[1] - gy;
    const float dist = sqrtf(dxr * dxr + dyr * dyr);
    const bool reached = dist < c.next_goal_threshold, leave = dist > c.leave_goal_threshold;
    if (reached) timer += 1.0f;
    io.cur_goal_idx[e] = gi;
    io.reach_goal_timer[e] = timer;
    io.reached_goal[e] = reached;
    const float tx = gx - rs[0], ty = gy - rs[1];
    const float tn = sqrtf(tx * tx + ty * ty);
    const float tvx = tx / (tn + 1e-5f), tvy = ty / (tn + 1e-5f);
    const float target_yaw = atan2f(tvy, tvx);
    const float ux = nx - rs[0], uy = ny - rs[1];
    const float un = sqrtf(ux * ux + uy * uy);
    const float next_yaw = atan2f(uy / (un + 1e-5f), ux / (un + 1e-5f));
    io.target_pos_rel[e * 2 + 0] = tx; io.target_pos_rel[e * 2 + 1] = ty;
    io.next_target_pos_rel[e * 2 + 0] = ux; io.next_target_pos_rel[e * 2 + 1] = uy;
    io.target_yaw[e] = target_yaw; io.next_target_yaw[e] = next_yaw;

    // the obstacle the current goal belongs to
    int64_t gclamp = gi < 0 ? 0 : gi;
    const int64_t gmax = c.num_goal_slots - c.last_goal_repeat - 1;
    gclamp = gclamp > gmax ? gmax : gclamp;
    int ob = (int)(gclamp / c.goals_per_obstacle);
    ob = ob >= c.num_obstacles ? c.num_obstacles - 1 : ob;
    const int64_t otype = io.obstacle_types[e * c.num_obstacles + ob];
    io.cur_obstacle_type[e] = otype;

    // termination
    bool reset = (over1 & term_mask) != 0;
    const bool goal_cut = gi >= (int64_t)(c.num_goal_slots - c.last_goal_repeat);
    const bool time_out = ((float)ep_len > c.max_episode_length) || goal_cut;
    reset = reset || time_out || fabsf(roll) > 1.5f || fabsf(pitch) > 1.5f || rs[2] < -0.25f || leave;
    if (c.use_camera) {
        const float *lg = io.env_goals + (e * c.num_goal_slots + (c.num_goal_slots - c.last_goal_repeat)) * 3;
        const float lx = rs[0] - lg[0], ly = rs[1] - lg[1];
        reset = reset || sqrtf(lx * lx + ly * ly) < c.next_goal_threshold;
    }
    io.reset_buf[e] = reset; io.time_out_buf[e] = time_out; io.reach_goal_cutoff[e] = goal_cut;

    // rewards, summed in the reference's order
    float term[QA_TSC_NUM_REWARDS];
    if (io.action_hl_history) {
        const float *h = io.action_hl_history + e * c.history_len * c.history_width;
        const float *h1 = h + (c.history_len - 1) * c.history_width, *h2 = h + (c.history_len - 2) * c.history_width,
                    *h3 = h + (c.history_len - 3) * c.history_width;
        float ss = 0.0f;
        for (int k = 0; k < c.history_width; ++k) { const float d = h2[k] - h1[k]; ss += d * d; }
        term[QA_TSC_REW_ACTION_HL_RATE] = sqrtf(ss);
        term[QA_TSC_REW_LATENT_C_RATE] = 0.5f * (fabsf(h3[0] - h1[0]) + fabsf(h2[0] - h1[0]));
    } else {
        term[QA_TSC_REW_ACTION_HL_RATE] = 0.0f;
        term[QA_TSC_REW_LATENT_C_RATE] = 0.0f;
    }
    term[QA_TSC_REW_COLLISION] = (float)__popc(over01 & pen_mask);
    int edge = 0;
#pragma unroll
    for (int f = 0; f < 4; ++f) {
        const float *fp = io.rigid_body_states + (e * c.num_bodies + c.feet_bodies[f]) * 13;
        int64_t ix = (int64_t)rintf((fp[0] + c.border_size) / c.horizontal_scale);      // torch.round: half to even
        int64_t iy = (int64_t)rintf((fp[1] + c.border_size) / c.horizontal_scale);
        ix = ix < 0 ? 0 : (ix > c.mask_rows - 1 ? c.mask_rows - 1 : ix);
        iy = iy < 0 ? 0 : (iy > c.mask_cols - 1 ? c.mask_cols - 1 : iy);
        const uint8_t at_edge = io.x_edge_mask[ix * c.mask_cols + iy];           // unconditional: no load behind a branch
        edge += (int)filt[f] & (int)(at_edge != 0);
    }
    term[QA_TSC_REW_FEET_EDGE] = (float)edge;
    term[QA_TSC_REW_REACH_GOAL] = reached ? 1.0f : 0.0f;
    const float vt = (otype == 0 || otype == 4) ? 2.5f : c.target_lin_vel;
    term[QA_TSC_REW_TRACKING_GOAL_VEL] = fminf(tvx * rs[7] + tvy * rs[8], vt) / (vt + 1e-5f);
    const float PI_F = 3.14159265358979323846f;
    const float dyaw = floor_mod((target_yaw - yaw) + PI_F, 2.0f * PI_F) - PI_F;
    term[QA_TSC_REW_TRACKING_YAW] = expf(-fabsf(dyaw));
    term[QA_TSC_REW_TERMINATION] = (reset && !time_out) ? 1.0f : 0.0f;
    float total = 0.0f;
#pragma unroll
    for (int k = 0; k < QA_TSC_NUM_REWARDS - 1; ++k) {
        const float r = term[k] * c.reward_scales[k];
        total += r;
        io.episode_sums[k * N + e] += r;
    }
    total = fmaxf(total, 0.0f);
    const float rt = term[QA_TSC_REW_TERMINATION] * c.reward_scales[QA_TSC_REW_TERMINATION];
    total += rt;
    io.episode_sums[(int64_t)QA_TSC_REW_TERMINATION * N + e] += rt;
    io.rew_buf[e] = total;

    // gather of the goals for the next step
    int64_t g0 = gi < 0 ? 0 : (gi > c.num_goal_slots - 1 ? c.num_goal_slots - 1 : gi);
    int64_t g1 = gi + 1 < 0 ? 0 : (gi + 1 > c.num_goal_slots - 1 ? c.num_goal_slots - 1 : gi + 1);
    const float *eg = io.env_goals + e * c.num_goal_slots * 3;
#pragma unroll
    for (int k = 0; k < 3; ++k) { io.cur_goals[e * 3 + k] = eg[g0 * 3 + k]; io.next_goals[e * 3 + k] = eg[g1 * 3 + k]; }
}


// ---------------------------------------------------------------------------------------------------------------
// Height scan + observation assembly: one wavefront per env, 4 envs per workgroup.  The 800-wide row is built in LDS
// (segments filled by the lanes that own them, the 570-float history streamed in), then the three observation rows and
// the pushed history go out as contiguous, lane-strided stores: 8.4 KB written and 2.9 KB read per env, HBM-bound.
constexpr int OBS_WAVES = 4;
constexpr int OBS_ROW = QA_TSC_NUM_OBS + 12;            // the 800 row + [commands, latent_eps, latent_c] + 1 pad (16-byte rows)
constexpr int OFF_YAW = 57, OFF_TYPE = 59, OFF_SCAN = 65, OFF_PRIV = 197, OFF_LATENT = 201, OFF_HIST = 230, OFF_CMD = 800;

__device__ inline float clampf(float v, float lo, float hi) { return fminf(fmaxf(v, lo), hi); }

__global__ void __launch_bounds__(64 * OBS_WAVES) qa_tsc_observations_kernel(qa_tsc_obs_cfg c, qa_tsc_obs_io io) {
    // A wavefront's lifetime is a chain of memory latencies, so the code is written load-first: every lane picks its source
    // ADDRESS with selects, all loads of a phase are issued back to back with no branch between them, and only then do the
    // values go through the (select-based) arithmetic into LDS.  A load inside each arm of an if-chain costs one exposed
    // round trip per arm (the first version: ~40 of them, 37.7 us per 8192-env launch).
    __shared__ __align__(16) float s_row[OBS_WAVES][OBS_ROW];
    __shared__ float s_meas[OBS_WAVES][QA_TSC_NUM_SCAN];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const int64_t N = c.num_envs;
    const int64_t e_raw = (int64_t)blockIdx.x * OBS_WAVES + wave;
    const bool valid = e_raw < N;
    const int64_t e = valid ? e_raw : N - 1;          // a ragged tail recomputes the last env and stores nothing
    float *row = s_row[wave], *meas = s_meas[wave];
    const float *rs = io.root_states + e * 13;
    const float PI_F = 3.14159265358979323846f;

    // ---- phase A loads
    const float r0 = rs[0], r1 = rs[1], rz = rs[2], q0 = rs[3], q1 = rs[4], q2 = rs[5], q3 = rs[6];
    // proprio: one float source per lane (< 41), value = (x - sub) * mul
    const float *psrc = io.rpy + e * 3 + (lane < 2 ? lane : 0);
    float psub = 0.0f, pmul = 1.0f;
    const float dall = c.default_dof_pos_all[lane >= 5 && lane < 17 ? lane - 5 : 0];
    if (lane >= 2 && lane < 5) { psrc = io.base_ang_vel + e * 3 + (lane - 2); pmul = c.ang_vel; }
    if (lane >= 5 && lane < 17) { psrc = io.dof_pos + e * 12 + (lane - 5); psub = dall; pmul = c.dof_pos; }
    if (lane >= 17 && lane < 29) { psrc = io.dof_vel + e * 12 + (lane - 17); pmul = c.dof_vel; }
    if (lane >= 29 && lane < 41) psrc = io.last_action + e * c.action_stride + (lane - 29);
    const float px_ = *psrc;
    const uint8_t pcf2 = io.contact_filt[e * 4 + ((lane + 3) & 3)];      // lanes 41..44 read foot (lane - 41) = (lane + 3) & 3
    // yaw errors (lanes 0, 1)
    float *ykeep = (lane & 1) ? io.delta_next_yaw : io.delta_yaw;
    const float yold = ykeep[e];
    const float ytarget = ((lane & 1) ? io.next_target_yaw : io.target_yaw)[e];
    const float yaw = io.rpy[e * 3 + 2];
    const int64_t otype = io.cur_obstacle_type[e];
    // scan points: 3 per lane (132 = 64 + 64 + 4), addresses clamped, stores predicated
    float hbx[3], hby[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int p = lane + 64 * k < QA_TSC_NUM_SCAN ? lane + 64 * k : QA_TSC_NUM_SCAN - 1;
        const float *hp = io.height_points + e * c.points_env_stride + p * c.point_stride;
        hbx[k] = hp[0]; hby[k] = hp[1];
    }
    const float plin = io.base_lin_vel[e * 3 + (lane < 3 ? lane : 0)];
    // privileged latent (lanes < 29): value = x - sub
    const float *lsrc = io.mass_params + e * 4 + (lane < 4 ? lane : 0);
    if (lane == 4) lsrc = io.friction + e;
    if (lane >= 5 && lane < 17) lsrc = io.motor_strength + e * 12 + (lane - 5);
    if (lane >= 17 && lane < 29) lsrc = io.motor_strength + (N + e) * 12 + (lane - 17);
    const float lx_ = *lsrc;
    // commands block (lanes < 11)
    const float *csrc = io.commands + e * 5 + (lane < 5 ? lane : 0);
    if (lane == 5) csrc = io.latent_eps + e;
    if (lane >= 6 && lane < 11) csrc = io.latent_c + e * 5 + (lane - 6);
    const float cx_ = *csrc;
    // the history before this step's push: 285 8-byte words, 5 per lane
    float2 hv[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        const int i = lane + 64 * k < 285 ? lane + 64 * k : 284;
        hv[k] = reinterpret_cast<const float2 *>(io.obs_history + e * 570)[i];
    }
    const int64_t ep_len = io.episode_length[e];
    // discriminator row sources (lanes < 49), used in phase B but loaded here
    const float *dsrc = io.rpy + e * 3 + (lane < 2 ? lane : 0);
    float dsub = 0.0f, dmul = 1.0f;
    const float dd = c.default_dof_pos[lane >= 9 && lane < 21 ? lane - 9 : 0];
    if (lane >= 3 && lane < 6) { dsrc = io.base_lin_vel + e * 3 + (lane - 3); dmul = c.lin_vel_dist; }
    if (lane >= 6 && lane < 9) { dsrc = io.base_ang_vel + e * 3 + (lane - 6); dmul = c.ang_vel_dist; }
    if (lane >= 9 && lane < 21) { dsrc = io.dof_pos + e * 12 + (lane - 9); dsub = dd; dmul = c.dof_pos; }
    if (lane >= 21 && lane < 33) { dsrc = io.dof_vel + e * 12 + (lane - 21); dmul = c.dof_vel; }
    const float dx_ = *dsrc;
    const int kl = lane >= 33 && lane < 45 ? lane - 33 : 0;
    const int kb = kl / 3, kax = kl - kb * 3;
    const float *bp = io.rigid_body_states + (e * c.num_bodies + c.key_bodies[kb]) * 13;
    const float b0 = bp[0], b1 = bp[1], b2 = bp[2];
    const uint8_t dcf = io.contact_filt[e * 4 + ((lane + 3) & 3)];      // lanes 45..48 read foot (lane - 45) = (lane + 3) & 3

    // ---- scan gathers (depend on the points only)
    const float qn = fmaxf(sqrtf(q2 * q2 + q3 * q3), 1e-9f);
    const float qz = q2 / qn, qw = q3 / qn;
    int16_t g1[3], g2[3], g3[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float bx = hbx[k], by = hby[k];
        const float t0 = (0.0f - qz * by) * 2.0f, t1 = (qz * bx - 0.0f) * 2.0f;
        const float wx = (bx + qw * t0) + (0.0f - qz * t1) + r0;
        const float wy = (by + qw * t1) + (qz * t0 - 0.0f) + r1;
        int64_t px = (int64_t)((wx + c.border_size) / c.horizontal_scale);
        int64_t py = (int64_t)((wy + c.border_size) / c.horizontal_scale);
        px = px < 0 ? 0 : (px > c.map_rows - 2 ? c.map_rows - 2 : px);
        py = py < 0 ? 0 : (py > c.map_cols - 2 ? c.map_cols - 2 : py);
        g1[k] = io.height_samples[px * c.map_cols + py];
        g2[k] = io.height_samples[(px + 1) * c.map_cols + py];
        g3[k] = io.height_samples[px * c.map_cols + py + 1];
    }

    // ---- phase A arithmetic into LDS
    if (lane < QA_TSC_NUM_PROPRIO) {
        const float v = lane < 41 ? (px_ - psub) * pmul : (lane < 45 ? (pcf2 ? 1.0f : 0.0f) - 0.5f : 0.0f);
        row[lane] = v;
    }
    if (lane < 2) {
        float d = yold;
        if (c.update_yaw) {
            d = floor_mod((ytarget - yaw) + PI_F, 2.0f * PI_F) - PI_F;
            if (valid) ykeep[e] = d;
        }
        row[OFF_YAW + lane] = d;
    }
    if (lane < QA_TSC_NUM_OBSTACLE_CLASSES) row[OFF_TYPE + lane] = otype == lane ? 1.0f : 0.0f;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int p = lane + 64 * k;
        const int16_t hm = g1[k] < g2[k] ? (g1[k] < g3[k] ? g1[k] : g3[k]) : (g2[k] < g3[k] ? g2[k] : g3[k]);
        const float h = (float)hm * c.vertical_scale;
        if (p < QA_TSC_NUM_SCAN) {
            meas[p] = h;
            if (valid) io.measured_heights[e * QA_TSC_NUM_SCAN + p] = h;
            row[OFF_SCAN + p] = clampf((rz - 0.3f) - h, -1.0f, 1.0f);
        }
    }
    if (lane < 3) row[OFF_PRIV + 1 + lane] = plin * c.lin_vel;
    if (lane < 29) row[OFF_LATENT + lane] = lx_ - (lane < 5 ? 0.0f : 1.0f);
#pragma unroll
    for (int k = 0; k < 5; ++k)
        if (lane + 64 * k < 285) *reinterpret_cast<float2 *>(row + OFF_HIST + 2 * (lane + 64 * k)) = hv[k];
    if (lane < 11) row[OFF_CMD + lane] = cx_;
    __syncthreads();

    // ---- phase B: what needs the scan's centre height
    const float root_h = rz - meas[QA_TSC_NUM_SCAN / 2 + 1];
    if (lane == 0) row[OFF_PRIV] = c.root_height_obs ? root_h : 0.0f;
    if (lane < QA_TSC_NUM_OBS_DISC) {          // the imitation discriminator's view
        float v = (dx_ - dsub) * dmul;
        if (lane == 2) v = root_h;
        if (lane >= 33 && lane < 45) {
            // key body relative to the root, rotated by the inverse heading (compute_flat_key_pos)
            const float s = 2.0f * q3 * q3 - 1.0f;                      // heading = atan2 of the rotated x axis (quat_rotate)
            const float hx = s + q0 * q0 * 2.0f, hy = q2 * q3 * 2.0f + q1 * q0 * 2.0f;
            const float half = -atan2f(hy, hx) / 2.0f;
            float hz = sinf(half), hw = cosf(half);
            const float hn = fmaxf(sqrtf(hz * hz + hw * hw), 1e-9f);
            hz /= hn; hw /= hn;
            const float lx = b0 - r0, ly = b1 - r1, lz = b2 - rz;
            const float s2 = 2.0f * hw * hw - 1.0f;
            const float rx = lx * s2 + (0.0f - hz * ly) * hw * 2.0f, ry = ly * s2 + (hz * lx - 0.0f) * hw * 2.0f, rzz = lz * s2 + hz * (hz * lz) * 2.0f;
            v = (kax == 0 ? rx : (kax == 1 ? ry : rzz)) * c.key_pos;
        }
        if (lane >= 45) v = (dcf ? 1.0f : 0.0f) * c.foot_contact;
        if (valid) io.obs_disc_buf[e * QA_TSC_NUM_OBS_DISC + lane] = v;
    }
    __syncthreads();
    if (!valid) return;

    // ---- phase C: rows out
    const float cl = c.clip_observations;
    for (int i = lane; i < QA_TSC_NUM_OBS / 4; i += 64) {      // 16-byte stores: 800 floats per env, rows 16-byte aligned
        float4 v = reinterpret_cast<const float4 *>(row)[i];
        v.x = clampf(v.x, -cl, cl); v.y = clampf(v.y, -cl, cl); v.z = clampf(v.z, -cl, cl); v.w = clampf(v.w, -cl, cl);
        reinterpret_cast<float4 *>(io.obs_buf + e * QA_TSC_NUM_OBS)[i] = v;
    }
    for (int i = lane; i < QA_TSC_NUM_OBS_BBC; i += 64) {
        const int src = i < 57 ? i : (i < 90 ? OFF_PRIV + (i - 57) : (i < 660 ? OFF_HIST + (i - 90) : OFF_CMD + (i - 660)));
        io.obs_bbc_buf[e * QA_TSC_NUM_OBS_BBC + i] = clampf(row[src], -cl, cl);
    }
    const bool first = ep_len <= 1;
    for (int i2 = lane; i2 < 285; i2 += 64) {
        float2 v;
        {
            const int i = 2 * i2, slot = i / 57, j = i - slot * 57;
            v.x = clampf((first || slot == QA_TSC_HISTORY_LEN - 1) ? row[j] : row[OFF_HIST + i + 57], -cl, cl);
        }
        {
            const int i = 2 * i2 + 1, slot = i / 57, j = i - slot * 57;
            v.y = clampf((first || slot == QA_TSC_HISTORY_LEN - 1) ? row[j] : row[OFF_HIST + i + 57], -cl, cl);
        }
        reinterpret_cast<float2 *>(io.obs_history + e * 570)[i2] = v;
    }
}

}  // namespace


// ---------------------------------------------------------------------------------------------------------------------------------
// reset bookkeeping of a step in one launch: any_reset = OR of the flags (what decides the reference's extra simulate, legged_robot.py
// :382-384) and the `extras["episode"]` means of the resetting envs (:396-404: mean episode sum / episode length in seconds), kept from
// the last step on which anyone reset.  ONE workgroup, fixed summation order (thread t adds envs t, t + 1024, ... then a fixed LDS
// tree): bit-reproducible, no atomics -- it replaces torch's sum(dim=1) + an index_add_ inside the recorded rollout.
__global__ void __launch_bounds__(1024) qa_tsc_reset_stats_kernel(const uint8_t *__restrict__ flags, const float *__restrict__ sums, int64_t n, int num_terms,
                                                                  float inv_len_s, float *__restrict__ means, uint8_t *__restrict__ any_reset) {
    __shared__ float red[1024];
    __shared__ float cnt_s;
    const int t = threadIdx.x;
    float c = 0.f;
    for (int64_t e = t; e < n; e += 1024) c += flags[e] ? 1.f : 0.f;
    red[t] = c;
    __syncthreads();
    for (int w = 512; w > 0; w >>= 1) { if (t < w) red[t] += red[t + w]; __syncthreads(); }
    if (t == 0) { cnt_s = red[0]; any_reset[0] = red[0] > 0.f ? 1 : 0; }
    __syncthreads();
    const float cnt = cnt_s;
    if (cnt <= 0.f) return;                      // nobody reset: the means keep their values
    // r6: eight terms per pass -- their loads in flight together, one barrier ladder for all of them (was: per term a pass over the envs and a ten-step
    // ladder, eight dependent rounds, 11 us per env step of the task-level rollout).  Per term the same lanes add the same envs in the same order.
    constexpr int G = 8;
    __shared__ float redk[G][1024];
    for (int k0 = 0; k0 < num_terms; k0 += G) {
        float a[G];
#pragma unroll
        for (int j = 0; j < G; ++j) a[j] = 0.f;
        for (int64_t e = t; e < n; e += 1024) {
            const bool f = flags[e] != 0;
#pragma unroll
            for (int j = 0; j < G; ++j) if (k0 + j < num_terms) a[j] += f ? sums[(int64_t)(k0 + j) * n + e] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < G; ++j) redk[j][t] = a[j];
        __syncthreads();
        for (int w = 512; w > 0; w >>= 1) {
            if (t < w) {
#pragma unroll
                for (int j = 0; j < G; ++j) redk[j][t] += redk[j][t + w];
            }
            __syncthreads();
        }
        if (t < G && k0 + t < num_terms) means[k0 + t] = redk[t][0] / cnt * inv_len_s;
    }
}

extern "C" {

int qa_tsc_set_commands(const float *actions, const int64_t *episode_length, int64_t num_envs, int32_t num_d, int32_t num_c, int32_t dim_c,
                        int32_t interval, const int32_t *mocap_index, const float *vel_ranges, const float *jump_range,
                        const float *height_range, const float *noise, float *commands, float *latent_eps, float *latent_c,
                        float *next_commands, void *stream) {
    if (!actions || !episode_length || !mocap_index || !vel_ranges || !jump_range || !height_range || !commands || !latent_eps ||
        !latent_c || !next_commands || num_envs <= 0) { snprintf(g_terr, sizeof(g_terr), "qa_tsc_set_commands: null pointer or no envs"); return QA_E_ARG; }
    if (num_c != TSC_NUM_C || dim_c < 1 || dim_c > TSC_MAX_DIM_C || num_d < 1 || num_d > TSC_MAX_DIM_C || interval < 1) {
        snprintf(g_terr, sizeof(g_terr), "qa_tsc_set_commands: num_c must be %d, 1 <= num_d, dim_c <= %d, interval >= 1", TSC_NUM_C, TSC_MAX_DIM_C); return QA_E_ARG; }
    SetCmdArgs a{};
    a.actions = actions; a.episode_length = episode_length; a.noise = noise;
    a.commands = commands; a.latent_eps = latent_eps; a.latent_c = latent_c; a.next_commands = next_commands;
    a.n = num_envs; a.num_d = num_d; a.dim_c = dim_c; a.interval = interval;
    for (int i = 0; i < num_d; ++i) {
        if (mocap_index[i] < 0 || mocap_index[i] >= dim_c) { snprintf(g_terr, sizeof(g_terr), "qa_tsc_set_commands: mocap_index[%d] out of range", i); return QA_E_ARG; }
        a.mocap_index[i] = mocap_index[i];
    }
    for (int k = 0; k < 3; ++k)
        for (int g = 0; g < dim_c; ++g) { a.vel[k][g][0] = vel_ranges[(k * dim_c + g) * 2]; a.vel[k][g][1] = vel_ranges[(k * dim_c + g) * 2 + 1]; }
    a.jump[0] = jump_range[0]; a.jump[1] = jump_range[1]; a.height[0] = height_range[0]; a.height[1] = height_range[1];
    hipLaunchKernelGGL(qa_tsc_set_commands_kernel, dim3((unsigned)((num_envs + TSC_BLOCK - 1) / TSC_BLOCK)), dim3(TSC_BLOCK), 0, (hipStream_t)stream, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { snprintf(g_terr, sizeof(g_terr), "qa_tsc_set_commands: %s", hipGetErrorString(e)); return QA_E_DEVICE; }
    return QA_OK;
}

int qa_tsc_goal_step(const qa_tsc_goal_cfg *cfg, const qa_tsc_goal_io *io, void *stream) {
    if (!cfg || !io) { snprintf(g_terr, sizeof(g_terr), "qa_tsc_goal_step: null argument"); return QA_E_ARG; }
    const void *need[] = {io->root_states, io->contact_forces, io->rigid_body_states, io->env_goals, io->obstacle_types, io->x_edge_mask,
                          io->episode_length, io->cur_goal_idx, io->reach_goal_timer, io->last_contacts, io->cur_goals, io->next_goals,
                          io->episode_sums, io->base_lin_vel, io->base_ang_vel, io->projected_gravity, io->rpy, io->contact_filt,
                          io->target_pos_rel, io->next_target_pos_rel, io->target_yaw, io->next_target_yaw, io->reached_goal,
                          io->cur_obstacle_type, io->reset_buf, io->time_out_buf, io->reach_goal_cutoff, io->rew_buf};
    for (size_t i = 0; i < sizeof(need) / sizeof(need[0]); ++i)
        if (!need[i]) { snprintf(g_terr, sizeof(g_terr), "qa_tsc_goal_step: io pointer %zu is null", i); return QA_E_ARG; }
    const qa_tsc_goal_cfg &c = *cfg;
    bool ok = c.num_envs > 0 && c.num_bodies > 0 && c.num_bodies <= 32 && c.num_goal_slots > c.last_goal_repeat && c.last_goal_repeat >= 0 && c.goals_per_obstacle > 0 &&
              c.num_obstacles > 0 && c.mask_rows > 0 && c.mask_cols > 0 && c.horizontal_scale > 0.0f &&
              c.num_termination_bodies >= 0 && c.num_termination_bodies <= QA_TSC_MAX_BODY_IDS &&
              c.num_penalised_bodies >= 0 && c.num_penalised_bodies <= QA_TSC_MAX_BODY_IDS &&
              (!io->action_hl_history || (c.history_len >= 3 && c.history_width >= 1));
    for (int k = 0; ok && k < c.num_termination_bodies; ++k) ok = c.termination_bodies[k] >= 0 && c.termination_bodies[k] < c.num_bodies;
    for (int k = 0; ok && k < c.num_penalised_bodies; ++k) ok = c.penalised_bodies[k] >= 0 && c.penalised_bodies[k] < c.num_bodies;
    for (int k = 0; ok && k < 4; ++k) ok = c.feet_bodies[k] >= 0 && c.feet_bodies[k] < c.num_bodies;
    if (!ok) { snprintf(g_terr, sizeof(g_terr), "qa_tsc_goal_step: inconsistent configuration"); return QA_E_ARG; }
    uint32_t term_mask = 0, pen_mask = 0;
    for (int k = 0; k < c.num_termination_bodies; ++k) term_mask |= 1u << c.termination_bodies[k];
    for (int k = 0; k < c.num_penalised_bodies; ++k) {
        if (pen_mask & (1u << c.penalised_bodies[k])) { snprintf(g_terr, sizeof(g_terr), "qa_tsc_goal_step: body %d listed twice as penalised", c.penalised_bodies[k]); return QA_E_ARG; }
        pen_mask |= 1u << c.penalised_bodies[k];
    }
    hipLaunchKernelGGL(qa_tsc_goal_step_kernel, dim3((unsigned)((c.num_envs + TSC_BLOCK - 1) / TSC_BLOCK)), dim3(TSC_BLOCK), 0, (hipStream_t)stream, c, *io,
                       term_mask, pen_mask);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { snprintf(g_terr, sizeof(g_terr), "qa_tsc_goal_step: %s", hipGetErrorString(e)); return QA_E_DEVICE; }
    return QA_OK;
}

int qa_tsc_observations(const qa_tsc_obs_cfg *cfg, const qa_tsc_obs_io *io, void *stream) {
    if (!cfg || !io) { snprintf(g_terr, sizeof(g_terr), "qa_tsc_observations: null argument"); return QA_E_ARG; }
    const void *need[] = {io->root_states, io->rpy, io->base_lin_vel, io->base_ang_vel, io->contact_filt, io->dof_pos, io->dof_vel, io->last_action,
                          io->rigid_body_states, io->mass_params, io->friction, io->motor_strength, io->cur_obstacle_type, io->target_yaw,
                          io->next_target_yaw, io->height_samples, io->height_points, io->commands, io->latent_eps, io->latent_c,
                          io->episode_length, io->delta_yaw, io->delta_next_yaw, io->obs_history, io->measured_heights, io->obs_buf,
                          io->obs_bbc_buf, io->obs_disc_buf};
    for (size_t i = 0; i < sizeof(need) / sizeof(need[0]); ++i)
        if (!need[i]) { snprintf(g_terr, sizeof(g_terr), "qa_tsc_observations: io pointer %zu is null", i); return QA_E_ARG; }
    const qa_tsc_obs_cfg &c = *cfg;
    bool ok = c.num_envs > 0 && c.num_bodies > 0 && c.map_rows >= 2 && c.map_cols >= 2 && c.horizontal_scale > 0.0f && c.action_stride >= 12 && c.point_stride >= 2 && c.points_env_stride >= 0 &&
              c.clip_observations > 0.0f;
    for (int k = 0; ok && k < 4; ++k) ok = c.key_bodies[k] >= 0 && c.key_bodies[k] < c.num_bodies;
    if (!ok) { snprintf(g_terr, sizeof(g_terr), "qa_tsc_observations: inconsistent configuration"); return QA_E_ARG; }
    hipLaunchKernelGGL(qa_tsc_observations_kernel, dim3((unsigned)((c.num_envs + OBS_WAVES - 1) / OBS_WAVES)), dim3(64 * OBS_WAVES), 0,
                       (hipStream_t)stream, c, *io);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { snprintf(g_terr, sizeof(g_terr), "qa_tsc_observations: %s", hipGetErrorString(e)); return QA_E_DEVICE; }
    return QA_OK;
}


int qa_tsc_reset_stats(const uint8_t *reset_flags, const float *episode_sums, int64_t num_envs, int32_t num_terms, float max_episode_length_s,
                       float *episode_means, uint8_t *any_reset, void *stream) {
    if (!reset_flags || !episode_sums || !episode_means || !any_reset || num_envs <= 0 || num_terms <= 0 || !(max_episode_length_s > 0.f)) {
        snprintf(g_terr, sizeof(g_terr), "qa_tsc_reset_stats: bad argument"); return QA_E_ARG; }
    hipLaunchKernelGGL(qa_tsc_reset_stats_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, reset_flags, episode_sums, num_envs, (int)num_terms,
                       1.0f / max_episode_length_s, episode_means, any_reset);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { snprintf(g_terr, sizeof(g_terr), "qa_tsc_reset_stats: %s", hipGetErrorString(e)); return QA_E_DEVICE; }
    return QA_OK;
}
}  // extern "C"
