/* qa_oracle.c -- CPU oracle for the quadrupedal-agility hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library; the product (quadrupedal_agility_amd/) never links, imports or calls it.
 *
 * What it restates, and what pins it:
 *  (1) env-side math of bbc/legged_gym/envs/base/legged_robot.py -- PD torques (:547-579),
 *      post_physics_step (:124-166), check_termination (:168-176), the 14 active reward
 *      terms (:242-259, :1231-1374), command/latent resampling (:474-545), default-pose
 *      reset (:581-596, :614-634), compute_observations (:261-331) and compute_flat_key_pos
 *      (:1377-1396) -- in fp32, following the torch op order.  PINNED by golden vectors
 *      generated from the reference Python itself (tests/golden/, tools/gen_golden.py).
 *  (2) GAE of bbc/rsl_rl/storage/rollout_storage.py:97-111.  PINNED by golden vectors.
 *  (2b) plain-C twins of the fused learner kernels (qo_ppo_loss, qo_elu_backward_bias, qo_normalizer_*, qo_clip_adam_step,
 *      qo_rollout_act / _post, qo_disc_loss, qo_disc_prepare, qo_mlp_*): each restates the PyTorch expression the reference's
 *      learner evaluates (gail.py, utils.py, discriminator.py; cited at the functions).  PINNED by tests/test_fused_learner.py
 *      against those eager expressions, which tests/test_golden_learner.py in turn pins to the reference's own numbers.
 *  (3) the physics the reference delegates to Isaac Gym / PhysX (legged_robot.py:103-106,
 *      129-131).  That binary is closed and absent, so this part follows the build's OWN
 *      stated model: floating base + 12 revolute DoF rigid-body dynamics
 *      (composite-rigid-body mass matrix + recursive Newton-Euler bias), semi-implicit Euler
 *      at dt = 5 ms, velocity-level contact with projected Gauss-Seidel.  PARITY UNPINNED
 *      against the reference; pinned instead by analytic known-answer tests
 *      (tests/test_oracle_physics.py) and by cross-checking the HIP kernels against it.
 *
 * The physics here is written in double precision with a DENSE 18x18 mass matrix and a
 * generic Cholesky -- on purpose a different algebraic route from the HIP kernel (which
 * eliminates the legs into a 6x6 base Schur complement, one leg per lane), so agreement
 * between the two checks the optimised algebra, not a shared implementation.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/qa_sim.h"
#include "qa_go2_model.h"

/* ------------------------------------------------------------------ layout */
typedef struct {
    int64_t off[QA_T_COUNT];
    int64_t shape[QA_T_COUNT][3];
    int32_t ndim[QA_T_COUNT];
    int32_t dtype[QA_T_COUNT];
    int64_t total;
} Layout;

static int64_t dtype_size(int d) { return d == QA_F32 ? 4 : (d == QA_I64 || d == QA_F64) ? 8 : d == QA_I32 ? 4 : d == QA_I16 ? 2 : 1; }

static void set_t(Layout *L, int t, int dt, int nd, int64_t a, int64_t b, int64_t c) {
    L->dtype[t] = dt; L->ndim[t] = nd; L->shape[t][0] = a; L->shape[t][1] = b; L->shape[t][2] = c;
}

static void make_layout(const qa_config *cfg, Layout *L) {
    int64_t N = cfg->num_envs;
    int64_t F = cfg->num_mocap_frames > 0 ? cfg->num_mocap_frames : 1;
    memset(L, 0, sizeof(*L));
    set_t(L, QA_T_ROOT_STATES, QA_F32, 2, N, 13, 1);
    set_t(L, QA_T_DOF_STATE, QA_F32, 3, N, 12, 2);
    set_t(L, QA_T_CONTACT_FORCES, QA_F32, 3, N, 19, 3);
    set_t(L, QA_T_RIGID_BODY_POS, QA_F32, 3, N, 19, 3);
    set_t(L, QA_T_TORQUES, QA_F32, 2, N, 12, 1);
    set_t(L, QA_T_TORQUES_ORG, QA_F32, 2, N, 12, 1);
    set_t(L, QA_T_ACTIONS, QA_F32, 2, N, 12, 1);
    set_t(L, QA_T_LAST_ACTIONS, QA_F32, 2, N, 12, 1);
    set_t(L, QA_T_LAST_DOF_VEL, QA_F32, 2, N, 12, 1);
    set_t(L, QA_T_LAST_TORQUES_ORG, QA_F32, 2, N, 12, 1);
    set_t(L, QA_T_LAST_ROOT_VEL, QA_F32, 2, N, 6, 1);
    set_t(L, QA_T_ACTION_HISTORY, QA_F32, 3, N, QA_ACTION_BUF_LEN, 12);
    set_t(L, QA_T_OBS, QA_F32, 2, N, QA_NUM_OBS, 1);
    set_t(L, QA_T_OBS_DISC, QA_F32, 2, N, QA_NUM_OBS_DISC, 1);
    set_t(L, QA_T_OBS_DISC_TERM, QA_F32, 2, N, QA_NUM_OBS_DISC, 1);
    set_t(L, QA_T_COMMANDS, QA_F32, 2, N, 5, 1);
    set_t(L, QA_T_LATENT_EPS, QA_F32, 2, N, 1, 1);
    set_t(L, QA_T_LATENT_C, QA_F32, 2, N, QA_NUM_GAITS, 1);
    set_t(L, QA_T_REW, QA_F32, 1, N, 1, 1);
    set_t(L, QA_T_RESET, QA_I64, 1, N, 1, 1);
    set_t(L, QA_T_TIME_OUT, QA_U8, 1, N, 1, 1);
    set_t(L, QA_T_EPISODE_LENGTH, QA_I64, 1, N, 1, 1);
    set_t(L, QA_T_EPISODE_SUMS, QA_F32, 2, QA_NUM_REWARDS, N, 1);
    set_t(L, QA_T_EPISODE_STATS, QA_F32, 2, 2, 16, 1);
    set_t(L, QA_T_LAST_CONTACTS, QA_U8, 2, N, 4, 1);
    set_t(L, QA_T_CONTACT_FILT, QA_U8, 2, N, 4, 1);
    set_t(L, QA_T_FEET_FORCE, QA_F32, 2, N, 4, 1);
    set_t(L, QA_T_BASE_LIN_VEL, QA_F32, 2, N, 3, 1);
    set_t(L, QA_T_BASE_ANG_VEL, QA_F32, 2, N, 3, 1);
    set_t(L, QA_T_PROJECTED_GRAVITY, QA_F32, 2, N, 3, 1);
    set_t(L, QA_T_RPY, QA_F32, 2, N, 3, 1);
    set_t(L, QA_T_MOTOR_STRENGTH, QA_F32, 3, 2, N, 12);
    set_t(L, QA_T_MASS_PARAMS, QA_F32, 2, N, 4, 1);
    set_t(L, QA_T_FRICTION, QA_F32, 1, N, 1, 1);
    set_t(L, QA_T_ENV_ORIGINS, QA_F32, 2, N, 3, 1);
    set_t(L, QA_T_BASE_INERTIA, QA_F32, 2, N, 10, 1);
    set_t(L, QA_T_PRIOR_PARAMETERS, QA_F32, 1, QA_NUM_GAITS, 1, 1);
    set_t(L, QA_T_MOCAP_FRAMES, QA_F32, 2, F, QA_MOCAP_FRAME, 1);
    { int64_t hr = cfg->terrain_type == 1 ? cfg->hf_rows : 1, hc = cfg->terrain_type == 1 ? cfg->hf_cols : 1;
      set_t(L, QA_T_HEIGHT_SAMPLES, QA_I16, 2, hr, hc, 1); }
    set_t(L, QA_T_SCAN_HEIGHT, QA_F32, 1, N, 1, 1);
    set_t(L, QA_T_FOOT_IMPULSE, QA_F32, 3, N, 4, 3);
    set_t(L, QA_T_MOCAP_CLIPS, QA_F64, 2, QA_MAX_MOCAP_CLIPS, QA_MOCAP_CLIP, 1);
    set_t(L, QA_T_RIGID_BODY_STATE, QA_F32, 3, cfg->export_body_state ? N : 1, QA_NUM_BODIES_ABI, 13);
    set_t(L, QA_T_STEP_TICKET, QA_I32, 1, 4, 1, 1);
    { int ce = cfg->terrain_type == 1 && cfg->hf_ceiling;
      set_t(L, QA_T_CEILING_SAMPLES, QA_I16, 2, ce ? cfg->hf_rows : 1, ce ? cfg->hf_cols : 1, 1); }
    { int ao = cfg->terrain_type == 1 && cfg->articulated_obstacles;
      set_t(L, QA_T_OBST_DESC, QA_F32, 3, ao ? N : 1, QA_OBST_PER_ENV, QA_OBST_DESC);
      set_t(L, QA_T_OBST_STATE, QA_F32, 3, ao ? N : 1, QA_OBST_PER_ENV, QA_OBST_STATE); }
    int64_t off = 0;
    for (int t = 0; t < QA_T_COUNT; ++t) {
        off = (off + 255) & ~(int64_t)255;
        L->off[t] = off;
        off += L->shape[t][0] * L->shape[t][1] * L->shape[t][2] * dtype_size(L->dtype[t]);
    }
    L->total = (off + 255) & ~(int64_t)255;
}

struct qo_sim {
    qa_config cfg;
    Layout L;
    char *arena;
    int32_t mocap_first[QA_NUM_GAITS + 1];
};
typedef struct qo_sim qo_sim;

#define TP(sim, t, type) ((type *)((sim)->arena + (sim)->L.off[t]))

/* ------------------------------------------------------------------ Philox4x32-10 */
static void philox(uint64_t seed, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t out[4]) {
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
/* stream ids (shared spec with the HIP kernels) */
enum { RS_INIT_BUCKET = 1, RS_INIT_FRICTION = 2, RS_INIT_MASS = 3, RS_INIT_MOTOR = 4,
       RS_NOISE = 8, RS_CMD = 9, RS_CMD_RESET = 10, RS_PUSH = 11, RS_RESET = 12 };

static void rng4(const qo_sim *s, uint32_t env, int64_t step, int stream, int block, float u[4]) {
    uint32_t o[4];
    if (stream != RS_INIT_BUCKET) env += (uint32_t)s->cfg.env_id_offset;      /* key = GLOBAL env id (friction buckets are shared, not per env) */
    philox(s->cfg.seed, env, (uint32_t)step, (uint32_t)(stream * 256 + block), (uint32_t)((uint64_t)step >> 32), o);
    for (int i = 0; i < 4; ++i) u[i] = (float)(o[i] >> 8) * (1.0f / 16777216.0f);
}

/* ------------------------------------------------------------------ fp32 helpers (torch op order) */
/* isaacgym.torch_utils quat_rotate / quat_rotate_inverse (xyzw), as used at legged_robot.py:138-140 */
static void quat_rotate_f(const float q[4], const float v[3], float sign, float out[3]) {
    float qw = q[3];
    float s = 2.0f * qw * qw - 1.0f;
    float a[3] = {v[0] * s, v[1] * s, v[2] * s};
    float cx = q[1] * v[2] - q[2] * v[1], cy = q[2] * v[0] - q[0] * v[2], cz = q[0] * v[1] - q[1] * v[0];
    float b[3] = {cx * qw * 2.0f, cy * qw * 2.0f, cz * qw * 2.0f};
    float d = q[0] * v[0] + q[1] * v[1] + q[2] * v[2];
    float c[3] = {q[0] * d * 2.0f, q[1] * d * 2.0f, q[2] * d * 2.0f};
    for (int i = 0; i < 3; ++i) out[i] = sign > 0 ? a[i] + b[i] + c[i] : a[i] - b[i] + c[i];
}
static float clipf(float x, float lo, float hi) { return x < lo ? lo : (x > hi ? hi : x); }

/* ------------------------------------------------------------------ double helpers for physics */
typedef double v3[3];
typedef double m3[3][3];
static void cross(const v3 a, const v3 b, v3 o) {
    double x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
    o[0] = x; o[1] = y; o[2] = z;
}
static double dot3(const v3 a, const v3 b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static void mv(const m3 A, const v3 x, v3 o) {
    v3 t; for (int i = 0; i < 3; ++i) t[i] = A[i][0] * x[0] + A[i][1] * x[1] + A[i][2] * x[2];
    o[0] = t[0]; o[1] = t[1]; o[2] = t[2];
}
static void mtv(const m3 A, const v3 x, v3 o) {
    v3 t; for (int i = 0; i < 3; ++i) t[i] = A[0][i] * x[0] + A[1][i] * x[1] + A[2][i] * x[2];
    o[0] = t[0]; o[1] = t[1]; o[2] = t[2];
}
static void mm(const m3 A, const m3 B, m3 C) {
    m3 T; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) T[i][j] = A[i][0] * B[0][j] + A[i][1] * B[1][j] + A[i][2] * B[2][j];
    memcpy(C, T, sizeof(m3));
}
static void quat_to_mat(const double q[4], m3 R) { /* xyzw, body->world */
    double x = q[0], y = q[1], z = q[2], w = q[3];
    R[0][0] = 1 - 2 * (y * y + z * z); R[0][1] = 2 * (x * y - z * w); R[0][2] = 2 * (x * z + y * w);
    R[1][0] = 2 * (x * y + z * w); R[1][1] = 1 - 2 * (x * x + z * z); R[1][2] = 2 * (y * z - x * w);
    R[2][0] = 2 * (x * z - y * w); R[2][1] = 2 * (y * z + x * w); R[2][2] = 1 - 2 * (x * x + y * y);
}

/* rigid-body inertia about the base origin, base axes: mass, first moment h = m c, I_O (sym 3x3) */
typedef struct { double m; v3 h; m3 I; } RB;
static void rb_make(double m, const v3 c, const m3 Ic, RB *o) {
    o->m = m; for (int i = 0; i < 3; ++i) o->h[i] = m * c[i];
    double cc = dot3(c, c);
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) o->I[i][j] = Ic[i][j] + m * ((i == j ? cc : 0.0) - c[i] * c[j]);
}
static void rb_add(RB *a, const RB *b) {
    a->m += b->m; for (int i = 0; i < 3; ++i) { a->h[i] += b->h[i]; for (int j = 0; j < 3; ++j) a->I[i][j] += b->I[i][j]; }
}
/* F = I V with V = (w; v), F = (n; f):  n = I_O w + h x v,  f = m v - h x w */
static void rb_apply(const RB *I, const double V[6], double F[6]) {
    v3 t, u;
    mv(I->I, V, t); cross(I->h, V + 3, u);
    for (int i = 0; i < 3; ++i) F[i] = t[i] + u[i];
    cross(I->h, V, u);
    for (int i = 0; i < 3; ++i) F[3 + i] = I->m * V[3 + i] - u[i];
}
/* spatial cross products: motion x motion and motion x* force */
static void crm(const double V[6], const double M[6], double o[6]) {
    v3 a, b, c; cross(V, M, a); cross(V, M + 3, b); cross(V + 3, M, c);
    for (int i = 0; i < 3; ++i) { o[i] = a[i]; o[3 + i] = b[i] + c[i]; }
}
static void crf(const double V[6], const double F[6], double o[6]) {
    v3 a, b, c; cross(V, F, a); cross(V + 3, F + 3, b); cross(V, F + 3, c);
    for (int i = 0; i < 3; ++i) { o[i] = a[i] + b[i]; o[3 + i] = c[i]; }
}
static double dot6(const double a[6], const double b[6]) { double s = 0; for (int i = 0; i < 6; ++i) s += a[i] * b[i]; return s; }

/* ------------------------------------------------------------------ kinematics of one env (base frame) */
typedef struct {
    m3 Rl[4][3];        /* link rotations, base <- link */
    v3 o[4][3];         /* joint origins (= link frame origins) */
    v3 a[4][3];         /* joint axes */
    v3 foot[4];         /* foot body origin */
    double S[4][3][6];  /* joint motion vectors (a; o x a) */
    RB link[4][3];      /* single-link inertias */
} Kin;

static void leg_kin(int l, const double q[3], Kin *K) {
    double c1 = cos(q[0]), s1 = sin(q[0]);
    double c2 = cos(q[1]), s2 = sin(q[1]);
    double c23 = cos(q[1] + q[2]), s23 = sin(q[1] + q[2]);
    m3 R1 = {{1, 0, 0}, {0, c1, -s1}, {0, s1, c1}};
    m3 Ry2 = {{c2, 0, s2}, {0, 1, 0}, {-s2, 0, c2}};
    m3 Ry23 = {{c23, 0, s23}, {0, 1, 0}, {-s23, 0, c23}};
    memcpy(K->Rl[l][0], R1, sizeof(m3));
    mm(R1, Ry2, K->Rl[l][1]);
    mm(R1, Ry23, K->Rl[l][2]);
    v3 t, off;
    for (int i = 0; i < 3; ++i) K->o[l][0][i] = QA_HIP_ORG[l][i];
    for (int i = 0; i < 3; ++i) off[i] = QA_THIGH_ORG[l][i];
    mv(K->Rl[l][0], off, t); for (int i = 0; i < 3; ++i) K->o[l][1][i] = K->o[l][0][i] + t[i];
    for (int i = 0; i < 3; ++i) off[i] = QA_CALF_ORG[l][i];
    mv(K->Rl[l][1], off, t); for (int i = 0; i < 3; ++i) K->o[l][2][i] = K->o[l][1][i] + t[i];
    for (int i = 0; i < 3; ++i) off[i] = QA_FOOT_ORG[l][i];
    mv(K->Rl[l][2], off, t); for (int i = 0; i < 3; ++i) K->foot[l][i] = K->o[l][2][i] + t[i];
    K->a[l][0][0] = 1; K->a[l][0][1] = 0; K->a[l][0][2] = 0;
    for (int k = 1; k < 3; ++k) { K->a[l][k][0] = 0; K->a[l][k][1] = c1; K->a[l][k][2] = s1; }
    for (int k = 0; k < 3; ++k) {
        v3 b; cross(K->o[l][k], K->a[l][k], b);
        for (int i = 0; i < 3; ++i) { K->S[l][k][i] = K->a[l][k][i]; K->S[l][k][3 + i] = b[i]; }
        /* link inertia into base frame */
        const float *I6 = QA_LINK_I[l][k];
        m3 Il = {{I6[0], I6[3], I6[4]}, {I6[3], I6[1], I6[5]}, {I6[4], I6[5], I6[2]}}, Rt, Ib;
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) Rt[i][j] = K->Rl[l][k][j][i];
        mm(K->Rl[l][k], Il, Ib); mm(Ib, Rt, Ib);
        v3 cl = {QA_LINK_COM[l][k][0], QA_LINK_COM[l][k][1], QA_LINK_COM[l][k][2]}, cb;
        mv(K->Rl[l][k], cl, cb); for (int i = 0; i < 3; ++i) cb[i] += K->o[l][k][i];
        rb_make(QA_LINK_MASS[l][k], cb, Ib, &K->link[l][k]);
    }
}

static void base_rb(const float bi[10], RB *o) {
    o->m = bi[0]; for (int i = 0; i < 3; ++i) o->h[i] = bi[1 + i];
    o->I[0][0] = bi[4]; o->I[1][1] = bi[5]; o->I[2][2] = bi[6];
    o->I[0][1] = o->I[1][0] = bi[7]; o->I[0][2] = o->I[2][0] = bi[8]; o->I[1][2] = o->I[2][1] = bi[9];
}

/* dense mass matrix (18x18; index 0..2 base angular, 3..5 base linear, 6+3l+k joints) and bias */
static void dynamics_terms(const Kin *K, const RB *base, const double ub[6], const double qd[12],
                           const v3 gB, double M[18][18], double h[18]) {
    memset(M, 0, sizeof(double) * 18 * 18);
    RB tot = *base;
    for (int l = 0; l < 4; ++l) {
        RB Ic[3];
        Ic[2] = K->link[l][2];
        Ic[1] = K->link[l][1]; rb_add(&Ic[1], &Ic[2]);
        Ic[0] = K->link[l][0]; rb_add(&Ic[0], &Ic[1]);
        rb_add(&tot, &Ic[0]);
        for (int j = 0; j < 3; ++j) {
            double F[6]; rb_apply(&Ic[j], K->S[l][j], F);
            int cj = 6 + 3 * l + j;
            for (int i = 0; i < 6; ++i) { M[i][cj] = F[i]; M[cj][i] = F[i]; }
            for (int i = 0; i <= j; ++i) { double v = dot6(K->S[l][i], F); int ci = 6 + 3 * l + i; M[ci][cj] = v; M[cj][ci] = v; }
        }
    }
    /* base block = spatial inertia of the whole robot */
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) M[i][j] = tot.I[i][j];
    double hx[3][3] = {{0, -tot.h[2], tot.h[1]}, {tot.h[2], 0, -tot.h[0]}, {-tot.h[1], tot.h[0], 0}};
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { M[i][3 + j] = hx[i][j]; M[3 + j][i] = hx[i][j]; }
    for (int i = 0; i < 3; ++i) M[3 + i][3 + i] = tot.m;

    /* bias: recursive Newton-Euler with zero joint acceleration, base acceleration = -gravity */
    double A0[6] = {0, 0, 0, -gB[0], -gB[1], -gB[2]}, t6[6], f0[6];
    rb_apply(base, A0, f0);
    rb_apply(base, ub, t6); { double c[6]; crf(ub, t6, c); for (int i = 0; i < 6; ++i) f0[i] += c[i]; }
    for (int l = 0; l < 4; ++l) {
        double V[6], A[6], f[3][6];
        memcpy(V, ub, sizeof(V)); memcpy(A, A0, sizeof(A));
        for (int k = 0; k < 3; ++k) {
            double Sq[6], c[6];
            for (int i = 0; i < 6; ++i) Sq[i] = K->S[l][k][i] * qd[3 * l + k];
            for (int i = 0; i < 6; ++i) V[i] += Sq[i];
            crm(V, Sq, c); for (int i = 0; i < 6; ++i) A[i] += c[i];
            double IA[6], IV[6], cf[6];
            rb_apply(&K->link[l][k], A, IA); rb_apply(&K->link[l][k], V, IV); crf(V, IV, cf);
            for (int i = 0; i < 6; ++i) f[k][i] = IA[i] + cf[i];
        }
        for (int i = 0; i < 6; ++i) { f[1][i] += f[2][i]; f[0][i] += f[1][i]; f0[i] += f[0][i]; }
        for (int k = 0; k < 3; ++k) h[6 + 3 * l + k] = dot6(K->S[l][k], f[k]);
    }
    for (int i = 0; i < 6; ++i) h[i] = f0[i];
}

/* in-place Cholesky (lower) and solve for n <= 18 */
static int chol(double A[18][18], int n) {
    for (int j = 0; j < n; ++j) {
        double d = A[j][j];
        for (int k = 0; k < j; ++k) d -= A[j][k] * A[j][k];
        if (!(d > 0)) return -1;
        d = sqrt(d); A[j][j] = d;
        for (int i = j + 1; i < n; ++i) {
            double s = A[i][j];
            for (int k = 0; k < j; ++k) s -= A[i][k] * A[j][k];
            A[i][j] = s / d;
        }
    }
    return 0;
}
static void chol_solve(double Lc[18][18], int n, const double b[18], double x[18]) {
    double y[18];
    for (int i = 0; i < n; ++i) { double s = b[i]; for (int k = 0; k < i; ++k) s -= Lc[i][k] * y[k]; y[i] = s / Lc[i][i]; }
    for (int i = n - 1; i >= 0; --i) { double s = y[i]; for (int k = i + 1; k < n; ++k) s -= Lc[k][i] * x[k]; x[i] = s / Lc[i][i]; }
}

/* ------------------------------------------------------------------ one physics substep */
typedef struct { double J[18], W[18], dinv, bias, lam, lo_mul, hi_mul; int kind; /*0 normal,1 tangent,2 limit*/ int parent; } Row;

#define LIMIT_MARGIN 0.2   /* rad: beyond 30.1 rad/s x 5 ms = 0.15 rad a stop cannot bind within one substep */
#define LIMIT_DEPEN 1.0    /* rad/s cap on limit-violation recovery speed */
#define CFM 1e-6

/* Terrain under world point (x, y): height and unit normal.  Height field = two triangles per cell split along the
 * (i,j)-(i+1,j+1) diagonal, as isaacgym's heightfield->trimesh conversion the reference uses (terrain.py:41-45). */
static void ground_query(const qo_sim *s, double x, double y, double *h, double n[3]) {
    const qa_config *c = &s->cfg;
    if (c->terrain_type != 1) { *h = 0.0; n[0] = 0; n[1] = 0; n[2] = 1; return; }
    const int16_t *hs = TP(s, QA_T_HEIGHT_SAMPLES, int16_t);
    double fx = (x + c->hf_border) / c->hf_hscale, fy = (y + c->hf_border) / c->hf_hscale;
    int ix = (int)floor(fx), iy = (int)floor(fy);
    if (ix < 0) ix = 0; if (ix > c->hf_rows - 2) ix = c->hf_rows - 2;
    if (iy < 0) iy = 0; if (iy > c->hf_cols - 2) iy = c->hf_cols - 2;
    double u = fx - ix, v = fy - iy;
    if (u < 0) u = 0; if (u > 1) u = 1; if (v < 0) v = 0; if (v > 1) v = 1;
    double vs = c->hf_vscale;
    double h00 = vs * hs[(int64_t)ix * c->hf_cols + iy], h10 = vs * hs[(int64_t)(ix + 1) * c->hf_cols + iy];
    double h01 = vs * hs[(int64_t)ix * c->hf_cols + iy + 1], h11 = vs * hs[(int64_t)(ix + 1) * c->hf_cols + iy + 1];
    double gx, gy;
    if (u >= v) { gx = h10 - h00; gy = h11 - h10; *h = h00 + u * gx + v * gy; }
    else { gy = h01 - h00; gx = h11 - h01; *h = h00 + v * gy + u * gx; }
    gx /= c->hf_hscale; gy /= c->hf_hscale;
    double inv = 1.0 / sqrt(gx * gx + gy * gy + 1.0);
    n[0] = -gx * inv; n[1] = -gy * inv; n[2] = inv;
}
/* Gap and contact normal of a sphere (radius rad, centre pw, world frame) against the terrain: the distance to the plane of the
 * floor triangle under it, or -- with cfg.hf_ceiling -- to the plane of the CEILING triangle above it if that is nearer (tunnel roof,
 * upper arc of the tyre; QA_T_CEILING_SAMPLES, same triangulation, a triangle with a QA_NO_CEILING corner does not exist).  The
 * ceiling's normal points down, away from the obstacle. */
static double terrain_contact(const qo_sim *s, const double pw[3], double rad, double n[3]) {
    const qa_config *c = &s->cfg;
    double gh; ground_query(s, pw[0], pw[1], &gh, n);
    double gap = (pw[2] - gh) * n[2] - rad;
    if (c->terrain_type != 1 || !c->hf_ceiling) return gap;
    const int16_t *cs = TP(s, QA_T_CEILING_SAMPLES, int16_t);
    double fx = (pw[0] + c->hf_border) / c->hf_hscale, fy = (pw[1] + c->hf_border) / c->hf_hscale;
    int ix = (int)floor(fx), iy = (int)floor(fy);
    if (ix < 0) ix = 0; if (ix > c->hf_rows - 2) ix = c->hf_rows - 2;
    if (iy < 0) iy = 0; if (iy > c->hf_cols - 2) iy = c->hf_cols - 2;
    double u = fx - ix, v = fy - iy;
    if (u < 0) u = 0; if (u > 1) u = 1; if (v < 0) v = 0; if (v > 1) v = 1;
    int16_t s00 = cs[(int64_t)ix * c->hf_cols + iy], s10 = cs[(int64_t)(ix + 1) * c->hf_cols + iy];
    int16_t s01 = cs[(int64_t)ix * c->hf_cols + iy + 1], s11 = cs[(int64_t)(ix + 1) * c->hf_cols + iy + 1];
    double vs = c->hf_vscale, h00 = vs * s00, h10 = vs * s10, h01 = vs * s01, h11 = vs * s11, gx, gy, ch;
    if (u >= v) { if (s00 == QA_NO_CEILING || s10 == QA_NO_CEILING || s11 == QA_NO_CEILING) return gap; gx = h10 - h00; gy = h11 - h10; ch = h00 + u * gx + v * gy; }
    else { if (s00 == QA_NO_CEILING || s01 == QA_NO_CEILING || s11 == QA_NO_CEILING) return gap; gy = h01 - h00; gx = h11 - h01; ch = h00 + v * gy + u * gx; }
    gx /= c->hf_hscale; gy /= c->hf_hscale;
    double inv = 1.0 / sqrt(gx * gx + gy * gy + 1.0);
    double cgap = (ch - pw[2]) * inv - rad;
    /* an overhang is a thin shell: its underside acts on points below it or at most QA_CEILING_SHELL above it (a point well above
     * is ON the obstacle, which this field does not describe), and only where it lies above the floor map */
    if (!(ch > gh) || (ch - pw[2]) * inv < -QA_CEILING_SHELL) return gap;
    if (cgap < gap) { gap = cgap; n[0] = gx * inv; n[1] = gy * inv; n[2] = -inv; }
    return gap;
}
/* Contact of a sphere with the terrain AND the env's articulated course obstacles (cfg.articulated_obstacles; DESIGN.md 3.3,
 * tsc/legged_gym/envs/base/legged_robot.py:792-794, 812-823, 1411-1427): inside the see-saw's footprint the height map's static tent is
 * replaced by {flat ground, the plank's top plane through the pivot at tilt q}; inside the bar's / tyre's footprint the map (and the
 * tyre's ceiling arc) is shifted vertically by the joint offset q.  *vs = velocity of the contacted surface along n (the normal row's
 * bias is gap/dt - vs), *ca = d vs / d q_dot (the contact's generalised force on the joint is -f_n ca), *ob = the obstacle slot or -1. */
static double contact_query(const qo_sim *s, int e, const double pw[3], double rad, double n[3], double *vs, double *ca, int *ob) {
    const qa_config *c = &s->cfg;
    *vs = 0; *ca = 0; *ob = -1;
    if (c->terrain_type != 1 || !c->articulated_obstacles) return terrain_contact(s, pw, rad, n);
    const float *desc = TP(s, QA_T_OBST_DESC, float) + (int64_t)e * QA_OBST_PER_ENV * QA_OBST_DESC;
    const float *stt = TP(s, QA_T_OBST_STATE, float) + (int64_t)e * QA_OBST_PER_ENV * QA_OBST_STATE;
    int mode = 0, slot = -1; double xl = 0, cpsi = 1, spsi = 0, oq = 0, oqd = 0, oh0 = 0;
    for (int k = 0; k < QA_OBST_PER_ENV; ++k) {
        const float *d = desc + QA_OBST_DESC * k;
        if (d[7] == 0.0f) continue;
        double dx = pw[0] - d[0], dy = pw[1] - d[1];
        double lx = d[2] * dx + d[3] * dy, ly = d[2] * dy - d[3] * dx;
        if (fabs(lx) <= d[4] && fabs(ly) <= d[5]) {
            mode = d[7] == (float)QA_OBST_SEESAW ? 1 : 2; slot = k; xl = lx; cpsi = d[2]; spsi = d[3]; oh0 = d[6];
            oq = stt[QA_OBST_STATE * k]; oqd = stt[QA_OBST_STATE * k + 1];
        }
    }
    if (mode == 2) {
        double pq[3] = {pw[0], pw[1], pw[2] - oq};          /* a surface raised by q == the point lowered by q */
        double gap = terrain_contact(s, pq, rad, n);
        *ca = n[2]; *vs = n[2] * oqd; *ob = slot;
        return gap;
    }
    if (mode == 1) {
        double sq = sin(oq), cq = cos(oq), zl = pw[2] - oh0, pgap = xl * sq + zl * cq - rad;
        double gap = pw[2] - rad;                            /* the ground under the see-saw (the map's tent is not collided with) */
        n[0] = 0; n[1] = 0; n[2] = 1;
        if (pgap >= -(double)QA_SEESAW_SHELL && pgap < gap) {
            gap = pgap; n[0] = cpsi * sq; n[1] = spsi * sq; n[2] = cq;
            *ca = sq * zl - cq * xl; *vs = *ca * oqd; *ob = slot;
        }
        return gap;
    }
    return terrain_contact(s, pw, rad, n);
}
/* joint dynamics of an articulated obstacle over one env step (h = decimation x dt) under the mean generalised contact force F of its
 * substeps; semi-implicit in the damper and the position drive (csrc/qa_physics.h obstacle_joint_step) */
static void obstacle_joint_step(double kind, double damping, double F, double h, double *q, double *qd) {
    if (kind == QA_OBST_SEESAW) {
        *qd = (*qd + h * F / QA_SEESAW_INERTIA) / (1.0 + h * damping / QA_SEESAW_INERTIA);
        if (*qd > QA_SEESAW_MAX_VEL) *qd = QA_SEESAW_MAX_VEL;
        if (*qd < -QA_SEESAW_MAX_VEL) *qd = -QA_SEESAW_MAX_VEL;
        *q += h * *qd;
        if (*q > QA_SEESAW_MAX_TILT) { *q = QA_SEESAW_MAX_TILT; *qd = 0; }
        if (*q < -QA_SEESAW_MAX_TILT) { *q = -QA_SEESAW_MAX_TILT; *qd = 0; }
    } else if (kind != 0) {
        double m = kind == QA_OBST_BAR ? QA_BAR_MASS : QA_TYRE_MASS;
        *qd = (*qd + h * (F - QA_OBST_STIFFNESS * *q) / m) / (1.0 + h * QA_OBST_DAMPING / m + h * h * QA_OBST_STIFFNESS / m);
        *q += h * *qd;
    }
}
static void obstacles_begin_step(qo_sim *s, int e) {
    if (s->cfg.terrain_type != 1 || !s->cfg.articulated_obstacles) return;
    float *stt = TP(s, QA_T_OBST_STATE, float) + (int64_t)e * QA_OBST_PER_ENV * QA_OBST_STATE;
    for (int k = 0; k < QA_OBST_PER_ENV; ++k) stt[QA_OBST_STATE * k + 2] = 0.0f;
}
static void obstacles_end_step(qo_sim *s, int e) {
    const qa_config *c = &s->cfg;
    if (c->terrain_type != 1 || !c->articulated_obstacles) return;
    const float *desc = TP(s, QA_T_OBST_DESC, float) + (int64_t)e * QA_OBST_PER_ENV * QA_OBST_DESC;
    float *stt = TP(s, QA_T_OBST_STATE, float) + (int64_t)e * QA_OBST_PER_ENV * QA_OBST_STATE;
    for (int k = 0; k < QA_OBST_PER_ENV; ++k) {
        double q = stt[QA_OBST_STATE * k], qd = stt[QA_OBST_STATE * k + 1];
        obstacle_joint_step(desc[QA_OBST_DESC * k + 7], stt[QA_OBST_STATE * k + 3], (double)stt[QA_OBST_STATE * k + 2] / c->decimation,
                            (double)c->sim_dt * c->decimation, &q, &qd);
        stt[QA_OBST_STATE * k] = (float)q; stt[QA_OBST_STATE * k + 1] = (float)qd;
    }
}
/* tangent basis of a contact: t1 = x-axis projected onto the tangent plane, t2 = n x t1 */
static void tangent_basis(const double n[3], double t1[3], double t2[3]) {
    double d = n[0];
    t1[0] = 1.0 - d * n[0]; t1[1] = -d * n[1]; t1[2] = -d * n[2];
    double inv = 1.0 / sqrt(t1[0] * t1[0] + t1[1] * t1[1] + t1[2] * t1[2]);
    t1[0] *= inv; t1[1] *= inv; t1[2] *= inv;
    cross(n, t1, t2);
}
/* the height-scan sample the BBC env actually uses: point 94 of the 17 x 11 scan = (0.0, 0.1) in the yaw frame,
 * looked up the way _get_heights does (legged_robot.py:1209-1228): truncate to a cell, min of three samples */
static float scan_center_height(const qo_sim *s, const float *root) {
    const qa_config *c = &s->cfg;
    if (c->terrain_type != 1) return 0.0f;
    const int16_t *hs = TP(s, QA_T_HEIGHT_SAMPLES, int16_t);
    float qz = root[5], qw = root[6];
    float nrm = sqrtf(qz * qz + qw * qw); if (nrm < 1e-9f) nrm = 1e-9f;
    qz /= nrm; qw /= nrm;
    /* quat_apply of (0,0,qz,qw) to (0, 0.1, 0):  v + 2 w (q x v) + 2 q x (q x v) */
    float vx = 0.0f, vy = 0.1f;
    float tx = -qz * vy * 2.0f, ty = qz * vx * 2.0f;
    float px = vx + qw * tx - qz * ty + root[0], py = vy + qw * ty + qz * tx + root[1];
    px += c->hf_border; py += c->hf_border;
    long ix = (long)(px / c->hf_hscale), iy = (long)(py / c->hf_hscale);
    if (ix < 0) ix = 0; if (ix > c->hf_rows - 2) ix = c->hf_rows - 2;
    if (iy < 0) iy = 0; if (iy > c->hf_cols - 2) iy = c->hf_cols - 2;
    int16_t a = hs[ix * c->hf_cols + iy], b = hs[(ix + 1) * c->hf_cols + iy], d = hs[ix * c->hf_cols + iy + 1];
    int16_t m = a < b ? a : b; m = m < d ? m : d;
    return (float)m * c->hf_vscale;
}

/* closest points of segments A0 + s (A1 - A0), B0 + t (B1 - B0), s, t in [0, 1] (Ericson, Real-Time Collision Detection 5.1.9) */
static void segment_closest(const double A0[3], const double A1[3], const double B0[3], const double B1[3], double *sp, double *tp) {
    double d1[3], d2[3], r[3];
    for (int i = 0; i < 3; ++i) { d1[i] = A1[i] - A0[i]; d2[i] = B1[i] - B0[i]; r[i] = A0[i] - B0[i]; }
    double a = dot3(d1, d1), e = dot3(d2, d2), f = dot3(d2, r), c = dot3(d1, r), b = dot3(d1, d2), den = a * e - b * b;
    double s = den > 1e-12 ? (b * f - c * e) / den : 0.0;
    s = s < 0 ? 0 : (s > 1 ? 1 : s);
    double t = (b * s + f) / e;
    if (t < 0) { t = 0; s = -c / a; s = s < 0 ? 0 : (s > 1 ? 1 : s); }
    else if (t > 1) { t = 1; s = (b - c) / a; s = s < 0 ? 0 : (s > 1 ? 1 : s); }
    *sp = s; *tp = t;
}
#define CALF_RADIUS 0.013
#define FOOT_RADIUS 0.022
/* r5: the horizontal position of the root is carried across the substeps of ONE env step as an offset from the world position the step
 * starts at (double), not re-read from the fp32 arena after every substep: 900 m from the origin -- where the 8192-env course puts its last
 * envs -- an fp32 world coordinate resolves 6e-5 m, and four roundings per env step moved contact gaps by more than the kernel's and this
 * restatement's arithmetic ever differ.  The kernel does the same in fp32 offsets (csrc/qa_physics.h, TerrainView); the arena still
 * receives the rounded world position after every substep (it is what the gym tensors show), it is just not read back. */
typedef struct { double ax, ay, lx, ly; } StepAnchor;
static void phys_substep_acc(qo_sim *s, int e, const float tau_in[12], int accumulate, StepAnchor *an);
static void phys_substep(qo_sim *s, int e, const float tau_in[12]) { phys_substep_acc(s, e, tau_in, 0, NULL); }
static void phys_substep_acc(qo_sim *s, int e, const float tau_in[12], int accumulate, StepAnchor *an) {
    const qa_config *cfg = &s->cfg;
    float *root = TP(s, QA_T_ROOT_STATES, float) + 13 * e;
    float *dof = TP(s, QA_T_DOF_STATE, float) + 24 * e;
    float *cf = TP(s, QA_T_CONTACT_FORCES, float) + 57 * e;
    float *rbp = TP(s, QA_T_RIGID_BODY_POS, float) + 57 * e;
    const float *binert = TP(s, QA_T_BASE_INERTIA, float) + 10 * e;
    double dt = cfg->sim_dt;
    double pos[3] = {an ? an->ax + an->lx : (double)root[0], an ? an->ay + an->ly : (double)root[1], root[2]}, quat[4] = {root[3], root[4], root[5], root[6]};
    m3 R; quat_to_mat(quat, R);
    v3 vw = {root[7], root[8], root[9]}, ww = {root[10], root[11], root[12]}, gw = {0, 0, cfg->gravity_z}, gB;
    double ub[6]; mtv(R, ww, ub); mtv(R, vw, ub + 3); mtv(R, gw, gB);
    double q[12], qd[12];
    for (int j = 0; j < 12; ++j) { q[j] = dof[2 * j]; qd[j] = dof[2 * j + 1]; }

    Kin K;
    for (int l = 0; l < 4; ++l) leg_kin(l, q + 3 * l, &K);
    RB base; base_rb(binert, &base);
    double M[18][18], h[18], Lc[18][18];
    dynamics_terms(&K, &base, ub, qd, gB, M, h);
    memcpy(Lc, M, sizeof(M));
    if (chol(Lc, 18) != 0) return; /* singular: leave state untouched (never happens for a physical robot) */

    /* unconstrained velocity: u* = u + dt (M^-1 (tau - h) + [0; w x v; 0]) */
    double rhs[18], acc[18], u[18];
    for (int i = 0; i < 6; ++i) rhs[i] = -h[i];
    for (int j = 0; j < 12; ++j) rhs[6 + j] = (double)tau_in[j] - h[6 + j];
    chol_solve(Lc, 18, rhs, acc);
    v3 wxv; cross(ub, ub + 3, wxv);
    for (int i = 0; i < 6; ++i) u[i] = ub[i] + dt * acc[i];
    for (int i = 0; i < 3; ++i) u[3 + i] += dt * wxv[i];
    for (int j = 0; j < 12; ++j) u[6 + j] = qd[j] + dt * acc[6 + j];

    /* ---- constraint rows, in the fixed Gauss-Seidel order: for leg l: foot(n,t1,t2), then up to TWO "extra" contacts (n,t1,t2
     * each) out of three candidates -- group 0 = the lowest point of the hip link or of this leg's share of the base / head points,
     * group 1 = the lowest thigh point, group 2 = the lowest non-foot calf point; when all three are inside the contact offset the
     * one with the largest gap waits -- then the joint-limit rows.  One candidate per body group, so thigh and calf (the 8 bodies
     * _reward_collision counts, legged_robot.py:1275-1278) and hip / base (check_termination, :168-176) report forces
     * independently of each other. */
    Row rows[4 * 15 + 4];
    int nrows = 0;
    int foot_row[4], extra_row[4][3], extra_body[4][3];
    double mu = 0.5 * ((double)TP(s, QA_T_FRICTION, float)[e] + cfg->ground_friction);
    double cdirs[16][3][3];     /* world-frame (n, t1, t2) of each contact slot (4 per leg), for the force report */
    double slot_ca[16]; int slot_ob[16];   /* articulated obstacle under each contact slot: joint lever, obstacle slot (-1 none) */
    for (int i = 0; i < 16; ++i) { slot_ca[i] = 0; slot_ob[i] = -1; }
    int row_leg_first[5] = {0, 0, 0, 0, 0};
    for (int l = 0; l < 4; ++l) {
        row_leg_first[l] = nrows;
        foot_row[l] = -1;
        double best_gap[3] = {1e30, 1e30, 1e30}; v3 best_p[3], best_n[3]; int best_depth[3] = {-1, -1, -1}, best_body[3] = {-1, -1, -1};
        double best_vs[3] = {0, 0, 0}, best_ca[3] = {0, 0, 0}, foot_vs = 0, foot_ca = 0; int best_ob[3] = {-1, -1, -1}, foot_ob = -1;
        for (int sl = 0; sl < 3; ++sl) { extra_row[l][sl] = -1; extra_body[l][sl] = -1; for (int i = 0; i < 3; ++i) { best_p[sl][i] = 0; best_n[sl][i] = i == 2; } }
        double foot_gap = 0; v3 foot_p = {0, 0, 0}, foot_n = {0, 0, 1};
        for (int c = 0; c < QA_NUM_LEG_PTS; ++c) {
            int k = QA_LEG_PT_LINK[l][c];
            v3 pl = {QA_LEG_PT_POS[l][c][0], QA_LEG_PT_POS[l][c][1], QA_LEG_PT_POS[l][c][2]}, p, pwld;
            mv(K.Rl[l][k], pl, p); for (int i = 0; i < 3; ++i) p[i] += K.o[l][k][i];
            mv(R, p, pwld); for (int i = 0; i < 3; ++i) pwld[i] += pos[i];
            v3 gn; double cvs, cca; int cob;
            double gap = contact_query(s, e, pwld, QA_LEG_PT_RAD[l][c], gn, &cvs, &cca, &cob);      /* distance to the terrain triangle's plane (floor, or ceiling), or to an articulated obstacle */
            if (c == 0) { foot_gap = gap; memcpy(foot_p, p, sizeof(v3)); memcpy(foot_n, gn, sizeof(v3)); foot_vs = cvs; foot_ca = cca; foot_ob = cob; }
            else if (gap < best_gap[k]) { best_gap[k] = gap; memcpy(best_p[k], p, sizeof(v3)); memcpy(best_n[k], gn, sizeof(v3)); best_depth[k] = k; best_body[k] = QA_LEG_PT_BODY[l][c];
                                          best_vs[k] = cvs; best_ca[k] = cca; best_ob[k] = cob; }
        }
        for (int c = l; c < QA_NUM_BASE_PTS; c += 4) { /* base points are dealt round-robin to the four legs; they share slot 0 with the hip link */
            v3 p = {QA_BASE_PT_POS[c][0], QA_BASE_PT_POS[c][1], QA_BASE_PT_POS[c][2]}, pwld;
            mv(R, p, pwld); for (int i = 0; i < 3; ++i) pwld[i] += pos[i];
            v3 gn; double cvs, cca; int cob;
            double gap = contact_query(s, e, pwld, QA_BASE_PT_RAD[c], gn, &cvs, &cca, &cob);
            if (gap < best_gap[0]) { best_gap[0] = gap; memcpy(best_p[0], p, sizeof(v3)); memcpy(best_n[0], gn, sizeof(v3)); best_depth[0] = -1; best_body[0] = QA_BASE_PT_BODY[c];
                                     best_vs[0] = cvs; best_ca[0] = cca; best_ob[0] = cob; }
        }
        if (cfg->contact_slots == 1) {      /* only the lowest non-foot point of the leg makes contact (the round-1 model) */
            int win = 0;
            if (best_gap[1] < best_gap[0] && !(best_gap[2] < best_gap[1])) win = 1;
            if (best_gap[2] < best_gap[0] && best_gap[2] < best_gap[1]) win = 2;
            for (int g = 0; g < 3; ++g) if (g != win) best_gap[g] = 1e30;
        } else if (best_gap[0] < cfg->contact_offset && best_gap[1] < cfg->contact_offset && best_gap[2] < cfg->contact_offset) {
            /* three candidates, two contact slots per leg: the one with the largest gap waits */
            int drop = (best_gap[0] >= best_gap[1] && best_gap[0] >= best_gap[2]) ? 0 : (best_gap[1] >= best_gap[2] ? 1 : 2);
            best_gap[drop] = 1e30;
        }
        for (int slot = 0; slot < 4; ++slot) {
            double gap = slot == 0 ? foot_gap : best_gap[slot - 1];
            const double *p = slot == 0 ? foot_p : best_p[slot - 1];
            int depth = slot == 0 ? 2 : best_depth[slot - 1];
            if (!(gap < cfg->contact_offset)) continue;
            if (slot == 0) foot_row[l] = nrows; else { extra_row[l][slot - 1] = nrows; extra_body[l][slot - 1] = best_body[slot - 1]; }
            const double surf_vs = slot == 0 ? foot_vs : best_vs[slot - 1];
            slot_ca[4 * l + slot] = slot == 0 ? foot_ca : best_ca[slot - 1]; slot_ob[4 * l + slot] = slot == 0 ? foot_ob : best_ob[slot - 1];
            double (*cw)[3] = cdirs[4 * l + slot];
            memcpy(cw[0], slot == 0 ? foot_n : best_n[slot - 1], sizeof(v3));
            tangent_basis(cw[0], cw[1], cw[2]);
            v3 dB[3]; for (int d = 0; d < 3; ++d) mtv(R, cw[d], dB[d]);       /* contact frame in base coordinates */
            const double *dirs[3] = {dB[0], dB[1], dB[2]};
            for (int d = 0; d < 3; ++d) {
                Row *r = &rows[nrows];
                memset(r, 0, sizeof(*r));
                v3 pxd; cross(p, dirs[d], pxd);
                for (int i = 0; i < 3; ++i) { r->J[i] = pxd[i]; r->J[3 + i] = dirs[d][i]; }
                for (int k = 0; k <= depth; ++k) {
                    v3 rel = {p[0] - K.o[l][k][0], p[1] - K.o[l][k][1], p[2] - K.o[l][k][2]}, axr;
                    cross(K.a[l][k], rel, axr);
                    r->J[6 + 3 * l + k] = dot3(dirs[d], axr);
                }
                r->kind = d == 0 ? 0 : 1; r->parent = nrows - d;
                if (d == 0) {
                    double mdv = cfg->max_depenetration_velocity;
                    r->bias = (gap >= 0 ? gap / dt : (gap / dt > -mdv ? gap / dt : -mdv)) - surf_vs;      /* a surface that moves along the normal */
                }
                nrows++;
            }
        }
        for (int k = 0; k < 3; ++k) {
            int j = 3 * l + k;
            double glo = q[j] - QA_DOF_LOWER[l][k], ghi = QA_DOF_UPPER[l][k] - q[j];
            double sgn = 0, gap = 0;
            if (glo < LIMIT_MARGIN) { sgn = 1; gap = glo; } else if (ghi < LIMIT_MARGIN) { sgn = -1; gap = ghi; }
            if (sgn == 0) continue;
            Row *r = &rows[nrows++];
            memset(r, 0, sizeof(*r));
            r->J[6 + j] = sgn; r->kind = 2;
            r->bias = gap >= 0 ? gap / dt : (gap / dt > -LIMIT_DEPEN ? gap / dt : -LIMIT_DEPEN);
        }
    }
    row_leg_first[4] = nrows;
    /* ---- self-collision (cfg.self_collision; csrc/qa_physics.h): the lower legs as capsules (knee -> foot centre, radius 13 mm at the knee
     * growing to the foot sphere's 22 mm), left/right pairs (0,1), (2,3) and front/rear pairs (0,2), (1,3); one frictionless row per pair
     * at the closest points, normal from the second capsule to the first.  The points move with the same base: J has leg columns only. */
    static const int PAIRS[4][2] = {{0, 1}, {2, 3}, {0, 2}, {1, 3}};
    int pair_row[4] = {-1, -1, -1, -1}; double pair_n[4][3];
    if (cfg->self_collision) {
        for (int pi = 0; pi < 4; ++pi) {
            const int la = PAIRS[pi][0], lb = PAIRS[pi][1];
            v3 fa, fb, pl = {QA_LEG_PT_POS[la][0][0], QA_LEG_PT_POS[la][0][1], QA_LEG_PT_POS[la][0][2]}, pl2 = {QA_LEG_PT_POS[lb][0][0], QA_LEG_PT_POS[lb][0][1], QA_LEG_PT_POS[lb][0][2]};
            mv(K.Rl[la][2], pl, fa); mv(K.Rl[lb][2], pl2, fb);
            for (int i = 0; i < 3; ++i) { fa[i] += K.o[la][2][i]; fb[i] += K.o[lb][2][i]; }
            double sa, tb; segment_closest(K.o[la][2], fa, K.o[lb][2], fb, &sa, &tb);
            v3 pA, pB, dv;
            for (int i = 0; i < 3; ++i) { pA[i] = K.o[la][2][i] + sa * (fa[i] - K.o[la][2][i]); pB[i] = K.o[lb][2][i] + tb * (fb[i] - K.o[lb][2][i]); dv[i] = pA[i] - pB[i]; }
            double dist = sqrt(dot3(dv, dv)), n[3] = {0, 1, 0};
            if (dist > 1e-6) for (int i = 0; i < 3; ++i) n[i] = dv[i] / dist;
            double gap = dist - (CALF_RADIUS + sa * (FOOT_RADIUS - CALF_RADIUS)) - (CALF_RADIUS + tb * (FOOT_RADIUS - CALF_RADIUS));
            if (!(gap < cfg->contact_offset)) continue;
            Row *r = &rows[nrows];
            memset(r, 0, sizeof(*r));
            for (int k = 0; k < 3; ++k) {
                v3 ra = {pA[0] - K.o[la][k][0], pA[1] - K.o[la][k][1], pA[2] - K.o[la][k][2]}, rb = {pB[0] - K.o[lb][k][0], pB[1] - K.o[lb][k][1], pB[2] - K.o[lb][k][2]}, axr;
                cross(K.a[la][k], ra, axr); r->J[6 + 3 * la + k] = dot3(n, axr);
                cross(K.a[lb][k], rb, axr); r->J[6 + 3 * lb + k] = -dot3(n, axr);
            }
            double mdv = cfg->max_depenetration_velocity;
            r->bias = gap >= 0 ? gap / dt : (gap / dt > -mdv ? gap / dt : -mdv);
            r->kind = 0;
            memcpy(pair_n[pi], n, sizeof(v3));
            pair_row[pi] = nrows++;
        }
    }
    for (int i = 0; i < nrows; ++i) {
        chol_solve(Lc, 18, rows[i].J, rows[i].W);
        double d = 0; for (int k = 0; k < 18; ++k) d += rows[i].J[k] * rows[i].W[k];
        rows[i].dinv = 1.0 / (d + CFM);
    }
    /* warm start: the foot rows start from the impulses of the previous substep (applied to u first);
     * a foot that is not in contact this substep forgets its impulse */
    float *fimp = TP(s, QA_T_FOOT_IMPULSE, float) + 12 * e;
    for (int l = 0; l < 4; ++l) {
        if (foot_row[l] < 0) { fimp[3 * l] = fimp[3 * l + 1] = fimp[3 * l + 2] = 0.0f; continue; }
        for (int d = 0; d < 3; ++d) {
            Row *r = &rows[foot_row[l] + d];
            r->lam = fimp[3 * l + d];
            for (int k = 0; k < 18; ++k) u[k] += r->W[k] * r->lam;
        }
    }
    /* Projected Gauss-Seidel with a two-colour ordering over the legs: the diagonal pairs {FL, RR} and {FR, RL} couple
     * only weakly through the base (their lever arms cancel in the rotational term), so the two legs of a colour are
     * updated from the SAME base velocity and their velocity changes are summed (block Jacobi inside a colour),
     * while the colours, and the rows inside a leg, follow each other Gauss-Seidel fashion.  On the GPU this is
     * 2 instead of 4 serial passes per sweep (one lane per leg).  Inside a leg: foot (normal, then both tangents
     * together), extra contact (same), joint-limit rows. */
    for (int it = 0; it < cfg->solver_iterations; ++it) {
        for (int color = 0; color < 2; ++color) {
            double u0[18], du[18]; memcpy(u0, u, sizeof(u0)); memset(du, 0, sizeof(du));
            for (int l = 0; l < 4; ++l) {
                if (((l == 0 || l == 3) ? 0 : 1) != color) continue;
                double ul[18]; memcpy(ul, u0, sizeof(ul));
                for (int i = row_leg_first[l]; i < row_leg_first[l + 1]; ++i) {
                    Row *r = &rows[i];
                    if (r->kind == 1) {
                        if (i - r->parent != 1) continue;
                        Row *r2 = &rows[i + 1];
                        double res1 = 0, res2 = 0;
                        for (int k = 0; k < 18; ++k) { res1 += r->J[k] * ul[k]; res2 += r2->J[k] * ul[k]; }
                        double lim = mu * rows[r->parent].lam;
                        double l1 = r->lam - res1 * r->dinv, l2 = r2->lam - res2 * r2->dinv;
                        l1 = l1 < -lim ? -lim : (l1 > lim ? lim : l1); l2 = l2 < -lim ? -lim : (l2 > lim ? lim : l2);
                        double d1 = l1 - r->lam, d2 = l2 - r2->lam; r->lam = l1; r2->lam = l2;
                        for (int k = 0; k < 18; ++k) ul[k] += r->W[k] * d1 + r2->W[k] * d2;
                        continue;
                    }
                    double res = r->bias; for (int k = 0; k < 18; ++k) res += r->J[k] * ul[k];
                    double lam = r->lam - res * r->dinv; if (lam < 0) lam = 0;
                    double dl = lam - r->lam; r->lam = lam;
                    for (int k = 0; k < 18; ++k) ul[k] += r->W[k] * dl;
                }
                for (int k = 0; k < 18; ++k) du[k] += ul[k] - u0[k];
            }
            for (int k = 0; k < 18; ++k) u[k] = u0[k] + du[k];
        }
        /* self-collision pairs: the two left/right pairs from the same velocity (changes summed), then the two front/rear pairs */
        for (int ph = 0; ph < 2; ++ph) {
            double u0[18], du[18]; memcpy(u0, u, sizeof(u0)); memset(du, 0, sizeof(du));
            for (int pi = 2 * ph; pi < 2 * ph + 2; ++pi) {
                if (pair_row[pi] < 0) continue;
                Row *r = &rows[pair_row[pi]];
                double res = r->bias; for (int k = 0; k < 18; ++k) res += r->J[k] * u0[k];
                double lam = r->lam - res * r->dinv; if (lam < 0) lam = 0;
                double dl = lam - r->lam; r->lam = lam;
                for (int k = 0; k < 18; ++k) du[k] += r->W[k] * dl;
            }
            for (int k = 0; k < 18; ++k) u[k] = u0[k] + du[k];
        }
    }
    /* joint velocity clamp (PhysX maxJointVelocity = URDF velocity limit) */
    for (int l = 0; l < 4; ++l) for (int k = 0; k < 3; ++k) {
        double vl = QA_DOF_VELLIM[l][k]; int j = 6 + 3 * l + k;
        u[j] = u[j] < -vl ? -vl : (u[j] > vl ? vl : u[j]);
    }

    /* ---- integrate: new world velocities expressed through the old frame, then the pose */
    v3 wn, vn; mv(R, u, wn); mv(R, u + 3, vn);
    for (int i = 0; i < 3; ++i) pos[i] += dt * vn[i];
    double wb[3] = {u[0], u[1], u[2]};
    double ang = sqrt(dot3(wb, wb)) * dt, dq[4];
    if (ang > 1e-12) { double sc = sin(0.5 * ang) / (ang / dt); dq[0] = wb[0] * sc; dq[1] = wb[1] * sc; dq[2] = wb[2] * sc; dq[3] = cos(0.5 * ang); }
    else { dq[0] = 0.5 * dt * wb[0]; dq[1] = 0.5 * dt * wb[1]; dq[2] = 0.5 * dt * wb[2]; dq[3] = 1; }
    double qn[4] = { /* quat (x) dq  (body-frame increment => right multiplication) */
        quat[3] * dq[0] + quat[0] * dq[3] + quat[1] * dq[2] - quat[2] * dq[1],
        quat[3] * dq[1] - quat[0] * dq[2] + quat[1] * dq[3] + quat[2] * dq[0],
        quat[3] * dq[2] + quat[0] * dq[1] - quat[1] * dq[0] + quat[2] * dq[3],
        quat[3] * dq[3] - quat[0] * dq[0] - quat[1] * dq[1] - quat[2] * dq[2]};
    double nn = 1.0 / sqrt(qn[0] * qn[0] + qn[1] * qn[1] + qn[2] * qn[2] + qn[3] * qn[3]);
    if (an) { an->lx = pos[0] - an->ax; an->ly = pos[1] - an->ay; }
    for (int i = 0; i < 3; ++i) { root[i] = (float)pos[i]; root[7 + i] = (float)vn[i]; root[10 + i] = (float)wn[i]; }
    for (int i = 0; i < 4; ++i) root[3 + i] = (float)(qn[i] * nn);
    for (int j = 0; j < 12; ++j) { dof[2 * j] = (float)(q[j] + dt * u[6 + j]); dof[2 * j + 1] = (float)u[6 + j]; }

    for (int l = 0; l < 4; ++l) if (foot_row[l] >= 0) for (int d = 0; d < 3; ++d) fimp[3 * l + d] = (float)rows[foot_row[l] + d].lam;
    /* ---- contact forces per body, world frame: lam_n n + lam_t1 t1 + lam_t2 t2 (plane: t1, t2, n are world x, y, z) */
    memset(cf, 0, sizeof(float) * 57);
    for (int l = 0; l < 4; ++l) for (int slot = 0; slot < 4; ++slot) {
        int r0 = slot == 0 ? foot_row[l] : extra_row[l][slot - 1];
        if (r0 < 0) continue;
        int b = slot == 0 ? QA_LEG_PT_BODY[l][0] : extra_body[l][slot - 1];
        double (*cw)[3] = cdirs[4 * l + slot];
        for (int i = 0; i < 3; ++i)
            cf[3 * b + i] += (float)((rows[r0].lam * cw[0][i] + rows[r0 + 1].lam * cw[1][i] + rows[r0 + 2].lam * cw[2][i]) / dt);
    }
    for (int pi = 0; pi < 4; ++pi) if (pair_row[pi] >= 0) {      /* self-collision: +n on the first leg's calf, -n on the second's */
        v3 nw; mv(R, pair_n[pi], nw);
        const int ba = QA_LEG_PT_BODY[PAIRS[pi][0]][11], bb = QA_LEG_PT_BODY[PAIRS[pi][1]][11];
        for (int i = 0; i < 3; ++i) { cf[3 * ba + i] += (float)(rows[pair_row[pi]].lam * nw[i] / dt); cf[3 * bb + i] -= (float)(rows[pair_row[pi]].lam * nw[i] / dt); }
    }
    /* ---- what the contacts do to the articulated obstacles' joints: generalised force -f_n ca, accumulated over the env step's substeps */
    if (accumulate && cfg->terrain_type == 1 && cfg->articulated_obstacles) {
        float *stt = TP(s, QA_T_OBST_STATE, float) + (int64_t)e * QA_OBST_PER_ENV * QA_OBST_STATE;
        double ga[QA_OBST_PER_ENV] = {0, 0, 0};
        for (int l = 0; l < 4; ++l) for (int slot = 0; slot < 4; ++slot) {
            int r0 = slot == 0 ? foot_row[l] : extra_row[l][slot - 1];
            if (r0 < 0 || slot_ob[4 * l + slot] < 0) continue;
            ga[slot_ob[4 * l + slot]] += -rows[r0].lam / dt * slot_ca[4 * l + slot];
        }
        for (int k = 0; k < QA_OBST_PER_ENV; ++k) stt[QA_OBST_STATE * k + 2] += (float)ga[k];
    }
    /* ---- body-origin positions with the NEW state (what refresh_rigid_body_state_tensor returns) */
    {
        double q2[12]; for (int j = 0; j < 12; ++j) q2[j] = dof[2 * j];
        double qq[4] = {root[3], root[4], root[5], root[6]}; m3 R2; quat_to_mat(qq, R2);
        Kin K2; for (int l = 0; l < 4; ++l) leg_kin(l, q2 + 3 * l, &K2);
        v3 pts[19]; memset(pts, 0, sizeof(pts));
        pts[1][0] = 0.285; pts[1][2] = 0.01; pts[2][0] = 0.293; pts[2][2] = -0.06;
        for (int l = 0; l < 4; ++l) { for (int k = 0; k < 3; ++k) memcpy(pts[3 + 4 * l + k], K2.o[l][k], sizeof(v3)); memcpy(pts[3 + 4 * l + 3], K2.foot[l], sizeof(v3)); }
        for (int b = 0; b < 19; ++b) { v3 w; mv(R2, pts[b], w); for (int i = 0; i < 3; ++i) rbp[3 * b + i] = (float)(w[i] + (double)root[i]); }
        if (cfg->export_body_state) {
            /* QA_T_RIGID_BODY_STATE: origin position, orientation (xyzw), origin velocity, angular velocity, world frame.
             * Link k of a leg is moved by joints 0..k, its origin (joint k) by joints 0..k-1; the foot is fixed to the calf;
             * Head_upper / Head_lower are fixed to the base. */
            float *rbs = TP(s, QA_T_RIGID_BODY_STATE, float) + (int64_t)e * (QA_NUM_BODIES_ABI * 13);
            v3 wB, vB; { v3 a = {root[10], root[11], root[12]}, b2 = {root[7], root[8], root[9]}; mtv(R2, a, wB); mtv(R2, b2, vB); }
            for (int b = 0; b < 19; ++b) {
                int l = b < 3 ? -1 : (b - 3) / 4, kb = b < 3 ? -1 : (b - 3) % 4, kq = kb > 2 ? 2 : kb;
                v3 w = {wB[0], wB[1], wB[2]}, v, t; cross(wB, pts[b], t); for (int i = 0; i < 3; ++i) v[i] = vB[i] + t[i];
                double qbody[4] = {qq[0], qq[1], qq[2], qq[3]};
                if (l >= 0) {
                    for (int j = 0; j < 3; ++j) {
                        double qdj = dof[2 * (3 * l + j) + 1];
                        if (j <= kq) for (int i = 0; i < 3; ++i) w[i] += qdj * K2.a[l][j][i];
                        if (j < kb) { v3 rel = {pts[b][0] - K2.o[l][j][0], pts[b][1] - K2.o[l][j][1], pts[b][2] - K2.o[l][j][2]}, axr; cross(K2.a[l][j], rel, axr); for (int i = 0; i < 3; ++i) v[i] += qdj * axr[i]; }
                    }
                    double hq = 0.5 * q2[3 * l], ha = 0.5 * (kq == 0 ? 0.0 : (kq == 1 ? q2[3 * l + 1] : q2[3 * l + 1] + q2[3 * l + 2]));
                    double qx_[4] = {sin(hq), 0, 0, cos(hq)}, qy_[4] = {0, sin(ha), 0, cos(ha)}, ql[4], qt[4];
                    /* ql = qx (x) qy, qbody = qbase (x) ql   (xyzw Hamilton products) */
                    ql[0] = qx_[3] * qy_[0] + qx_[0] * qy_[3] + qx_[1] * qy_[2] - qx_[2] * qy_[1];
                    ql[1] = qx_[3] * qy_[1] - qx_[0] * qy_[2] + qx_[1] * qy_[3] + qx_[2] * qy_[0];
                    ql[2] = qx_[3] * qy_[2] + qx_[0] * qy_[1] - qx_[1] * qy_[0] + qx_[2] * qy_[3];
                    ql[3] = qx_[3] * qy_[3] - qx_[0] * qy_[0] - qx_[1] * qy_[1] - qx_[2] * qy_[2];
                    qt[0] = qq[3] * ql[0] + qq[0] * ql[3] + qq[1] * ql[2] - qq[2] * ql[1];
                    qt[1] = qq[3] * ql[1] - qq[0] * ql[2] + qq[1] * ql[3] + qq[2] * ql[0];
                    qt[2] = qq[3] * ql[2] + qq[0] * ql[1] - qq[1] * ql[0] + qq[2] * ql[3];
                    qt[3] = qq[3] * ql[3] - qq[0] * ql[0] - qq[1] * ql[1] - qq[2] * ql[2];
                    memcpy(qbody, qt, sizeof(qt));
                }
                v3 vw, ww2; mv(R2, v, vw); mv(R2, w, ww2);
                float *r = rbs + 13 * b;
                for (int i = 0; i < 3; ++i) { r[i] = rbp[3 * b + i]; r[7 + i] = (float)vw[i]; r[10 + i] = (float)ww2[i]; }
                for (int i = 0; i < 4; ++i) r[3 + i] = (float)qbody[i];
            }
        }
    }
}

/* ------------------------------------------------------------------ env-side math (fp32) */
static void compute_torques(const qo_sim *s, int e, const float *act, float tau[12], float tau_org[12]) {
    /* legged_robot.py:547-579, control_type 'P', randomize_motor */
    const qa_config *c = &s->cfg;
    const float *dof = TP(s, QA_T_DOF_STATE, float) + 24 * e;
    const float *ms = TP(s, QA_T_MOTOR_STRENGTH, float);
    int N = c->num_envs;
    for (int j = 0; j < 12; ++j) {
        float a = act[j] * c->action_scale;
        if (j % 3 == 0) a *= c->hip_scale_reduction;
        float sp = c->randomize_motor ? ms[(0 * N + e) * 12 + j] : 1.0f, sd = c->randomize_motor ? ms[(1 * N + e) * 12 + j] : 1.0f;
        float t;
        if (c->randomize_motor) t = sp * c->kp * (a + c->default_dof_pos[j] - dof[2 * j]) - sd * c->kd * dof[2 * j + 1];
        else t = c->kp * (a + c->default_dof_pos[j] - dof[2 * j]) - c->kd * dof[2 * j + 1];
        tau_org[j] = t;
        float lim = QA_DOF_EFFORT[j / 3][j % 3];
        tau[j] = clipf(t, -lim, lim);
    }
}

static void resample_commands(qo_sim *s, int e, int64_t step, int stream) {
    /* legged_robot.py:532-540 (latents) and :474-530 (commands) */
    const qa_config *c = &s->cfg;
    float u0[4], u1[4];
    rng4(s, e, step, stream, 0, u0); rng4(s, e, step, stream, 1, u1);
    float *eps = TP(s, QA_T_LATENT_EPS, float) + e, *lc = TP(s, QA_T_LATENT_C, float) + 5 * e, *cmd = TP(s, QA_T_COMMANDS, float) + 5 * e;
    const float *prior = TP(s, QA_T_PRIOR_PARAMETERS, float);
    eps[0] = u0[1] * 2.0f - 1.0f;
    float z[5], zmax = -1e30f, sum = 0;
    for (int g = 0; g < 5; ++g) { z[g] = prior[g] / c->latent_temperature; if (z[g] > zmax) zmax = z[g]; }
    for (int g = 0; g < 5; ++g) { z[g] = expf(z[g] - zmax); sum += z[g]; }
    int gait = 4; float acc = 0;
    for (int g = 0; g < 5; ++g) { acc += z[g] / sum; if (u0[0] < acc) { gait = g; break; } }
    for (int g = 0; g < 5; ++g) lc[g] = g == gait ? 1.0f : 0.0f;
    float vx = (c->lin_vel_x[gait][1] - c->lin_vel_x[gait][0]) * u0[2] + c->lin_vel_x[gait][0];
    float vy = (c->lin_vel_y[gait][1] - c->lin_vel_y[gait][0]) * u0[3] + c->lin_vel_y[gait][0];
    float wz = (c->ang_vel_yaw[gait][1] - c->ang_vel_yaw[gait][0]) * u1[0] + c->ang_vel_yaw[gait][0];
    int jump = gait == QA_NUM_GAITS - 1;
    float hj = ((c->jump_height[1] - c->jump_height[0]) * u1[1] + c->jump_height[0]) * (jump ? 1.0f : 0.0f);
    float hl = ((c->locomotion_height[1] - c->locomotion_height[0]) * u1[2] + c->locomotion_height[0]) * (jump ? 0.0f : 1.0f);
    cmd[0] = vx * (fabsf(vx) > c->lin_vel_x_clip ? 1.0f : 0.0f);
    cmd[1] = vy * (fabsf(vy) > c->lin_vel_y_clip ? 1.0f : 0.0f);
    cmd[2] = wz * (fabsf(wz) > c->ang_vel_yaw_clip ? 1.0f : 0.0f);
    cmd[3] = hj; cmd[4] = hl;
}

/* quaternion_slerp of bbc/rsl_rl/utils/utils.py:126-159 for one pair, fp32, masks in the reference's order (see the HIP twin) */
static void mocap_slerp(const float *q0, const float *q1, float f, float out[4]) {
    float d = q0[0] * q1[0] + q0[1] * q1[1] + q0[2] * q1[2] + q0[3] * q1[3];
    int at_zero = fabsf(f) <= 1e-8f, at_one = fabsf(f - 1.0f) <= (1e-8f + 1e-5f);
    int same = fabsf(fabsf(d) - 1.0f) < 8.8817842e-16f;
    float sg = d < 0.f ? -1.0f : 1.0f;
    d = clipf(d * sg, -1.0f, 1.0f);
    float ang = acosf(d);
    int tiny = fabsf(ang) < 8.8817842e-16f;
    float isin = 1.0f / ang, w0 = sinf((1.0f - f) * ang) * isin, w1 = sinf(f * ang) * isin * sg;
    for (int i = 0; i < 4; ++i) out[i] = (same || tiny) ? q0[i] : (at_one ? q1[i] : (at_zero ? q0[i] : q0[i] * w0 + q1[i] * w1));
}

/* one draw of MotionLoader.get_full_frame_batch for gait `gait` from the uniforms (u0, u1): the blended 37-float frame */
static void mocap_sample(const qo_sim *s, int gait, float u0, float u1, float f[QA_MOCAP_FRAME]) {
    const double *clips = TP(s, QA_T_MOCAP_CLIPS, double);
    int c0 = s->mocap_first[gait], c1 = s->mocap_first[gait + 1], clip = c1 - 1;
    for (int i = c0; i < c1; ++i) if ((double)u0 < clips[QA_MOCAP_CLIP * i + 4]) { clip = i; break; }
    const double *ct = clips + QA_MOCAP_CLIP * clip;
    double t = ct[3] * (double)u1; if (t < 1e-7) t = 1e-7;
    double pn = t / ct[2] * ct[1], lo = floor(pn), hi = ceil(pn);
    float b = (float)(pn - lo);
    const float *f0 = TP(s, QA_T_MOCAP_FRAMES, float) + ((int64_t)ct[0] + (int64_t)lo) * QA_MOCAP_FRAME;
    const float *f1 = TP(s, QA_T_MOCAP_FRAMES, float) + ((int64_t)ct[0] + (int64_t)hi) * QA_MOCAP_FRAME;
    for (int i = 0; i < QA_MOCAP_FRAME; ++i) f[i] = (1.0f - b) * f0[i] + b * f1[i];
    mocap_slerp(f0 + 3, f1 + 3, b, f + 3);
}

static void reset_env(qo_sim *s, int e, int64_t step, int stats_parity, int report) {
    /* legged_robot.py:178-240 */
    const qa_config *c = &s->cfg;
    int N = c->num_envs;
    resample_commands(s, e, step, RS_CMD_RESET);
    float *root = TP(s, QA_T_ROOT_STATES, float) + 13 * e, *dof = TP(s, QA_T_DOF_STATE, float) + 24 * e;
    const float *org = TP(s, QA_T_ENV_ORIGINS, float) + 3 * e;
    if (c->reset_mode == 1 && s->mocap_first[QA_NUM_GAITS] > 0) {
        /* mocap frame reset (:205-214, :598-612, :660-680) = MotionLoader.get_full_frame_batch (motion_loader.py:461-474):
         * clip ~ MotionWeight inside the env's gait (np.random.choice: first clip whose cumulative probability exceeds u),
         * t = max(1e-7, range u) (:333-342), p n = t / length n in float64 (:411-416), frames floor / ceil, blend in fp32,
         * quaternion_slerp (utils.py:126-159) for the root orientation, root velocities rotated into the world frame */
        const float *lc = TP(s, QA_T_LATENT_C, float) + 5 * e;
        int gait = 0; for (int g = 1; g < 5; ++g) if (lc[g] > lc[gait]) gait = g;
        float u[4]; rng4(s, e, step, RS_RESET, 0, u);
        float f[QA_MOCAP_FRAME];
        mocap_sample(s, gait, u[0], u[1], f);
        for (int i = 0; i < 3; ++i) root[i] = f[i] + org[i];
        for (int i = 0; i < 4; ++i) root[3 + i] = f[3 + i];
        quat_rotate_f(f + 3, f + 19, +1.0f, root + 7);
        quat_rotate_f(f + 3, f + 22, +1.0f, root + 10);
        for (int j = 0; j < 12; ++j) { dof[2 * j] = f[7 + j]; dof[2 * j + 1] = f[25 + j]; }
    } else {
        /* default pose (:581-596, :614-634; plane => custom_origins False) */
        float u[20];
        for (int b = 0; b < 5; ++b) rng4(s, e, step, RS_RESET, b, u + 4 * b);
        for (int j = 0; j < 12; ++j) { dof[2 * j] = c->default_dof_pos[j] * ((1.5f - 0.5f) * u[j] + 0.5f); dof[2 * j + 1] = 0.0f; }
        for (int i = 0; i < 3; ++i) root[i] = c->init_pos[i] + org[i];
        if (c->reset_xy_jitter > 0.0f) { root[0] += (2.0f * u[18] - 1.0f) * c->reset_xy_jitter; root[1] += (2.0f * u[19] - 1.0f) * c->reset_xy_jitter; }   /* :622-625 */
        root[3] = 0; root[4] = 0; root[5] = 0; root[6] = 1;
        for (int i = 0; i < 6; ++i) root[7 + i] = (0.5f - -0.5f) * u[12 + i] + -0.5f;
    }
    memset(TP(s, QA_T_FOOT_IMPULSE, float) + 12 * e, 0, 48);
    memset(TP(s, QA_T_LAST_ACTIONS, float) + 12 * e, 0, 48);
    memset(TP(s, QA_T_LAST_DOF_VEL, float) + 12 * e, 0, 48);
    memset(TP(s, QA_T_LAST_ROOT_VEL, float) + 6 * e, 0, 24);
    memset(TP(s, QA_T_LAST_TORQUES_ORG, float) + 12 * e, 0, 48);
    TP(s, QA_T_EPISODE_LENGTH, int64_t)[e] = 0;
    TP(s, QA_T_RESET, int64_t)[e] = 1;
    memset(TP(s, QA_T_ACTION_HISTORY, float) + 96 * e, 0, 96 * 4);
    memset(TP(s, QA_T_OBS, float) + (int64_t)QA_NUM_OBS * e + 90, 0, 570 * 4);      /* obs_history_buf lives in the obs row */
    float *st = TP(s, QA_T_EPISODE_STATS, float) + 16 * stats_parity, *es = TP(s, QA_T_EPISODE_SUMS, float);
    if (report)
#pragma omp critical(qo_stats)
    {
        for (int r = 0; r < QA_NUM_REWARDS; ++r) st[r] += es[(int64_t)r * N + e];
        st[14] += 1.0f;
    }
    for (int r = 0; r < QA_NUM_REWARDS; ++r) es[(int64_t)r * N + e] = 0;
}

static void compute_observations(qo_sim *s, int e, int64_t step) {
    /* legged_robot.py:261-331 + compute_flat_key_pos :1377-1396 (plane: measured heights are 0) */
    const qa_config *c = &s->cfg;
    const float *root = TP(s, QA_T_ROOT_STATES, float) + 13 * e, *dof = TP(s, QA_T_DOF_STATE, float) + 24 * e;
    const float *rpy = TP(s, QA_T_RPY, float) + 3 * e, *blv = TP(s, QA_T_BASE_LIN_VEL, float) + 3 * e, *bav = TP(s, QA_T_BASE_ANG_VEL, float) + 3 * e;
    const float *rbp = TP(s, QA_T_RIGID_BODY_POS, float) + 57 * e;
    const uint8_t *cfilt = TP(s, QA_T_CONTACT_FILT, uint8_t) + 4 * e;
    int N = c->num_envs;
    float root_h = root[2] - TP(s, QA_T_SCAN_HEIGHT, float)[e];      /* measured height is the pre-reset one, like the reference's */
    /* heading-inverse rotation of the feet (torch_jit_utils.py:23-76) */
    float xdir[3] = {1, 0, 0}, rot_dir[3];
    quat_rotate_f(root + 3, xdir, +1.0f, rot_dir);
    float heading = atan2f(rot_dir[1], rot_dir[0]);
    float th = -heading; /* quat_from_angle_axis(-heading, z): xyz = axis*sin(th/2), w = cos(th/2), then normalize */
    float hq[4] = {0.0f * sinf(th / 2), 0.0f * sinf(th / 2), 1.0f * sinf(th / 2), cosf(th / 2)};
    float hn = sqrtf(hq[0] * hq[0] + hq[1] * hq[1] + hq[2] * hq[2] + hq[3] * hq[3]); if (hn < 1e-9f) hn = 1e-9f;
    for (int i = 0; i < 4; ++i) hq[i] /= hn;
    float key[12];
    for (int l = 0; l < 4; ++l) {
        const float *fp = rbp + 3 * (3 + 4 * l + 3);
        float rel[3] = {fp[0] - root[0], fp[1] - root[1], fp[2] - root[2]};
        quat_rotate_f(hq, rel, +1.0f, key + 3 * l);
    }
    float *od = TP(s, QA_T_OBS_DISC, float) + QA_NUM_OBS_DISC * e;
    od[0] = rpy[0]; od[1] = rpy[1]; od[2] = root_h;
    for (int i = 0; i < 3; ++i) { od[3 + i] = blv[i] * c->s_lin_vel_dist; od[6 + i] = bav[i] * c->s_ang_vel_dist; }
    for (int j = 0; j < 12; ++j) { od[9 + j] = (dof[2 * j] - c->default_dof_pos[j]) * c->s_dof_pos; od[21 + j] = dof[2 * j + 1] * c->s_dof_vel; od[33 + j] = key[j] * c->s_key_pos; }
    for (int l = 0; l < 4; ++l) od[45 + l] = (cfilt[l] ? 1.0f : 0.0f) * c->s_foot_contact;

    float prop[QA_NUM_PROP];
    const float *ah = TP(s, QA_T_ACTION_HISTORY, float) + 96 * e + 12 * (QA_ACTION_BUF_LEN - 1);
    prop[0] = rpy[0]; prop[1] = rpy[1];
    for (int i = 0; i < 3; ++i) prop[2 + i] = bav[i] * c->s_ang_vel;
    for (int j = 0; j < 12; ++j) { prop[5 + j] = (dof[2 * j] - c->default_dof_pos[j]) * c->s_dof_pos; prop[17 + j] = dof[2 * j + 1] * c->s_dof_vel; prop[29 + j] = ah[j]; }
    for (int l = 0; l < 4; ++l) prop[41 + l] = (cfilt[l] ? 1.0f : 0.0f) - 0.5f;
    for (int j = 0; j < 12; ++j) prop[45 + j] = key[j] * 0.0f;

    float *hist = TP(s, QA_T_OBS, float) + (int64_t)QA_NUM_OBS * e + 90;      /* previous step's history, shifted in place */
    int64_t epl = TP(s, QA_T_EPISODE_LENGTH, int64_t)[e];
    if (epl <= 1) { for (int t = 0; t < QA_HISTORY_LEN; ++t) memcpy(hist + 57 * t, prop, 57 * 4); }
    else { memmove(hist, hist + 57, 57 * 9 * 4); memcpy(hist + 57 * 9, prop, 57 * 4); }

    float *o = TP(s, QA_T_OBS, float) + (int64_t)QA_NUM_OBS * e;
    memcpy(o, prop, 57 * 4);
    o[57] = root_h; for (int i = 0; i < 3; ++i) o[58 + i] = blv[i] * c->s_lin_vel;
    const float *mp = TP(s, QA_T_MASS_PARAMS, float) + 4 * e, *ms = TP(s, QA_T_MOTOR_STRENGTH, float);
    for (int i = 0; i < 4; ++i) o[61 + i] = mp[i];
    o[65] = TP(s, QA_T_FRICTION, float)[e];
    for (int j = 0; j < 12; ++j) { o[66 + j] = ms[(0 * N + e) * 12 + j] - 1.0f; o[78 + j] = ms[(1 * N + e) * 12 + j] - 1.0f; }
    memcpy(o + 660, TP(s, QA_T_COMMANDS, float) + 5 * e, 20);
    o[665] = TP(s, QA_T_LATENT_EPS, float)[e];
    memcpy(o + 666, TP(s, QA_T_LATENT_C, float) + 5 * e, 20);
    if (c->add_noise) {
        /* only these 32 entries of noise_scale_vec are non-zero (:721-740); draw index i -> obs index */
        float u[32];
        for (int b = 0; b < 8; ++b) rng4(s, e, step, RS_NOISE, b, u + 4 * b);
        for (int i = 0; i < 32; ++i) {
            int idx = i < 29 ? i : 58 + (i - 29);
            float sc = idx < 2 ? c->noise_roll_pitch : idx < 5 ? c->noise_ang_vel : idx < 17 ? c->noise_dof_pos : idx < 29 ? c->noise_dof_vel : c->noise_lin_vel;
            o[idx] += (2.0f * u[i] - 1.0f) * sc;
        }
    }
    for (int i = 0; i < QA_NUM_OBS; ++i) o[i] = clipf(o[i], -c->clip_obs, c->clip_obs);
    for (int i = 0; i < 570; ++i) hist[i] = clipf(hist[i], -c->clip_obs, c->clip_obs);
}

static void post_physics(qo_sim *s, int e, int64_t step, float *term_disc_tmp) {
    const qa_config *c = &s->cfg;
    int N = c->num_envs;
    float *root = TP(s, QA_T_ROOT_STATES, float) + 13 * e;
    const float *dof = TP(s, QA_T_DOF_STATE, float) + 24 * e, *cfo = TP(s, QA_T_CONTACT_FORCES, float) + 57 * e;
    int64_t *epl = TP(s, QA_T_EPISODE_LENGTH, int64_t) + e;
    *epl += 1;
    int64_t common = step + 1; /* common_step_counter after the increment (:134) */
    float *blv = TP(s, QA_T_BASE_LIN_VEL, float) + 3 * e, *bav = TP(s, QA_T_BASE_ANG_VEL, float) + 3 * e, *pg = TP(s, QA_T_PROJECTED_GRAVITY, float) + 3 * e, *rpy = TP(s, QA_T_RPY, float) + 3 * e;
    float gvec[3] = {0, 0, -1};
    quat_rotate_f(root + 3, root + 7, -1.0f, blv); quat_rotate_f(root + 3, root + 10, -1.0f, bav); quat_rotate_f(root + 3, gvec, -1.0f, pg);
    { /* euler_from_quaternion, torch_jit_utils.py:169-192 */
        float x = root[3], y = root[4], z = root[5], w = root[6];
        rpy[0] = atan2f(2.0f * (w * x + y * z), 1.0f - 2.0f * (x * x + y * y));
        rpy[1] = asinf(clipf(2.0f * (w * y - z * x), -1.0f, 1.0f));
        rpy[2] = atan2f(2.0f * (w * z + x * y), 1.0f - 2.0f * (y * y + z * z));
    }
    float *ff = TP(s, QA_T_FEET_FORCE, float) + 4 * e; uint8_t *lastc = TP(s, QA_T_LAST_CONTACTS, uint8_t) + 4 * e, *cfilt = TP(s, QA_T_CONTACT_FILT, uint8_t) + 4 * e;
    for (int l = 0; l < 4; ++l) {
        const float *f = cfo + 3 * (3 + 4 * l + 3);
        ff[l] = sqrtf(f[0] * f[0] + f[1] * f[1] + f[2] * f[2]);
        uint8_t ct = ff[l] > 2.0f; cfilt[l] = ct | lastc[l]; lastc[l] = ct;
    }
    /* _post_physics_step_callback :449-472 */
    if (*epl % c->resampling_steps == 0) resample_commands(s, e, step, RS_CMD);
    TP(s, QA_T_SCAN_HEIGHT, float)[e] = scan_center_height(s, root);      /* self.measured_heights = self._get_heights() (:469-470) */
    if (c->push_robots && common % c->push_interval == 0) {
        float u[4]; rng4(s, e, step, RS_PUSH, 0, u);
        root[7] = (c->max_push_vel_xy - -c->max_push_vel_xy) * u[0] + -c->max_push_vel_xy;
        root[8] = (c->max_push_vel_xy - -c->max_push_vel_xy) * u[1] + -c->max_push_vel_xy;
    }
    /* check_termination :168-176 : bodies whose name contains "base" or "hip" */
    int reset = 0;
    { const int tb[5] = {0, 3, 7, 11, 15};
      for (int i = 0; i < 5; ++i) { const float *f = cfo + 3 * tb[i]; if (sqrtf(f[0] * f[0] + f[1] * f[1] + f[2] * f[2]) > 1.0f) reset = 1; } }
    int timeout = (*epl > c->max_episode_length) || (root[2] < -6.0f);
    for (int i = 0; i < 13; ++i) if (!isfinite(root[i])) timeout = 1;   /* build-added failure detection */
    reset |= timeout;
    TP(s, QA_T_TIME_OUT, uint8_t)[e] = (uint8_t)timeout;
    TP(s, QA_T_RESET, int64_t)[e] = reset;

    /* compute_reward :242-259 -- terms in alphabetical order */
    const float *act = TP(s, QA_T_ACTIONS, float) + 12 * e, *lact = TP(s, QA_T_LAST_ACTIONS, float) + 12 * e;
    const float *torg = TP(s, QA_T_TORQUES_ORG, float) + 12 * e, *ltorg = TP(s, QA_T_LAST_TORQUES_ORG, float) + 12 * e, *ldv = TP(s, QA_T_LAST_DOF_VEL, float) + 12 * e;
    const float *cmd = TP(s, QA_T_COMMANDS, float) + 5 * e;
    float dtp = c->sim_dt * (float)c->decimation; /* self.dt */
    float term[QA_NUM_REWARDS]; for (int i = 0; i < QA_NUM_REWARDS; ++i) term[i] = 0;
    for (int j = 0; j < 12; ++j) {
        int l = j / 3, k = j % 3;
        float qj = dof[2 * j], qdj = dof[2 * j + 1], d;
        d = lact[j] - act[j]; term[QA_R_ACTION_RATE] += d * d;
        d = torg[j] - ltorg[j]; term[QA_R_DELTA_TORQUES] += d * d;
        d = (ldv[j] - qdj) / dtp; term[QA_R_DOF_ACC] += d * d;
        d = qj - c->default_dof_pos[j]; term[QA_R_DOF_ERROR] += d * d; if (k == 0) term[QA_R_HIP_POS] += d * d;
        { float lo = QA_DOF_LOWER[l][k], hi = QA_DOF_UPPER[l][k], m = (lo + hi) / 2, r = hi - lo;
          float slo = m - 0.5f * r * c->soft_dof_pos_limit, shi = m + 0.5f * r * c->soft_dof_pos_limit;
          float a = qj - slo; a = a > 0 ? 0 : a; float b = qj - shi; b = b < 0 ? 0 : b; term[QA_R_DOF_POS_LIMITS] += -a + b; }
        term[QA_R_DOF_VEL_LIMITS] += clipf(fabsf(qdj) - QA_DOF_VELLIM[l][k] * c->soft_dof_vel_limit, 0.0f, 1.0f);
        { float a = fabsf(torg[j]) - QA_DOF_EFFORT[l][k] * c->soft_torque_limit; term[QA_R_TORQUE_LIMITS] += a < 0 ? 0 : a; }
        term[QA_R_TORQUES] += torg[j] * torg[j];
    }
    for (int l = 0; l < 4; ++l) for (int k = 1; k < 3; ++k) { /* thigh, calf bodies */
        const float *f = cfo + 3 * (3 + 4 * l + k);
        if (sqrtf(f[0] * f[0] + f[1] * f[1] + f[2] * f[2]) > 0.1f) term[QA_R_COLLISION] += 1.0f;
    }
    { float root_h = root[2] - TP(s, QA_T_SCAN_HEIGHT, float)[e];
      float ej = sqrtf((cmd[3] - root_h) * (cmd[3] - root_h));
      term[QA_R_JUMP_UP_HEIGHT] = (ej < 0.05f && cmd[3] >= c->jump_height[0]) ? c->jump_goal : 0.0f;
      float el = sqrtf((cmd[4] - root_h) * (cmd[4] - root_h));
      float rl = expf(-10.0f * (el * el) / c->tracking_sigma);
      term[QA_R_LOCOMOTION_HEIGHT] = (cmd[3] > c->jump_height[0]) ? 0.0f : rl;
      float ea = (cmd[2] - bav[2]) * (cmd[2] - bav[2]);
      term[QA_R_TRACKING_ANG_VEL] = expf(-ea / c->tracking_sigma);
      float elv = (cmd[0] - blv[0]) * (cmd[0] - blv[0]) + (cmd[1] - blv[1]) * (cmd[1] - blv[1]);
      term[QA_R_TRACKING_LIN_VEL] = expf(-elv / c->tracking_sigma); }
    float rew = 0; float *es = TP(s, QA_T_EPISODE_SUMS, float);
    for (int r = 0; r < QA_NUM_REWARDS; ++r) {
        if (c->reward_scale_dt[r] == 0.0f) continue;
        float v = term[r] * c->reward_scale_dt[r]; rew += v; es[(int64_t)r * N + e] += v;
    }
    if (c->only_positive_rewards && rew < 0) rew = 0;
    TP(s, QA_T_REW, float)[e] = rew;

    /* terminal disc obs = obs_disc_buf of the previous compute_observations (:153-154) */
    memcpy(term_disc_tmp, TP(s, QA_T_OBS_DISC, float) + QA_NUM_OBS_DISC * e, QA_NUM_OBS_DISC * 4);
    if (reset) reset_env(s, e, step, (int)(step & 1), 1);
    compute_observations(s, e, step);
    float *odt = TP(s, QA_T_OBS_DISC_TERM, float) + QA_NUM_OBS_DISC * e;
    memcpy(odt, reset ? term_disc_tmp : TP(s, QA_T_OBS_DISC, float) + QA_NUM_OBS_DISC * e, QA_NUM_OBS_DISC * 4);
    /* :158-161 */
    memcpy(TP(s, QA_T_LAST_ACTIONS, float) + 12 * e, act, 48);
    for (int j = 0; j < 12; ++j) TP(s, QA_T_LAST_DOF_VEL, float)[12 * e + j] = dof[2 * j + 1];
    memcpy(TP(s, QA_T_LAST_ROOT_VEL, float) + 6 * e, root + 7, 24);
    memcpy(TP(s, QA_T_LAST_TORQUES_ORG, float) + 12 * e, torg, 48);
}

/* ------------------------------------------------------------------ C ABI (qo_ = oracle twin of qa_) */
int64_t qo_arena_bytes(const qa_config *cfg) { Layout L; if (!cfg || cfg->num_envs <= 0) return QA_E_ARG; make_layout(cfg, &L); return L.total; }

int qo_tensor_info(const qa_config *cfg, int which, int64_t *off, int64_t shape[3], int32_t *ndim, int32_t *dtype) {
    if (!cfg || which < 0 || which >= QA_T_COUNT) return QA_E_ARG;
    Layout L; make_layout(cfg, &L);
    if (off) *off = L.off[which];
    if (shape) for (int i = 0; i < 3; ++i) shape[i] = L.shape[which][i];
    if (ndim) *ndim = L.ndim[which];
    if (dtype) *dtype = L.dtype[which];
    return QA_OK;
}

static float normal_from(float u1, float u2) { /* Box-Muller */
    if (u1 < 1e-7f) u1 = 1e-7f;
    return sqrtf(-2.0f * logf(u1)) * cosf(6.28318530717958647692f * u2);
}

int qo_create(const qa_config *cfg, void *arena, int64_t arena_bytes, void *stream, qo_sim **out) {
    (void)stream;
    if (!cfg || !arena || !out || cfg->num_envs <= 0) return QA_E_ARG;
    if (cfg->abi_version != QA_ABI_VERSION) return QA_E_VERSION;
    qo_sim *s = (qo_sim *)calloc(1, sizeof(qo_sim));
    s->cfg = *cfg; make_layout(cfg, &s->L); s->arena = (char *)arena;
    if (arena_bytes < s->L.total || ((uintptr_t)arena & 255)) { free(s); return QA_E_ARENA; }
    memset(arena, 0, (size_t)s->L.total);
    if (cfg->terrain_type == 1 && cfg->hf_ceiling) { int16_t *cs = TP(s, QA_T_CEILING_SAMPLES, int16_t); for (int64_t i = 0; i < (int64_t)cfg->hf_rows * cfg->hf_cols; ++i) cs[i] = QA_NO_CEILING; }
    int N = cfg->num_envs;
    /* env origins: grid (legged_robot.py:1126-1136) */
    int ncols = (int)floor(sqrt((double)(cfg->num_envs_global > 0 ? cfg->num_envs_global : N)));
    for (int e = 0; e < N; ++e) {
        float *o = TP(s, QA_T_ENV_ORIGINS, float) + 3 * e;
        int ge = e + cfg->env_id_offset;                      /* spawn-grid slot by global env id */
        o[0] = cfg->env_spacing * (float)(ge / ncols); o[1] = cfg->env_spacing * (float)(ge % ncols); o[2] = 0;
        float u[4];
        /* friction: 64 buckets, legged_robot.py:386-401 */
        float fr = 1.0f;
        if (cfg->randomize_friction) {
            rng4(s, e, 0, RS_INIT_FRICTION, 0, u); int b = (int)(u[0] * 64.0f); if (b > 63) b = 63;
            rng4(s, (uint32_t)b, 0, RS_INIT_BUCKET, 0, u);
            fr = (cfg->friction_range[1] - cfg->friction_range[0]) * u[0] + cfg->friction_range[0];
        }
        TP(s, QA_T_FRICTION, float)[e] = fr;
        /* base mass / CoM: legged_robot.py:432-447 */
        rng4(s, e, 0, RS_INIT_MASS, 0, u);
        float *mp = TP(s, QA_T_MASS_PARAMS, float) + 4 * e;
        mp[0] = cfg->randomize_base_mass ? (cfg->added_mass_range[1] - cfg->added_mass_range[0]) * u[0] + cfg->added_mass_range[0] : 0.0f;
        for (int i = 0; i < 3; ++i) mp[1 + i] = cfg->randomize_base_com ? (cfg->added_com_range[1] - cfg->added_com_range[0]) * u[1 + i] + cfg->added_com_range[0] : 0.0f;
        { /* base link inertia about the base origin: mass m0+dm at c0+dc, CoM inertia scaled with mass */
            double m = (double)QA_BASE_MASS + mp[0], sc = m / (double)QA_BASE_MASS;
            v3 cc = {(double)QA_BASE_COM[0] + mp[1], (double)QA_BASE_COM[1] + mp[2], (double)QA_BASE_COM[2] + mp[3]};
            m3 Ic = {{QA_BASE_I[0] * sc, QA_BASE_I[3] * sc, QA_BASE_I[4] * sc}, {QA_BASE_I[3] * sc, QA_BASE_I[1] * sc, QA_BASE_I[5] * sc}, {QA_BASE_I[4] * sc, QA_BASE_I[5] * sc, QA_BASE_I[2] * sc}};
            RB rb; rb_make(m, cc, Ic, &rb);
            float *bi = TP(s, QA_T_BASE_INERTIA, float) + 10 * e;
            bi[0] = (float)rb.m; for (int i = 0; i < 3; ++i) bi[1 + i] = (float)rb.h[i];
            bi[4] = (float)rb.I[0][0]; bi[5] = (float)rb.I[1][1]; bi[6] = (float)rb.I[2][2]; bi[7] = (float)rb.I[0][1]; bi[8] = (float)rb.I[0][2]; bi[9] = (float)rb.I[1][2];
        }
        /* motor strength: legged_robot.py:799-807, 861-888 */
        float uu[48];
        for (int b = 0; b < 12; ++b) rng4(s, e, 0, RS_INIT_MOTOR, b, uu + 4 * b);
        float *ms = TP(s, QA_T_MOTOR_STRENGTH, float);
        for (int j = 0; j < 12; ++j) {
            int pi = 2 * (j % 3);
            float sp, sd;
            if (!cfg->randomize_motor) { sp = sd = 1.0f; }
            else if (cfg->use_easi) { sp = cfg->easi_mean[pi] + cfg->easi_var[pi] * normal_from(uu[j], uu[12 + j]);
                                      sd = cfg->easi_mean[pi + 1] + cfg->easi_var[pi + 1] * normal_from(uu[24 + j], uu[36 + j]); }
            else { sp = (cfg->motor_strength_range[1] - cfg->motor_strength_range[0]) * uu[j] + cfg->motor_strength_range[0];
                   sd = (cfg->motor_strength_range[1] - cfg->motor_strength_range[0]) * uu[12 + j] + cfg->motor_strength_range[0]; }
            ms[(0 * N + e) * 12 + j] = sp; ms[(1 * N + e) * 12 + j] = sd;
        }
        TP(s, QA_T_RESET, int64_t)[e] = 1;
        TP(s, QA_T_ROOT_STATES, float)[13 * e + 6] = 1.0f;
    }
    for (int g = 0; g < QA_NUM_GAITS; ++g) TP(s, QA_T_PRIOR_PARAMETERS, float)[g] = 1.0f / QA_NUM_GAITS; /* :822-823 */
    *out = s;
    return QA_OK;
}

int qo_destroy(qo_sim *s) { free(s); return QA_OK; }

int qo_set_mocap(qo_sim *s, const float *frames, int32_t nf, const double *clips, int32_t nc, const int32_t first[QA_NUM_GAITS + 1], void *stream) {
    (void)stream;
    if (!s || !frames || !clips || !first || nf <= 0 || nf > s->cfg.num_mocap_frames || nc <= 0 || nc > QA_MAX_MOCAP_CLIPS) return QA_E_ARG;
    if (first[0] != 0 || first[QA_NUM_GAITS] != nc) return QA_E_ARG;
    for (int g = 0; g < QA_NUM_GAITS; ++g) if (first[g + 1] <= first[g]) return QA_E_ARG;
    memcpy(TP(s, QA_T_MOCAP_FRAMES, float), frames, (size_t)nf * QA_MOCAP_FRAME * 4);
    double *ct = TP(s, QA_T_MOCAP_CLIPS, double);
    memset(ct, 0, sizeof(double) * QA_MAX_MOCAP_CLIPS * QA_MOCAP_CLIP);
    for (int i = 0; i < nc; ++i) for (int k = 0; k < 5; ++k) ct[QA_MOCAP_CLIP * i + k] = clips[QA_MOCAP_CLIP * i + k];
    for (int g = 0; g <= QA_NUM_GAITS; ++g) ct[QA_MOCAP_CLIP * g + 5] = (double)first[g];
    memcpy(s->mocap_first, first, sizeof(s->mocap_first));
    return QA_OK;
}

int qo_simulate(qo_sim *s, const float *torques, void *stream) {
    (void)stream;
    if (!s || !torques) return QA_E_ARG;
    for (int e = 0; e < s->cfg.num_envs; ++e) {
        float tau[12];
        for (int j = 0; j < 12; ++j) tau[j] = clipf(torques[12 * e + j], -QA_DOF_EFFORT[j / 3][j % 3], QA_DOF_EFFORT[j / 3][j % 3]);
        phys_substep(s, e, tau);
    }
    return QA_OK;
}

int qo_simulate_if(qo_sim *s, const float *torques, const uint8_t *cond, void *stream) {
    if (!s || !cond) return QA_E_ARG;
    if (*cond == 0) return QA_OK;
    return qo_simulate(s, torques ? torques : TP(s, QA_T_TORQUES, float), stream);
}

/* tsc/legged_gym/envs/base/legged_robot.py:118-139 + the refreshes of :231-234 -- the step without post_physics_step */
int qo_env_physics_step(qo_sim *s, const float *actions, int32_t delay_steps, void *stream) {
    (void)stream;
    if (!s || !actions || delay_steps < 0 || delay_steps >= QA_ACTION_BUF_LEN) return QA_E_ARG;
    const qa_config *c = &s->cfg;
#pragma omp parallel for schedule(static)
    for (int e = 0; e < c->num_envs; ++e) {
        float *act = TP(s, QA_T_ACTIONS, float) + 12 * e, *torg = TP(s, QA_T_TORQUES_ORG, float) + 12 * e, *tau = TP(s, QA_T_TORQUES, float) + 12 * e;
        const float *root = TP(s, QA_T_ROOT_STATES, float) + 13 * e, *dof = TP(s, QA_T_DOF_STATE, float) + 24 * e;
        /* last_* = what the previous step's end left (:275-278) = the values this step starts from */
        memcpy(TP(s, QA_T_LAST_ACTIONS, float) + 12 * e, act, 48);
        memcpy(TP(s, QA_T_LAST_TORQUES_ORG, float) + 12 * e, torg, 48);
        for (int j = 0; j < 12; ++j) TP(s, QA_T_LAST_DOF_VEL, float)[12 * e + j] = dof[2 * j + 1];
        memcpy(TP(s, QA_T_LAST_ROOT_VEL, float) + 6 * e, root + 7, 24);
        float *ah = TP(s, QA_T_ACTION_HISTORY, float) + 96 * e;
        memmove(ah, ah + 12, 7 * 12 * 4);
        memcpy(ah + 7 * 12, actions + 12 * e, 48);
        const float *src = ah + 12 * (QA_ACTION_BUF_LEN - 1 - delay_steps);
        float clipa = c->clip_actions / c->action_scale;
        for (int j = 0; j < 12; ++j) act[j] = clipf(src[j], -clipa, clipa);
        obstacles_begin_step(s, e);
        StepAnchor an = {(double)(TP(s, QA_T_ROOT_STATES, float) + 13 * e)[0], (double)(TP(s, QA_T_ROOT_STATES, float) + 13 * e)[1], 0.0, 0.0};
        for (int d = 0; d < c->decimation; ++d) { compute_torques(s, e, act, tau, torg); phys_substep_acc(s, e, tau, 1, &an); }
        obstacles_end_step(s, e);
    }
    return QA_OK;
}

/* the simulator part of the task-level reset_idx (tsc/legged_gym/envs/base/legged_robot.py:348-410, 796-884) */
int qo_tsc_reset(qo_sim *s, const uint8_t *flags, const float *start_xy, const float *start_yaw, float yaw_range, float x_range, float y_range,
                 float pitch_range, int64_t step, void *stream);
int qo_tsc_reset_dev(qo_sim *s, const uint8_t *flags, const float *start_xy, const float *start_yaw, float yaw_range, float x_range, float y_range,
                     float pitch_range, const int64_t *step_dev, void *stream) {
    if (!step_dev) return QA_E_ARG;
    return qo_tsc_reset(s, flags, start_xy, start_yaw, yaw_range, x_range, y_range, pitch_range, *step_dev, stream);
}
int qo_tsc_reset(qo_sim *s, const uint8_t *flags, const float *start_xy, const float *start_yaw, float yaw_range, float x_range, float y_range,
                 float pitch_range, int64_t step, void *stream) {
    (void)stream;
    if (!s || !flags || !start_xy || !start_yaw) return QA_E_ARG;
    const qa_config *c = &s->cfg;
    for (int e = 0; e < c->num_envs; ++e) {
        memset(TP(s, QA_T_LAST_ROOT_VEL, float) + 6 * e, 0, 24);
        if (!flags[e]) continue;
        float u[4]; rng4(s, e, step, RS_RESET, 0, u);
        float yaw = start_yaw[e] + yaw_range * (2.0f * u[0] - 1.0f), pitch = pitch_range * (2.0f * u[3] - 1.0f);
        float *rt = TP(s, QA_T_ROOT_STATES, float) + 13 * e, *dof = TP(s, QA_T_DOF_STATE, float) + 24 * e;
        rt[0] = c->init_pos[0] + start_xy[2 * e] + x_range * (u[1] - 1.0f);
        rt[1] = c->init_pos[1] + start_xy[2 * e + 1] + y_range * (2.0f * u[2] - 1.0f);
        rt[2] = c->init_pos[2];
        float sy = sinf(0.5f * yaw), cy = cosf(0.5f * yaw), sp = sinf(0.5f * pitch), cp = cosf(0.5f * pitch);
        rt[3] = -sy * sp; rt[4] = cy * sp; rt[5] = sy * cp; rt[6] = cy * cp;
        for (int i = 7; i < 13; ++i) rt[i] = 0.0f;
        for (int j = 0; j < 12; ++j) { dof[2 * j] = c->default_dof_pos[j]; dof[2 * j + 1] = 0.0f; }
        memset(TP(s, QA_T_LAST_ACTIONS, float) + 12 * e, 0, 48); memset(TP(s, QA_T_LAST_DOF_VEL, float) + 12 * e, 0, 48);
        memset(TP(s, QA_T_LAST_TORQUES_ORG, float) + 12 * e, 0, 48); memset(TP(s, QA_T_FOOT_IMPULSE, float) + 12 * e, 0, 48);
        memset(TP(s, QA_T_ACTION_HISTORY, float) + 96 * e, 0, 96 * 4);
        TP(s, QA_T_RESET, int64_t)[e] = 1;
    }
    return QA_OK;
}

int qo_reset_all(qo_sim *s, int64_t step, void *stream) {
    (void)stream;
    if (!s) return QA_E_ARG;
    memset(TP(s, QA_T_EPISODE_STATS, float) + 16 * (step & 1), 0, 64);
    for (int e = 0; e < s->cfg.num_envs; ++e) reset_env(s, e, step, (int)(step & 1), 0);   /* a full reset reports no episode stats */
    return QA_OK;
}

int qo_env_step(qo_sim *s, const float *actions, int32_t delay_steps, int64_t step, void *stream) {
    (void)stream;
    if (!s || !actions || delay_steps < 0 || delay_steps >= QA_ACTION_BUF_LEN) return QA_E_ARG;
    const qa_config *c = &s->cfg;
    /* EPISODE_STATS[step & 1] accumulates this step; the bin of the NEXT step is cleared here (the HIP kernel
     * cannot clear its own bin without a grid-wide barrier, so this is the ABI's rule for both sides) */
    memset(TP(s, QA_T_EPISODE_STATS, float) + 16 * ((step + 1) & 1), 0, 64);
#pragma omp parallel for schedule(static)
    for (int e = 0; e < c->num_envs; ++e) {
        /* legged_robot.py:84-98 */
        float *ah = TP(s, QA_T_ACTION_HISTORY, float) + 96 * e;
        memmove(ah, ah + 12, 7 * 12 * 4);
        memcpy(ah + 7 * 12, actions + 12 * e, 48);
        const float *src = ah + 12 * (QA_ACTION_BUF_LEN - 1 - delay_steps);
        float *act = TP(s, QA_T_ACTIONS, float) + 12 * e;
        float clipa = c->clip_actions / c->action_scale;
        for (int j = 0; j < 12; ++j) act[j] = clipf(src[j], -clipa, clipa);
        float *tau = TP(s, QA_T_TORQUES, float) + 12 * e, *torg = TP(s, QA_T_TORQUES_ORG, float) + 12 * e;
        obstacles_begin_step(s, e);
        StepAnchor an = {(double)(TP(s, QA_T_ROOT_STATES, float) + 13 * e)[0], (double)(TP(s, QA_T_ROOT_STATES, float) + 13 * e)[1], 0.0, 0.0};
        for (int d = 0; d < c->decimation; ++d) {     /* :101-106 */
            compute_torques(s, e, act, tau, torg);
            phys_substep_acc(s, e, tau, 1, &an);
        }
        obstacles_end_step(s, e);
        float tmp[QA_NUM_OBS_DISC];
        post_physics(s, e, step, tmp);
    }
    return QA_OK;
}

int qo_env_step_dev(qo_sim *s, const float *actions, int32_t delay_steps, int64_t *step_counter, void *stream) {
    if (!step_counter) return QA_E_ARG;
    int rc = qo_env_step(s, actions, delay_steps, *step_counter, stream);
    if (rc == QA_OK) *step_counter += 1;
    return rc;
}

/* GAE: rollout_storage.py:97-111 (fp32 like the reference; mean/std accumulated in double) */
int qo_gae(const float *rewards, const float *values, const uint8_t *dones, const float *last_values,
           float *returns, float *advantages, int32_t T, int32_t N, float gamma, float lam, int32_t normalize,
           void *scratch, void *stream) {
    (void)scratch; (void)stream;
    if (!rewards || !values || !dones || !last_values || !returns || !advantages || T <= 0 || N <= 0) return QA_E_ARG;
    for (int e = 0; e < N; ++e) {
        float adv = 0;
        for (int t = T - 1; t >= 0; --t) {
            float nv = t == T - 1 ? last_values[e] : values[(int64_t)(t + 1) * N + e];
            float nt = 1.0f - (float)dones[(int64_t)t * N + e];
            float delta = rewards[(int64_t)t * N + e] + nt * gamma * nv - values[(int64_t)t * N + e];
            adv = delta + nt * gamma * lam * adv;
            returns[(int64_t)t * N + e] = adv + values[(int64_t)t * N + e];
        }
    }
    int64_t n = (int64_t)T * N; double sum = 0, sq = 0;
    for (int64_t i = 0; i < n; ++i) { advantages[i] = returns[i] - values[i]; sum += advantages[i]; }
    if (normalize) {
        double mean = sum / (double)n;
        for (int64_t i = 0; i < n; ++i) { double d = advantages[i] - mean; sq += d * d; }
        double sd = sqrt(sq / (double)(n - 1));
        for (int64_t i = 0; i < n; ++i) advantages[i] = (float)(((double)advantages[i] - mean) / (sd + 1e-8));
    }
    return QA_OK;
}

/* PPO minibatch objective + gradient (gail.py:333-345, 363-403; torch.distributions.Normal log_prob / entropy;
 * torch.max / clamp sub-gradients: ties split evenly, clamp passes the gradient on the closed interval).
 * Double arithmetic; same signature as qa_ppo_loss, host pointers. */
int qo_ppo_loss(const float *mu, const float *std, const float *value, const float *actions, const float *old_logp,
                const float *old_mu, const float *old_sigma, const float *advantages, const float *returns,
                const float *target_values, int64_t B, int32_t D, float clip_f, float c_surr, float c_value,
                float c_bound, float c_entropy, int32_t clipped_value, float *dmu, float *dstd, float *dvalue, float *out,
                void *scratch, int64_t scratch_bytes, void *stream) {
    (void)scratch; (void)scratch_bytes; (void)stream;
    if (!mu || !std || !value || !actions || !old_logp || !old_mu || !old_sigma || !advantages || !returns || !target_values ||
        !dmu || !dstd || !dvalue || !out || B <= 0 || D != 12) return QA_E_ARG;
    const double HALF_LOG_2PI = 0.91893853320467274178, clip = clip_f, invB = 1.0 / (double)B;
    double sums[5] = {0, 0, 0, 0, 0}, gs[12] = {0};
    for (int64_t i = 0; i < B; ++i) {
        double logp = 0, ent = 0, kl = 0, bl = 0, dl_dmu[12], dl_dsd[12], dbl[12];
        for (int j = 0; j < 12; ++j) {
            double s = std[j], m = mu[i * 12 + j], d = (double)actions[i * 12 + j] - m, os = old_sigma[i * 12 + j], dm = (double)old_mu[i * 12 + j] - m;
            logp += -(d * d) / (2 * s * s) - log(s) - HALF_LOG_2PI;
            ent += 0.5 + HALF_LOG_2PI + log(s);
            kl += log(s / os + 1.0e-5) + (os * os + dm * dm) / (2 * s * s) - 0.5;
            dl_dmu[j] = d / (s * s); dl_dsd[j] = d * d / (s * s * s) - 1.0 / s;
            double lo = m + 1.0 < 0 ? m + 1.0 : 0.0, hi = m - 1.0 > 0 ? m - 1.0 : 0.0;
            bl += lo * lo + hi * hi; dbl[j] = 2 * lo + 2 * hi;
        }
        double A = advantages[i], ratio = exp(logp - (double)old_logp[i]);
        double rc = ratio < 1 - clip ? 1 - clip : (ratio > 1 + clip ? 1 + clip : ratio);
        double s1 = -A * ratio, s2 = -A * rc, pass = (ratio >= 1 - clip && ratio <= 1 + clip) ? 1.0 : 0.0;
        double surr = s1 > s2 ? s1 : s2;
        double dsr = s1 > s2 ? -A : (s1 == s2 ? 0.5 * (-A) + 0.5 * (-A) * pass : (-A) * pass);
        double dsl = dsr * ratio;
        double v = value[i], R = returns[i], tv = target_values[i], vl, dvl;
        if (clipped_value) {
            double dv = v - tv, dvc = dv < -clip ? -clip : (dv > clip ? clip : dv), vc = tv + dvc, pv = (dv >= -clip && dv <= clip) ? 1.0 : 0.0;
            double l1 = (v - R) * (v - R), l2 = (vc - R) * (vc - R);
            vl = l1 > l2 ? l1 : l2;
            dvl = l1 > l2 ? 2 * (v - R) : (l1 == l2 ? (v - R) + (vc - R) * pv : 2 * (vc - R) * pv);
        } else { vl = (R - v) * (R - v); dvl = 2 * (v - R); }
        for (int j = 0; j < 12; ++j) {
            dmu[i * 12 + j] = (float)(invB * (c_surr * dsl * dl_dmu[j] + c_bound * dbl[j]));
            gs[j] += c_surr * dsl * dl_dsd[j] - (double)c_entropy / (double)std[j];
        }
        dvalue[i] = (float)(invB * c_value * dvl);
        sums[0] += surr; sums[1] += vl; sums[2] += bl; sums[3] += ent; sums[4] += kl;
    }
    for (int k = 0; k < 5; ++k) out[1 + k] = (float)(sums[k] * invB);
    out[0] = (float)((c_surr * sums[0] + c_value * sums[1] + c_bound * sums[2] - c_entropy * sums[3]) * invB);
    out[6] = 0; out[7] = 0;
    for (int j = 0; j < 12; ++j) dstd[j] = (float)(gs[j] * invB);
    return QA_OK;
}

int64_t qo_ppo_loss_scratch_bytes(int64_t B) { return B <= 0 ? -1 : 0; }

/* torch's elu_backward with is_result=True (alpha e^x = y + alpha for y <= 0) + the column sum of the bias gradient */
int64_t qo_elu_backward_bias_scratch_bytes(int64_t rows, int32_t cols) { return (rows <= 0 || cols <= 0) ? -1 : 0; }
int qo_elu_backward_bias(const float *grad_out, const float *out, float *grad_in, float *grad_bias, int64_t rows, int32_t cols,
                         float alpha, void *scratch, int64_t scratch_bytes, void *stream) {
    (void)scratch; (void)scratch_bytes; (void)stream;
    if (!grad_out || !out || !grad_in || rows <= 0 || cols <= 0) return QA_E_ARG;
    for (int c = 0; c < cols; ++c) {
        double acc = 0;
        for (int64_t r = 0; r < rows; ++r) {
            float y = out[r * cols + c], g = grad_out[r * cols + c] * (y > 0.0f ? 1.0f : y + alpha);
            grad_in[r * cols + c] = g; acc += g;
        }
        if (grad_bias) grad_bias[c] = (float)acc;           /* NULL (r5): the caller adds the parts itself; this twin keeps none */
    }
    return QA_OK;
}

/* CPU twin of qa_hybrid_ppo_loss (tsc/rsl_rl/algorithms/ppo.py:222-262), double precision, any num_d <= 8 / num_c <= 32 */
int64_t qo_hybrid_ppo_loss_scratch_bytes(int64_t B) { (void)B; return 16; }
static void hyb_surr(double logp, double ologp, double adv, double clip, double *surr, double *dsurr_dlogp) {
    double ratio = exp(logp - ologp), rc = ratio < 1 - clip ? 1 - clip : (ratio > 1 + clip ? 1 + clip : ratio);
    double s1 = -adv * ratio, s2 = -adv * rc;
    *surr = s1 > s2 ? s1 : s2;
    int inside = ratio >= 1 - clip && ratio <= 1 + clip;
    double d = s1 > s2 ? -adv : (s1 == s2 ? (inside ? -adv : -0.5 * adv) : 0.0);
    *dsurr_dlogp = d * ratio;
}
int qo_hybrid_ppo_loss(const float *logits, const float *mean, const float *std, const float *value, const float *actions, const float *old_logp_d,
                       const float *old_logp_c, const float *old_mu, const float *old_sigma, const float *advantages, const float *returns,
                       const float *target_values, int64_t B, int32_t ND, int32_t NC, float clip_f, float c_value, float c_entropy,
                       int32_t clipped_value, float *dlogits, float *dmean, float *dstd, float *dvalue, float *out, void *scratch,
                       int64_t scratch_bytes, void *stream) {
    (void)scratch; (void)scratch_bytes; (void)stream;
    if (!logits || !mean || !std || !value || !actions || B <= 0 || ND <= 0 || ND > 8 || NC <= 0 || NC > 32) return QA_E_ARG;
    const double HALF_LOG_2PI = 0.91893853320467274178, EPS = 1.1920928955078125e-07, clip = clip_f, invB = 1.0 / (double)B;
    double S[5] = {0, 0, 0, 0, 0}, dsd[32] = {0};
    for (int64_t r = 0; r < B; ++r) {
        double z[8], p[8], lp[8], fl[8], zmax = -1e300, den = 0, Hd = 0, sum_pf = 0, plp = 0;
        for (int k = 0; k < ND; ++k) { z[k] = logits[r * ND + k]; if (z[k] > zmax) zmax = z[k]; }
        for (int k = 0; k < ND; ++k) { p[k] = exp(z[k] - zmax); den += p[k]; }
        for (int k = 0; k < ND; ++k) {
            p[k] /= den; fl[k] = (p[k] > EPS && p[k] < 1 - EPS) ? 1 : 0;
            double c = p[k] < EPS ? EPS : (p[k] > 1 - EPS ? 1 - EPS : p[k]);
            lp[k] = log(c); Hd -= p[k] * lp[k]; sum_pf += p[k] * fl[k]; plp += p[k] * lp[k];
        }
        const float *act = actions + r * (1 + NC);
        int ad = (int)act[0]; if (ad < 0) ad = 0; if (ad >= ND) ad = ND - 1;
        double adv = advantages[r], surr_d, dd, surr_c, dc;
        hyb_surr(lp[ad], old_logp_d[r], adv, clip, &surr_d, &dd);
        double logp = 0, ent_c = 0, kl = 0, dmu[32], dls[32];
        for (int j = 0; j < NC; ++j) {
            double s = std[j], mu = mean[r * NC + j], d = act[1 + j] - mu, var = s * s, ls = log(s), osd = old_sigma[r * NC + j], dm = old_mu[r * NC + j] - mu;
            logp += -(d * d) / (2 * var) - ls - HALF_LOG_2PI; ent_c += 0.5 + HALF_LOG_2PI + ls;
            kl += log(s / osd + 1e-5) + (osd * osd + dm * dm) / (2 * var) - 0.5;
            dmu[j] = d / var; dls[j] = d * d / (s * s * s) - 1 / s;
        }
        ent_c /= NC;
        hyb_surr(logp, old_logp_c[r], adv, clip, &surr_c, &dc);
        double v = value[r], ret = returns[r], tv = target_values[r], vl, dvl;
        if (clipped_value) {
            double dv = v - tv, dvc = dv < -clip ? -clip : (dv > clip ? clip : dv), vc = tv + dvc, l1 = (v - ret) * (v - ret), l2 = (vc - ret) * (vc - ret);
            double pass = (dv >= -clip && dv <= clip) ? 1 : 0;
            vl = l1 > l2 ? l1 : l2;
            dvl = l1 > l2 ? 2 * (v - ret) : (l1 == l2 ? (v - ret) + (vc - ret) * pass : 2 * (vc - ret) * pass);
        } else { vl = (ret - v) * (ret - v); dvl = 2 * (v - ret); }
        for (int k = 0; k < ND; ++k) {
            double dlogpa = fl[ad] * ((k == ad ? 1.0 : 0.0) - p[k]);
            double dH = -p[k] * lp[k] + p[k] * plp - p[k] * fl[k] + p[k] * sum_pf;
            dlogits[r * ND + k] = (float)(invB * (dd * dlogpa - c_entropy * dH));
        }
        for (int j = 0; j < NC; ++j) { dmean[r * NC + j] = (float)(invB * dc * dmu[j]); dsd[j] += dc * dls[j] - c_entropy / (std[j] * NC); }
        dvalue[r] = (float)(invB * c_value * dvl);
        S[0] += surr_d; S[1] += surr_c; S[2] += vl; S[3] += Hd + ent_c; S[4] += kl;
    }
    out[0] = (float)((S[0] + S[1] + c_value * S[2] - c_entropy * S[3]) * invB);
    out[1] = (float)((S[0] + S[1]) * invB); out[2] = (float)(S[2] * invB); out[3] = (float)(S[3] * invB); out[4] = (float)(S[4] * invB);
    out[5] = (float)(S[0] * invB); out[6] = (float)(S[1] * invB); out[7] = 0.f;
    for (int j = 0; j < NC; ++j) dstd[j] = (float)(dsd[j] * invB);
    return QA_OK;
}

/* CPU twin of qa_narrow_wgrad: F.linear's weight / bias gradient for a layer with few outputs, plain double loops */
int64_t qo_narrow_wgrad_scratch_bytes(int64_t rows, int32_t out_features, int32_t in_features) { (void)rows; (void)out_features; (void)in_features; return 16; }
int qo_narrow_wgrad(const float *grad_out, const float *x, int64_t rows, int32_t O, int32_t K, float *grad_weight, float *grad_bias,
                    void *scratch, int64_t scratch_bytes, void *stream) {
    (void)scratch; (void)scratch_bytes; (void)stream;
    if (!grad_out || !x || !grad_weight || !grad_bias || rows <= 0 || O <= 0 || O > QA_NARROW_MAX_OUT || K <= 0) return QA_E_ARG;
    for (int o = 0; o < O; ++o) {
        double b = 0;
        for (int64_t r = 0; r < rows; ++r) b += grad_out[r * O + o];
        grad_bias[o] = (float)b;
        for (int k = 0; k < K; ++k) {
            double acc = 0;
            for (int64_t r = 0; r < rows; ++r) acc += (double)grad_out[r * O + o] * x[r * K + k];
            grad_weight[(int64_t)o * K + k] = (float)acc;
        }
    }
    return QA_OK;
}

/* CPU twins of the learner's dense-layer kernels (csrc/qa_gemm.hip): one Linear(+activation) layer of the reference's MLPs
 * (bbc/rsl_rl/modules/actor_critic.py:92-139, estimator.py:12-36) forward, and the two halves of its backward as autograd
 * evaluates them for gail.py:328-413 -- plain double-precision loops, OpenMP over rows / outputs.  act: 0 none, 1 ELU(alpha), 2 ReLU. */
int qo_linear_forward(const float *x, int64_t ldx, const float *weight, int64_t ldw, const float *bias, float *y, int64_t ldy, int64_t rows,
                      int32_t in_features, int32_t out_features, int32_t act, float alpha, void *stream) {
    (void)stream;
    if (!x || !weight || !y || rows <= 0 || in_features <= 0 || out_features <= 0 || ldx < in_features || ldw < in_features || ldy < out_features ||
        act < 0 || act > 2) return QA_E_ARG;
#pragma omp parallel for
    for (int64_t r = 0; r < rows; ++r)
        for (int o = 0; o < out_features; ++o) {
            double acc = bias ? bias[o] : 0.0;
            for (int k = 0; k < in_features; ++k) acc += (double)x[r * ldx + k] * weight[(int64_t)o * ldw + k];
            if (act == 1) acc = acc > 0 ? acc : alpha * (exp(acc) - 1.0);
            else if (act == 2) acc = acc > 0 ? acc : 0.0;
            y[r * ldy + o] = (float)acc;
        }
    return QA_OK;
}
/* grad_in = (grad_out W) * act'(y_prev), the derivative taken from the previous layer's activation OUTPUT (ELU: y > 0 ? 1 : y + alpha) */
int qo_linear_backward_input(const float *grad_out, int64_t ldg, const float *weight, int64_t ldw, const float *y_prev, int64_t ldyp,
                             float *grad_in, int64_t ldgi, int64_t rows, int32_t in_features, int32_t out_features, int32_t act_prev, float alpha,
                             void *stream) {
    (void)stream;
    if (!grad_out || !weight || !grad_in || rows <= 0 || in_features <= 0 || out_features <= 0 || ldg < out_features || ldw < in_features ||
        ldgi < in_features || act_prev < 0 || act_prev > 2 || (act_prev != 0 && (!y_prev || ldyp < in_features))) return QA_E_ARG;
#pragma omp parallel for
    for (int64_t r = 0; r < rows; ++r)
        for (int k = 0; k < in_features; ++k) {
            double acc = 0;
            for (int o = 0; o < out_features; ++o) acc += (double)grad_out[r * ldg + o] * weight[(int64_t)o * ldw + k];
            if (act_prev) {
                float yv = y_prev[r * ldyp + k];
                acc *= act_prev == 1 ? (yv > 0.0f ? 1.0 : (double)yv + alpha) : (yv > 0.0f ? 1.0 : 0.0);
            }
            grad_in[r * ldgi + k] = (float)acc;
        }
    return QA_OK;
}
int64_t qo_linear_backward_weight_scratch_bytes(int64_t rows, int32_t in_features, int32_t out_features) {
    return (rows <= 0 || in_features <= 0 || out_features <= 0) ? 0 : 16 + 4 * ((int64_t)in_features * out_features + out_features + 8);
}
/* twin of qa_linear_backward_weight_layout (ABI 17): the twin's "parts" are ONE finished weight gradient and ONE finished bias gradient */
int qo_linear_backward_weight_layout(int64_t rows, int32_t in_features, int32_t out_features, int64_t layout[5]) {
    if (!layout || rows <= 0 || in_features <= 0 || out_features <= 0) return QA_E_ARG;
    const int64_t nw = ((int64_t)in_features * out_features + 3) & ~(int64_t)3;
    layout[0] = 1; layout[1] = nw; layout[2] = 1; layout[3] = (out_features + 3) & ~3; layout[4] = nw;
    return QA_OK;
}
int qo_linear_backward_weight(const float *grad_out, int64_t ldg, const float *x, int64_t ldx, float *grad_weight, float *grad_bias, int64_t rows,
                              int32_t in_features, int32_t out_features, void *scratch, int64_t scratch_bytes, void *stream) {
    (void)stream;
    if (!grad_weight && !grad_bias) {          /* ABI 17: in parts, inside `scratch` (qo_linear_backward_weight_layout) */
        int64_t lay[5];
        if (!scratch || qo_linear_backward_weight_layout(rows, in_features, out_features, lay) != QA_OK ||
            scratch_bytes < qo_linear_backward_weight_scratch_bytes(rows, in_features, out_features)) return QA_E_ARG;
        grad_weight = (float *)scratch; grad_bias = (float *)scratch + lay[4];
    }
    if (!grad_out || !x || !grad_weight || !grad_bias || rows <= 0 || in_features <= 0 || out_features <= 0 || ldg < out_features || ldx < in_features)
        return QA_E_ARG;
#pragma omp parallel for
    for (int o = 0; o < out_features; ++o) {
        double b = 0;
        for (int64_t r = 0; r < rows; ++r) b += grad_out[r * ldg + o];
        grad_bias[o] = (float)b;
        for (int k = 0; k < in_features; ++k) {
            double acc = 0;
            for (int64_t r = 0; r < rows; ++r) acc += (double)grad_out[r * ldg + o] * x[r * ldx + k];
            grad_weight[(int64_t)o * in_features + k] = (float)acc;
        }
    }
    return QA_OK;
}

/* twins of the batch entry points (ABI 17): the same products, one after the other */
int64_t qo_linear_backward_weight_batch_scratch_bytes(int64_t rows, int32_t in_features, int32_t out_features) {
    return qo_linear_backward_weight_scratch_bytes(rows, in_features, out_features);
}
int qo_linear_backward_weight_batch_layout(int64_t rows, int32_t in_features, int32_t out_features, int64_t layout[5]) {
    return qo_linear_backward_weight_layout(rows, in_features, out_features, layout);
}
int qo_linear_backward_weight_batch(const qa_wgrad_desc *descs, int32_t count, void *stream) {
    if (!descs || count <= 0 || count > 32) return QA_E_ARG;
    for (int i = 0; i < count; ++i) {
        const qa_wgrad_desc *d = &descs[i];
        int rc = qo_linear_backward_weight(d->grad_out, d->ldg, d->x, d->ldx, d->grad_weight, d->grad_bias, d->rows, d->in_features, d->out_features, d->scratch,
                                           d->scratch_bytes, stream);
        if (rc != QA_OK) return rc;
    }
    return QA_OK;
}

int64_t qo_linear_forward_split_scratch_bytes(int64_t rows, int32_t in_features, int32_t out_features) {
    (void)in_features; return (rows > 0 && out_features > 0) ? 16 : 0;
}
/* twin of qa_linear_forward_split: the same layer as qo_linear_forward (the split only changes the summation order on the device) */
int qo_linear_forward_split(const float *x, int64_t ldx, const float *weight, int64_t ldw, const float *bias, float *y, int64_t ldy, int64_t rows,
                            int32_t in_features, int32_t out_features, int32_t act, float alpha, void *scratch, int64_t scratch_bytes, void *stream) {
    (void)scratch; (void)scratch_bytes;
    if (out_features % 4 || ldy % 4) return QA_E_ARG;
    return qo_linear_forward(x, ldx, weight, ldw, bias, y, ldy, rows, in_features, out_features, act, alpha, stream);
}

/* twin of qa_disc_sample_prepare: the generators' row selection (bbc/rsl_rl/storage/replay_buffer.py:38-47, motion_loader feed_forward_generator_lb / _ulb)
 * from index tables + qo_disc_prepare's arithmetic */
int qo_disc_sample_prepare(const qa_disc_sample_io *io, int32_t dim, int32_t c_dim, const float *task_mask, const float *frame_mult,
                           const float *task_weight_dev, const double *mean, const double *var, float epsilon, float clip, float *out, void *stream) {
    (void)stream;
    if (!io || !task_mask || !frame_mult || !out || dim <= 0 || c_dim < 0 || ((mean == NULL) != (var == NULL)) || !io->block_dev) return QA_E_ARG;
    const int64_t blk = io->block_dev[0];
    int64_t o = 0;
    for (int b = 0; b < 3; ++b) {
        if (!io->src[b] || !io->index[b] || io->rows[b] <= 0) return QA_E_ARG;
        for (int64_t r = 0; r < io->rows[b]; ++r) {
            const int64_t row = io->index[b][blk * io->rows[b] + r];
            for (int c = 0; c < dim; ++c) {
                float x = io->src[b][row * dim + c];
                if (task_weight_dev && task_mask[c] != 0.0f) x *= task_weight_dev[0];
                x *= frame_mult[c];
                if (mean) { float m = (float)mean[c], sd = sqrtf((float)(var[c] + (double)epsilon)); x = (x - m) / sd; x = x < -clip ? -clip : (x > clip ? clip : x); }
                out[o++] = x;
            }
        }
    }
    if (io->eps_src)
        for (int64_t j = 0; j < io->rows[1]; ++j) {
            const int64_t row = io->index[1][blk * io->rows[1] + j];
            io->eps_out[j] = io->eps_src[row];
            for (int k = 0; k < c_dim; ++k) io->c_out[j * c_dim + k] = io->c_src[row * c_dim + k];
        }
    if (io->label_src)
        for (int64_t j = 0; j < io->rows[0]; ++j) io->label_out[j] = io->label_src[blk * io->rows[0] + j];
    return QA_OK;
}

int64_t qo_disc_step_tail_scratch_bytes(void) { return 16; }
/* twin of qa_disc_step_tail (bbc/rsl_rl/algorithms/gail.py:486-504, 520-533): double sums */
int qo_disc_step_tail(const float *head_stats, const float *input_grad, int64_t grad_rows, int32_t grad_cols, const float *const *weights,
                      const int64_t *weight_counts, int32_t num_weights, float *out, float *acc, int64_t *step_counter, float *prior, int32_t prior_dim,
                      float prior_soft_coef, void *scratch, int64_t scratch_bytes, void *stream) {
    (void)scratch; (void)scratch_bytes; (void)stream;
    if (!head_stats || !input_grad || grad_rows <= 0 || grad_cols <= 0 || !weights || !weight_counts || num_weights <= 0 || num_weights > 7 || !out ||
        (prior && (prior_dim <= 0 || prior_dim > 5))) return QA_E_ARG;
    double gp = 0.0, wd = 0.0, last = 0.0;
    for (int64_t i = 0; i < grad_rows * grad_cols; ++i) gp += (double)input_grad[i] * input_grad[i];
    for (int t = 0; t < num_weights; ++t) {
        double s = 0.0;
        for (int64_t i = 0; i < weight_counts[t]; ++i) s += (double)weights[t][i] * weights[t][i];
        wd += s; last = s;
    }
    float o[11] = {head_stats[1], head_stats[2], head_stats[3], head_stats[4], (float)(gp * (double)(1.0f / (float)grad_rows)), (float)last, (float)wd,
                   head_stats[5], head_stats[6], head_stats[7], head_stats[8]};
    for (int k = 0; k < 11; ++k) { out[k] = o[k]; if (acc) acc[k] += o[k]; }
    if (step_counter) step_counter[0] += 1;
    if (prior) for (int k = 0; k < prior_dim; ++k) prior[k] = fmaf(prior_soft_coef, head_stats[9 + k], prior[k] * (1.0f - prior_soft_coef));
    return QA_OK;
}

int qo_slab_sum(const float *slabs, int64_t slab_stride, int32_t num_slabs, int64_t n, float *out, void *stream) {
    (void)stream;
    if (!slabs || !out || num_slabs <= 0 || n <= 0 || slab_stride < n) return QA_E_ARG;
    for (int64_t i = 0; i < n; ++i) {
        float s = 0.0f;                                 /* fp32, slab order: what the kernel does */
        for (int z = 0; z < num_slabs; ++z) s += slabs[z * slab_stride + i];
        out[i] = s;
    }
    return QA_OK;
}

/* CPU twins of the depth encoder's image stem (csrc/qa_conv.hip, csrc/qa_gemm.hip), restating
 * tsc/rsl_rl/modules/depth_backbone.py:63-75 (Conv2d(1, 32, 5) -> MaxPool2d(2, 2) -> ELU -> Conv2d(32, 64, 3) -> ELU) and its gradients as
 * plain loops over channels-last tensors; double accumulation (the kernels accumulate in fp32 in a different order: the tests compare at a
 * tolerance).  Pinned against torch's conv2d / max_pool2d / elu forward and autograd by tests/test_depth_stem.py. */
static float act_fwd(int act, float alpha, float v) { return act == 1 ? (v > 0.f ? v : alpha * (expf(v) - 1.f)) : (act == 2 ? (v > 0.f ? v : 0.f) : v); }
static float act_deriv_from_output(int act, float alpha, float y) { return act == 1 ? (y > 0.f ? 1.f : y + alpha) : (act == 2 ? (y > 0.f ? 1.f : 0.f) : 1.f); }

int qo_depth_stem_forward(const float *images, const float *weight, const float *bias, float *y, uint8_t *argmax, int64_t n_img, int32_t ih, int32_t iw,
                          float alpha, void *stream) {
    (void)stream;
    if (!images || !weight || !bias || !y || !argmax || n_img <= 0 || ih < 6 || iw < 6 || iw > 126) return QA_E_ARG;
    const int ph = (ih - 4) / 2, pw = (iw - 4) / 2;
    for (int64_t n = 0; n < n_img; ++n)
        for (int py = 0; py < ph; ++py)
            for (int px = 0; px < pw; ++px)
                for (int c = 0; c < 32; ++c) {
                    float best = 0.f; int arg = 0;
                    for (int d = 0; d < 4; ++d) {
                        const int cy = 2 * py + (d >> 1), cx = 2 * px + (d & 1);
                        double acc = 0.0;
                        for (int ky = 0; ky < 5; ++ky)
                            for (int kx = 0; kx < 5; ++kx) acc += (double)weight[c * 25 + ky * 5 + kx] * images[(n * ih + cy + ky) * iw + cx + kx];
                        const float v = (float)acc + bias[c];
                        if (d == 0 || v > best) { best = v; arg = d; }          /* first maximum in row-major window order (PyTorch) */
                    }
                    const int64_t o = ((n * ph + py) * pw + px) * 32 + c;
                    y[o] = act_fwd(1, alpha, best);
                    argmax[o] = (uint8_t)arg;
                }
    return QA_OK;
}

int64_t qo_depth_stem_backward_scratch_bytes(void) { return 16; }

int qo_depth_stem_backward(const float *images, const uint8_t *argmax, const float *grad_pre, float *grad_wb, int64_t n_img, int32_t ih, int32_t iw,
                           void *scratch, int64_t scratch_bytes, void *stream) {
    (void)stream; (void)scratch; (void)scratch_bytes;
    if (!images || !argmax || !grad_pre || !grad_wb || n_img <= 0 || ih < 6 || iw < 6 || iw > 126) return QA_E_ARG;
    const int ph = (ih - 4) / 2, pw = (iw - 4) / 2;
    double acc[832];
    for (int i = 0; i < 832; ++i) acc[i] = 0.0;
    for (int64_t n = 0; n < n_img; ++n)
        for (int py = 0; py < ph; ++py)
            for (int px = 0; px < pw; ++px)
                for (int c = 0; c < 32; ++c) {
                    const int64_t o = ((n * ph + py) * pw + px) * 32 + c;
                    const double g = grad_pre[o];
                    const int cy = 2 * py + (argmax[o] >> 1), cx = 2 * px + (argmax[o] & 1);
                    acc[800 + c] += g;
                    for (int ky = 0; ky < 5; ++ky)
                        for (int kx = 0; kx < 5; ++kx) acc[c * 25 + ky * 5 + kx] += g * images[(n * ih + cy + ky) * iw + cx + kx];
                }
    for (int i = 0; i < 832; ++i) grad_wb[i] = (float)acc[i];
    return QA_OK;
}

static int conv_args_ok(int64_t n_img, int ih, int iw, int cin, int kh, int kw, int cout) {
    const int oh = ih - kh + 1, ow = iw - kw + 1;
    return n_img > 0 && cin >= 16 && !(cin & (cin - 1)) && kh >= 1 && kw >= 2 && oh >= 1 && ow >= 2 && (int64_t)oh * ow > 256 && cout > 0 && cout % 4 == 0;
}
/* mode 0: act(conv + bias); mode 1: conv * act'(deriv_of) */
static void conv_nhwc(int mode, const float *x, const float *w, const float *bias, const float *deriv_of, float *y, int64_t n_img, int ih, int iw, int cin,
                      int kh, int kw, int cout, int act, float alpha) {
    const int oh = ih - kh + 1, ow = iw - kw + 1;
    for (int64_t n = 0; n < n_img; ++n)
        for (int oy = 0; oy < oh; ++oy)
            for (int ox = 0; ox < ow; ++ox)
                for (int o = 0; o < cout; ++o) {
                    double acc = 0.0;
                    for (int ky = 0; ky < kh; ++ky)
                        for (int kx = 0; kx < kw; ++kx) {
                            const float *xp = x + ((n * ih + oy + ky) * iw + ox + kx) * cin, *wp = w + ((int64_t)(o * kh + ky) * kw + kx) * cin;
                            for (int c = 0; c < cin; ++c) acc += (double)xp[c] * wp[c];
                        }
                    const int64_t i = ((n * oh + oy) * ow + ox) * cout + o;
                    y[i] = mode == 0 ? act_fwd(act, alpha, (float)acc + (bias ? bias[o] : 0.f))
                                     : (float)acc * ((deriv_of && act) ? act_deriv_from_output(act, alpha, deriv_of[i]) : 1.f);
                }
}
int qo_conv_nhwc_forward(const float *x, const float *weight, const float *bias, float *y, int64_t n_img, int32_t ih, int32_t iw, int32_t cin,
                         int32_t kh, int32_t kw, int32_t cout, int32_t act, float alpha, void *stream) {
    (void)stream;
    if (!x || !weight || !y || act < 0 || act > 2 || !conv_args_ok(n_img, ih, iw, cin, kh, kw, cout)) return QA_E_ARG;
    conv_nhwc(0, x, weight, bias, NULL, y, n_img, ih, iw, cin, kh, kw, cout, act, alpha);
    return QA_OK;
}
int qo_conv_nhwc_backward_input(const float *grad_padded, const float *weight_flipped, const float *x_act, float *grad_in, int64_t n_img, int32_t ihp,
                                int32_t iwp, int32_t cout, int32_t kh, int32_t kw, int32_t cin, int32_t act_prev, float alpha, void *stream) {
    (void)stream;
    if (!grad_padded || !weight_flipped || !grad_in || act_prev < 0 || act_prev > 2 || (act_prev && !x_act) || !conv_args_ok(n_img, ihp, iwp, cout, kh, kw, cin))
        return QA_E_ARG;
    conv_nhwc(1, grad_padded, weight_flipped, NULL, x_act, grad_in, n_img, ihp, iwp, cout, kh, kw, cin, act_prev, alpha);
    return QA_OK;
}
int64_t qo_conv_nhwc_backward_weight_scratch_bytes(int64_t n_img, int32_t ih, int32_t iw, int32_t cin, int32_t kh, int32_t kw, int32_t cout) {
    return conv_args_ok(n_img, ih, iw, cin, kh, kw, cout) ? (int64_t)kh * kw * cin * cout * 8 + cout * 8 : 0;
}
int qo_conv_nhwc_backward_weight(const float *x, const float *grad_out, float *grad_weight, float *grad_bias, int64_t n_img, int32_t ih, int32_t iw,
                                 int32_t cin, int32_t kh, int32_t kw, int32_t cout, void *scratch, int64_t scratch_bytes, void *stream) {
    (void)stream;
    if (!x || !grad_out || !grad_weight || !grad_bias || !scratch || !conv_args_ok(n_img, ih, iw, cin, kh, kw, cout) ||
        scratch_bytes < qo_conv_nhwc_backward_weight_scratch_bytes(n_img, ih, iw, cin, kh, kw, cout)) return QA_E_ARG;
    const int oh = ih - kh + 1, ow = iw - kw + 1, kred = kh * kw * cin;
    double *aw = (double *)scratch, *ab = aw + (int64_t)kred * cout;
    for (int64_t i = 0; i < (int64_t)kred * cout + cout; ++i) aw[i] = 0.0;
    for (int64_t n = 0; n < n_img; ++n)
        for (int oy = 0; oy < oh; ++oy)
            for (int ox = 0; ox < ow; ++ox) {
                const float *gp = grad_out + ((n * oh + oy) * ow + ox) * cout;
                for (int o = 0; o < cout; ++o) {
                    const double g = gp[o];
                    ab[o] += g;
                    for (int ky = 0; ky < kh; ++ky)
                        for (int kx = 0; kx < kw; ++kx) {
                            const float *xp = x + ((n * ih + oy + ky) * iw + ox + kx) * cin;
                            double *wp = aw + ((int64_t)(o * kh + ky) * kw + kx) * cin;
                            for (int c = 0; c < cin; ++c) wp[c] += g * xp[c];
                        }
                }
            }
    for (int64_t i = 0; i < (int64_t)kred * cout; ++i) grad_weight[i] = (float)aw[i];
    for (int o = 0; o < cout; ++o) grad_bias[o] = (float)ab[o];
    return QA_OK;
}
int qo_elu_backward_pad(const float *grad_out, const float *y, float *grad_pre, float *grad_pre_padded, int64_t n_img, int32_t oh, int32_t ow,
                        int32_t channels, int32_t pad, int32_t act, float alpha, void *stream) {
    (void)stream;
    if (!grad_out || !y || !grad_pre || !grad_pre_padded || n_img <= 0 || oh <= 0 || ow <= 0 || channels <= 0 || channels % 4 || pad < 0 || act < 0 || act > 2)
        return QA_E_ARG;
    const int ph = oh + 2 * pad, pw = ow + 2 * pad;
    for (int64_t n = 0; n < n_img; ++n)
        for (int yy = 0; yy < ph; ++yy)
            for (int xx = 0; xx < pw; ++xx)
                for (int c = 0; c < channels; ++c) {
                    const int iy = yy - pad, ix = xx - pad;
                    float v = 0.f;
                    if (iy >= 0 && iy < oh && ix >= 0 && ix < ow) {
                        const int64_t o = ((n * oh + iy) * ow + ix) * channels + c;
                        v = grad_out[o] * act_deriv_from_output(act, alpha, y[o]);
                        grad_pre[o] = v;
                    }
                    grad_pre_padded[((n * ph + yy) * pw + xx) * channels + c] = v;
                }
    return QA_OK;
}

/* CPU twin of qa_tsc_reset_stats (tsc/legged_gym/envs/base/legged_robot.py:382-384, 396-404): the kernel's summation order restated
 * (1024 strided partial sums, then a binary tree), fp32, so the two agree bit for bit */
static float tree1024(float *red) {
    for (int w = 512; w > 0; w >>= 1) for (int t = 0; t < w; ++t) red[t] += red[t + w];
    return red[0];
}
int qo_tsc_reset_stats(const uint8_t *reset_flags, const float *episode_sums, int64_t num_envs, int32_t num_terms, float max_episode_length_s,
                       float *episode_means, uint8_t *any_reset, void *stream) {
    (void)stream;
    if (!reset_flags || !episode_sums || !episode_means || !any_reset || num_envs <= 0 || num_terms <= 0 || !(max_episode_length_s > 0.f)) return QA_E_ARG;
    float red[1024];
    for (int t = 0; t < 1024; ++t) { float c = 0.f; for (int64_t e = t; e < num_envs; e += 1024) c += reset_flags[e] ? 1.f : 0.f; red[t] = c; }
    const float cnt = tree1024(red);
    any_reset[0] = cnt > 0.f ? 1 : 0;
    if (cnt <= 0.f) return QA_OK;
    const float inv = 1.0f / max_episode_length_s;
    for (int k = 0; k < num_terms; ++k) {
        for (int t = 0; t < 1024; ++t) {
            float a = 0.f;
            for (int64_t e = t; e < num_envs; e += 1024) a += reset_flags[e] ? episode_sums[(int64_t)k * num_envs + e] : 0.f;
            red[t] = a;
        }
        episode_means[k] = tree1024(red) / cnt * inv;
    }
    return QA_OK;
}

/* RunningMeanStd.update / update_from_moments (utils.py:62-84) on host memory, plain double loops */
int qo_normalizer_update(const float *const *batches, const int64_t *rows, int32_t num_batches, int32_t dim,
                         double *mean, double *var, double *count, void *stream) {
    (void)stream;
    if (!batches || !rows || !mean || !var || !count || num_batches <= 0 || num_batches > 4 || dim <= 0 || dim > 128) return QA_E_ARG;
    for (int b = 0; b < num_batches; ++b) {
        const float *x = batches[b]; int64_t n = rows[b];
        if (!x || n <= 0) return QA_E_ARG;
        double total = *count + (double)n;
        for (int c = 0; c < dim; ++c) {
            double bm = 0, bv = 0;
            for (int64_t r = 0; r < n; ++r) bm += x[r * dim + c];
            bm /= (double)n;
            for (int64_t r = 0; r < n; ++r) { double d = x[r * dim + c] - bm; bv += d * d; }
            bv /= (double)n;
            double delta = bm - mean[c];
            double m2 = var[c] * *count + bv * (double)n + delta * delta * *count * (double)n / total;
            mean[c] += delta * (double)n / total;
            var[c] = m2 / total;
        }
        *count = total;
    }
    return QA_OK;
}
int qo_normalizer_apply(const float *x, float *y, int64_t rows, int32_t dim, const double *mean, const double *var,
                        float epsilon, float clip, void *stream) {
    (void)stream;
    if (!x || !y || !mean || !var || rows <= 0 || dim <= 0) return QA_E_ARG;
    for (int64_t i = 0; i < rows * dim; ++i) {
        int c = (int)(i % dim);
        float m = (float)mean[c], sd = sqrtf((float)(var[c] + (double)epsilon)), v = (x[i] - m) / sd;
        y[i] = v < -clip ? -clip : (v > clip ? clip : v);
    }
    return QA_OK;
}

/* clip_grad_norm_ + Adam.step on host pointer tables (same signature as qa_clip_adam_step; fp32 state, double norm) */
int qo_clip_adam_step(float *const *params, const float *const *grads, float *const *exp_avg, float *const *exp_avg_sq,
                      float *const *steps, int32_t num_tensors, const int32_t *chunk_tensor, const int32_t *chunk_start,
                      const int32_t *chunk_len, int32_t num_chunks, const float *weight_decay, const float *lr, float beta1,
                      float beta2, float eps, float max_norm, float *scratch, int64_t scratch_floats, void *stream) {
    (void)stream;
    if (!params || !grads || !exp_avg || !exp_avg_sq || !steps || !chunk_tensor || !chunk_start || !chunk_len || !weight_decay || !lr ||
        !scratch || num_tensors <= 0 || num_chunks <= 0 || scratch_floats < 4 + (int64_t)num_chunks) return QA_E_ARG;
    double ss = 0;
    for (int c = 0; c < num_chunks; ++c) { const float *g = grads[chunk_tensor[c]] + chunk_start[c]; for (int i = 0; i < chunk_len[c]; ++i) ss += (double)g[i] * g[i]; }
    float norm = (float)sqrt(ss), coef = max_norm > 0.0f ? fminf(1.0f, max_norm / (norm + 1e-6f)) : 1.0f;
    float step = steps[0][0] + 1.0f;
    for (int t = 0; t < num_tensors; ++t) steps[t][0] = step;
    float bc1 = 1.0f - powf(beta1, step), bc2s = sqrtf(1.0f - powf(beta2, step)), step_size = lr[0] / bc1;
    for (int c = 0; c < num_chunks; ++c) {
        int t = chunk_tensor[c], s0 = chunk_start[c];
        float *p = params[t] + s0, *m = exp_avg[t] + s0, *v = exp_avg_sq[t] + s0; const float *g = grads[t] + s0;
        for (int i = 0; i < chunk_len[c]; ++i) {
            float gi = weight_decay[t] * p[i] + g[i] * coef;
            m[i] = beta1 * m[i] + (1.0f - beta1) * gi;
            v[i] = beta2 * v[i] + (1.0f - beta2) * gi * gi;
            p[i] -= step_size * m[i] / (sqrtf(v[i]) / bc2s + eps);
        }
    }
    scratch[0] = coef; scratch[1] = bc1; scratch[2] = bc2s; scratch[3] = norm;
    return QA_OK;
}

int qo_clip_adam_step_hostgrads(float *const *params, const float *const *grads_host, float *const *exp_avg, float *const *exp_avg_sq,
                                float *const *steps, int32_t num_tensors, const int32_t *chunk_tensor, const int32_t *chunk_start,
                                const int32_t *chunk_len, int32_t num_chunks, const float *weight_decay, const float *lr, float beta1,
                                float beta2, float eps, float max_norm, float *scratch, int64_t scratch_floats, void *stream) {
    if (num_tensors > QA_ADAM_MAX_INLINE) return QA_E_ARG;          /* on the host both forms read the same array */
    return qo_clip_adam_step(params, grads_host, exp_avg, exp_avg_sq, steps, num_tensors, chunk_tensor, chunk_start, chunk_len, num_chunks, weight_decay, lr,
                             beta1, beta2, eps, max_norm, scratch, scratch_floats, stream);
}

/* r5: gradients still in parts (split-K slabs of a weight gradient, row-block column sums of a bias gradient) added in order, then the step */
int qo_grad_reduce(float *const *dst, const float *const *src, const int64_t *stride, const int32_t *parts, const int32_t *numel, int32_t num_tensors, void *stream) {
    (void)stream;
    if (!dst || !src || !stride || !parts || !numel || num_tensors <= 0 || num_tensors > QA_ADAM_MAX_INLINE) return QA_E_ARG;
    for (int t = 0; t < num_tensors; ++t) {
        if (!dst[t] || !src[t] || stride[t] <= 0 || parts[t] <= 0 || numel[t] <= 0) return QA_E_ARG;
        for (int32_t i = 0; i < numel[t]; ++i) {
            double acc = 0;
            for (int32_t z = 0; z < parts[t]; ++z) acc += src[t][(int64_t)z * stride[t] + i];
            dst[t][i] = (float)acc;
        }
    }
    return QA_OK;
}
int qo_clip_adam_step_reduce(float *const *params, const float *const *grads_host, float *const *exp_avg, float *const *exp_avg_sq,
                             float *const *steps, int32_t num_tensors, const int32_t *chunk_tensor, const int32_t *chunk_start,
                             const int32_t *chunk_len, int32_t num_chunks, const float *weight_decay, const float *lr, float beta1,
                             float beta2, float eps, float max_norm, float *scratch, int64_t scratch_floats,
                             const float *const *red_src, const int64_t *red_stride, const int32_t *red_parts, void *stream) {
    if (!red_src || !red_stride || !red_parts || num_tensors > QA_ADAM_MAX_INLINE || !(max_norm > 0.f)) return QA_E_ARG;
    for (int t = 0; t < num_tensors; ++t) {
        if (red_parts[t] <= 0) continue;
        int64_t n = 0;
        for (int c = 0; c < num_chunks; ++c) if (chunk_tensor[c] == t && chunk_start[c] + chunk_len[c] > n) n = chunk_start[c] + chunk_len[c];
        float *dst[1] = {(float *)grads_host[t]}; const float *src[1] = {red_src[t]}; int32_t numel[1] = {(int32_t)n};
        int rc = qo_grad_reduce(dst, src, &red_stride[t], &red_parts[t], numel, 1, stream);
        if (rc != QA_OK) return rc;
    }
    return qo_clip_adam_step(params, grads_host, exp_avg, exp_avg_sq, steps, num_tensors, chunk_tensor, chunk_start, chunk_len, num_chunks, weight_decay, lr,
                             beta1, beta2, eps, max_norm, scratch, scratch_floats, stream);
}

/* ABI 18: two clipping optimisers with the KL rule between them (gail.py:359-362, 367-379, 405-408) over one pair of tables: the first optimiser's
 * tensors / chunks first */
int qo_clip_adam_pair_step(float *const *params, const float *const *grads_host, float *const *exp_avg, float *const *exp_avg_sq,
                           float *const *steps, int32_t num_tensors, const int32_t *chunk_tensor, const int32_t *chunk_start,
                           const int32_t *chunk_len, int32_t num_chunks, const float *weight_decay, const float *lr, float beta1,
                           float beta2, float eps, float max_norm, float *scratch, int64_t scratch_floats,
                           const float *const *red_src, const int64_t *red_stride, const int32_t *red_parts, const qa_adam_pair *pair, void *stream) {
    if (!pair || pair->split_tensor <= 0 || pair->split_tensor >= num_tensors || pair->split_chunk <= 0 || pair->split_chunk >= num_chunks || !pair->lr2 ||
        !(max_norm > 0.f) || !(pair->max_norm2 > 0.f) || num_tensors > QA_ADAM_MAX_INLINE || scratch_floats < 4 + (int64_t)num_chunks + 1 + 4) return QA_E_ARG;
    const int nt = pair->split_tensor, nc = pair->split_chunk;
    int32_t ct2[4096];
    if (num_chunks - nc > 4096) return QA_E_ARG;
    for (int c = nc; c < num_chunks; ++c) ct2[c - nc] = chunk_tensor[c] - nt;
    int32_t zero_parts[QA_ADAM_MAX_INLINE] = {0}; int64_t zero_stride[QA_ADAM_MAX_INLINE] = {0}; const float *zero_src[QA_ADAM_MAX_INLINE] = {0};
    const float *const *rs = red_src ? red_src : zero_src; const int64_t *rst = red_stride ? red_stride : zero_stride; const int32_t *rp = red_parts ? red_parts : zero_parts;
    int rc = qo_clip_adam_step_reduce(params, grads_host, exp_avg, exp_avg_sq, steps, nt, chunk_tensor, chunk_start, chunk_len, nc, weight_decay, lr, beta1, beta2, eps,
                                      max_norm, scratch, scratch_floats, rs, rst, rp, stream);
    if (rc != QA_OK) return rc;
    if (pair->kl) {
        float k = pair->kl[0], cur = pair->lr2[0], out = cur;
        if (k > pair->desired_kl * 2.0f) out = fmaxf(pair->lr_min, cur / pair->kl_factor);
        else if (k < pair->desired_kl / 2.0f && k > 0.0f) out = fminf(pair->lr_max, cur * pair->kl_factor);
        pair->lr2[0] = out;
    }
    float head1[4] = {scratch[0], scratch[1], scratch[2], scratch[3]};
    rc = qo_clip_adam_step_reduce(params + nt, grads_host + nt, exp_avg + nt, exp_avg_sq + nt, steps + nt, num_tensors - nt, ct2, chunk_start + nc, chunk_len + nc,
                                  num_chunks - nc, weight_decay + nt, pair->lr2, beta1, beta2, eps, pair->max_norm2, scratch, scratch_floats, rs + nt, rst + nt, rp + nt, stream);
    float *h2 = scratch + 4 + num_chunks + 1;
    for (int i = 0; i < 4; ++i) { h2[i] = scratch[i]; scratch[i] = head1[i]; }
    return rc;
}

/* ABI 18: a tensor's gradient put together (parts + alpha2 * second parts + reg * W), then its Adam states in order (the three optimisers of
 * bbc/rsl_rl/algorithms/gail.py:107-132 stepped one after the other, gail.py:518-520; regularisers gail.py:503-511, penalty gail.py:487-501) */
int qo_adam_stack_step(const qa_adam_stack_tensor *tensors, int32_t count, float beta1, float beta2, float eps, uint32_t *ticket, void *stream) {
    (void)stream;
    if (!tensors || count <= 0 || count > QA_ADAM_STACK_MAX_TENSORS || !ticket) return QA_E_ARG;
    for (int t = 0; t < count; ++t) {
        const qa_adam_stack_tensor *T = &tensors[t];
        if (!T->param || !T->grad || T->numel <= 0 || T->num_states < 1 || T->num_states > QA_ADAM_STACK_MAX_STATES || T->parts1 < 0 || T->parts2 < 0 ||
            (T->parts1 > 0 && (!T->src1 || T->stride1 < T->numel)) || (T->parts2 > 0 && (!T->src2 || !T->tmp || T->stride2 < T->numel))) return QA_E_ARG;
        for (int s = 0; s < T->num_states; ++s) if (!T->state[s].exp_avg || !T->state[s].exp_avg_sq || !T->state[s].step || !T->state[s].lr) return QA_E_ARG;
    }
    for (int t = 0; t < count; ++t) {
        const qa_adam_stack_tensor *T = &tensors[t];
        for (int32_t i = 0; i < T->numel; ++i) {
            float gi = T->grad[i];
            if (T->parts1 > 0) { double acc = 0; for (int z = 0; z < T->parts1; ++z) acc += T->src1[(int64_t)z * T->stride1 + i]; gi = (float)acc; }
            if (T->parts2 > 0) { double acc = 0; for (int z = 0; z < T->parts2; ++z) acc += T->src2[(int64_t)z * T->stride2 + i]; T->tmp[i] = (float)acc; gi = fmaf(T->alpha2, T->tmp[i], gi); }
            gi = fmaf(T->reg, T->param[i], gi);
            T->grad[i] = gi;
            float pi = T->param[i];
            for (int s = 0; s < T->num_states; ++s) {
                const qa_adam_stack_state *S = &T->state[s];
                float step = S->step[0] + 1.0f, bc1 = 1.0f - powf(beta1, step), bc2s = sqrtf(1.0f - powf(beta2, step)), step_size = S->lr[0] / bc1;
                float ge = S->weight_decay * pi + gi;
                S->exp_avg[i] = beta1 * S->exp_avg[i] + (1.0f - beta1) * ge;
                S->exp_avg_sq[i] = beta2 * S->exp_avg_sq[i] + (1.0f - beta2) * ge * ge;
                pi -= step_size * S->exp_avg[i] / (sqrtf(S->exp_avg_sq[i]) / bc2s + eps);
            }
            T->param[i] = pi;
        }
    }
    for (int t = 0; t < count; ++t) for (int s = 0; s < tensors[t].num_states; ++s) tensors[t].state[s].step[0] += 1.0f;
    return QA_OK;
}

/* rollout bookkeeping twins (host pointers) */
int qo_rollout_act(const float *mean, const float *std, const float *value, const float *noise, uint64_t seed, const int64_t *step_dev,
                   int64_t step, int32_t num_envs, int32_t env_id_offset, float *actions, float *st_actions, float *st_mu, float *st_sigma, float *st_logp,
                   float *st_values, void *stream) {
    (void)stream;
    if (!mean || !std || !value || !actions || !st_actions || !st_mu || !st_sigma || !st_logp || !st_values || num_envs <= 0) return QA_E_ARG;
    if (step_dev) step = *step_dev;
    for (int e = 0; e < num_envs; ++e) {
        float eps[12];
        if (noise) for (int j = 0; j < 12; ++j) eps[j] = noise[(int64_t)e * 12 + j];
        else for (int b = 0; b < 3; ++b) {
            uint32_t o[4]; float u[4];
            philox(seed, (uint32_t)(e + env_id_offset), (uint32_t)step, (uint32_t)(20 * 256 + b), (uint32_t)((uint64_t)step >> 32), o);
            for (int i = 0; i < 4; ++i) u[i] = (float)(o[i] >> 8) * (1.0f / 16777216.0f);
            float r0 = sqrtf(-2.0f * logf(fmaxf(u[0], 1e-7f))), r1 = sqrtf(-2.0f * logf(fmaxf(u[2], 1e-7f)));
            eps[4 * b] = r0 * cosf(6.28318530717958647692f * u[1]); eps[4 * b + 1] = r0 * sinf(6.28318530717958647692f * u[1]);
            eps[4 * b + 2] = r1 * cosf(6.28318530717958647692f * u[3]); eps[4 * b + 3] = r1 * sinf(6.28318530717958647692f * u[3]);
        }
        float logp = 0.0f;
        for (int j = 0; j < 12; ++j) {
            int64_t i = (int64_t)e * 12 + j;
            float m = mean[i], s = std[j], a = m + s * eps[j], d = a - m;
            logp += -(d * d) / (2.0f * s * s) - logf(s) - 0.91893853320467274178f;
            actions[i] = a; st_actions[i] = a; st_mu[i] = m; st_sigma[i] = s;
        }
        st_logp[e] = logp; st_values[e] = value[e];
    }
    return QA_OK;
}

int qo_rollout_act_store(const float *mean, const float *std, const float *value, const float *noise, uint64_t seed, const int64_t *step_dev,
                         int64_t step, int32_t num_envs, int32_t env_id_offset, float *actions, float *st_actions, float *st_mu, float *st_sigma, float *st_logp,
                         float *st_values, const float *obs, int64_t obs_stride, int32_t obs_width, float *st_obs, int64_t st_obs_stride, void *stream) {
    if (!obs || !st_obs || obs_width <= 0 || obs_stride < obs_width || st_obs_stride < obs_width) return QA_E_ARG;
    int rc = qo_rollout_act(mean, std, value, noise, seed, step_dev, step, num_envs, env_id_offset, actions, st_actions, st_mu, st_sigma, st_logp, st_values, stream);
    if (rc != QA_OK) return rc;
    for (int64_t r = 0; r < num_envs; ++r) for (int c = 0; c < obs_width; ++c) st_obs[r * st_obs_stride + c] = obs[r * obs_stride + c];
    return QA_OK;
}
int qo_tsc_push(float *root_states, int64_t num_envs, int64_t *step_dev, int32_t *ticket, int32_t push_interval, float max_push_vel_xy, uint64_t seed,
                int32_t env_id_offset, void *stream) {
    (void)stream; (void)ticket;         /* tsc/legged_gym/envs/base/legged_robot.py:905-915 */
    if (!root_states || !step_dev || !ticket || num_envs <= 0) return QA_E_ARG;
    const int64_t s = *step_dev + 1;
    if (push_interval > 0 && s % push_interval == 0)
        for (int64_t e = 0; e < num_envs; ++e) {
            uint32_t o[4];
            philox(seed, (uint32_t)(e + env_id_offset), (uint32_t)s, (uint32_t)(22 * 256 + 0), (uint32_t)((uint64_t)s >> 32), o);
            root_states[e * 13 + 7] = ((float)(o[0] >> 8) * (1.0f / 16777216.0f) * 2.0f - 1.0f) * max_push_vel_xy;
            root_states[e * 13 + 8] = ((float)(o[1] >> 8) * (1.0f / 16777216.0f) * 2.0f - 1.0f) * max_push_vel_xy;
        }
    *step_dev = s;
    return QA_OK;
}

int qo_tsc_start_pose(const uint8_t *flags, int64_t *cur_obst_idx, const float *env_goals, const float *obst_angs, int64_t num_envs, int32_t num_goal_slots,
                      int32_t num_obstacles, int32_t goals_per_obstacle, int32_t randomize_start, float frame_yaw0, uint64_t seed, const int64_t *step_dev,
                      int32_t env_id_offset, float *start_xy, float *start_yaw, int64_t *start_goal, void *stream) {
    (void)stream;                       /* :352-366 */
    if (!flags || !env_goals || !start_xy || !start_yaw || !start_goal || num_envs <= 0 || num_goal_slots <= 0 ||
        (randomize_start && (!cur_obst_idx || !obst_angs || !step_dev || num_obstacles <= 0 || goals_per_obstacle <= 0))) return QA_E_ARG;
    for (int64_t e = 0; e < num_envs; ++e) {
        int64_t sg = 0; float yaw = frame_yaw0;
        if (randomize_start) {
            int64_t ob = cur_obst_idx[e];
            if (flags[e]) {
                uint32_t o[4];
                philox(seed, (uint32_t)(e + env_id_offset), (uint32_t)*step_dev, (uint32_t)(23 * 256 + 0), (uint32_t)((uint64_t)*step_dev >> 32), o);
                int d = (int)((float)(o[0] >> 8) * (1.0f / 16777216.0f) * (float)num_obstacles);
                ob = d >= num_obstacles ? num_obstacles - 1 : d;
                cur_obst_idx[e] = ob;
            }
            sg = ob * goals_per_obstacle; yaw = obst_angs[e * num_obstacles + ob];
        }
        const int64_t g = sg < 0 ? 0 : (sg >= num_goal_slots ? num_goal_slots - 1 : sg);
        start_xy[2 * e] = env_goals[(e * num_goal_slots + g) * 3]; start_xy[2 * e + 1] = env_goals[(e * num_goal_slots + g) * 3 + 1];
        start_yaw[e] = yaw; start_goal[e] = sg;
    }
    return QA_OK;
}

int qo_tsc_reset_where(const uint8_t *flags, const uint8_t *any_reset, const int64_t *start_goal, int64_t *cur_goal_idx, float *reach_goal_timer,
                       float *episode_sums, int32_t num_terms, int64_t *episode_length, const float *env_goals, int32_t num_goal_slots, float *cur_goals,
                       float *next_goals, float *obst_state, float seesaw_rest, const int64_t *cur_obst_idx, const int64_t *seesaw_order, int64_t num_envs,
                       void *stream) {
    (void)stream;                       /* :367-376, 396-404, 812-823 */
    if (!flags || !start_goal || !cur_goal_idx || !reach_goal_timer || !episode_sums || !episode_length || !env_goals || !cur_goals || !next_goals ||
        num_envs <= 0 || num_terms <= 0 || num_goal_slots <= 0 || (obst_state && (!any_reset || (seesaw_order && !cur_obst_idx)))) return QA_E_ARG;
    for (int64_t e = 0; e < num_envs; ++e) {
        int64_t g = cur_goal_idx[e];
        if (flags[e]) {
            g = start_goal[e]; cur_goal_idx[e] = g; reach_goal_timer[e] = 0.0f; episode_length[e] = 0;
            for (int k = 0; k < num_terms; ++k) episode_sums[(int64_t)k * num_envs + e] = 0.0f;
        }
        const int64_t g0 = g < 0 ? 0 : (g >= num_goal_slots ? num_goal_slots - 1 : g), g1 = g + 1 < 0 ? 0 : (g + 1 >= num_goal_slots ? num_goal_slots - 1 : g + 1);
        for (int k = 0; k < 3; ++k) { cur_goals[e * 3 + k] = env_goals[(e * num_goal_slots + g0) * 3 + k]; next_goals[e * 3 + k] = env_goals[(e * num_goal_slots + g1) * 3 + k]; }
        if (obst_state) {
            float *st = obst_state + e * 12;
            if (flags[e]) st[0] = (seesaw_order && cur_obst_idx[e] > seesaw_order[e]) ? -seesaw_rest : seesaw_rest;
            if (any_reset[0]) { st[1] = 0.0f; st[5] = 0.0f; st[9] = 0.0f; }
        }
    }
    return QA_OK;
}

int qo_rollout_act_hybrid(const float *logits, const float *mean, const float *std, const float *value, uint64_t seed, const int64_t *step_dev, int64_t step,
                          int32_t num_envs, int32_t env_id_offset, int32_t nd, int32_t nc_all, float *actions, float *st_actions, float *st_mu, float *st_sigma,
                          float *st_logp_d, float *st_logp_c, float *st_values, const float *action_history_in, float *action_history, int32_t hist_len,
                          void *stream) {
    (void)stream;       /* tsc/rsl_rl/modules/actor_critic.py:252-261, algorithms/ppo.py:101-125 */
    if (!logits || !mean || !std || !value || !actions || !st_actions || !st_mu || !st_sigma || !st_logp_d || !st_logp_c || !st_values || num_envs <= 0 ||
        nd <= 0 || nd > 16 || nc_all <= 0 || nc_all > 32 || (action_history && hist_len <= 0) || (action_history_in && !action_history)) return QA_E_ARG;
    if (!action_history_in) action_history_in = action_history;
    if (step_dev) step = *step_dev;
    const int w = 1 + nc_all;
    const float EPS = 1.1920928955078125e-07f;
    for (int e = 0; e < num_envs; ++e) {
        const float *lg = logits + (int64_t)e * nd;
        float mx = lg[0], p[16], sum = 0.0f;
        for (int i = 1; i < nd; ++i) if (lg[i] > mx) mx = lg[i];
        for (int i = 0; i < nd; ++i) { p[i] = expf(lg[i] - mx); sum += p[i]; }
        const float inv = 1.0f / sum;
        uint32_t o[4];
        philox(seed, (uint32_t)(e + env_id_offset), (uint32_t)step, (uint32_t)(21 * 256 + 0), (uint32_t)((uint64_t)step >> 32), o);
        const float u = (float)(o[0] >> 8) * (1.0f / 16777216.0f);
        int choice = nd - 1; float cdf = 0.0f, pa = 0.0f; int found = 0;
        for (int i = 0; i < nd; ++i) {
            const float pi = p[i] * inv;
            cdf += pi;
            if (!found && (u < cdf || i == nd - 1)) { choice = i; pa = pi; found = 1; }
        }
        float cl = pa < EPS ? EPS : (pa > 1.0f - EPS ? 1.0f - EPS : pa);
        float *act = actions + (int64_t)e * w, *sa = st_actions + (int64_t)e * w;
        act[0] = (float)choice; sa[0] = (float)choice;
        float logp_c = 0.0f;
        for (int b = 0; 4 * b < nc_all; ++b) {
            float uu[4];
            philox(seed, (uint32_t)(e + env_id_offset), (uint32_t)step, (uint32_t)(20 * 256 + b), (uint32_t)((uint64_t)step >> 32), o);
            for (int i = 0; i < 4; ++i) uu[i] = (float)(o[i] >> 8) * (1.0f / 16777216.0f);
            const float r0 = sqrtf(-2.0f * logf(fmaxf(uu[0], 1e-7f))), r1 = sqrtf(-2.0f * logf(fmaxf(uu[2], 1e-7f)));
            const float eps4[4] = {r0 * cosf(6.28318530717958647692f * uu[1]), r0 * sinf(6.28318530717958647692f * uu[1]),
                                   r1 * cosf(6.28318530717958647692f * uu[3]), r1 * sinf(6.28318530717958647692f * uu[3])};
            for (int k = 0; k < 4; ++k) {
                const int j = 4 * b + k;
                if (j >= nc_all) break;
                const float m = mean[(int64_t)e * nc_all + j], s = std[j], v = m + s * eps4[k], d = v - m;
                logp_c += -(d * d) / (2.0f * s * s) - logf(s) - 0.91893853320467274178f;
                act[1 + j] = v; sa[1 + j] = v;
                st_mu[(int64_t)e * nc_all + j] = m; st_sigma[(int64_t)e * nc_all + j] = s;
            }
        }
        st_logp_d[e] = logf(cl); st_logp_c[e] = logp_c; st_values[e] = value[e];
        if (action_history) {
            float *h = action_history + (int64_t)e * hist_len * w;
            memmove(h, action_history_in + (int64_t)e * hist_len * w + w, sizeof(float) * (size_t)(hist_len - 1) * (size_t)w);
            memcpy(h + (int64_t)(hist_len - 1) * w, act, sizeof(float) * (size_t)w);
        }
    }
    return QA_OK;
}
int qo_rollout_post(const float *rew, const int64_t *reset, const uint8_t *time_out, const float *values, float reward_coef, float gamma,
                    int32_t num_envs, float *st_rewards, uint8_t *st_dones, float *cur, float *fin_vals, uint8_t *fin_mask, void *stream) {
    (void)stream;
    if (!rew || !reset || !time_out || !values || !st_rewards || !st_dones || num_envs <= 0 || (cur && (!fin_vals || !fin_mask))) return QA_E_ARG;
    int N = num_envs;
    for (int e = 0; e < N; ++e) {
        float r_t = rew[e], r = reward_coef * r_t; int done = reset[e] > 0;
        st_rewards[e] = r + gamma * values[e] * (time_out[e] ? 1.0f : 0.0f);
        st_dones[e] = (uint8_t)done;
        if (cur) {
            const float add[6] = {r, 0, 0, 0, r_t, 1.0f};
            for (int k = 0; k < 6; ++k) { float c = cur[(int64_t)k * N + e] + add[k]; fin_vals[(int64_t)k * N + e] = c; cur[(int64_t)k * N + e] = done ? 0.0f : c; }
            fin_mask[e] = (uint8_t)done;
        }
    }
    return QA_OK;
}

/* discriminator head losses + gradient (gail.py:452-520, MSELoss variant); double arithmetic, host pointers */
int64_t qo_disc_loss_scratch_bytes(int64_t rows) { return rows <= 0 ? -1 : 0; }
int qo_disc_loss(const float *d, const float *eps, const float *c, const int64_t *label_lb, const float *policy_eps, const float *policy_c,
                 int32_t b_lb, int32_t b_pi, int32_t b_ulb, float c_ss, const float *info_coef_dev, float c_disc, float c_us,
                 float *grad_d, float *grad_eps, float *grad_c, float *out, void *scratch, int64_t scratch_bytes, void *stream) {
    (void)scratch; (void)scratch_bytes; (void)stream;
    if (!d || !eps || !c || !label_lb || !policy_eps || !policy_c || !info_coef_dev || !grad_d || !grad_eps || !grad_c || !out ||
        b_lb <= 0 || b_pi <= 0 || b_ulb <= 0) return QA_E_ARG;
    int B = b_lb + b_pi + b_ulb;
    double ss = 0, info = 0, dpi = 0, dex = 0, us = 0, acc[4] = {0, 0, 0, 0}, pm[5] = {0, 0, 0, 0, 0}, c_info = info_coef_dev[0];
    for (int i = 0; i < B; ++i) {
        const float *craw = c + (int64_t)i * 5; float *gc = grad_c + (int64_t)i * 5;
        float ci[5]; int pass[5];                   /* torch.clamp(c, 1e-20) of Discriminator.forward (discriminator.py:52): clamped entries pass no gradient */
        for (int j = 0; j < 5; ++j) { pass[j] = craw[j] >= 1e-20f; ci[j] = pass[j] ? craw[j] : 1e-20f; }
        int arg = 0; for (int j = 1; j < 5; ++j) if (ci[j] > ci[arg]) arg = j;
        for (int j = 0; j < 5; ++j) gc[j] = 0; grad_d[i] = 0; grad_eps[i] = 0;
        if (i < b_lb) {
            double m = ci[0]; for (int j = 1; j < 5; ++j) if (ci[j] > m) m = ci[j];
            double e[5], se = 0; for (int j = 0; j < 5; ++j) { e[j] = exp((double)ci[j] - m); se += e[j]; }
            int lab = (int)label_lb[i];
            ss += m + log(se) - ci[lab];
            for (int j = 0; j < 5; ++j) gc[j] = (float)((double)c_ss / b_lb * (e[j] / se - (lab == j ? 1.0 : 0.0)));
            acc[0] += arg == lab;
        } else if (i < b_lb + b_pi) {
            int r = i - b_lb; double dd = d[i];
            dpi += (dd + 1) * (dd + 1); grad_d[i] = (float)((double)c_disc * (dd + 1) / b_pi);
            double de = (double)eps[i] - policy_eps[r];
            us += fabs(de); grad_eps[i] = (float)((double)c_us * (de > 0 ? 1.0 : (de < 0 ? -1.0 : 0.0)) / b_pi);
            acc[1] += dd < 0;
            int pa = 0; for (int j = 1; j < 5; ++j) if (policy_c[(int64_t)r * 5 + j] > policy_c[(int64_t)r * 5 + pa]) pa = j;
            acc[3] += arg == pa;
        } else {
            double dd = d[i];
            dex += (dd - 1) * (dd - 1); grad_d[i] = (float)((double)c_disc * (dd - 1) / b_ulb);
            for (int j = 0; j < 5; ++j) {
                double p = ci[j], l = log(p + 1e-20);
                info -= p * l; gc[j] = (float)(-c_info / b_ulb * (l + p / (p + 1e-20))); pm[j] += p;
            }
            acc[2] += dd > 0;
        }
        for (int j = 0; j < 5; ++j) if (!pass[j]) gc[j] = 0;
    }
    ss /= b_lb; info /= b_ulb; us /= b_pi;
    double disc = 0.5 * (dpi / b_pi + dex / b_ulb);
    out[0] = (float)(c_ss * ss + c_info * info + c_disc * disc + c_us * us);
    out[1] = (float)ss; out[2] = (float)info; out[3] = (float)disc; out[4] = (float)us;
    out[5] = (float)(acc[0] / b_lb); out[6] = (float)(acc[1] / b_pi); out[7] = (float)(acc[2] / b_ulb); out[8] = (float)(acc[3] / b_pi);
    for (int j = 0; j < 5; ++j) out[9 + j] = (float)(pm[j] / b_ulb);
    out[14] = 0; out[15] = 0;
    return QA_OK;
}

/* ABI 18: class logits in, gradient w.r.t. the logits out (softmax + its backward around qo_disc_loss; discriminator.py:66) */
int qo_disc_loss_logits(const float *d, const float *eps, const float *logits, const int64_t *label_lb, const float *policy_eps, const float *policy_c,
                        int32_t b_lb, int32_t b_pi, int32_t b_ulb, float c_ss, const float *info_coef_dev, float c_disc, float c_us,
                        float *grad_d, float *grad_eps, float *grad_logits, float *out, void *scratch, int64_t scratch_bytes, void *stream) {
    if (!logits || !grad_logits || b_lb <= 0 || b_pi <= 0 || b_ulb <= 0) return QA_E_ARG;
    const int64_t B = (int64_t)b_lb + b_pi + b_ulb;
    float *p = (float *)malloc(sizeof(float) * 5 * (size_t)B);
    if (!p) return QA_E_ARG;
    for (int64_t r = 0; r < B; ++r) {
        float m = logits[r * 5], s = 0.f;
        for (int j = 1; j < 5; ++j) m = fmaxf(m, logits[r * 5 + j]);
        for (int j = 0; j < 5; ++j) { p[r * 5 + j] = expf(logits[r * 5 + j] - m); s += p[r * 5 + j]; }
        for (int j = 0; j < 5; ++j) p[r * 5 + j] = p[r * 5 + j] / s;
    }
    int rc = qo_disc_loss(d, eps, p, label_lb, policy_eps, policy_c, b_lb, b_pi, b_ulb, c_ss, info_coef_dev, c_disc, c_us, grad_d, grad_eps, grad_logits, out,
                          scratch, scratch_bytes, stream);
    if (rc == QA_OK)
        for (int64_t r = 0; r < B; ++r) {
            float dot = 0.f;
            for (int j = 0; j < 5; ++j) dot += grad_logits[r * 5 + j] * p[r * 5 + j];
            for (int j = 0; j < 5; ++j) grad_logits[r * 5 + j] = p[r * 5 + j] * (grad_logits[r * 5 + j] - dot);
        }
    free(p);
    return rc;
}

int qo_disc_prepare(const float *const *batches, const int64_t *rows, int32_t num_batches, int32_t dim, const float *task_mask,
                    const float *frame_mult, const float *task_weight_dev, const double *mean, const double *var, float epsilon, float clip,
                    float *out, void *stream) {
    (void)stream;
    if (!batches || !rows || !task_mask || !frame_mult || !out || num_batches <= 0 || num_batches > 3 || dim <= 0 || ((mean == NULL) != (var == NULL))) return QA_E_ARG;
    int64_t o = 0;
    for (int b = 0; b < num_batches; ++b) {
        if (!batches[b] || rows[b] <= 0) return QA_E_ARG;
        for (int64_t r = 0; r < rows[b]; ++r)
            for (int c = 0; c < dim; ++c) {
                float x = batches[b][r * dim + c];
                if (task_weight_dev && task_mask[c] != 0.0f) x *= task_weight_dev[0];
                x *= frame_mult[c];
                if (mean) { float m = (float)mean[c], sd = sqrtf((float)(var[c] + (double)epsilon)); x = (x - m) / sd; x = x < -clip ? -clip : (x > clip ? clip : x); }
                out[o++] = x;
            }
    }
    return QA_OK;
}

int qo_rollout_post_amp(const float *rew, const int64_t *reset, const uint8_t *time_out, const float *values, const float *d, const float *eps,
                        const float *logits, int32_t dim_c, const float *obs, int64_t obs_stride, int32_t num_obs, float c_i, float c_us,
                        float c_ss, float c_t, float dt, float gamma, int32_t num_envs, float *st_rewards, uint8_t *st_dones, float *cur,
                        float *fin_vals, uint8_t *fin_mask, void *stream) {
    (void)stream;
    if (!rew || !reset || !time_out || !values || !d || !eps || !logits || !obs || !st_rewards || !st_dones || num_envs <= 0 || dim_c <= 0 || dim_c > 8 ||
        num_obs < dim_c + 1 || obs_stride < num_obs || (cur && (!fin_vals || !fin_mask))) return QA_E_ARG;
    const int N = num_envs;
    for (int e = 0; e < N; ++e) {               /* Discriminator.predict_disc_reward, discriminator.py:88-118 (MSELoss mapping) */
        const float *o = obs + (int64_t)e * obs_stride;
        const double label_eps = o[num_obs - dim_c - 1];
        int label = 0;
        for (int k = 1; k < dim_c; ++k) if (o[num_obs - dim_c + k] > o[num_obs - dim_c + label]) label = k;
        double p[8], m = logits[(int64_t)e * dim_c], z = 0.0, lse = 0.0;
        for (int k = 1; k < dim_c; ++k) if (logits[(int64_t)e * dim_c + k] > m) m = logits[(int64_t)e * dim_c + k];
        for (int k = 0; k < dim_c; ++k) { p[k] = exp((double)logits[(int64_t)e * dim_c + k] - m); z += p[k]; }
        for (int k = 0; k < dim_c; ++k) { p[k] /= z; if (p[k] < 1e-20) p[k] = 1e-20; }
        for (int k = 0; k < dim_c; ++k) lse += exp(p[k]);
        lse = log(lse);
        const double dd = (double)d[e] - 1.0, r_i0 = 1.0 - 0.25 * dd * dd;
        const double r_i = (r_i0 > 0.0 ? r_i0 : 0.0) * dt, r_us = -fabs((double)eps[e] - label_eps) * dt, r_ss = (p[label] - lse) * dt, r_t = rew[e];
        const double total = c_i * r_i + c_us * r_us + c_ss * r_ss + c_t * r_t;
        const int done = reset[e] > 0;
        st_rewards[e] = (float)(total + (double)gamma * values[e] * (time_out[e] ? 1.0 : 0.0));
        st_dones[e] = done ? 1 : 0;
        if (cur) {
            const double add[6] = {total, r_i, r_us, r_ss, r_t, 1.0};
            for (int k = 0; k < 6; ++k) {
                const float c = (float)((double)cur[(int64_t)k * N + e] + add[k]);
                fin_vals[(int64_t)k * N + e] = c;
                cur[(int64_t)k * N + e] = done ? 0.0f : c;
            }
            fin_mask[e] = done ? 1 : 0;
        }
    }
    return QA_OK;
}

int64_t qo_pair_loss_scratch_bytes(int64_t rows) { return rows <= 0 ? -1 : (int64_t)sizeof(float) * ((rows + 255) / 256); }

int qo_pair_loss(const float *a, const float *b, int64_t rows, int32_t cols, int64_t b_stride, int32_t mode, float *grad_a, float *out,
                 void *scratch, int64_t scratch_bytes, void *stream) {
    (void)scratch; (void)scratch_bytes; (void)stream;
    if (!a || !b || !grad_a || !out || rows <= 0 || cols <= 0 || b_stride < cols || (mode != QA_PAIR_ROW_L2 && mode != QA_PAIR_MSE)) return QA_E_ARG;
    double acc = 0.0;
    for (int64_t r = 0; r < rows; ++r) {
        double ss = 0.0;
        for (int c = 0; c < cols; ++c) { const double d = (double)a[r * cols + c] - (double)b[r * b_stride + c]; ss += d * d; }
        if (mode == QA_PAIR_ROW_L2) {               /* (p - h).norm(p=2, dim=1).mean(), gail.py:348 */
            const double n = sqrt(ss), sc = n > 0.0 ? 1.0 / (n * (double)rows) : 0.0;
            for (int c = 0; c < cols; ++c) grad_a[r * cols + c] = (float)(((double)a[r * cols + c] - (double)b[r * b_stride + c]) * sc);
            acc += n;
        } else {                                    /* (e - t).pow(2).mean(), gail.py:357 */
            const double sc = 2.0 / ((double)rows * (double)cols);
            for (int c = 0; c < cols; ++c) grad_a[r * cols + c] = (float)(((double)a[r * cols + c] - (double)b[r * b_stride + c]) * sc);
            acc += ss;
        }
    }
    out[0] = (float)(mode == QA_PAIR_ROW_L2 ? acc / (double)rows : acc / ((double)rows * (double)cols));
    return QA_OK;
}

int qo_accumulate_scalars(float *acc, const float *const *src, int32_t count, void *stream) {
    (void)stream;
    if (!acc || !src || count <= 0 || count > QA_ACC_MAX) return QA_E_ARG;
    for (int i = 0; i < count; ++i) { if (!src[i]) return QA_E_ARG; acc[i] += src[i][0]; }
    return QA_OK;
}

/* ABI 18: several pair losses in one call, the gradient optionally times a device scalar (the regulariser's coefficient, gail.py:353-357) */
int64_t qo_pair_losses_scratch_bytes(const qa_pair_job *jobs, int32_t count) {
    if (!jobs || count <= 0 || count > QA_PAIR_MAX_JOBS) return -1;
    int64_t blocks = 0;
    for (int t = 0; t < count; ++t) { if (jobs[t].rows <= 0) return -1; blocks += (jobs[t].rows + 255) / 256; }
    return (int64_t)sizeof(float) * (blocks + 1);
}
int qo_pair_losses(const qa_pair_job *jobs, int32_t count, void *scratch, int64_t scratch_bytes, void *stream) {
    if (!jobs || count <= 0 || count > QA_PAIR_MAX_JOBS || !scratch || scratch_bytes < qo_pair_losses_scratch_bytes(jobs, count)) return QA_E_ARG;
    for (int t = 0; t < count; ++t) {
        const qa_pair_job *J = &jobs[t];
        int rc = qo_pair_loss(J->a, J->b, J->rows, J->cols, J->b_stride, J->mode, J->grad_a, J->out, scratch, scratch_bytes, stream);
        if (rc != QA_OK) return rc;
        if (J->grad_scale) for (int64_t i = 0; i < J->rows * J->cols; ++i) J->grad_a[i] *= J->grad_scale[0];
    }
    return QA_OK;
}

int qo_gather_rows(const int64_t *idx, const int64_t *idx_block, int64_t rows, int32_t num_tensors, const float *const *src, const int64_t *src_strides,
                   const int32_t *widths, float *const *dst, void *stream) {
    (void)stream;
    if (idx && idx_block) idx += idx_block[0] * rows;
    if (!idx || !src || !src_strides || !widths || !dst || rows <= 0 || num_tensors <= 0 || num_tensors > QA_GATHER_MAX) return QA_E_ARG;
    for (int t = 0; t < num_tensors; ++t) {
        if (!src[t] || !dst[t] || widths[t] <= 0 || src_strides[t] < widths[t]) return QA_E_ARG;
        for (int64_t r = 0; r < rows; ++r) memcpy(dst[t] + r * widths[t], src[t] + idx[r] * src_strides[t], sizeof(float) * (size_t)widths[t]);
    }
    return QA_OK;
}

int qo_kl_lr_rule(const float *kl, float desired_kl, float factor, float lr_min, float lr_max, float *lr, void *stream) {
    (void)stream;
    if (!kl || !lr || !(desired_kl > 0.f) || !(factor > 1.f) || !(lr_min > 0.f) || !(lr_max >= lr_min)) return QA_E_ARG;
    const float k = kl[0];                                  /* gail.py:367-379 */
    if (k > desired_kl * 2.0f) { const float v = lr[0] / factor; lr[0] = v > lr_min ? v : lr_min; }
    else if (k < desired_kl / 2.0f && k > 0.0f) { const float v = lr[0] * factor; lr[0] = v < lr_max ? v : lr_max; }
    return QA_OK;
}

int qo_set_lean_exports(qa_sim *sim, int32_t mask) {
    /* the twin always exports (tests compare lean HIP runs with it on the tensors the mode keeps) */
    if (!sim || (mask != 0 && mask != 1 && mask != 3)) return QA_E_ARG;
    return QA_OK;
}

int qo_episode_means(const float *episode_stats, const int64_t *step_dev, int64_t step, int32_t num_terms, float max_episode_length_s, float *means,
                     float *snapshot, void *stream) {
    (void)stream;                                           /* legged_robot.py:229-240 */
    if (!episode_stats || !means || !snapshot || num_terms <= 0 || num_terms > 14 || !(max_episode_length_s > 0.f)) return QA_E_ARG;
    const int64_t st = step_dev ? step_dev[0] : step;
    const float *b = episode_stats + 16 * (int)((st - 1) & 1);
    const float cnt = b[14];
    for (int i = 0; i < num_terms; ++i) {
        if (cnt > 0.f) means[i] = b[i] / (cnt > 1.0f ? cnt : 1.0f) / max_episode_length_s;
        snapshot[i] = means[i];
    }
    return QA_OK;
}

/* ---- policy inference chain (include/qa_sim.h "policy inference"): Estimator.forward (bbc/rsl_rl/modules/estimator.py:35-36),
 * ActorCritic.update_distribution / evaluate (actor_critic.py:171-196,222-225) as nn.Linear + ELU layers and column copies.
 * The twin keeps the weights row-major inside `packed` (the caller's offsets leave room: the device layout is padded) and
 * accumulates every dot product in double. */
int64_t qo_mlp_packed_floats(const qa_mlp_op *ops, int32_t num_ops) {
    if (!ops || num_ops <= 0) return -1;
    int64_t end = 0;
    for (int i = 0; i < num_ops; ++i) {
        if (ops[i].kind != QA_MLP_LAYER) continue;
        const int64_t nt = (ops[i].n + 15) / 16, per = (nt + 7) / 8, pf = per > 2 ? 4 : 8;     /* the device layout's padding */
        const int64_t kb = ((ops[i].k + 15) / 16 + pf - 1) / pf * pf;
        const int64_t we = ops[i].w_off + nt * kb * 256, be = ops[i].b_off + nt * 16;
        if (we > end) end = we;
        if (be > end) end = be;
    }
    return end;
}

int qo_mlp_pack(const qa_mlp_op *ops, int32_t num_ops, const float *const *weights, const float *const *biases, float *packed, int64_t packed_floats,
                void *stream) {
    (void)stream;
    if (!ops || num_ops <= 0 || num_ops > QA_MLP_MAX_OPS || !weights || !biases || !packed || packed_floats < qo_mlp_packed_floats(ops, num_ops)) return QA_E_ARG;
    for (int i = 0; i < num_ops; ++i) {
        if (ops[i].kind != QA_MLP_LAYER) continue;
        if (!weights[i]) return QA_E_ARG;
        if (ops[i].flags & QA_MLP_F_TRANSPOSED) {         /* the forward layer's (k, n) matrix, used transposed: keep (n, k) row-major here */
            for (int r = 0; r < ops[i].n; ++r)
                for (int c = 0; c < ops[i].k; ++c) packed[ops[i].w_off + (int64_t)r * ops[i].k + c] = weights[i][(int64_t)c * ops[i].n + r];
        } else
            memcpy(packed + ops[i].w_off, weights[i], sizeof(float) * (size_t)ops[i].n * (size_t)ops[i].k);
        for (int c = 0; c < ops[i].n; ++c) packed[ops[i].b_off + c] = biases[i] ? biases[i][c] : 0.0f;
    }
    return QA_OK;
}

/* The partition of a chain into strands (include/qa_sim.h, ABI 15): checker-side restatement by label propagation.  Op i DEPENDS on op j < i when
 * i reads a scratch column whose most recent write before i was j's; a group is a connected set of ops, named by its earliest op; groups are dealt
 * largest cost first (ties: earliest group first) to the least loaded strand (ties: lowest strand).  cost(op) = k n for a layer, + 4096 per op. */
int qo_mlp_strands(const qa_mlp_op *ops, int32_t num_ops, int32_t max_strands, int32_t *strand_of) {
    if (!ops || !strand_of || num_ops <= 0 || num_ops > QA_MLP_MAX_OPS || max_strands < 1) return QA_E_ARG;
    if (max_strands > 4) max_strands = 4;
    int label[QA_MLP_MAX_OPS];
    unsigned char dep[QA_MLP_MAX_OPS][QA_MLP_MAX_OPS];
    memset(dep, 0, sizeof(dep));
    for (int i = 0; i < num_ops; ++i) {
        label[i] = i; strand_of[i] = 0;
        if (ops[i].src_buf <= 0) continue;                       /* the input tile is everybody's */
        const int rd = ops[i].kind == QA_MLP_LAYER ? ops[i].k : ops[i].n;
        for (int c = ops[i].src_col; c < ops[i].src_col + rd; ++c)
            for (int j = i - 1; j >= 0; --j)                     /* the most recent earlier op that wrote column c of that buffer */
                if (ops[j].dst_buf == ops[i].src_buf && c >= ops[j].dst_col && c < ops[j].dst_col + ops[j].n) { dep[i][j] = 1; break; }
    }
    for (int changed = 1; changed;) {
        changed = 0;
        for (int i = 0; i < num_ops; ++i)
            for (int j = 0; j < num_ops; ++j)
                if ((dep[i][j] || dep[j][i]) && label[j] < label[i]) { label[i] = label[j]; changed = 1; }
    }
    double cost[QA_MLP_MAX_OPS];
    int ngroups = 0;
    for (int g = 0; g < num_ops; ++g) {
        cost[g] = -1.0;
        for (int i = 0; i < num_ops; ++i)
            if (label[i] == g) cost[g] = (cost[g] < 0 ? 0.0 : cost[g]) + (ops[i].kind == QA_MLP_LAYER ? (double)ops[i].k * ops[i].n : 0.0) + 4096.0;
        if (cost[g] >= 0) ++ngroups;
    }
    if (max_strands < 2 || ngroups < 2) return 1;
    const int ns = ngroups < max_strands ? ngroups : max_strands;
    double load[4] = {0, 0, 0, 0};
    int group_strand[QA_MLP_MAX_OPS];
    for (int dealt = 0; dealt < ngroups; ++dealt) {
        int g = -1;
        for (int c = 0; c < num_ops; ++c) if (cost[c] >= 0 && (g < 0 || cost[c] > cost[g])) g = c;      /* largest left; ties keep the earliest */
        int best = 0;
        for (int t = 1; t < ns; ++t) if (load[t] < load[best]) best = t;
        group_strand[g] = best; load[best] += cost[g]; cost[g] = -1.0;
    }
    for (int i = 0; i < num_ops; ++i) strand_of[i] = group_strand[label[i]];
    return ns;
}

/* The two-group launch's LDS plan (include/qa_sim.h, ABI 16), checker-side restatement: buffer 0 as wide as anyone reads it (at least x_cols); then,
 * group by group, buffers 1, 2, 3 in that order, each as wide as the furthest column one of the group's ops reads or writes; 16 rows per buffer, row
 * stride = width rounded up to 32 floats, + 4; a buffer that is first touched after an earlier (and at least as wide) buffer of its group was last
 * touched takes that buffer's place instead of new room; behind everything, room for the padded k-blocks of the layer that reads furthest.  The kernel's k-block
 * padding rule (qa_policy.hip mlp_kb) is restated here too. */
static int qo_mlp_kblocks(int k, int n) {
    const int tiles = (n + 15) / 16, per_wave = (tiles + 7) / 8, depth = per_wave > 2 ? 4 : 8, blocks = (k + 15) / 16;
    return (blocks + depth - 1) / depth * depth;
}
int qo_mlp_groups(const qa_mlp_op *ops, int32_t num_ops, int32_t x_cols, int32_t *strand_of, int32_t *base, int32_t *stride, int32_t *lds_floats) {
    if (!ops || !strand_of || !base || !stride || !lds_floats || num_ops <= 0 || num_ops > QA_MLP_MAX_OPS || x_cols <= 0 || x_cols > QA_MLP_BUF0_COLS) return QA_E_ARG;
    if (qo_mlp_strands(ops, num_ops, 2, strand_of) != 2) return 0;
    int far[2][4];
    memset(far, 0, sizeof(far));
    int far0 = x_cols;
    for (int i = 0; i < num_ops; ++i) {
        const int g = strand_of[i], reach_src = ops[i].src_col + (ops[i].kind == QA_MLP_LAYER ? ops[i].k : ops[i].n);
        if (ops[i].src_buf == 0) { if (reach_src > far0) far0 = reach_src; }
        else if (reach_src > far[g][ops[i].src_buf]) far[g][ops[i].src_buf] = reach_src;
        if (ops[i].dst_buf > 0 && ops[i].dst_col + ops[i].n > far[g][ops[i].dst_buf]) far[g][ops[i].dst_buf] = ops[i].dst_col + ops[i].n;
    }
    const int s0 = (far0 + 31) / 32 * 32 + 4;
    int at = 16 * s0;
    for (int g = 0; g < 2; ++g) {
        base[4 * g] = 0; stride[4 * g] = s0;
        int busy_until[4] = {0, 0, 0, 0};                      /* index of the last op that needs what lies in the buffer's region */
        for (int b = 1; b < 4; ++b) {
            stride[4 * g + b] = far[g][b] ? (far[g][b] + 31) / 32 * 32 + 4 : 0;
            int born = num_ops, dies = -1;
            for (int i = 0; i < num_ops; ++i)
                if (strand_of[i] == g && (ops[i].src_buf == b || ops[i].dst_buf == b)) { if (i < born) born = i; dies = i; }
            /* laid over the lowest-numbered buffer of the group that is at least as wide and whose contents nobody needs any more when this one is
             * first touched */
            int host = 0;
            for (int e = 1; e < b && !host; ++e)
                if (stride[4 * g + b] > 0 && stride[4 * g + e] >= stride[4 * g + b] && busy_until[e] < born) host = e;
            if (host) { base[4 * g + b] = base[4 * g + host]; busy_until[host] = dies; busy_until[b] = dies; }
            else { base[4 * g + b] = at; at += 16 * stride[4 * g + b]; busy_until[b] = dies; }
        }
    }
    for (int i = 0; i < num_ops; ++i) {
        if (ops[i].kind != QA_MLP_LAYER) continue;
        const int g = strand_of[i], b = ops[i].src_buf;
        const int last_read = base[4 * g + b] + 15 * stride[4 * g + b] + ops[i].src_col + 16 * qo_mlp_kblocks(ops[i].k, ops[i].n);
        if (last_read > at) at = last_read;
    }
    *lds_floats = (at + 3) / 4 * 4;
    return *lds_floats * 4 <= 160 * 1024 - 256;
}
static int qo_groups_setting = 2;
int qo_mlp_set_groups(int32_t groups) { const int prev = qo_groups_setting; qo_groups_setting = groups >= 2 ? 2 : 1; return prev; }

int qo_mlp_forward(const float *x, int64_t x_stride, int32_t rows, int32_t x_cols, const qa_mlp_op *ops, int32_t num_ops, const float *packed,
                   float *const *outs, const int64_t *out_strides, int32_t num_outs, void *stream) {
    (void)stream;
    static const int cols[4] = {QA_MLP_BUF0_COLS, QA_MLP_BUF1_COLS, QA_MLP_BUF2_COLS, QA_MLP_BUF3_COLS};
    if (!x || !ops || !packed || rows <= 0 || x_cols <= 0 || x_cols > cols[0] || num_ops <= 0 || num_ops > QA_MLP_MAX_OPS || num_outs < 0 ||
        num_outs > QA_MLP_MAX_OUTPUTS) return QA_E_ARG;
    float *buf[4];
    for (int b = 0; b < 4; ++b) buf[b] = (float *)malloc(sizeof(float) * (size_t)cols[b]);
    int rc = QA_OK;
    for (int64_t r = 0; r < rows && rc == QA_OK; ++r) {
        for (int b = 0; b < 4; ++b) memset(buf[b], 0, sizeof(float) * (size_t)cols[b]);
        memcpy(buf[0], x + r * x_stride, sizeof(float) * (size_t)x_cols);
        for (int i = 0; i < num_ops; ++i) {
            const qa_mlp_op *o = &ops[i];
            if (o->src_buf < 0 || o->src_buf > 3 || o->dst_buf > 3 || o->n <= 0) { rc = QA_E_ARG; break; }
            const int save = (o->flags & QA_MLP_F_SAVE) != 0;
            if (save && (o->out_index < 0 || o->out_index >= num_outs || !outs || !outs[o->out_index] || out_strides[o->out_index] < o->out_col + o->n)) { rc = QA_E_ARG; break; }
            const int deriv = (o->kind == QA_MLP_LAYER || o->kind == QA_MLP_GRAD) && o->act >= 4;
            if (deriv && (o->act > 6 || o->aux_index < 0 || o->aux_index >= num_outs || !outs || !outs[o->aux_index] || out_strides[o->aux_index] < o->aux_col + o->n)) { rc = QA_E_ARG; break; }
            const float *ysaved = deriv ? outs[o->aux_index] + r * out_strides[o->aux_index] + o->aux_col : NULL;
            if (o->kind == QA_MLP_LOAD) {                                 /* ABI 17: global -> scratch buffer */
                if (o->dst_buf < 1 || o->dst_col + o->n > cols[o->dst_buf] || o->aux_index < 0 || o->aux_index >= num_outs || !outs || !outs[o->aux_index] ||
                    out_strides[o->aux_index] < o->aux_col + o->n || o->act != 0) { rc = QA_E_ARG; break; }
                memcpy(buf[o->dst_buf] + o->dst_col, outs[o->aux_index] + r * out_strides[o->aux_index] + o->aux_col, sizeof(float) * (size_t)o->n);
                if (save) memcpy(outs[o->out_index] + r * out_strides[o->out_index] + o->out_col, buf[o->dst_buf] + o->dst_col, sizeof(float) * (size_t)o->n);
                continue;
            }
            if (o->kind == QA_MLP_COPY || o->kind == QA_MLP_GRAD) {       /* ABI 17: QA_MLP_GRAD = copy (+ dst) x act'(saved y) */
                if (o->dst_buf < 1 || o->src_col + o->n > cols[o->src_buf] || o->dst_col + o->n > cols[o->dst_buf] ||
                    (o->kind == QA_MLP_GRAD && o->act != 0 && !deriv)) { rc = QA_E_ARG; break; }
                if (o->kind == QA_MLP_COPY) memmove(buf[o->dst_buf] + o->dst_col, buf[o->src_buf] + o->src_col, sizeof(float) * (size_t)o->n);
                else
                    for (int c = 0; c < o->n; ++c) {
                        double v = buf[o->src_buf][o->src_col + c];
                        if (o->flags & QA_MLP_F_ADD) v = (double)(float)v + (double)buf[o->dst_buf][o->dst_col + c];
                        float f = (float)v;
                        if (deriv) { const float y = ysaved[c]; f *= o->act == 4 ? (y > 0.0f ? 1.0f : y + 1.0f) : o->act == 5 ? (y > 0.0f ? 1.0f : 0.0f) : (1.0f - y * y); }
                        buf[o->dst_buf][o->dst_col + c] = f;
                    }
                if (save) memcpy(outs[o->out_index] + r * out_strides[o->out_index] + o->out_col, buf[o->dst_buf] + o->dst_col, sizeof(float) * (size_t)o->n);
                continue;
            }
            if (o->kind != QA_MLP_LAYER || o->src_col + o->k > cols[o->src_buf] || o->dst_buf == o->src_buf || o->dst_buf == 0 ||
                (o->dst_buf > 0 && o->dst_col + o->n > cols[o->dst_buf]) ||
                (o->dst_buf < 0 && (o->out_index < 0 || o->out_index >= num_outs || !outs || !outs[o->out_index]))) { rc = QA_E_ARG; break; }
            const float *w = packed + o->w_off, *bias = packed + o->b_off, *in = buf[o->src_buf] + o->src_col;
            for (int c = 0; c < o->n; ++c) {
                double acc = bias[c];
                for (int k = 0; k < o->k; ++k) acc += (double)w[(int64_t)c * o->k + k] * (double)in[k];
                float v = (float)acc;
                if (o->act == 1) v = v > 0.0f ? v : (float)(exp((double)v) - 1.0);
                else if (o->act == 2) v = v > 0.0f ? v : 0.0f;
                else if (o->act == 3) v = (float)tanh((double)v);
                else if (deriv) { const float y = ysaved[c]; v = (float)(acc * (o->act == 4 ? (y > 0.0f ? 1.0 : (double)y + 1.0) : o->act == 5 ? (y > 0.0f ? 1.0 : 0.0) : 1.0 - (double)y * y)); }
                if (o->dst_buf > 0) buf[o->dst_buf][o->dst_col + c] = v;
                if (o->dst_buf < 0 || save) outs[o->out_index][r * out_strides[o->out_index] + o->out_col + c] = v;
            }
        }
    }
    for (int b = 0; b < 4; ++b) free(buf[b]);
    return rc;
}

/* ---- debug entry points used only by the physics known-answer tests ---- */
/* mass matrix and bias for a configuration: ub = base twist (w; v) in the base frame */
int qo_debug_dynamics(const float q[12], const float qd[12], const float ub[6], const float quat[4], double Mout[18 * 18], double hout[18]) {
    Kin K; double qq[12], qdd[12], u[6];
    for (int j = 0; j < 12; ++j) { qq[j] = q[j]; qdd[j] = qd[j]; }
    for (int i = 0; i < 6; ++i) u[i] = ub[i];
    for (int l = 0; l < 4; ++l) leg_kin(l, qq + 3 * l, &K);
    v3 c0 = {QA_BASE_COM[0], QA_BASE_COM[1], QA_BASE_COM[2]};
    m3 Ic = {{QA_BASE_I[0], QA_BASE_I[3], QA_BASE_I[4]}, {QA_BASE_I[3], QA_BASE_I[1], QA_BASE_I[5]}, {QA_BASE_I[4], QA_BASE_I[5], QA_BASE_I[2]}};
    RB base; rb_make(QA_BASE_MASS, c0, Ic, &base);
    double qd4[4] = {quat[0], quat[1], quat[2], quat[3]}; m3 R; quat_to_mat(qd4, R);
    v3 gw = {0, 0, -9.81}, gB; mtv(R, gw, gB);
    double M[18][18], h[18];
    dynamics_terms(&K, &base, u, qdd, gB, M, h);
    memcpy(Mout, M, sizeof(M)); memcpy(hout, h, sizeof(h));
    return QA_OK;
}
/* per-body kinematics (CoM position and velocity in the base frame, mass, inertia) for an independent energy check */
int qo_debug_bodies(const float q[12], const float qd[12], const float ub[6], double com[13 * 3], double vel[13 * 3], double omg[13 * 3], double mass[13], double Icom[13 * 9]) {
    Kin K; double qq[12];
    for (int j = 0; j < 12; ++j) qq[j] = q[j];
    for (int l = 0; l < 4; ++l) leg_kin(l, qq + 3 * l, &K);
    for (int i = 0; i < 3; ++i) { com[i] = QA_BASE_COM[i]; omg[i] = ub[i]; }
    { v3 w = {ub[0], ub[1], ub[2]}, c = {com[0], com[1], com[2]}, t; cross(w, c, t); for (int i = 0; i < 3; ++i) vel[i] = ub[3 + i] + t[i]; }
    mass[0] = QA_BASE_MASS;
    { double I9[9] = {QA_BASE_I[0], QA_BASE_I[3], QA_BASE_I[4], QA_BASE_I[3], QA_BASE_I[1], QA_BASE_I[5], QA_BASE_I[4], QA_BASE_I[5], QA_BASE_I[2]}; memcpy(Icom, I9, sizeof(I9)); }
    for (int l = 0; l < 4; ++l) {
        v3 w = {ub[0], ub[1], ub[2]};
        /* velocity of a point p on link k: v_B + w_B x p + sum_j qd_j a_j x (p - o_j) */
        for (int k = 0; k < 3; ++k) {
            int b = 1 + 3 * l + k;
            v3 cl = {QA_LINK_COM[l][k][0], QA_LINK_COM[l][k][1], QA_LINK_COM[l][k][2]}, c;
            mv(K.Rl[l][k], cl, c); for (int i = 0; i < 3; ++i) c[i] += K.o[l][k][i];
            v3 v, t, wk = {w[0], w[1], w[2]};
            cross(w, c, t); for (int i = 0; i < 3; ++i) v[i] = ub[3 + i] + t[i];
            for (int j = 0; j <= k; ++j) {
                v3 rel = {c[0] - K.o[l][j][0], c[1] - K.o[l][j][1], c[2] - K.o[l][j][2]};
                cross(K.a[l][j], rel, t);
                for (int i = 0; i < 3; ++i) { v[i] += qd[3 * l + j] * t[i]; wk[i] += qd[3 * l + j] * K.a[l][j][i]; }
            }
            const float *I6 = QA_LINK_I[l][k];
            m3 Il = {{I6[0], I6[3], I6[4]}, {I6[3], I6[1], I6[5]}, {I6[4], I6[5], I6[2]}}, Rt, Ib;
            for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) Rt[i][j] = K.Rl[l][k][j][i];
            mm(K.Rl[l][k], Il, Ib); mm(Ib, Rt, Ib);
            for (int i = 0; i < 3; ++i) { com[3 * b + i] = c[i]; vel[3 * b + i] = v[i]; omg[3 * b + i] = wk[i]; }
            mass[b] = QA_LINK_MASS[l][k];
            for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) Icom[9 * b + 3 * i + j] = Ib[i][j];
        }
    }
    return QA_OK;
}

/* ---- debug entry points used by the golden-vector tests (tests/test_golden_env.py) ---- */
/* legged_robot.py:84-106 only: action-history roll, delay, clip, decimation x (PD torque + physics).  Leaves the arena
 * in the state the reference has when it enters post_physics_step (after the refresh_* calls, :129-131). */
int qo_debug_pre_physics(qo_sim *s, const float *actions, int32_t delay_steps) {
    const qa_config *c = &s->cfg;
    for (int e = 0; e < c->num_envs; ++e) {
        float *ah = TP(s, QA_T_ACTION_HISTORY, float) + 96 * e;
        memmove(ah, ah + 12, 7 * 12 * 4);
        memcpy(ah + 7 * 12, actions + 12 * e, 48);
        const float *src = ah + 12 * (QA_ACTION_BUF_LEN - 1 - delay_steps);
        float *act = TP(s, QA_T_ACTIONS, float) + 12 * e;
        float clipa = c->clip_actions / c->action_scale;
        for (int j = 0; j < 12; ++j) act[j] = clipf(src[j], -clipa, clipa);
        float *tau = TP(s, QA_T_TORQUES, float) + 12 * e, *torg = TP(s, QA_T_TORQUES_ORG, float) + 12 * e;
        obstacles_begin_step(s, e);
        StepAnchor an = {(double)(TP(s, QA_T_ROOT_STATES, float) + 13 * e)[0], (double)(TP(s, QA_T_ROOT_STATES, float) + 13 * e)[1], 0.0, 0.0};
        for (int d = 0; d < c->decimation; ++d) { compute_torques(s, e, act, tau, torg); phys_substep_acc(s, e, tau, 1, &an); }
        obstacles_end_step(s, e);
    }
    return QA_OK;
}
/* legged_robot.py:124-166 only, on whatever state is in the arena */
int qo_debug_post_physics(qo_sim *s, int64_t step) {
    memset(TP(s, QA_T_EPISODE_STATS, float) + 16 * ((step + 1) & 1), 0, 64);
    for (int e = 0; e < s->cfg.num_envs; ++e) { float tmp[QA_NUM_OBS_DISC]; post_physics(s, e, step, tmp); }
    return QA_OK;
}
/* the mocap reset's frame sampling alone, for given uniforms (tests/test_mocap_reset.py against the reference's own frames):
 * out = root state 13 (origin not added) | joint positions 12 | joint velocities 12 */
int qo_debug_mocap_reset(qo_sim *s, int gait, float u0, float u1, float out[37]) {
    if (!s || gait < 0 || gait >= QA_NUM_GAITS || s->mocap_first[QA_NUM_GAITS] <= 0) return QA_E_ARG;
    float f[QA_MOCAP_FRAME];
    mocap_sample(s, gait, u0, u1, f);
    for (int i = 0; i < 7; ++i) out[i] = f[i];
    quat_rotate_f(f + 3, f + 19, +1.0f, out + 7);
    quat_rotate_f(f + 3, f + 22, +1.0f, out + 10);
    for (int j = 0; j < 12; ++j) { out[13 + j] = f[7 + j]; out[25 + j] = f[25 + j]; }
    return QA_OK;
}
/* legged_robot.py:547-579 on the current dof state */
int qo_debug_torques(qo_sim *s, const float *actions, float *tau, float *tau_org) {
    for (int e = 0; e < s->cfg.num_envs; ++e) compute_torques(s, e, actions + 12 * e, tau + 12 * e, tau_org + 12 * e);
    return QA_OK;
}

/* ================================================================================================================
 * Task-level (TSC) env-side math: CPU twins of qa_tsc_set_commands / qa_tsc_goal_step (include/qa_sim.h), following
 * tsc/legged_gym/envs/base/legged_robot.py:699-760 (set_commands) and :204-262, 322-346, 412-430, 1779-1925 (goal update,
 * termination, the 8 active rewards).  Pinned by tests/golden/tsc_env.npz, which tools/gen_golden_tsc_env.py makes by running
 * the reference's own set_commands / post_physics_step on CPU.
 * ================================================================================================================ */
int qo_tsc_set_commands(const float *actions, const int64_t *episode_length, int64_t num_envs, int32_t num_d, int32_t num_c, int32_t dim_c,
                        int32_t interval, const int32_t *mocap_index, const float *vel_ranges, const float *jump_range,
                        const float *height_range, const float *noise, float *commands, float *latent_eps, float *latent_c,
                        float *next_commands, void *stream) {
    (void)stream;
    if (!actions || !episode_length || !mocap_index || !vel_ranges || !jump_range || !height_range || !commands || !latent_eps ||
        !latent_c || !next_commands || num_envs <= 0) return QA_E_ARG;
    if (num_c != 6 || dim_c < 1 || dim_c > 8 || num_d < 1 || num_d > 8 || interval < 1) return QA_E_ARG;
    for (int i = 0; i < num_d; ++i) if (mocap_index[i] < 0 || mocap_index[i] >= dim_c) return QA_E_ARG;
    const int width = 1 + num_d * num_c;
    for (int64_t e = 0; e < num_envs; ++e) {
        float *cmd = commands + e * 5, *lc = latent_c + e * dim_c;
        if (episode_length[e] % interval == 0) {                                   /* :700-701 */
            const float *row = actions + e * width;
            int id = (int)row[0];                                                  /* :702 */
            if (id < 0) id = 0;
            if (id >= num_d) id = num_d - 1;
            const int g = mocap_index[id];                                         /* :704 */
            float u[6];
            for (int k = 0; k < 6; ++k) {
                const float p = clipf(row[1 + id * num_c + k], -1.0f, 1.0f);       /* :705-711 */
                if (k == 5) latent_eps[e] = p;                                     /* :716 */
                u[k] = (p + 1.0f) / 2.0f;                                          /* :718 */
            }
            for (int k = 0; k < dim_c; ++k) lc[k] = k == g ? 1.0f : 0.0f;          /* :713-714 */
            for (int k = 0; k < 3; ++k) {                                          /* :740-742 */
                const float lo = vel_ranges[(k * dim_c + g) * 2], hi = vel_ranges[(k * dim_c + g) * 2 + 1];
                cmd[k] = lo + (hi - lo) * u[k];
            }
            const int jump = g == dim_c - 1;                                       /* :749 */
            cmd[3] = (jump_range[0] + (jump_range[1] - jump_range[0]) * u[3]) * (jump ? 1.0f : 0.0f);
            cmd[4] = (height_range[0] + (height_range[1] - height_range[0]) * u[4]) * (jump ? 0.0f : 1.0f);
        }
        if (noise) for (int k = 0; k < 5; ++k) cmd[k] *= noise[e * 5 + k];         /* :756-759, every env */
        float *nx = next_commands + e * (6 + dim_c);                               /* :760 */
        for (int k = 0; k < 5; ++k) nx[k] = cmd[k];
        nx[5] = latent_eps[e];
        for (int k = 0; k < dim_c; ++k) nx[6 + k] = lc[k];
    }
    return QA_OK;
}

static void tsc_rotate_inverse(const float q[4], const float v[3], float out[3]) {
    const float w = q[3], s = 2.0f * w * w - 1.0f;
    const float cx = q[1] * v[2] - q[2] * v[1], cy = q[2] * v[0] - q[0] * v[2], cz = q[0] * v[1] - q[1] * v[0];
    const float d = q[0] * v[0] + q[1] * v[1] + q[2] * v[2];
    out[0] = v[0] * s - cx * w * 2.0f + q[0] * d * 2.0f;
    out[1] = v[1] * s - cy * w * 2.0f + q[1] * d * 2.0f;
    out[2] = v[2] * s - cz * w * 2.0f + q[2] * d * 2.0f;
}
static float tsc_norm3(const float *f) { return sqrtf(f[0] * f[0] + f[1] * f[1] + f[2] * f[2]); }
static float tsc_floor_mod(float a, float b) {
    float r = fmodf(a, b);
    if (r != 0.0f && ((r < 0.0f) != (b < 0.0f))) r += b;
    return r;
}

int qo_tsc_goal_step(const qa_tsc_goal_cfg *c, const qa_tsc_goal_io *io, void *stream) {
    (void)stream;
    if (!c || !io || c->num_envs <= 0) return QA_E_ARG;
    const int64_t N = c->num_envs;
    for (int64_t e = 0; e < N; ++e) {
        const float *rs = io->root_states + e * 13;
        const float *cf = io->contact_forces + e * c->num_bodies * 3;
        const int64_t ep_len = ++io->episode_length[e];                            /* :236 */
        const float q[4] = {rs[3], rs[4], rs[5], rs[6]}, grav[3] = {0.0f, 0.0f, -1.0f};
        tsc_rotate_inverse(q, rs + 7, io->base_lin_vel + e * 3);                   /* :240-243 */
        tsc_rotate_inverse(q, rs + 10, io->base_ang_vel + e * 3);
        tsc_rotate_inverse(q, grav, io->projected_gravity + e * 3);
        const float x = q[0], y = q[1], z = q[2], w = q[3];                        /* :32-55 */
        const float roll = atan2f(2.0f * (w * x + y * z), 1.0f - 2.0f * (x * x + y * y));
        const float pitch = asinf(clipf(2.0f * (w * y - z * x), -1.0f, 1.0f));
        const float yaw = atan2f(2.0f * (w * z + x * y), 1.0f - 2.0f * (y * y + z * z));
        io->rpy[e * 3] = roll; io->rpy[e * 3 + 1] = pitch; io->rpy[e * 3 + 2] = yaw;
        int filt[4];
        for (int f = 0; f < 4; ++f) {                                              /* :247-249 */
            const int now = tsc_norm3(cf + c->feet_bodies[f] * 3) > 2.0f;
            filt[f] = now || io->last_contacts[e * 4 + f];
            io->last_contacts[e * 4 + f] = (uint8_t)now;
            io->contact_filt[e * 4 + f] = (uint8_t)filt[f];
        }
        /* _update_goals :204-224 */
        int64_t gi = io->cur_goal_idx[e];
        float timer = io->reach_goal_timer[e];
        if (timer > c->reach_goal_delay_steps) { gi += 1; timer = 0.0f; }
        const float gx = io->cur_goals[e * 3], gy = io->cur_goals[e * 3 + 1];
        const float nx = io->next_goals[e * 3], ny = io->next_goals[e * 3 + 1];
        const float dxr = rs[0] - gx, dyr = rs[1] - gy;
        const float dist = sqrtf(dxr * dxr + dyr * dyr);
        const int reached = dist < c->next_goal_threshold, leave = dist > c->leave_goal_threshold;
        if (reached) timer += 1.0f;
        io->cur_goal_idx[e] = gi; io->reach_goal_timer[e] = timer; io->reached_goal[e] = (uint8_t)reached;
        const float tx = gx - rs[0], ty = gy - rs[1];
        const float tn = sqrtf(tx * tx + ty * ty);
        const float tvx = tx / (tn + 1e-5f), tvy = ty / (tn + 1e-5f);
        const float target_yaw = atan2f(tvy, tvx);
        const float ux = nx - rs[0], uy = ny - rs[1];
        const float un = sqrtf(ux * ux + uy * uy);
        const float next_yaw = atan2f(uy / (un + 1e-5f), ux / (un + 1e-5f));
        io->target_pos_rel[e * 2] = tx; io->target_pos_rel[e * 2 + 1] = ty;
        io->next_target_pos_rel[e * 2] = ux; io->next_target_pos_rel[e * 2 + 1] = uy;
        io->target_yaw[e] = target_yaw; io->next_target_yaw[e] = next_yaw;
        /* :255-258 */
        int64_t gclamp = gi < 0 ? 0 : gi;
        const int64_t gmax = c->num_goal_slots - c->last_goal_repeat - 1;
        if (gclamp > gmax) gclamp = gmax;
        int ob = (int)(gclamp / c->goals_per_obstacle);
        if (ob >= c->num_obstacles) ob = c->num_obstacles - 1;
        const int64_t otype = io->obstacle_types[e * c->num_obstacles + ob];
        io->cur_obstacle_type[e] = otype;
        /* check_termination :322-346 */
        int reset = 0;
        for (int k = 0; k < c->num_termination_bodies; ++k) reset |= tsc_norm3(cf + c->termination_bodies[k] * 3) > 1.0f;
        const int goal_cut = gi >= (int64_t)(c->num_goal_slots - c->last_goal_repeat);
        const int time_out = ((float)ep_len > c->max_episode_length) || goal_cut;
        reset |= time_out || fabsf(roll) > 1.5f || fabsf(pitch) > 1.5f || rs[2] < -0.25f || leave;
        if (c->use_camera) {
            const float *lg = io->env_goals + (e * c->num_goal_slots + (c->num_goal_slots - c->last_goal_repeat)) * 3;
            const float lx = rs[0] - lg[0], ly = rs[1] - lg[1];
            reset |= sqrtf(lx * lx + ly * ly) < c->next_goal_threshold;
        }
        io->reset_buf[e] = (uint8_t)reset; io->time_out_buf[e] = (uint8_t)time_out; io->reach_goal_cutoff[e] = (uint8_t)goal_cut;
        /* rewards :1779-1925, order and clipping of compute_reward :412-430 */
        float term[QA_TSC_NUM_REWARDS];
        term[QA_TSC_REW_ACTION_HL_RATE] = 0.0f; term[QA_TSC_REW_LATENT_C_RATE] = 0.0f;
        if (io->action_hl_history) {
            const float *h = io->action_hl_history + e * c->history_len * c->history_width;
            const float *h1 = h + (c->history_len - 1) * c->history_width, *h2 = h + (c->history_len - 2) * c->history_width,
                        *h3 = h + (c->history_len - 3) * c->history_width;
            float ss = 0.0f;
            for (int k = 0; k < c->history_width; ++k) { const float d = h2[k] - h1[k]; ss += d * d; }
            term[QA_TSC_REW_ACTION_HL_RATE] = sqrtf(ss);
            term[QA_TSC_REW_LATENT_C_RATE] = 0.5f * (fabsf(h3[0] - h1[0]) + fabsf(h2[0] - h1[0]));
        }
        int hits = 0;
        for (int k = 0; k < c->num_penalised_bodies; ++k) hits += tsc_norm3(cf + c->penalised_bodies[k] * 3) > 0.1f;
        term[QA_TSC_REW_COLLISION] = (float)hits;
        int edge = 0;
        for (int f = 0; f < 4; ++f) {
            const float *fp = io->rigid_body_states + (e * c->num_bodies + c->feet_bodies[f]) * 13;
            int64_t ix = (int64_t)rintf((fp[0] + c->border_size) / c->horizontal_scale);
            int64_t iy = (int64_t)rintf((fp[1] + c->border_size) / c->horizontal_scale);
            if (ix < 0) ix = 0; if (ix > c->mask_rows - 1) ix = c->mask_rows - 1;
            if (iy < 0) iy = 0; if (iy > c->mask_cols - 1) iy = c->mask_cols - 1;
            edge += (filt[f] && io->x_edge_mask[ix * c->mask_cols + iy]) ? 1 : 0;
        }
        term[QA_TSC_REW_FEET_EDGE] = (float)edge;
        term[QA_TSC_REW_REACH_GOAL] = reached ? 1.0f : 0.0f;
        const float vt = (otype == 0 || otype == 4) ? 2.5f : c->target_lin_vel;
        term[QA_TSC_REW_TRACKING_GOAL_VEL] = fminf(tvx * rs[7] + tvy * rs[8], vt) / (vt + 1e-5f);
        const float PI_F = 3.14159265358979323846f;
        const float dyaw = tsc_floor_mod((target_yaw - yaw) + PI_F, 2.0f * PI_F) - PI_F;
        term[QA_TSC_REW_TRACKING_YAW] = expf(-fabsf(dyaw));
        term[QA_TSC_REW_TERMINATION] = (reset && !time_out) ? 1.0f : 0.0f;
        float total = 0.0f;
        for (int k = 0; k < QA_TSC_NUM_REWARDS - 1; ++k) {
            const float r = term[k] * c->reward_scales[k];
            total += r;
            io->episode_sums[k * N + e] += r;
        }
        total = fmaxf(total, 0.0f);
        const float rt = term[QA_TSC_REW_TERMINATION] * c->reward_scales[QA_TSC_REW_TERMINATION];
        total += rt;
        io->episode_sums[(int64_t)QA_TSC_REW_TERMINATION * N + e] += rt;
        io->rew_buf[e] = total;
        /* :272-273 */
        int64_t g0 = gi < 0 ? 0 : (gi > c->num_goal_slots - 1 ? c->num_goal_slots - 1 : gi);
        int64_t g1 = gi + 1 < 0 ? 0 : (gi + 1 > c->num_goal_slots - 1 ? c->num_goal_slots - 1 : gi + 1);
        const float *eg = io->env_goals + e * c->num_goal_slots * 3;
        for (int k = 0; k < 3; ++k) { io->cur_goals[e * 3 + k] = eg[g0 * 3 + k]; io->next_goals[e * 3 + k] = eg[g1 * 3 + k]; }
    }
    return QA_OK;
}

/* CPU twin of qa_tsc_observations: _get_heights (tsc/legged_gym/envs/base/legged_robot.py:1708-1755), compute_observations
 * (:432-515), compute_flat_key_pos (:1929-1947).  Pinned by tests/golden/tsc_env.npz (`obs_*` arrays). */
int qo_tsc_observations(const qa_tsc_obs_cfg *c, const qa_tsc_obs_io *io, void *stream) {
    (void)stream;
    if (!c || !io || c->num_envs <= 0 || c->map_rows < 2 || c->map_cols < 2 || c->point_stride < 2) return QA_E_ARG;
    const int64_t N = c->num_envs;
    const float PI_F = 3.14159265358979323846f, cl = c->clip_observations;
    for (int64_t e = 0; e < N; ++e) {
        const float *rs = io->root_states + e * 13;
        float row[QA_TSC_NUM_OBS], prop[QA_TSC_NUM_PROPRIO], priv[33], *meas = io->measured_heights + e * QA_TSC_NUM_SCAN;
        /* proprio :463-469 */
        prop[0] = io->rpy[e * 3]; prop[1] = io->rpy[e * 3 + 1];
        for (int k = 0; k < 3; ++k) prop[2 + k] = io->base_ang_vel[e * 3 + k] * c->ang_vel;
        for (int k = 0; k < 12; ++k) {
            prop[5 + k] = (io->dof_pos[e * 12 + k] - c->default_dof_pos_all[k]) * c->dof_pos;
            prop[17 + k] = io->dof_vel[e * 12 + k] * c->dof_vel;
            prop[29 + k] = io->last_action[e * c->action_stride + k];
            prop[45 + k] = 0.0f;
        }
        for (int k = 0; k < 4; ++k) prop[41 + k] = (io->contact_filt[e * 4 + k] ? 1.0f : 0.0f) - 0.5f;
        /* delta yaws :445-452 */
        if (c->update_yaw) {
            io->delta_yaw[e] = tsc_floor_mod((io->target_yaw[e] - io->rpy[e * 3 + 2]) + PI_F, 2.0f * PI_F) - PI_F;
            io->delta_next_yaw[e] = tsc_floor_mod((io->next_target_yaw[e] - io->rpy[e * 3 + 2]) + PI_F, 2.0f * PI_F) - PI_F;
        }
        /* scan :1708-1755 */
        const float qn = fmaxf(sqrtf(rs[5] * rs[5] + rs[6] * rs[6]), 1e-9f);
        const float qz = rs[5] / qn, qw = rs[6] / qn;
        for (int p = 0; p < QA_TSC_NUM_SCAN; ++p) {
            const float *hp = io->height_points + e * c->points_env_stride + p * c->point_stride;
            const float bx = hp[0], by = hp[1];
            const float t0 = (0.0f - qz * by) * 2.0f, t1 = (qz * bx - 0.0f) * 2.0f;
            const float wx = (bx + qw * t0) + (0.0f - qz * t1) + rs[0];
            const float wy = (by + qw * t1) + (qz * t0 - 0.0f) + rs[1];
            int64_t px = (int64_t)((wx + c->border_size) / c->horizontal_scale), py = (int64_t)((wy + c->border_size) / c->horizontal_scale);
            if (px < 0) px = 0; if (px > c->map_rows - 2) px = c->map_rows - 2;
            if (py < 0) py = 0; if (py > c->map_cols - 2) py = c->map_cols - 2;
            int16_t hm = io->height_samples[px * c->map_cols + py];
            const int16_t h2 = io->height_samples[(px + 1) * c->map_cols + py], h3 = io->height_samples[px * c->map_cols + py + 1];
            if (h2 < hm) hm = h2;
            if (h3 < hm) hm = h3;
            meas[p] = (float)hm * c->vertical_scale;
        }
        const float root_h = rs[2] - meas[QA_TSC_NUM_SCAN / 2 + 1];                 /* :436-438 */
        /* privileged values :471-478 */
        priv[0] = c->root_height_obs ? root_h : 0.0f;
        for (int k = 0; k < 3; ++k) priv[1 + k] = io->base_lin_vel[e * 3 + k] * c->lin_vel;
        for (int k = 0; k < 4; ++k) priv[4 + k] = io->mass_params[e * 4 + k];
        priv[8] = io->friction[e];
        for (int k = 0; k < 12; ++k) { priv[9 + k] = io->motor_strength[e * 12 + k] - 1.0f; priv[21 + k] = io->motor_strength[(N + e) * 12 + k] - 1.0f; }
        /* discriminator view :456-461 with compute_flat_key_pos */
        float *d = io->obs_disc_buf + e * QA_TSC_NUM_OBS_DISC;
        d[0] = prop[0]; d[1] = prop[1]; d[2] = root_h;
        for (int k = 0; k < 3; ++k) { d[3 + k] = io->base_lin_vel[e * 3 + k] * c->lin_vel_dist; d[6 + k] = io->base_ang_vel[e * 3 + k] * c->ang_vel_dist; }
        for (int k = 0; k < 12; ++k) { d[9 + k] = (io->dof_pos[e * 12 + k] - c->default_dof_pos[k]) * c->dof_pos; d[21 + k] = io->dof_vel[e * 12 + k] * c->dof_vel; }
        {
            const float *q = rs + 3;
            const float s = 2.0f * q[3] * q[3] - 1.0f;
            const float hx = s + q[0] * q[0] * 2.0f, hy = q[2] * q[3] * 2.0f + q[1] * q[0] * 2.0f;
            const float half = -atan2f(hy, hx) / 2.0f;
            float hz = sinf(half), hw = cosf(half);
            const float hn = fmaxf(sqrtf(hz * hz + hw * hw), 1e-9f);
            hz /= hn; hw /= hn;
            const float s2 = 2.0f * hw * hw - 1.0f;
            for (int kb = 0; kb < 4; ++kb) {
                const float *bp = io->rigid_body_states + (e * c->num_bodies + c->key_bodies[kb]) * 13;
                const float lx = bp[0] - rs[0], ly = bp[1] - rs[1], lz = bp[2] - rs[2];
                d[33 + kb * 3 + 0] = (lx * s2 + (0.0f - hz * ly) * hw * 2.0f) * c->key_pos;
                d[33 + kb * 3 + 1] = (ly * s2 + (hz * lx - 0.0f) * hw * 2.0f) * c->key_pos;
                d[33 + kb * 3 + 2] = (lz * s2 + hz * (hz * lz) * 2.0f) * c->key_pos;
            }
        }
        for (int k = 0; k < 4; ++k) d[45 + k] = (io->contact_filt[e * 4 + k] ? 1.0f : 0.0f) * c->foot_contact;
        /* rows :479-495, history BEFORE the push */
        float *hist = io->obs_history + e * 570;
        memcpy(row, prop, sizeof(prop));
        row[57] = io->delta_yaw[e]; row[58] = io->delta_next_yaw[e];
        for (int k = 0; k < QA_TSC_NUM_OBSTACLE_CLASSES; ++k) row[59 + k] = io->cur_obstacle_type[e] == k ? 1.0f : 0.0f;
        for (int p = 0; p < QA_TSC_NUM_SCAN; ++p) row[65 + p] = clipf((rs[2] - 0.3f) - meas[p], -1.0f, 1.0f);
        memcpy(row + 197, priv, sizeof(priv));
        memcpy(row + 230, hist, 570 * sizeof(float));
        float *ob = io->obs_buf + e * QA_TSC_NUM_OBS, *bb = io->obs_bbc_buf + e * QA_TSC_NUM_OBS_BBC;
        for (int i = 0; i < QA_TSC_NUM_OBS; ++i) ob[i] = clipf(row[i], -cl, cl);
        for (int i = 0; i < 57; ++i) bb[i] = clipf(prop[i], -cl, cl);
        for (int i = 0; i < 33; ++i) bb[57 + i] = clipf(priv[i], -cl, cl);
        for (int i = 0; i < 570; ++i) bb[90 + i] = clipf(hist[i], -cl, cl);
        for (int k = 0; k < 5; ++k) { bb[660 + k] = clipf(io->commands[e * 5 + k], -cl, cl); bb[666 + k] = clipf(io->latent_c[e * 5 + k], -cl, cl); }
        bb[665] = clipf(io->latent_eps[e], -cl, cl);
        /* history push :497-505, clipped :511 */
        if (io->episode_length[e] <= 1) {
            for (int s = 0; s < QA_TSC_HISTORY_LEN; ++s) for (int j = 0; j < 57; ++j) hist[s * 57 + j] = clipf(prop[j], -cl, cl);
        } else {
            memmove(hist, hist + 57, 9 * 57 * sizeof(float));
            for (int i = 0; i < 9 * 57; ++i) hist[i] = clipf(hist[i], -cl, cl);
            for (int j = 0; j < 57; ++j) hist[9 * 57 + j] = clipf(prop[j], -cl, cl);
        }
    }
    return QA_OK;
}

/* CPU twin of qa_tsc_depth_update (include/qa_sim.h): the vision student's depth camera, ray-cast against the course's height field
 * and ceiling field; update_depth_buffer + process_depth_image (tsc/legged_gym/envs/base/legged_robot.py:154-200), camera of
 * attach_camera (:1203-1226).  PARITY UNPINNED against the reference's images: those come out of Isaac Gym's rasteriser (absent
 * here) looking at obstacle meshes; this restates the camera model, the crop / clip / normalise / noise / ring arithmetic, and is
 * pinned by analytic scenes (tests/test_tsc_depth.py). */
static double depth_surface(const int16_t *m, int rows, int cols, double border, double hs, double vs, double x, double y, int *exists) {
    double fx = (x + border) / hs, fy = (y + border) / hs;
    int ix = (int)floor(fx), iy = (int)floor(fy);
    if (ix < 0) ix = 0; if (ix > rows - 2) ix = rows - 2;
    if (iy < 0) iy = 0; if (iy > cols - 2) iy = cols - 2;
    double u = fx - ix, v = fy - iy;
    if (u < 0) u = 0; if (u > 1) u = 1; if (v < 0) v = 0; if (v > 1) v = 1;
    int16_t s00 = m[(int64_t)ix * cols + iy], s10 = m[(int64_t)(ix + 1) * cols + iy], s01 = m[(int64_t)ix * cols + iy + 1], s11 = m[(int64_t)(ix + 1) * cols + iy + 1];
    double h00 = vs * s00, h10 = vs * s10, h01 = vs * s01, h11 = vs * s11;
    if (u >= v) { if (exists) *exists = s00 != QA_NO_CEILING && s10 != QA_NO_CEILING && s11 != QA_NO_CEILING; return h00 + u * (h10 - h00) + v * (h11 - h10); }
    if (exists) *exists = s00 != QA_NO_CEILING && s01 != QA_NO_CEILING && s11 != QA_NO_CEILING;
    return h00 + v * (h01 - h00) + u * (h11 - h01);
}
int qo_tsc_depth_update(const qa_tsc_depth_cfg *c, const qa_tsc_depth_io *io, void *stream) {
    (void)stream;
    if (!c || !io || !io->root_states || !io->camera_pitch || !io->height_samples || !io->episode_length || !io->depth_buffer) return QA_E_ARG;
    const int Wc = c->width - c->crop_left - c->crop_right, Hc = c->height - c->crop_top - c->crop_bottom;
    if (c->num_envs <= 0 || Wc <= 0 || Hc <= 0 || c->buffer_len < 1 || c->map_rows < 2 || c->map_cols < 2 || !(c->horizontal_scale > 0) ||
        !(c->far_clip > c->near_clip) || !(c->horizontal_fov_deg > 0 && c->horizontal_fov_deg < 180)) return QA_E_ARG;
    const double tan_h = tan(c->horizontal_fov_deg * 3.14159265358979323846 / 360.0), tan_v = tan_h * c->height / c->width;
    const double hs = c->horizontal_scale, vs = c->vertical_scale, far = c->far_clip, near = c->near_clip;
    const int64_t npix = (int64_t)Hc * Wc;
    #pragma omp parallel for schedule(dynamic, 1)
    for (int64_t e = 0; e < c->num_envs; ++e) {
        const float *rs = io->root_states + e * 13;
        double qx = rs[3], qy = rs[4], qz = rs[5], qw = rs[6];
        double R[9] = {1 - 2 * (qy * qy + qz * qz), 2 * (qx * qy - qz * qw), 2 * (qx * qz + qy * qw),
                       2 * (qx * qy + qz * qw), 1 - 2 * (qx * qx + qz * qz), 2 * (qy * qz - qx * qw),
                       2 * (qx * qz - qy * qw), 2 * (qy * qz + qx * qw), 1 - 2 * (qx * qx + qy * qy)};
        double a = io->camera_pitch[e], ca = cos(a), sa = sin(a);
        double o[3];
        for (int i = 0; i < 3; ++i) o[i] = rs[i] + R[3 * i] * c->position[0] + R[3 * i + 1] * c->position[1] + R[3 * i + 2] * c->position[2];
        uint32_t r0[4];
        const int64_t kstep = io->step_dev ? *io->step_dev : c->step;
        philox(c->seed, (uint32_t)(e + c->env_id_offset), (uint32_t)kstep, (uint32_t)(QA_TSC_DEPTH_STREAM * 256), (uint32_t)((uint64_t)kstep >> 32), r0);
        const float amp = c->depth_noise * ((float)(r0[0] >> 8) * (1.0f / 16777216.0f));
        const float offs = c->depth_noise * 2.0f * ((float)(r0[1] >> 8) * (1.0f / 16777216.0f) - 0.5f);
        float *buf = io->depth_buffer + e * c->buffer_len * npix;
        const int init = io->episode_length[e] <= 1;
        for (int64_t p = 0; p < npix; ++p) {
            int i = (int)(p / Wc), j = (int)(p % Wc);
            double sx = ((j + c->crop_left + 0.5) / c->width * 2.0 - 1.0) * tan_h, sy = ((i + c->crop_top + 0.5) / c->height * 2.0 - 1.0) * tan_v;
            double dt3[3] = {ca - sy * sa, -sx, -sa - sy * ca}, d[3];       /* fwd - sx left - sy up in the trunk frame */
            for (int k = 0; k < 3; ++k) d[k] = R[3 * k] * dt3[0] + R[3 * k + 1] * dt3[1] + R[3 * k + 2] * dt3[2];
            double dxy = sqrt(d[0] * d[0] + d[1] * d[1]), dt = 0.5 * hs / (dxy > 0.5 ? dxy : 0.5);
            int nsteps = (int)ceil(far / dt);
            double t_prev = 0.0, hit = far;
            int ex_prev = 0, ex = 0;
            double g_prev = o[2] - depth_surface(io->height_samples, c->map_rows, c->map_cols, c->border_size, hs, vs, o[0], o[1], NULL);
            double h_prev = io->ceiling_samples ? o[2] - depth_surface(io->ceiling_samples, c->map_rows, c->map_cols, c->border_size, hs, vs, o[0], o[1], &ex_prev) : 0.0;
            if (g_prev < 0) hit = 0.0;
            else for (int k = 1; k <= nsteps; ++k) {
                double t = k * dt; if (t > far) t = far;
                double x = o[0] + t * d[0], y = o[1] + t * d[1], z = o[2] + t * d[2];
                double g = z - depth_surface(io->height_samples, c->map_rows, c->map_cols, c->border_size, hs, vs, x, y, NULL), h = 0.0;
                double best = 1e30;
                if (g < 0) {                       /* the floor was crossed in (t_prev, t]: QA_TSC_DEPTH_BISECT halvings, then linear interpolation */
                    double lo = t_prev, hi = t, flo = g_prev, fhi = g;
                    for (int b = 0; b < QA_TSC_DEPTH_BISECT; ++b) {
                        double tm = 0.5 * (lo + hi);
                        double fm = o[2] + tm * d[2] - depth_surface(io->height_samples, c->map_rows, c->map_cols, c->border_size, hs, vs, o[0] + tm * d[0], o[1] + tm * d[1], NULL);
                        if (fm < 0) { hi = tm; fhi = fm; } else { lo = tm; flo = fm; }
                    }
                    best = lo + (hi - lo) * flo / (flo - fhi);
                }
                if (io->ceiling_samples) {
                    h = z - depth_surface(io->ceiling_samples, c->map_rows, c->map_cols, c->border_size, hs, vs, x, y, &ex);
                    if (ex && ex_prev && ((h_prev < 0) != (h < 0))) {
                        double lo = t_prev, hi = t, flo = h_prev, fhi = h;
                        for (int b = 0; b < QA_TSC_DEPTH_BISECT; ++b) {
                            double tm = 0.5 * (lo + hi); int exm;
                            double fm = o[2] + tm * d[2] - depth_surface(io->ceiling_samples, c->map_rows, c->map_cols, c->border_size, hs, vs, o[0] + tm * d[0], o[1] + tm * d[1], &exm);
                            if (!exm) break;                            /* a hole in the shell inside the bracket: keep the bracket */
                            if ((fm < 0) == (flo < 0)) { lo = tm; flo = fm; } else { hi = tm; fhi = fm; }
                        }
                        double tc = lo + (hi - lo) * flo / (flo - fhi);
                        if (tc < best) best = tc;
                    }
                }
                if (best < 1e29) { hit = best; break; }
                t_prev = t; g_prev = g; h_prev = h; ex_prev = ex;
            }
            double dd = hit < near ? near : (hit > far ? far : hit);
            float v = (float)((dd - near) / (far - near) - 0.5);
            uint32_t rp[4];
            philox(c->seed, (uint32_t)(e + c->env_id_offset), (uint32_t)kstep, (uint32_t)(QA_TSC_DEPTH_STREAM * 256 + 1 + (p >> 2)), (uint32_t)((uint64_t)kstep >> 32), rp);
            v += offs + amp * 2.0f * ((float)(rp[p & 3] >> 8) * (1.0f / 16777216.0f) - 0.5f);
            if (init) for (int s = 0; s < c->buffer_len; ++s) buf[s * npix + p] = v;
            else { for (int s = 0; s + 1 < c->buffer_len; ++s) buf[s * npix + p] = buf[(s + 1) * npix + p]; buf[(c->buffer_len - 1) * npix + p] = v; }
        }
    }
    return QA_OK;
}
