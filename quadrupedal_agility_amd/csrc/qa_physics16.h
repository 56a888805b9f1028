// qa_physics16.h -- the physics substep with SIXTEEN lanes per env (plane terrain): lane = 16 env + 4 leg + sub.
//
// Why: with one env per quad, 4096 envs are 256 wavefronts = one per CU, and the kernel time is the instruction count of
// one wavefront.  Here an env occupies a 16-lane DPP row, 4096 envs are 1024 wavefronts = one per SIMD,
// and the four sub-lanes of a leg share the parts of the substep that dominate the instruction stream:
//   * dynamics (kinematics, composite inertias, bias, Schur complement, 6x6 inverse): identical to qa_physics.h, run
//     redundantly by the four sub-lanes (cross-leg sums are row_ror:4/8 DPP adds instead of quad_perm);
//   * contact candidates: the 16 non-foot leg points and the leg's 3 base points are dealt to the sub-lanes, the closest
//     one is found with a quad min that carries the point code (ties -> smaller code, like the serial scan);
//   * constraint rows and the PGS state live DISTRIBUTED over the sub-lanes: sub 0 holds entries ub[0:3], sub 1 ub[3:6],
//     sub 2 w[0:3] (sub 3 holds zeros) of the state and of every row's Jacobian (jh | jl) and response (Binv jh | Linv jl).
//     A row residual is 3 FMAs + a quad sum, an impulse update 3 FMAs; the extra-contact rows stay in registers.
// The arithmetic is the same model as qa_physics.h (same rows, same two-colour ordering, same sweeps); sums are
// associated differently, so results agree with the 4-lane kernel and the oracle to fp32 rounding, not bitwise.
#pragma once
#include "qa_physics.h"

// one contact (normal, tangent 1, tangent 2) in distributed form
struct Contact16 {
    float j[3][3];     // this sub-lane's 3 entries of (jh | jl) per row
    float b[3][3];     // this sub-lane's 3 entries of (Binv jh | Linv jl) per row
    float dinv[3], bias0, lam[3];
};

QA_DEV float dot3(const float *a, const float *b) { return fmaf(a[2], b[2], fmaf(a[1], b[1], a[0] * b[0])); }

// rows of a contact at base-frame point p on a body of chain depth `depth` (0 = base), directions n, t1, t2 (base frame)
// Gs[k][i] = this lane's 3 columns of G (sub 0: 0..2, sub 1: 3..5), Bs[i][c] = its 3 rows of Binv (sub 0: 0..2, sub 1: 3..5)
QA_DEV void contact_rows16(Contact16 &C, V3 p, int depth, float gap, const V3 *o, const V3 *ax, V3 nB, V3 t1B, V3 t2B,
                           const float (*Gs)[3], const float *Linv, const float (*Bs)[6], const PhysParams &P, int sub) {
    const V3 dirs[3] = {nB, t1B, t2B};
    // joint columns: sub k (< 3) evaluates c_k = ax_k x (p - o_k) once and its dot with the three directions
    const V3 ok = sub == 0 ? o[0] : (sub == 1 ? o[1] : o[2]);
    const V3 ak = sub == 0 ? ax[0] : (sub == 1 ? ax[1] : ax[2]);
    const V3 ck = cross(ak, p - ok);
    const bool live = sub < depth;                           // sub 3 and joints below the contact body contribute nothing
    float mine[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) mine[d] = live ? dot(dirs[d], ck) : 0.f;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const float jl[3] = {quad_bcast<0>(mine[d]), quad_bcast<1>(mine[d]), quad_bcast<2>(mine[d])};
        const V3 pxd = cross(p, dirs[d]);
        // this lane's slice of jh = jb + G^T jl  (sub 0: angular part, sub 1: linear part)
        float jh_s[3];
        {
            const float jb0 = sub == 0 ? pxd.x : dirs[d].x, jb1 = sub == 0 ? pxd.y : dirs[d].y, jb2 = sub == 0 ? pxd.z : dirs[d].z;
            jh_s[0] = jb0 + Gs[0][0] * jl[0] + Gs[1][0] * jl[1] + Gs[2][0] * jl[2];
            jh_s[1] = jb1 + Gs[0][1] * jl[0] + Gs[1][1] * jl[1] + Gs[2][1] * jl[2];
            jh_s[2] = jb2 + Gs[0][2] * jl[0] + Gs[1][2] * jl[1] + Gs[2][2] * jl[2];
        }
        // full jh on every lane: slices of sub 0 and sub 1
        float jh[6];
#pragma unroll
        for (int i = 0; i < 3; ++i) { jh[i] = quad_bcast<0>(jh_s[i]); jh[3 + i] = quad_bcast<1>(jh_s[i]); }
        float js[3], bs[3];
        if (sub < 2) {
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                js[i] = jh_s[i];
                float s = 0.f;
#pragma unroll
                for (int c = 0; c < 6; ++c) s = fmaf(Bs[i][c], jh[c], s);
                bs[i] = s;
            }
        } else if (sub == 2) {
            js[0] = jl[0]; js[1] = jl[1]; js[2] = jl[2];
            bs[0] = Linv[0] * jl[0] + Linv[1] * jl[1] + Linv[2] * jl[2];
            bs[1] = Linv[1] * jl[0] + Linv[3] * jl[1] + Linv[4] * jl[2];
            bs[2] = Linv[2] * jl[0] + Linv[4] * jl[1] + Linv[5] * jl[2];
        } else {
            js[0] = js[1] = js[2] = 0.f; bs[0] = bs[1] = bs[2] = 0.f;
        }
        const float dd = quad_sum(dot3(js, bs));
        C.dinv[d] = 1.0f / (dd + QA_CFM);
#pragma unroll
        for (int i = 0; i < 3; ++i) { C.j[d][i] = js[i]; C.b[d][i] = bs[i]; }
        C.lam[d] = 0.f;
    }
    const float g = gap / P.dt;
    C.bias0 = gap >= 0.f ? g : fmaxf(g, -P.max_depen);
}

// Gauss-Seidel update of one contact on the lane-local state slice: normal row, then both tangent rows from the
// velocity the normal row left (pyramid friction), as contact_update() in qa_physics.h
QA_DEV void contact_update16(Contact16 &C, float *s, float mu) {
    {
        const float r = quad_sum(dot3(C.j[0], s)) + C.bias0;
        const float lam = fmaxf(C.lam[0] - r * C.dinv[0], 0.f), dl = lam - C.lam[0];
        C.lam[0] = lam;
#pragma unroll
        for (int i = 0; i < 3; ++i) s[i] = fmaf(C.b[0][i], dl, s[i]);
    }
    const float lim = mu * C.lam[0];
    const float r1 = quad_sum(dot3(C.j[1], s)), r2 = quad_sum(dot3(C.j[2], s));
    const float l1 = clampf(C.lam[1] - r1 * C.dinv[1], -lim, lim), l2 = clampf(C.lam[2] - r2 * C.dinv[2], -lim, lim);
    const float d1 = l1 - C.lam[1], d2 = l2 - C.lam[2];
    C.lam[1] = l1; C.lam[2] = l2;
#pragma unroll
    for (int i = 0; i < 3; ++i) s[i] = fmaf(C.b[2][i], d2, fmaf(C.b[1][i], d1, s[i]));
}

// One substep on the plane, 16 lanes per env.  `leg` = 0..3, `sub` = 0..3; everything per-leg in `st`, tau, fimp, co is
// replicated over the four sub-lanes (bitwise identical), everything per-env over the 16 lanes.
QA_DEV void phys_substep16(EnvState &st, const float *tbl, const float *btbl, const float *binert, const float tau[3],
                           float mu, int leg, int sub, const PhysParams &P, ContactOut &co, float fimp[3]) {
    const float dt = P.dt;
    M3 R = quat_to_mat(st.qx, st.qy, st.qz, st.qw);
    S6 V0 = s6(mulT(R, st.ww), mulT(R, st.vw));
    V3 gB = v3(R.m[6] * P.gz, R.m[7] * P.gz, R.m[8] * P.gz);

    QA_SUBSTAMP(0);
    // ---- leg kinematics in the base frame (as qa_physics.h)
    float s1, c1, s2, c2, s23, c23;
    __sincosf(st.q[0], &s1, &c1); __sincosf(st.q[1], &s2, &c2); __sincosf(st.q[1] + st.q[2], &s23, &c23);
    M3 Rl[3];
    Rl[0].m[0] = 1; Rl[0].m[1] = 0; Rl[0].m[2] = 0; Rl[0].m[3] = 0; Rl[0].m[4] = c1; Rl[0].m[5] = -s1; Rl[0].m[6] = 0; Rl[0].m[7] = s1; Rl[0].m[8] = c1;
    Rl[1].m[0] = c2; Rl[1].m[1] = 0; Rl[1].m[2] = s2; Rl[1].m[3] = s1 * s2; Rl[1].m[4] = c1; Rl[1].m[5] = -s1 * c2; Rl[1].m[6] = -c1 * s2; Rl[1].m[7] = s1; Rl[1].m[8] = c1 * c2;
    Rl[2].m[0] = c23; Rl[2].m[1] = 0; Rl[2].m[2] = s23; Rl[2].m[3] = s1 * s23; Rl[2].m[4] = c1; Rl[2].m[5] = -s1 * c23; Rl[2].m[6] = -c1 * s23; Rl[2].m[7] = s1; Rl[2].m[8] = c1 * c23;
    V3 o[3], ax[3];
    o[0] = v3(tbl[T_HIP_ORG], tbl[T_HIP_ORG + 1], tbl[T_HIP_ORG + 2]);
    o[1] = o[0] + mul(Rl[0], v3(tbl[T_THIGH_ORG], tbl[T_THIGH_ORG + 1], tbl[T_THIGH_ORG + 2]));
    o[2] = o[1] + mul(Rl[1], v3(tbl[T_CALF_ORG], tbl[T_CALF_ORG + 1], tbl[T_CALF_ORG + 2]));
    ax[0] = v3(1, 0, 0); ax[1] = v3(0, c1, s1); ax[2] = ax[1];
    S6 S[3];
    RB link[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        S[k] = s6(ax[k], cross(o[k], ax[k]));
        link[k] = link_rb(tbl[T_MASS + k], v3(tbl[T_COM + 3 * k], tbl[T_COM + 3 * k + 1], tbl[T_COM + 3 * k + 2]),
                          tbl + T_INERTIA + 6 * k, Rl[k], o[k]);
    }
    QA_SUBSTAMP(1);
    // ---- composite inertias, mass-matrix blocks
    RB Ic2 = link[2], Ic1 = link[1] + Ic2, Ic0 = link[0] + Ic1;
    S6 F[3] = {apply(Ic0, S[0]), apply(Ic1, S[1]), apply(Ic2, S[2])};
    float L00 = dot(S[0], F[0]), L01 = dot(S[0], F[1]), L02 = dot(S[0], F[2]);
    float L11 = dot(S[1], F[1]), L12 = dot(S[1], F[2]), L22 = dot(S[2], F[2]);
    RB base; base.m = binert[0]; base.h = v3(binert[1], binert[2], binert[3]);
    base.xx = binert[4]; base.yy = binert[5]; base.zz = binert[6]; base.xy = binert[7]; base.xz = binert[8]; base.yz = binert[9];
    RB tot = base + xsum<16>(Ic0);

    QA_SUBSTAMP(2);
    // ---- bias forces
    S6 A0 = s6(v3(0, 0, 0), v3(-gB.x, -gB.y, -gB.z));
    S6 fl[3];
    {
        S6 V = V0, A = A0;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            S6 Sq = st.qd[k] * S[k];
            V = V + Sq;
            A = A + crm(V, Sq);
            fl[k] = apply(link[k], A) + crf(V, apply(link[k], V));
        }
    }
    fl[1] = fl[1] + fl[2]; fl[0] = fl[0] + fl[1];
    float hl[3] = {dot(S[0], fl[0]), dot(S[1], fl[1]), dot(S[2], fl[2])};
    S6 f0 = apply(base, A0) + crf(V0, apply(base, V0)) + xsum<16>(fl[0]);

    QA_SUBSTAMP(3);
    // ---- leg elimination
    float Linv[6];
    {
        float cA = L11 * L22 - L12 * L12, cB = L02 * L12 - L01 * L22, cC = L01 * L12 - L02 * L11;
        float cD = L00 * L22 - L02 * L02, cE = L01 * L02 - L00 * L12, cF = L00 * L11 - L01 * L01;
        float idet = 1.0f / (L00 * cA + L01 * cB + L02 * cC);
        Linv[0] = cA * idet; Linv[1] = cB * idet; Linv[2] = cC * idet; Linv[3] = cD * idet; Linv[4] = cE * idet; Linv[5] = cF * idet;
    }
    float Fm[3][6];
#pragma unroll
    for (int k = 0; k < 3; ++k) { Fm[k][0] = F[k].a.x; Fm[k][1] = F[k].a.y; Fm[k][2] = F[k].a.z; Fm[k][3] = F[k].l.x; Fm[k][4] = F[k].l.y; Fm[k][5] = F[k].l.z; }
    float G[18];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        G[0 * 6 + i] = -(Linv[0] * Fm[0][i] + Linv[1] * Fm[1][i] + Linv[2] * Fm[2][i]);
        G[1 * 6 + i] = -(Linv[1] * Fm[0][i] + Linv[3] * Fm[1][i] + Linv[4] * Fm[2][i]);
        G[2 * 6 + i] = -(Linv[2] * Fm[0][i] + Linv[4] * Fm[1][i] + Linv[5] * Fm[2][i]);
    }
    float Bm[21];
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j <= i; ++j)
            Bm[SIDX(i, j)] = xsum<16>(Fm[0][i] * G[0 * 6 + j] + Fm[1][i] * G[1 * 6 + j] + Fm[2][i] * G[2 * 6 + j]);
    Bm[SIDX(0, 0)] += tot.xx; Bm[SIDX(1, 1)] += tot.yy; Bm[SIDX(2, 2)] += tot.zz;
    Bm[SIDX(1, 0)] += tot.xy; Bm[SIDX(2, 0)] += tot.xz; Bm[SIDX(2, 1)] += tot.yz;
    Bm[SIDX(3, 3)] += tot.m; Bm[SIDX(4, 4)] += tot.m; Bm[SIDX(5, 5)] += tot.m;
    Bm[SIDX(3, 1)] += tot.h.z; Bm[SIDX(3, 2)] += -tot.h.y;
    Bm[SIDX(4, 0)] += -tot.h.z; Bm[SIDX(4, 2)] += tot.h.x;
    Bm[SIDX(5, 0)] += tot.h.y; Bm[SIDX(5, 1)] += -tot.h.x;
    float Binv[21];
    spd6_inverse(Bm, Binv);

    QA_SUBSTAMP(5);
    // ---- unconstrained velocity
    float rl[3] = {tau[0] - hl[0], tau[1] - hl[1], tau[2] - hl[2]};
    float rb[6] = {-f0.a.x, -f0.a.y, -f0.a.z, -f0.l.x, -f0.l.y, -f0.l.z};
#pragma unroll
    for (int i = 0; i < 6; ++i) rb[i] += xsum<16>(G[0 * 6 + i] * rl[0] + G[1 * 6 + i] * rl[1] + G[2 * 6 + i] * rl[2]);
    float ab[6];
    sym6_mul(Binv, rb, ab);
    V3 wxv = cross(V0.a, V0.l);
    float ub[6] = {V0.a.x + dt * ab[0], V0.a.y + dt * ab[1], V0.a.z + dt * ab[2],
                   V0.l.x + dt * (ab[3] + wxv.x), V0.l.y + dt * (ab[4] + wxv.y), V0.l.z + dt * (ab[5] + wxv.z)};
    float w[3];
    {
        float lr0 = Linv[0] * rl[0] + Linv[1] * rl[1] + Linv[2] * rl[2];
        float lr1 = Linv[1] * rl[0] + Linv[3] * rl[1] + Linv[4] * rl[2];
        float lr2 = Linv[2] * rl[0] + Linv[4] * rl[1] + Linv[5] * rl[2];
        float lr[3] = {lr0, lr1, lr2};
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            float gab = 0.f, gub = 0.f;
#pragma unroll
            for (int i = 0; i < 6; ++i) { gab = fmaf(G[k * 6 + i], ab[i], gab); gub = fmaf(G[k * 6 + i], ub[i], gub); }
            float ul = st.qd[k] + dt * (lr[k] + gab);
            w[k] = ul - gub;
        }
    }

    QA_SUBSTAMP(6);
    // lane-local blocks for the distributed rows.  Blended arithmetically (x*1 + y*0 is exact): written as selects the
    // compiler folds `cond ? A[i] : A[j]` into A[cond ? i : j], a run-time index into a register array, which it then
    // lowers to a 21-way compare/select chain per access.
    float Gs[3][3], Bs[3][6];
    {
        const float m0 = sub == 1 ? 0.f : 1.f, m1 = sub == 1 ? 1.f : 0.f;
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
            for (int i = 0; i < 3; ++i) Gs[k][i] = fmaf(m1, G[k * 6 + 3 + i], m0 * G[k * 6 + i]);
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int c = 0; c < 6; ++c) Bs[i][c] = fmaf(m1, Binv[SIDX(3 + i, c)], m0 * Binv[SIDX(i, c)]);
    }

    // ---- contact candidates, dealt to the sub-lanes
    const V3 nB = v3(R.m[6], R.m[7], R.m[8]), t1B = v3(R.m[0], R.m[1], R.m[2]), t2B = v3(R.m[3], R.m[4], R.m[5]);
    V3 foot_p; float foot_gap;
    {
        const float *pt = tbl + T_POINTS;
        foot_p = mul(Rl[2], v3(pt[0], pt[1], pt[2])) + o[2];
        foot_gap = dot(nB, foot_p) + st.pos.z - pt[3];
    }
    float best_gap = 1e30f; int best_c = 0;                   // 1..16 leg point, 64 + c base point
    {
        V3 nk[3]; float hk[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) { nk[k] = mulT(Rl[k], nB); hk[k] = dot(nB, o[k]) + st.pos.z; }
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const int c = 1 + sub + 4 * jj;                   // 1..16
            const int k = (c < 3 ? 0 : (c < 11 ? 1 : 2));
            const float *pt = tbl + T_POINTS + 4 * c;
            const V3 n = k == 0 ? nk[0] : (k == 1 ? nk[1] : nk[2]);
            const float h = k == 0 ? hk[0] : (k == 1 ? hk[1] : hk[2]);
            const float gap = dot(n, v3(pt[0], pt[1], pt[2])) + h - pt[3];
            if (gap < best_gap) { best_gap = gap; best_c = c; }
        }
        const int cb = leg + 4 * sub;                         // the leg's base points: leg, leg + 4, leg + 8
        if (sub < 3 && cb < QA_BASE_PTS) {
            const float *pt = btbl + 4 * cb;
            const float gap = dot(nB, v3(pt[0], pt[1], pt[2])) + st.pos.z - pt[3];
            if (gap < best_gap) { best_gap = gap; best_c = 64 + cb; }
        }
        // quad min carrying the code; ties go to the smaller code (= the order of the serial scan)
        {
            float g1 = dpp_f<0xB1>(best_gap); int c1_ = dpp_i<0xB1>(best_c);
            if (g1 < best_gap || (g1 == best_gap && c1_ < best_c)) { best_gap = g1; best_c = c1_; }
            float g2 = dpp_f<0x4E>(best_gap); int c2_ = dpp_i<0x4E>(best_c);
            if (g2 < best_gap || (g2 == best_gap && c2_ < best_c)) { best_gap = g2; best_c = c2_; }
        }
    }
    V3 best_p = v3(0, 0, 0); int best_depth = 0, best_body = -1;
    if (best_c >= 64) {
        const int c = best_c - 64;
        const float *pt = btbl + 4 * c;
        best_p = v3(pt[0], pt[1], pt[2]); best_depth = 0; best_body = c < 8 ? 0 : (c < 10 ? 1 : 2);
    } else if (best_c > 0) {
        const int k = (best_c < 3 ? 0 : (best_c < 11 ? 1 : 2));
        const float *pt = tbl + T_POINTS + 4 * best_c;
        const M3 Rs = k == 0 ? Rl[0] : (k == 1 ? Rl[1] : Rl[2]);
        const V3 os = k == 0 ? o[0] : (k == 1 ? o[1] : o[2]);
        best_p = mul(Rs, v3(pt[0], pt[1], pt[2])) + os; best_depth = k + 1; best_body = 3 + 4 * leg + k;
    }
    const bool foot_on = foot_gap < P.contact_offset;
    const bool extra_on = best_gap < P.contact_offset;

    QA_SUBSTAMP(7);
    // ---- rows, distributed
    Contact16 cf;
    contact_rows16(cf, foot_p, 3, foot_gap, o, ax, nB, t1B, t2B, Gs, Linv, Bs, P, sub);
    const bool any_extra = __any(extra_on);
    Contact16 ce;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        ce.lam[d] = 0.f; ce.dinv[d] = 0.f;
#pragma unroll
        for (int i = 0; i < 3; ++i) { ce.j[d][i] = 0.f; ce.b[d][i] = 0.f; }
    }
    ce.bias0 = 0.f;
    if (any_extra) contact_rows16(ce, best_p, best_depth, best_gap, o, ax, nB, t1B, t2B, Gs, Linv, Bs, P, sub);
    // joint limits (row k: jl = sgn e_k, jh = sgn G[k,:], lj = sgn Linv[:,k])
    float lim_j[3][3], lim_b[3][3], lim_dinv[3], lim_bias[3], lim_lam[3];
    bool lim_on[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        float glo = st.q[k] - tbl[T_LOWER + k], ghi = tbl[T_UPPER + k] - st.q[k];
        bool lo = glo < QA_LIMIT_MARGIN, hi = !lo && (ghi < QA_LIMIT_MARGIN);
        lim_on[k] = lo || hi;
        const float sgn = lo ? 1.f : -1.f;
        float gap = lo ? glo : ghi, g = gap / dt;
        lim_bias[k] = gap >= 0.f ? g : fmaxf(g, -QA_LIMIT_DEPEN);
        lim_lam[k] = 0.f; lim_dinv[k] = 0.f;
#pragma unroll
        for (int i = 0; i < 3; ++i) { lim_j[k][i] = 0.f; lim_b[k][i] = 0.f; }
        lim_j[k][0] = sgn;                                   // placeholder so that sgn survives to the block below
    }
    const bool any_lim = __any(lim_on[0] || lim_on[1] || lim_on[2]);
    if (any_lim) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float sgn = lim_j[k][0];
            float jh[6];
#pragma unroll
            for (int i = 0; i < 6; ++i) jh[i] = sgn * G[k * 6 + i];
            float js[3], bs[3];
            if (sub < 2) {
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    js[i] = sgn * Gs[k][i];
                    float s = 0.f;
#pragma unroll
                    for (int c = 0; c < 6; ++c) s = fmaf(Bs[i][c], jh[c], s);
                    bs[i] = s;
                }
            } else if (sub == 2) {
                js[0] = k == 0 ? sgn : 0.f; js[1] = k == 1 ? sgn : 0.f; js[2] = k == 2 ? sgn : 0.f;
                bs[0] = sgn * (k == 0 ? Linv[0] : (k == 1 ? Linv[1] : Linv[2]));
                bs[1] = sgn * (k == 0 ? Linv[1] : (k == 1 ? Linv[3] : Linv[4]));
                bs[2] = sgn * (k == 0 ? Linv[2] : (k == 1 ? Linv[4] : Linv[5]));
            } else {
                js[0] = js[1] = js[2] = 0.f; bs[0] = bs[1] = bs[2] = 0.f;
            }
            const float dd = quad_sum(dot3(js, bs));
            lim_dinv[k] = 1.0f / (dd + QA_CFM);
#pragma unroll
            for (int i = 0; i < 3; ++i) { lim_j[k][i] = js[i]; lim_b[k][i] = bs[i]; }
        }
    }

    QA_SUBSTAMP(8);
    // ---- distributed state: sub 0 ub[0:3], sub 1 ub[3:6], sub 2 w, sub 3 zeros
    float s[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) s[i] = sub == 0 ? ub[i] : (sub == 1 ? ub[3 + i] : (sub == 2 ? w[i] : 0.f));
    const bool base_part = sub < 2;                          // entries that belong to the shared base velocity

    // ---- warm start of the foot rows
    {
        float d[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const float l0 = foot_on ? fimp[r] : 0.f;
            cf.lam[r] = l0;
#pragma unroll
            for (int i = 0; i < 3; ++i) d[i] = fmaf(cf.b[r][i], l0, d[i]);
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) { const float t = xsum<16>(d[i]); s[i] += base_part ? t : d[i]; }
    }

    QA_SUBSTAMP(4);
    // ---- projected Gauss-Seidel, two-colour ordering over the legs (as qa_physics.h)
    const bool any_foot = __any(foot_on);
    const bool colour_a = (leg == 0) || (leg == 3);
    for (int it = 0; it < P.iters; ++it) {
#pragma unroll
        for (int colour = 0; colour < 2; ++colour) {
            const bool mine = (colour == 0) == colour_a;
            float s2[3] = {s[0], s[1], s[2]};
            if (any_foot) {
                if (foot_on) {
                    const float l0 = cf.lam[0], l1 = cf.lam[1], l2 = cf.lam[2];
                    contact_update16(cf, s2, mu);
                    if (!mine) { cf.lam[0] = l0; cf.lam[1] = l1; cf.lam[2] = l2; }
                }
            }
            if (any_extra) {
                if (extra_on) {
                    const float l0 = ce.lam[0], l1 = ce.lam[1], l2 = ce.lam[2];
                    contact_update16(ce, s2, mu);
                    if (!mine) { ce.lam[0] = l0; ce.lam[1] = l1; ce.lam[2] = l2; }
                }
            }
            if (any_lim) {
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    if (lim_on[k]) {
                        const float res = quad_sum(dot3(lim_j[k], s2)) + lim_bias[k];
                        const float lam = fmaxf(lim_lam[k] - res * lim_dinv[k], 0.f), dl = lam - lim_lam[k];
                        if (mine) lim_lam[k] = lam;
#pragma unroll
                        for (int i = 0; i < 3; ++i) s2[i] = fmaf(lim_b[k][i], dl, s2[i]);
                    }
                }
            }
            // commit: the base entries take the SUM over the legs of the active colour, the leg entries are lane-local
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const float d = mine ? (s2[i] - s[i]) : 0.f;
                const float t = xsum<16>(d);
                s[i] += base_part ? t : d;
            }
        }
    }

    QA_SUBSTAMP(9);
    // ---- back to replicated (ub, w)
#pragma unroll
    for (int i = 0; i < 3; ++i) { ub[i] = quad_bcast<0>(s[i]); ub[3 + i] = quad_bcast<1>(s[i]); w[i] = quad_bcast<2>(s[i]); }

    // ---- leg velocity, clamp, integrate (as qa_physics.h)
    float ul[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        float g = w[k];
#pragma unroll
        for (int i = 0; i < 6; ++i) g = fmaf(G[k * 6 + i], ub[i], g);
        float vl = tbl[T_VELLIM + k];
        ul[k] = clampf(g, -vl, vl);
    }
    V3 wb = v3(ub[0], ub[1], ub[2]);
    V3 wn = mul(R, wb), vn = mul(R, v3(ub[3], ub[4], ub[5]));
    st.pos = st.pos + dt * vn;
    {
        float wn2 = dot(wb, wb), wnorm = sqrtf(wn2), ang = wnorm * dt;
        float dx, dy, dz, dw;
        if (ang > 1e-12f) { float sh, ch; sincosf(0.5f * ang, &sh, &ch); float sc = sh / wnorm; dx = wb.x * sc; dy = wb.y * sc; dz = wb.z * sc; dw = ch; }
        else { dx = 0.5f * dt * wb.x; dy = 0.5f * dt * wb.y; dz = 0.5f * dt * wb.z; dw = 1.f; }
        float nx = st.qw * dx + st.qx * dw + st.qy * dz - st.qz * dy;
        float ny = st.qw * dy - st.qx * dz + st.qy * dw + st.qz * dx;
        float nz = st.qw * dz + st.qx * dy - st.qy * dx + st.qz * dw;
        float nw = st.qw * dw - st.qx * dx - st.qy * dy - st.qz * dz;
        float inv = rsqrtf(nx * nx + ny * ny + nz * nz + nw * nw);
        st.qx = nx * inv; st.qy = ny * inv; st.qz = nz * inv; st.qw = nw * inv;
    }
    st.vw = vn; st.ww = wn;
#pragma unroll
    for (int k = 0; k < 3; ++k) { st.q[k] = fmaf(dt, ul[k], st.q[k]); st.qd[k] = ul[k]; }

#pragma unroll
    for (int d = 0; d < 3; ++d) fimp[d] = foot_on ? cf.lam[d] : 0.f;
    QA_SUBSTAMP(10);
    const float idt = 1.0f / dt;
    co.foot_f = foot_on ? v3(cf.lam[1] * idt, cf.lam[2] * idt, cf.lam[0] * idt) : v3(0, 0, 0);
    co.extra_f = (any_extra && extra_on) ? v3(ce.lam[1] * idt, ce.lam[2] * idt, ce.lam[0] * idt) : v3(0, 0, 0);
    co.extra_body = (any_extra && extra_on) ? best_body : -1;
}
