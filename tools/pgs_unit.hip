// Dev tool (r5): the packed rows / sweeps of csrc/qa_physics.h against the scalar ones on random inputs, one lane per sample.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I quadrupedal_agility_amd/csrc -I include tools/pgs_unit.hip -o tools/_prof/pgs_unit && tools/_prof/pgs_unit
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>
#include "qa_sim.h"
#include "qa_go2_model.h"
#include "qa_physics.h"

__global__ void k(const float *in, float *out) {
    const float *p = in + threadIdx.x * 128;
    float G[18], Linv[6], Binv[21], jb[3][6], jl[3][3], ub[6], w[3];
    for (int i = 0; i < 18; ++i) G[i] = p[i];
    for (int i = 0; i < 6; ++i) Linv[i] = p[18 + i];
    for (int i = 0; i < 21; ++i) Binv[i] = p[24 + i];
    for (int d = 0; d < 3; ++d) { for (int i = 0; i < 6; ++i) jb[d][i] = p[45 + 9 * d + i]; for (int k2 = 0; k2 < 3; ++k2) jl[d][k2] = p[45 + 9 * d + 6 + k2]; }
    for (int i = 0; i < 6; ++i) ub[i] = p[72 + i];
    for (int i = 0; i < 3; ++i) w[i] = p[78 + i];
    const float bias = p[81], mu = 0.8f;
    Row r[3];
    for (int d = 0; d < 3; ++d) { for (int k2 = 0; k2 < 3; ++k2) r[d].jl[k2] = jl[d][k2]; row_finish(r[d], G, Linv, Binv, jb[d]); r[d].bias = d == 0 ? bias : 0.f; }
    PSolve S; psolve_make(S, G, Linv, Binv);
    PRow q[3];
    for (int d = 0; d < 3; ++d) prow_finish(q[d], S, jb[d], jl[d], d == 0 ? bias : 0.f);
    float *o = out + threadIdx.x * 64;
    float e_rows = 0.f;
    for (int d = 0; d < 3; ++d) {
        const float pj[10] = {q[d].j[0].x, q[d].j[0].y, q[d].j[1].x, q[d].j[1].y, q[d].j[2].x, q[d].j[2].y, q[d].j[3].x, q[d].j[3].y, q[d].j[4].x, q[d].j[4].y};
        const float pm[10] = {q[d].m[0].x, q[d].m[0].y, q[d].m[1].x, q[d].m[1].y, q[d].m[2].x, q[d].m[2].y, q[d].m[3].x, q[d].m[3].y, q[d].m[4].x, q[d].m[4].y};
        for (int i = 0; i < 6; ++i) { e_rows = fmaxf(e_rows, fabsf(pj[i] - r[d].jh[i])); e_rows = fmaxf(e_rows, fabsf(pm[i] - r[d].bj[i])); }
        for (int i = 0; i < 3; ++i) { e_rows = fmaxf(e_rows, fabsf(pj[6 + i] - r[d].jl[i])); e_rows = fmaxf(e_rows, fabsf(pm[6 + i] - r[d].lj[i])); }
        e_rows = fmaxf(e_rows, fabsf(pj[9] - r[d].bias)); e_rows = fmaxf(e_rows, fabsf(pm[9]));
        e_rows = fmaxf(e_rows, fabsf(q[d].dinv - r[d].dinv) / fabsf(r[d].dinv));
    }
    o[0] = e_rows;
    // two sweeps of the contact update
    f2 x[5] = {f2{ub[0], ub[1]}, f2{ub[2], ub[3]}, f2{ub[4], ub[5]}, f2{w[0], w[1]}, f2{w[2], 1.0f}};
    for (int it = 0; it < 2; ++it) { contact_update(r, ub, w, mu); pcontact_update(q, x, mu); }
    const float xs[9] = {x[0].x, x[0].y, x[1].x, x[1].y, x[2].x, x[2].y, x[3].x, x[3].y, x[4].x};
    float e_x = 0.f;
    for (int i = 0; i < 6; ++i) e_x = fmaxf(e_x, fabsf(xs[i] - ub[i]));
    for (int i = 0; i < 3; ++i) e_x = fmaxf(e_x, fabsf(xs[6 + i] - w[i]));
    o[1] = e_x; o[2] = fabsf(x[4].y - 1.0f);
    o[3] = fmaxf(fabsf(q[0].lam - r[0].lam), fmaxf(fabsf(q[1].lam - r[1].lam), fabsf(q[2].lam - r[2].lam)));
    o[4] = r[0].lam; o[5] = q[0].lam;
    // LDS record round trip
    __shared__ __attribute__((aligned(16))) float rec[64 * 20];
    prow_store(rec + threadIdx.x * 20, q[1]);
    PRow t; prow_load(rec + threadIdx.x * 20, t);
    float e_l = fabsf(t.dinv - q[1].dinv);
    for (int i = 0; i < 5; ++i) { e_l = fmaxf(e_l, fabsf(t.j[i].x - q[1].j[i].x) + fabsf(t.j[i].y - q[1].j[i].y)); e_l = fmaxf(e_l, fabsf(t.m[i].x - q[1].m[i].x) + fabsf(t.m[i].y - q[1].m[i].y)); }
    o[6] = e_l;
    // joint-limit row of joint kk = lane % 3, sign by lane parity: scalar (round 4) against packed
    {
        const int kk = threadIdx.x % 3; const float sgn = (threadIdx.x & 4) ? 1.f : -1.f, lbias = -1.5f;
        float ub2[6], w2[3]; for (int i = 0; i < 6; ++i) ub2[i] = ub[i]; for (int i = 0; i < 3; ++i) w2[i] = w[i];
        f2 x2[5]; for (int i = 0; i < 5; ++i) x2[i] = x[i];
        float lam_s = 0.f, lam_p = 0.f;
        // scalar
        float jh[6], bj[6];
        for (int i = 0; i < 6; ++i) jh[i] = sgn * G[kk * 6 + i];
        sym6_mul(Binv, jh, bj);
        const float lkk = (kk == 0) ? Linv[0] : (kk == 1 ? Linv[3] : Linv[5]);
        float dd = lkk; for (int i = 0; i < 6; ++i) dd = fmaf(jh[i], bj[i], dd);
        const float dinv = 1.0f / (dd + QA_CFM);
        for (int it = 0; it < 2; ++it) {
            float uk = w2[kk]; for (int i = 0; i < 6; ++i) uk = fmaf(G[kk * 6 + i], ub2[i], uk);
            const float res = lbias + sgn * uk, lam = fmaxf(lam_s - res * dinv, 0.f), dl = lam - lam_s; lam_s = lam;
            for (int i = 0; i < 6; ++i) ub2[i] = fmaf(bj[i], dl, ub2[i]);
            const float sd = sgn * dl;
            w2[0] = fmaf(kk == 0 ? Linv[0] : (kk == 1 ? Linv[1] : Linv[2]), sd, w2[0]);
            w2[1] = fmaf(kk == 0 ? Linv[1] : (kk == 1 ? Linv[3] : Linv[4]), sd, w2[1]);
            w2[2] = fmaf(kk == 0 ? Linv[2] : (kk == 1 ? Linv[4] : Linv[5]), sd, w2[2]);
        }
        // packed (the code of phys_substep)
        f2 pjh[3], lim_m[3];
        for (int qq = 0; qq < 3; ++qq) pjh[qq] = f2s(sgn) * S.G2[kk][qq];
        pbinv_mul(S, pjh, lim_m);
        const f2 pd = pfma(pjh[2], lim_m[2], pfma(pjh[1], lim_m[1], pjh[0] * lim_m[0]));
        const float pdinv = 1.0f / (lkk + hsum(pd) + QA_CFM);
        for (int it = 0; it < 2; ++it) {
            const f2 gu = pfma(S.G2[kk][2], x2[2], pfma(S.G2[kk][1], x2[1], S.G2[kk][0] * x2[0]));
            const float wk = kk == 0 ? x2[3].x : (kk == 1 ? x2[3].y : x2[4].x);
            const float uk = hsum(gu) + wk;
            const float res = fmaf(sgn, uk, lbias);
            const float lam = fmaxf(fmaf(-res, pdinv, lam_p), 0.f), dl = lam - lam_p; lam_p = lam;
            for (int qq = 0; qq < 3; ++qq) x2[qq] = pfma(lim_m[qq], f2s(dl), x2[qq]);
            const float sd = sgn * dl;
            x2[3] = pfma(S.L01[kk], f2s(sd), x2[3]);
            x2[4].x = fmaf(S.L2[kk], sd, x2[4].x);
        }
        const float xs2[9] = {x2[0].x, x2[0].y, x2[1].x, x2[1].y, x2[2].x, x2[2].y, x2[3].x, x2[3].y, x2[4].x};
        float e = fabsf(lam_s - lam_p) + fabsf(dinv - pdinv) / dinv;
        for (int i = 0; i < 6; ++i) e = fmaxf(e, fabsf(xs2[i] - ub2[i]));
        for (int i = 0; i < 3; ++i) e = fmaxf(e, fabsf(xs2[6 + i] - w2[i]));
        o[8] = e; o[9] = lam_s;
    }
    // open row: no velocity overcomes the bias
    q[0].j[4].y = QA_OPEN_BIAS; q[0].lam = q[1].lam = q[2].lam = 0.f;
    f2 y[5]; for (int i = 0; i < 5; ++i) y[i] = x[i];
    pcontact_update(q, y, mu);
    float e_o = 0.f;
    for (int i = 0; i < 5; ++i) e_o = fmaxf(e_o, fabsf(y[i].x - x[i].x) + fabsf(y[i].y - x[i].y));
    o[7] = e_o;
}

int main() {
    std::vector<float> h(64 * 128);
    srand(1);
    auto rnd = []() { return (float)rand() / RAND_MAX * 2.f - 1.f; };
    for (int t = 0; t < 64; ++t) {
        float *p = h.data() + t * 128;
        for (int i = 0; i < 18; ++i) p[i] = rnd();
        float A[9]; for (int i = 0; i < 9; ++i) A[i] = rnd();
        // Linv = A A^T + I  packed 00 01 02 11 12 22
        auto L = [&](int i, int j) { float s = i == j ? 1.f : 0.f; for (int k = 0; k < 3; ++k) s += A[i * 3 + k] * A[j * 3 + k]; return s; };
        p[18] = L(0, 0); p[19] = L(0, 1); p[20] = L(0, 2); p[21] = L(1, 1); p[22] = L(1, 2); p[23] = L(2, 2);
        float Bm[36]; for (int i = 0; i < 36; ++i) Bm[i] = rnd();
        for (int i = 0; i < 6; ++i) for (int j = 0; j <= i; ++j) { float s = i == j ? 1.f : 0.f; for (int k = 0; k < 6; ++k) s += Bm[i * 6 + k] * Bm[j * 6 + k]; p[24 + i * (i + 1) / 2 + j] = s; }
        for (int i = 45; i < 81; ++i) p[i] = rnd();
        p[81] = -2.f + rnd();
    }
    float *din, *dout; hipMalloc(&din, h.size() * 4); hipMalloc(&dout, 64 * 64 * 4);
    hipMemcpy(din, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, din, dout);
    std::vector<float> o(64 * 64); hipMemcpy(o.data(), dout, o.size() * 4, hipMemcpyDeviceToHost);
    const char *names[10] = {"rows (jh bj jl lj bias dinv)", "state after 2 sweeps", "x[4].y - 1", "impulses", "lam0 scalar", "lam0 packed", "LDS record round trip", "open row leaves x", "joint-limit row, 2 sweeps", "limit impulse (scalar)"};
    for (int c = 0; c < 10; ++c) { float m = 0; for (int t = 0; t < 64; ++t) m = fmaxf(m, fabsf(o[t * 64 + c])); printf("%-32s max %.3e\n", names[c], m); }
    return 0;
}
