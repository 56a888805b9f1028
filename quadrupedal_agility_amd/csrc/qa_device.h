// qa_device.h -- device-side helpers for the gfx950 kernels: quad (4-lane) DPP exchange, small
// vector algebra, Philox4x32-10.  One Go2 env occupies one quad of a 64-wide wavefront: lane&3 is
// the leg (FL,FR,RL,RR), so every cross-leg reduction is a quad_perm DPP move -- no LDS, no
// ds_bpermute.  16 envs per wavefront.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define QA_DEV static __device__ __forceinline__

// ---------------------------------------------------------------- quad exchange (DPP quad_perm)
template <int CTRL>
QA_DEV float dpp_f(float x) {
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(x), CTRL, 0xF, 0xF, true));
}
template <int CTRL>
QA_DEV int dpp_i(int x) { return __builtin_amdgcn_mov_dpp(x, CTRL, 0xF, 0xF, true); }

// sum over the 4 lanes of a quad; all 4 lanes receive the bitwise-identical value
QA_DEV float quad_sum(float x) {
    x += dpp_f<0xB1>(x);   // quad_perm [1,0,3,2]
    x += dpp_f<0x4E>(x);   // quad_perm [2,3,0,1]
    return x;
}
QA_DEV float quad_min(float x) {
    x = fminf(x, dpp_f<0xB1>(x));
    x = fminf(x, dpp_f<0x4E>(x));
    return x;
}
QA_DEV int quad_or(int x) {
    x |= dpp_i<0xB1>(x);
    x |= dpp_i<0x4E>(x);
    return x;
}
// broadcast lane S of each quad to its 4 lanes
template <int S>
QA_DEV float quad_bcast(float x) { return dpp_f<S * 0x55>(x); }
template <int S>
QA_DEV int quad_bcast_i(int x) { return dpp_i<S * 0x55>(x); }

// ---------------------------------------------------------------- cross-leg reductions for both lane mappings
// LPE = lanes per env.  4: lane&3 = leg, the legs of an env are one quad.  16: lane = 16 env + 4 leg + sub, the legs of
// an env are the 4 quads of one 16-lane DPP row; row_ror:4 / row_ror:8 add the lanes that share `sub`.
template <int LPE>
QA_DEV float xsum(float x) {
    if (LPE == 4) return quad_sum(x);
    x += dpp_f<0x124>(x);      // row_ror:4
    x += dpp_f<0x128>(x);      // row_ror:8
    return x;
}
template <int LPE>
QA_DEV int xor_(int x) {
    if (LPE == 4) return quad_or(x);
    x |= dpp_i<0x124>(x);
    x |= dpp_i<0x128>(x);
    return x;
}

// ---------------------------------------------------------------- 3-vectors
struct V3 { float x, y, z; };
QA_DEV V3 v3(float x, float y, float z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
QA_DEV V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
QA_DEV V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
QA_DEV V3 operator*(float s, V3 a) { return v3(s * a.x, s * a.y, s * a.z); }
QA_DEV float dot(V3 a, V3 b) { return fmaf(a.x, b.x, fmaf(a.y, b.y, a.z * b.z)); }
QA_DEV V3 cross(V3 a, V3 b) { return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }

// row-major 3x3
struct M3 { float m[9]; };
QA_DEV V3 mul(const M3 &A, V3 x) {
    return v3(fmaf(A.m[0], x.x, fmaf(A.m[1], x.y, A.m[2] * x.z)),
              fmaf(A.m[3], x.x, fmaf(A.m[4], x.y, A.m[5] * x.z)),
              fmaf(A.m[6], x.x, fmaf(A.m[7], x.y, A.m[8] * x.z)));
}
QA_DEV V3 mulT(const M3 &A, V3 x) {
    return v3(fmaf(A.m[0], x.x, fmaf(A.m[3], x.y, A.m[6] * x.z)),
              fmaf(A.m[1], x.x, fmaf(A.m[4], x.y, A.m[7] * x.z)),
              fmaf(A.m[2], x.x, fmaf(A.m[5], x.y, A.m[8] * x.z)));
}
QA_DEV M3 quat_to_mat(float x, float y, float z, float w) {   // xyzw, body -> world
    M3 R;
    R.m[0] = 1 - 2 * (y * y + z * z); R.m[1] = 2 * (x * y - z * w); R.m[2] = 2 * (x * z + y * w);
    R.m[3] = 2 * (x * y + z * w); R.m[4] = 1 - 2 * (x * x + z * z); R.m[5] = 2 * (y * z - x * w);
    R.m[6] = 2 * (x * z - y * w); R.m[7] = 2 * (y * z + x * w); R.m[8] = 1 - 2 * (x * x + y * y);
    return R;
}

// ---------------------------------------------------------------- spatial (6D) algebra, (angular; linear)
struct S6 { V3 a, l; };
QA_DEV S6 s6(V3 a, V3 l) { S6 r; r.a = a; r.l = l; return r; }
QA_DEV S6 operator+(S6 p, S6 q) { return s6(p.a + q.a, p.l + q.l); }
QA_DEV S6 operator*(float s, S6 p) { return s6(s * p.a, s * p.l); }
QA_DEV float dot(S6 p, S6 q) { return dot(p.a, q.a) + dot(p.l, q.l); }
QA_DEV S6 crm(S6 V, S6 M) { return s6(cross(V.a, M.a), cross(V.a, M.l) + cross(V.l, M.a)); }   // motion x motion
QA_DEV S6 crf(S6 V, S6 F) { return s6(cross(V.a, F.a) + cross(V.l, F.l), cross(V.a, F.l)); }   // motion x* force
QA_DEV S6 quad_sum(S6 p) {
    return s6(v3(quad_sum(p.a.x), quad_sum(p.a.y), quad_sum(p.a.z)), v3(quad_sum(p.l.x), quad_sum(p.l.y), quad_sum(p.l.z)));
}
template <int LPE>
QA_DEV S6 xsum(S6 p) {
    return s6(v3(xsum<LPE>(p.a.x), xsum<LPE>(p.a.y), xsum<LPE>(p.a.z)), v3(xsum<LPE>(p.l.x), xsum<LPE>(p.l.y), xsum<LPE>(p.l.z)));
}

// rigid-body inertia about the base origin in base axes: mass, first moment h = m c, I_O (xx yy zz xy xz yz)
struct RB { float m; V3 h; float xx, yy, zz, xy, xz, yz; };
QA_DEV V3 sym_mul(const RB &I, V3 w) {
    return v3(fmaf(I.xx, w.x, fmaf(I.xy, w.y, I.xz * w.z)), fmaf(I.xy, w.x, fmaf(I.yy, w.y, I.yz * w.z)),
              fmaf(I.xz, w.x, fmaf(I.yz, w.y, I.zz * w.z)));
}
QA_DEV S6 apply(const RB &I, S6 V) {   // (n; f) = I (w; v):  n = I_O w + h x v,  f = m v - h x w
    return s6(sym_mul(I, V.a) + cross(I.h, V.l), I.m * V.l - cross(I.h, V.a));
}
QA_DEV RB operator+(const RB &a, const RB &b) {
    RB r; r.m = a.m + b.m; r.h = a.h + b.h; r.xx = a.xx + b.xx; r.yy = a.yy + b.yy; r.zz = a.zz + b.zz;
    r.xy = a.xy + b.xy; r.xz = a.xz + b.xz; r.yz = a.yz + b.yz; return r;
}
QA_DEV RB quad_sum(const RB &a) {
    RB r; r.m = quad_sum(a.m); r.h = v3(quad_sum(a.h.x), quad_sum(a.h.y), quad_sum(a.h.z));
    r.xx = quad_sum(a.xx); r.yy = quad_sum(a.yy); r.zz = quad_sum(a.zz); r.xy = quad_sum(a.xy); r.xz = quad_sum(a.xz); r.yz = quad_sum(a.yz);
    return r;
}
template <int LPE>
QA_DEV RB xsum(const RB &a) {
    RB r; r.m = xsum<LPE>(a.m); r.h = v3(xsum<LPE>(a.h.x), xsum<LPE>(a.h.y), xsum<LPE>(a.h.z));
    r.xx = xsum<LPE>(a.xx); r.yy = xsum<LPE>(a.yy); r.zz = xsum<LPE>(a.zz); r.xy = xsum<LPE>(a.xy); r.xz = xsum<LPE>(a.xz); r.yz = xsum<LPE>(a.yz);
    return r;
}
// link inertia (mass m, CoM c_l and CoM inertia I6 in the link frame) moved to the base frame:
// rotation R (base <- link), link origin o
QA_DEV RB link_rb(float m, V3 cl, const float *I6, const M3 &R, V3 o) {
    V3 c = mul(R, cl) + o;
    // R I R^T, column by column
    V3 c0 = mul(R, v3(I6[0], I6[3], I6[4])), c1 = mul(R, v3(I6[3], I6[1], I6[5])), c2 = mul(R, v3(I6[4], I6[5], I6[2]));
    // (R I)_{ij} = c_j[i];  (R I R^T)_{ik} = sum_j c_j[i] R[k][j]
    float bxx = c0.x * R.m[0] + c1.x * R.m[1] + c2.x * R.m[2];
    float bxy = c0.x * R.m[3] + c1.x * R.m[4] + c2.x * R.m[5];
    float bxz = c0.x * R.m[6] + c1.x * R.m[7] + c2.x * R.m[8];
    float byy = c0.y * R.m[3] + c1.y * R.m[4] + c2.y * R.m[5];
    float byz = c0.y * R.m[6] + c1.y * R.m[7] + c2.y * R.m[8];
    float bzz = c0.z * R.m[6] + c1.z * R.m[7] + c2.z * R.m[8];
    float cc = dot(c, c);
    RB r; r.m = m; r.h = m * c;
    r.xx = bxx + m * (cc - c.x * c.x); r.yy = byy + m * (cc - c.y * c.y); r.zz = bzz + m * (cc - c.z * c.z);
    r.xy = bxy - m * c.x * c.y; r.xz = bxz - m * c.x * c.z; r.yz = byz - m * c.y * c.z;
    return r;
}

// ---------------------------------------------------------------- Philox4x32-10 (same streams as oracle/qa_oracle.c)
struct U4 { uint32_t v[4]; };
QA_DEV U4 philox(uint64_t seed, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3) {
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        uint32_t h0 = __umulhi(0xD2511F53u, c0), l0 = 0xD2511F53u * c0;
        uint32_t h1 = __umulhi(0xCD9E8D57u, c2), l1 = 0xCD9E8D57u * c2;
        uint32_t n0 = h1 ^ c1 ^ k0, n2 = h0 ^ c3 ^ k1;
        c0 = n0; c1 = l1; c2 = n2; c3 = l0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    U4 o; o.v[0] = c0; o.v[1] = c1; o.v[2] = c2; o.v[3] = c3; return o;
}
struct F4 { float v[4]; };
QA_DEV F4 rng4(uint64_t seed, uint32_t env, int64_t step, int stream, int block) {
    U4 o = philox(seed, env, (uint32_t)step, (uint32_t)(stream * 256 + block), (uint32_t)((uint64_t)step >> 32));
    F4 f;
#pragma unroll
    for (int i = 0; i < 4; ++i) f.v[i] = (float)(o.v[i] >> 8) * (1.0f / 16777216.0f);
    return f;
}
enum { RS_INIT_BUCKET = 1, RS_INIT_FRICTION = 2, RS_INIT_MASS = 3, RS_INIT_MOTOR = 4,
       RS_NOISE = 8, RS_CMD = 9, RS_CMD_RESET = 10, RS_PUSH = 11, RS_RESET = 12 };

QA_DEV float clampf(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }
