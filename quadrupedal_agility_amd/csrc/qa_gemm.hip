// qa_gemm.hip -- the dense layers of the learner's networks as hand-written fp32-MFMA GEMMs for gfx950.
//
// What they replace: the Linear(+ELU) blocks of the reference's actor / critic / estimator / encoder MLPs under training
// (bbc/rsl_rl/modules/actor_critic.py:92-139 `_mlp` blocks and :171-225, estimator.py:12-36; exercised by
// SSInfoGAIL.update_actor_critic, bbc/rsl_rl/algorithms/gail.py:328-413, and by tsc/rsl_rl/algorithms/ppo.py:160-282) --
// in PyTorch: addmm + elu forward; elu_backward, mm, mm^T and sum(0) backward, i.e. library GEMMs with aten elementwise and
// reduction kernels between them.  Here one layer is
//   forward            y  = act(x W^T + b)                      one launch, bias + activation in the epilogue
//   backward (input)   gx = (g W) * act'(y_prev)                one launch, the PREVIOUS layer's activation derivative in the epilogue
//   backward (weight)  dW = g^T x,  db = column sums of g       one split-K launch (db rides on the MFMAs against a row of ones)
//                                                               + one fixed-order slab reduction
// fp32 in, fp32 accumulate (v_mfma_f32_16x16x4_f32: bit-for-bit a k-ordered fmaf chain -- the reference's dtype, no reduced precision).
//
// One kernel template covers the three products.  It computes
//     out[b * ldo + a] = epilogue( sum_k Aop(a, k) * Bop(b, k) )
// where the "A side" index `a` is the CONTIGUOUS index of the output (MFMA D rows: a lane holds 4 consecutive a for one b, so
// bias / activation / derivative / store are 16-byte operations) and each operand is addressed either k-contiguous
// (KC: P[idx * ld + k], an nn.Linear weight read along its input features, an activation matrix read along its features) or
// index-contiguous (MC: P[k * ld + idx], the same matrices when the reduction runs over their ROWS):
//     forward          a = out feature   A = W  (KC)   b = sample      B = x (KC)   k = in feature
//     backward input   a = in feature    A = W  (MC)   b = sample      B = g (KC)   k = out feature
//     backward weight  a = in feature    A = x  (MC)   b = out feature B = g (MC)   k = sample   (split over z)
// Tiles: BA x BB outputs per 256-thread workgroup (4 wavefronts as 2 x 2, each (BA/2) x (BB/2) = TA x TB MFMA tiles), BK = 16.
// Global -> registers -> LDS, double-buffered, one barrier per k-tile: the loads of tile t+1 are in flight while tile t is
// multiplied (fp32 MFMA is 32 cycles per instruction: a 128 x 128 x 16 tile is 2,048 MFMA cycles per wavefront against 16 KB of
// operands, so neither the LDS nor the L2 is near its limit and two resident workgroups per CU cover each other's barriers).
// LDS images are padded by 4 floats per row: KC rows of 16 k (+4) are read as ONE ds_read_b128 per fragment -- the lane's four
// values are k = 4 (lane>>4) + s for the four MFMA steps s of the tile, a permutation of k that both operands share -- and MC
// rows of BT indices (+4) as four ds_read_b32; both are bank-conflict-free.  Workgroup ids are remapped so that consecutive
// tiles (which share an operand tile) run on the same XCD and hit in its L2.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <type_traits>
#include <mutex>

#include "../../include/qa_sim.h"

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

// Convolution as a GEMM over a channels-last image: operand rows are the windows of an [image][y][x][channel] tensor, read in
// place -- row p (an output pixel) starts at element rowbase(p) = ((img * ipix) + oy * iw + ox) * cin, reduction index k = (tap, channel)
// adds koff(k) = ((tap / kw) * iw + tap % kw) * cin + channel.  cin is a power of two >= 16, so a 16-wide k-tile lies inside one tap.
// Divisions by the per-call constants are multiplications: p / opix with m = ceil(2^40 / opix) (256 < opix, exact for p < 2^40 / opix), the
// in-image ones with m = ceil(2^32 / d) and the high half of the product (2 <= d, exact for n < 2^32 / d).
struct ConvGeom {
    int opix, ow, iw, ipix, cin, cs, kw;
    uint32_t m_opix, m_ow, m_kw;
};
static __device__ __forceinline__ uint32_t conv_div(uint32_t n, uint32_t m) { return (uint32_t)(((uint64_t)n * m) >> 40); }
static __device__ __forceinline__ uint32_t conv_div_small(uint32_t n, uint32_t m) { return __umulhi(n, m); }      // m = ceil(2^32 / d), n < 2^32 / d
static __device__ __forceinline__ uint32_t conv_rowbase(const ConvGeom &c, int p) {
    const uint32_t img = conv_div((uint32_t)p, c.m_opix), r = (uint32_t)p - img * (uint32_t)c.opix;
    const uint32_t oy = conv_div_small(r, c.m_ow), ox = r - oy * (uint32_t)c.ow;
    return (img * (uint32_t)c.ipix + oy * (uint32_t)c.iw + ox) << c.cs;
}
static __device__ __forceinline__ uint32_t conv_koff(const ConvGeom &c, int k) {
    const uint32_t tap = (uint32_t)k >> c.cs, ch = (uint32_t)k & (uint32_t)(c.cin - 1);
    const uint32_t ky = conv_div_small(tap, c.m_kw), kx = tap - ky * (uint32_t)c.kw;
    return ((ky * (uint32_t)c.iw + kx) << c.cs) + ch;
}

struct GemmArgs {
    const float *A, *B;
    float *out;
    const float *bias;       // EPI 1: per a (may be null)
    const float *yprev;      // EPI 2: activation OUTPUT whose derivative multiplies the product, [b * ldy + a] (may be null)
    float *ones_out;         // ONES: per-split sums of Bop(b, k) over k, [z * ones_split_stride + b]
    int64_t lda, ldb, ldo, ldy, out_split_stride, ones_split_stride;
    int a_count, b_count, kred, k_per_split, na, nb, nsplit;
    int a_vec, b_vec, o_vec, act;
    int a_ld_count;          // > 0: the A operand's index count AS LOADED (a_count rounded up to 4: the rows have readable padding); products into a >= a_count are never stored
    float alpha;
    ConvGeom cg;             // CONV 1: the B side (k-contiguous) is a window matrix; CONV 2: the A side (index-contiguous, reduction over pixels)
};

#define GEMM_BK 16
#ifndef QA_GEMM_DMA_DEFAULT
#define QA_GEMM_DMA_DEFAULT 0
#endif

// global -> staging registers.  KC: element (idx, k) at P[idx * ld + k]; MC: at P[k * ld + idx].  The loads are UNCONDITIONAL from
// clamped (always valid) addresses; out-of-range elements are zeroed when the registers go to LDS (gemm_store), by a MULTIPLICATION with
// a 0/1 mask.  A load inside a conditional, a select on a loaded value (hipcc sinks the load into a branch) and even a mask applied next
// to the load in the loop body are all waited for one at a time -- 16 dependent round trips per tile instead of 16 loads in flight
// behind the tile's MFMAs.  (The clamped address holds an element of the same row / column, so mask * value never manufactures a NaN
// that the true operands do not contain.)
template <int BT, bool MC, int VEC>
static __device__ __forceinline__ void gemm_load(float (&r)[BT / 16], const float *__restrict__ P, int64_t ld, int idx0, int count, int k0,
                                                 int kend, int tid) {
    if (!MC) {
        if (VEC == 4 && BT == 32) {          // 32 rows x 16 k = two floats per thread (the 32-channel input gradient of the window GEMM)
            const int kc = min(k0 + (tid & 7) * 2, kend - 2);
            const f2 v = *(const f2 *)(P + (uint32_t)(min(idx0 + (tid >> 3), count - 1) * (int)ld + kc));
            r[0] = v.x; r[1] = v.y;
        } else if (VEC == 4) {
            const int kc = min(k0 + (tid & 3) * 4, kend - 4);
#pragma unroll
            for (int j = 0; j < BT / 64; ++j) {
                const int row = idx0 + (tid >> 2) + 64 * j;
                const f4 v = *(const f4 *)(P + (uint32_t)(min(row, count - 1) * (int)ld + kc));
                r[4 * j] = v.x; r[4 * j + 1] = v.y; r[4 * j + 2] = v.z; r[4 * j + 3] = v.w;
            }
        } else {
            const int kc = min(k0 + (tid & 15), kend - 1);
#pragma unroll
            for (int j = 0; j < BT / 16; ++j) {
                const int row = idx0 + (tid >> 4) + 16 * j;
                r[j] = P[(uint32_t)(min(row, count - 1) * (int)ld + kc)];
            }
        }
    } else {
        if (VEC == 4) {
            constexpr int TPR = BT / 4, RPP = 256 / TPR;
            const int ic = min(idx0 + (tid % TPR) * 4, count - 4);
#pragma unroll
            for (int j = 0; j < 16 / RPP; ++j) {
                const int k = k0 + tid / TPR + RPP * j;
                const f4 v = *(const f4 *)(P + (uint32_t)(min(k, kend - 1) * (int)ld + ic));
                r[4 * j] = v.x; r[4 * j + 1] = v.y; r[4 * j + 2] = v.z; r[4 * j + 3] = v.w;
            }
        } else {
            constexpr int RPP = 256 / BT;
            const int ic = min(idx0 + tid % BT, count - 1);
#pragma unroll
            for (int j = 0; j < 16 / RPP; ++j) {
                const int k = k0 + tid / BT + RPP * j;
                r[j] = P[(uint32_t)(min(k, kend - 1) * (int)ld + ic)];
            }
        }
    }
}

// CONV 1: k-contiguous window rows, 16-byte loads.  rb[j] = rowbase of the thread's j-th row (fixed for the whole launch); the k-tile's
// tap offset is uniform over the workgroup (16 | cin).
template <int BT>
static __device__ __forceinline__ void gemm_conv_rows(uint32_t (&rb)[BT / 64], const ConvGeom &cg, int idx0, int count, int tid) {
#pragma unroll
    for (int j = 0; j < BT / 64; ++j) rb[j] = conv_rowbase(cg, min(idx0 + (tid >> 2) + 64 * j, count - 1));
}
template <int BT>
static __device__ __forceinline__ void gemm_load_conv_kc(float (&r)[BT / 16], const float *__restrict__ P, const uint32_t (&rb)[BT / 64],
                                                         const ConvGeom &cg, int k0, int kend, int tid) {
    const uint32_t ko = conv_koff(cg, min(k0, kend - GEMM_BK)) + (tid & 3) * 4;
#pragma unroll
    for (int j = 0; j < BT / 64; ++j) {
        const f4 v = *(const f4 *)(P + (rb[j] + ko));
        r[4 * j] = v.x; r[4 * j + 1] = v.y; r[4 * j + 2] = v.z; r[4 * j + 3] = v.w;
    }
}
// CONV 2: the reduction runs over the pixels (rows of the window matrix), the thread's four consecutive (tap, channel) columns are fixed
template <int BT>
static __device__ __forceinline__ void gemm_load_conv_mc(float (&r)[BT / 16], const float *__restrict__ P, uint32_t coloff, const ConvGeom &cg,
                                                         int k0, int kend, int tid) {
    constexpr int TPR = BT / 4, RPP = 256 / TPR;
#pragma unroll
    for (int j = 0; j < 16 / RPP; ++j) {
        const int k = min(k0 + tid / TPR + RPP * j, kend - 1);
        const f4 v = *(const f4 *)(P + (conv_rowbase(cg, k) + coloff));
        r[4 * j] = v.x; r[4 * j + 1] = v.y; r[4 * j + 2] = v.z; r[4 * j + 3] = v.w;
    }
}

// staging registers -> LDS image, out-of-range elements zeroed.  KC image: [idx][k] with a row stride of 20 floats; MC image: [k][idx]
// with a row stride of BT + 4.  Same thread -> element mapping as gemm_load.
template <int BT, bool MC, int VEC>
static __device__ __forceinline__ void gemm_store(const float (&r)[BT / 16], float *__restrict__ S, int idx0, int count, int k0, int kend, int tid) {
    if (!MC) {
        if (VEC == 4 && BT == 32) {
            const float ok = (k0 + (tid & 7) * 2 < kend && idx0 + (tid >> 3) < count) ? 1.f : 0.f;
            f2 v = {r[0] * ok, r[1] * ok};
            *(f2 *)(S + (tid >> 3) * (GEMM_BK + 4) + (tid & 7) * 2) = v;
        } else if (VEC == 4) {
            const float kok = (k0 + (tid & 3) * 4 < kend) ? 1.f : 0.f;
#pragma unroll
            for (int j = 0; j < BT / 64; ++j) {
                const float ok = (idx0 + (tid >> 2) + 64 * j < count) ? kok : 0.f;
                f4 v = {r[4 * j] * ok, r[4 * j + 1] * ok, r[4 * j + 2] * ok, r[4 * j + 3] * ok};
                *(f4 *)(S + ((tid >> 2) + 64 * j) * (GEMM_BK + 4) + (tid & 3) * 4) = v;
            }
        } else {
            const float kok = (k0 + (tid & 15) < kend) ? 1.f : 0.f;
#pragma unroll
            for (int j = 0; j < BT / 16; ++j)
                S[((tid >> 4) + 16 * j) * (GEMM_BK + 4) + (tid & 15)] = r[j] * ((idx0 + (tid >> 4) + 16 * j < count) ? kok : 0.f);
        }
    } else {
        if (VEC == 4) {
            constexpr int TPR = BT / 4, RPP = 256 / TPR;
            const float iok = (idx0 + (tid % TPR) * 4 < count) ? 1.f : 0.f;
#pragma unroll
            for (int j = 0; j < 16 / RPP; ++j) {
                const float ok = (k0 + tid / TPR + RPP * j < kend) ? iok : 0.f;
                f4 v = {r[4 * j] * ok, r[4 * j + 1] * ok, r[4 * j + 2] * ok, r[4 * j + 3] * ok};
                *(f4 *)(S + (tid / TPR + RPP * j) * (BT + 4) + (tid % TPR) * 4) = v;
            }
        } else {
            constexpr int RPP = 256 / BT;
            const float iok = (idx0 + tid % BT < count) ? 1.f : 0.f;
#pragma unroll
            for (int j = 0; j < 16 / RPP; ++j)
                S[(tid / BT + RPP * j) * (BT + 4) + tid % BT] = r[j] * ((k0 + tid / BT + RPP * j < kend) ? iok : 0.f);
        }
    }
}

// the four MFMA-step values (k = 4 kq + s, s = 0..3) of the 16-index fragment starting at local index i0
template <int BT, bool MC>
static __device__ __forceinline__ void gemm_frag(float (&f)[4], const float *__restrict__ S, int i0, int li, int kq) {
    if (!MC) {
        const f4 v = *(const f4 *)(S + (i0 + li) * (GEMM_BK + 4) + kq * 4);
        f[0] = v.x; f[1] = v.y; f[2] = v.z; f[3] = v.w;
    } else {
#pragma unroll
        for (int s = 0; s < 4; ++s) f[s] = S[(kq * 4 + s) * (BT + 4) + i0 + li];
    }
}

// the epilogue of both kernels: the lane holds out[b][a .. a+3] for each of its TA x TB tiles (MFMA D rows = 4 consecutive a)
template <int BA, int BB, int EPI, bool ONES>
static __device__ __forceinline__ void gemm_epilogue(const GemmArgs &g, const f4 (&acc)[BA / 32][BB / 32], const f4 (&oacc)[BB / 32], int a_base, int b_base,
                                                     int at, int z, int wa, int wb, int li, int kq, bool ones_wave) {
    constexpr int TA = BA / 32, TB = BB / 32;
    float *out = g.out + (int64_t)z * g.out_split_stride;
#pragma unroll
    for (int i = 0; i < TA; ++i) {
        const int a = a_base + wa * (BA / 2) + i * 16 + 4 * kq;
        if (a >= g.a_count) continue;
        const bool full = g.o_vec == 4 && a + 3 < g.a_count;
        f4 bias = {0.f, 0.f, 0.f, 0.f};
        if (EPI == 1 && g.bias) {
            if (full) bias = *(const f4 *)(g.bias + a);
            else {
                bias.x = g.bias[a];
                if (a + 1 < g.a_count) bias.y = g.bias[a + 1];
                if (a + 2 < g.a_count) bias.z = g.bias[a + 2];
                if (a + 3 < g.a_count) bias.w = g.bias[a + 3];
            }
        }
#pragma unroll
        for (int j = 0; j < TB; ++j) {
            const int b = b_base + wb * (BB / 2) + j * 16 + li;
            if (b >= g.b_count) continue;
            f4 v = acc[i][j];
            if (EPI == 1) {
                v += bias;
                if (g.act == 1) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = v[r] > 0.f ? v[r] : g.alpha * (expf(v[r]) - 1.f);
                } else if (g.act == 2) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
                }
            }
            if (EPI == 2 && g.yprev && g.act != 0) {
                const float *yp = g.yprev + (int64_t)b * g.ldy + a;
                f4 y = {1.f, 1.f, 1.f, 1.f};
                if (full) y = *(const f4 *)yp;
                else {
                    y.x = yp[0];
                    if (a + 1 < g.a_count) y.y = yp[1];
                    if (a + 2 < g.a_count) y.z = yp[2];
                    if (a + 3 < g.a_count) y.w = yp[3];
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] *= (g.act == 1) ? (y[r] > 0.f ? 1.f : y[r] + g.alpha) : (y[r] > 0.f ? 1.f : 0.f);
            }
            float *o = out + (int64_t)b * g.ldo + a;
            if (full) *(f4 *)o = v;
            else {
                o[0] = v.x;
                if (a + 1 < g.a_count) o[1] = v.y;
                if (a + 2 < g.a_count) o[2] = v.z;
                if (a + 3 < g.a_count) o[3] = v.w;
            }
        }
    }
    if (ONES) {
        if (ones_wave && kq == 0) {       // D row 0 of the ones product: this workgroup's share of sum_k Bop(b, k)
#pragma unroll
            for (int j = 0; j < TB; ++j) {
                const int b = b_base + wb * (BB / 2) + j * 16 + li;
                if (b < g.b_count) g.ones_out[((int64_t)z * g.na + at) * g.ones_split_stride + b] = oacc[j].x;
            }
        }
    }
}

// EPI 0: plain store (split-K partial).  1: + bias, activation (act 0 none, 1 ELU(alpha), 2 ReLU).  2: * activation derivative taken from
// the activation's output yprev (ELU: y > 0 ? 1 : y + alpha; ReLU: y > 0; none: 1).
// ABL (tools/gemm_probe.hip only): 1 = no global loads / LDS writes after the first tile, 2 = no MFMAs, 3 = no LDS fragment reads
template <int BA, int BB, bool A_MC, bool B_MC, int AV, int BV, int EPI, bool ONES, int ABL = 0, int CONV = 0>
static __device__ __forceinline__ void qa_gemm_body(const GemmArgs &g, const int block_id) {
    static_assert(CONV == 0 || (CONV == 1 && !B_MC && BV == 4) || (CONV == 2 && A_MC && AV == 4), "window operands are read with 16-byte loads");
    constexpr int TA = BA / 32, TB = BB / 32;
    constexpr int A_TILE = A_MC ? GEMM_BK * (BA + 4) : BA * (GEMM_BK + 4);
    constexpr int B_TILE = B_MC ? GEMM_BK * (BB + 4) : BB * (GEMM_BK + 4);
    __shared__ __attribute__((aligned(16))) float lds[2 * (A_TILE + B_TILE)];

    // consecutive workgroup ids are dealt round-robin to the 8 XCDs: remap so that each XCD owns a contiguous run of tiles
    const int total = g.na * g.nb * g.nsplit;
    int id = block_id;
    {
        const int q = total >> 3, rmd = total & 7, xcd = id & 7, local = id >> 3;
        id = (xcd < rmd) ? xcd * (q + 1) + local : rmd * (q + 1) + (xcd - rmd) * q + local;
    }
    const int at = id % g.na, bt = (id / g.na) % g.nb, z = id / (g.na * g.nb);
    const int a_base = at * BA, b_base = bt * BB;
    const int kbeg = z * g.k_per_split;
    const int kend = min(g.kred, kbeg + g.k_per_split);
    const int T = (kend - kbeg + GEMM_BK - 1) / GEMM_BK;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wa = wave & 1, wb = wave >> 1, li = lane & 15, kq = lane >> 4;
    // the bias sums (ONES) ride on extra MFMAs against a fragment of ones: k-tile t of a split belongs to the a-tile t % na, so every
    // workgroup carries 1/na of them (all on the a-tile 0 workgroups they were +25 % on one workgroup in na -- and the launch waits for those)
    const bool ones_wave = ONES && wa == 0;

    f4 acc[TA][TB];
    f4 oacc[TB];
#pragma unroll
    for (int i = 0; i < TA; ++i)
#pragma unroll
        for (int j = 0; j < TB; ++j) acc[i][j] = (f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < TB; ++j) oacc[j] = (f4){0.f, 0.f, 0.f, 0.f};

    // Software pipeline, two tiles deep in registers: tile j is loaded into staging set j & 1 and its fragments live in fragment set
    // j & 1.  Iteration t issues the loads of tile t+2, multiplies tile t from registers, writes tile t+1 (loaded one iteration ago) to
    // the other LDS buffer between the two halves of its MFMAs, and after the barrier reads the fragments of tile t+1 -- so a load has
    // almost two iterations to land, the LDS writes ride inside the MFMA stream, and the fragment reads overlap the next iteration's
    // load issue instead of standing in front of its MFMAs.
    float sa[2][BA / 16], sb[2][BB / 16];
    // The 128 x 128 tile keeps ONE fragment set (the second costs 32 registers: 177 -> occupancy 2, and its 768 workgroups per
    // 24,576 x 512 product then run as 1.5 rounds of 512): it reads tile t+1's fragments after tile t's last MFMA, three waves per SIMD
    // covering the read latency.
    constexpr int NF = (BA * BB >= 128 * 128) ? 1 : 2;
    float af[NF][TA][4], bf[NF][TB][4];
    uint32_t conv_rb[CONV == 1 ? BB / 64 : 1];
    uint32_t conv_col = 0;
    if (CONV == 1) gemm_conv_rows<CONV == 1 ? BB : 64>(conv_rb, g.cg, b_base, g.b_count, tid);
    if (CONV == 2) conv_col = conv_koff(g.cg, min(a_base + (tid % (BA / 4)) * 4, g.a_count - 4));
    const int a_ldc = g.a_ld_count > 0 ? g.a_ld_count : g.a_count;
    auto load_a = [&](float (&r)[BA / 16], int k0) {
        if constexpr (CONV == 2) gemm_load_conv_mc<BA>(r, g.A, conv_col, g.cg, k0, kend, tid);
        else gemm_load<BA, A_MC, AV>(r, g.A, g.lda, a_base, a_ldc, k0, kend, tid);
    };
    auto load_b = [&](float (&r)[BB / 16], int k0) {
        if constexpr (CONV == 1) gemm_load_conv_kc<BB>(r, g.B, conv_rb, g.cg, k0, kend, tid);
        else gemm_load<BB, B_MC, BV>(r, g.B, g.ldb, b_base, g.b_count, k0, kend, tid);
    };
    if (T > 0) {
        load_a(sa[0], kbeg);
        load_b(sb[0], kbeg);
        load_a(sa[1], kbeg + GEMM_BK);
        load_b(sb[1], kbeg + GEMM_BK);
        gemm_store<BA, A_MC, AV>(sa[0], lds, a_base, a_ldc, kbeg, kend, tid);
        gemm_store<BB, B_MC, BV>(sb[0], lds + A_TILE, b_base, g.b_count, kbeg, kend, tid);
    }
    __syncthreads();
    if (T > 0) {
#pragma unroll
        for (int i = 0; i < TA; ++i) gemm_frag<BA, A_MC>(af[0][i], lds, wa * (BA / 2) + i * 16, li, kq);
#pragma unroll
        for (int j = 0; j < TB; ++j) gemm_frag<BB, B_MC>(bf[0][j], lds + A_TILE, wb * (BB / 2) + j * 16, li, kq);
    }
    auto mfma_steps = [&](auto ptag, auto s0tag, bool do_ones) {
        constexpr int P = decltype(ptag)::value, S0 = decltype(s0tag)::value;
#pragma unroll
        for (int s = S0; s < S0 + 2; ++s) {
#pragma unroll
            for (int i = 0; i < TA; ++i)
#pragma unroll
                for (int j = 0; j < TB; ++j)
                    if (ABL != 2) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[P % NF][i][s], bf[P % NF][j][s], acc[i][j], 0, 0, 0);
                    else acc[i][j][s] += af[P % NF][i][s] * bf[P % NF][j][s];
            if (ONES) {
                if (do_ones) {
#pragma unroll
                    for (int j = 0; j < TB; ++j) oacc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(1.0f, bf[P % NF][j][s], oacc[j], 0, 0, 0);
                }
            }
        }
    };
    auto body = [&](auto ptag, int t) {
        constexpr int P = decltype(ptag)::value, Q = P ^ 1;
        if (ABL != 1) {       // unconditional, also past the last tile (clamped addresses, result unused): a load inside `if (t + 2 < T)` makes
                              // hipcc count the waits of the stores below for the path that did NOT issue it, i.e. drain the new loads too
            load_a(sa[P], kbeg + (t + 2) * GEMM_BK);
            load_b(sb[P], kbeg + (t + 2) * GEMM_BK);
        }
        __builtin_amdgcn_sched_barrier(0);        // the loads stay in front of the MFMAs ...
        const bool do_ones = ones_wave && (t % g.na) == at;
        mfma_steps(ptag, std::integral_constant<int, 0>{}, do_ones);
        __builtin_amdgcn_sched_barrier(0);        // ... and nothing of the store section (its mask multiplies wait for tile t+1's loads) moves above them
        float *An = lds + Q * (A_TILE + B_TILE);
        if (ABL != 1) {       // past the last tile this writes zeros (k >= kend) into the idle buffer
            gemm_store<BA, A_MC, AV>(sa[Q], An, a_base, a_ldc, kbeg + (t + 1) * GEMM_BK, kend, tid);
            gemm_store<BB, B_MC, BV>(sb[Q], An + A_TILE, b_base, g.b_count, kbeg + (t + 1) * GEMM_BK, kend, tid);
        }
        if (NF == 2) {
            // every wave has written its share of tile t+1 (and finished reading tile t's fragments an iteration ago): read tile t+1's
            // fragments into the other register set NOW, under the second half of this tile's MFMAs
            __syncthreads();
            if (ABL != 3) {
#pragma unroll
                for (int i = 0; i < TA; ++i) gemm_frag<BA, A_MC>(af[Q % NF][i], An, wa * (BA / 2) + i * 16, li, kq);
#pragma unroll
                for (int j = 0; j < TB; ++j) gemm_frag<BB, B_MC>(bf[Q % NF][j], An + A_TILE, wb * (BB / 2) + j * 16, li, kq);
            }
            __builtin_amdgcn_sched_barrier(0);
            mfma_steps(ptag, std::integral_constant<int, 2>{}, do_ones);
            __builtin_amdgcn_sched_barrier(0);
        } else {
            __builtin_amdgcn_sched_barrier(0);
            mfma_steps(ptag, std::integral_constant<int, 2>{}, do_ones);
            __builtin_amdgcn_sched_barrier(0);
            __syncthreads();
            if (ABL != 3) {
#pragma unroll
                for (int i = 0; i < TA; ++i) gemm_frag<BA, A_MC>(af[0][i], An, wa * (BA / 2) + i * 16, li, kq);
#pragma unroll
                for (int j = 0; j < TB; ++j) gemm_frag<BB, B_MC>(bf[0][j], An + A_TILE, wb * (BB / 2) + j * 16, li, kq);
            }
        }
    };
    // an odd tile count runs one more (all-zero) tile: a conditional second half makes hipcc drain the load queue at the loop head
    for (int t = 0; t < T; t += 2) {
        body(std::integral_constant<int, 0>{}, t);
        body(std::integral_constant<int, 1>{}, t + 1);
    }

    gemm_epilogue<BA, BB, EPI, ONES>(g, acc, oacc, a_base, b_base, at, z, wa, wb, li, kq, ones_wave);
}
template <int BA, int BB, bool A_MC, bool B_MC, int AV, int BV, int EPI, bool ONES, int ABL = 0, int CONV = 0>
__global__ void __launch_bounds__(256, 2) qa_gemm_kernel(GemmArgs g) {
    qa_gemm_body<BA, BB, A_MC, B_MC, AV, BV, EPI, ONES, ABL, CONV>(g, (int)blockIdx.x);
}
// r6 (ABI 17): SEVERAL weight-gradient products in ONE launch (qa_linear_backward_weight_batch).  The products of a chain training step are a dozen
// (out x in) outputs over a few thousand rows -- 16-190 workgroups and ~15 us each when launched one after the other
// (profiles/r6_ppo_chain_step_sequence_512_one_stream.txt: 13 launches, 215 us of a 517 us step); side by side they are one launch of a few thousand
// workgroups.  Workgroup -> (product, tile) by a scan of the cumulative tile counts in the kernel arguments; every product of a launch shares the
// kernel's template instance (64 x 64 tiles, the operands' vector widths), its own GemmArgs otherwise.
constexpr int GEMM_GROUP_MAX = 12;
struct GemmGroup { GemmArgs g[GEMM_GROUP_MAX]; int start[GEMM_GROUP_MAX + 1]; int n; };
static_assert(sizeof(GemmGroup) <= 3800, "the group travels in the kernel arguments (4 KB)");
template <int AV, int BV>
__global__ void __launch_bounds__(256, 2) qa_wgrad_group_kernel(GemmGroup G) {
    int j = 0;
    while (j + 1 < G.n && (int)blockIdx.x >= G.start[j + 1]) ++j;
    qa_gemm_body<64, 64, true, true, AV, BV, 0, true>(G.g[j], (int)blockIdx.x - G.start[j]);
}

// ---- the same products with LDS-DMA staging (r4) ----
// For the wide trunk products (K and both index dimensions 16-byte friendly) the operand tiles go global -> LDS directly
// (`global_load_lds_dwordx4`: a wave-instruction lands 64 x 16 B as ONE contiguous KiB of LDS, wave-uniform base + 16 lane): no staging
// registers, no mask multiplies, no ds_write pass (a ds_write_b128 occupies the LDS data path for 13 cycles), and the loads of tile t+2
// are in flight across the barrier of tile t (three LDS stages, counted `s_waitcnt vmcnt`, raw `s_barrier` -- `__syncthreads()` would
// drain the DMA queue).  The LDS image is lane-linear, so it cannot be padded; bank conflicts of the fragment reads are removed by an
// XOR swizzle applied to the SOURCE address and to the read address alike (the destination stays linear):
//   KC image [row][16 k] (64-B rows, four 16-B chunks): slot = chunk ^ G[(row >> 2) & 3], G = {0, 3, 2, 1}  ->  every ds_read_b128 lane
//   group ({0-3,12-15,20-27}, ...) touches 16 distinct 4-bank runs;
//   MC image [16 k][BT idx]: chunk(idx / 4) ^ 4 * ((k >> 2) & 1), i.e. idx ^ 16 on the k rows of odd kq  ->  the two kq halves of a
//   32-lane ds_read_b32 group use disjoint banks.
// Requirements (checked by dma_ok(); everything else takes the register-staged kernel above): both operands 16-byte aligned with
// leading dimensions and index counts that are multiples of 4, the reduction length of every split a multiple of 16.  Out-of-range
// rows / columns of the last tiles are loaded from clamped (valid) addresses and only ever reach outputs that are not stored.
typedef const __attribute__((address_space(1))) void *qa_glb_ptr;
typedef __attribute__((address_space(3))) void *qa_lds_ptr;
static __device__ __forceinline__ void glds16(const float *src, float *dst_wave_uniform) {
    __builtin_amdgcn_global_load_lds((qa_glb_ptr)src, (qa_lds_ptr)dst_wave_uniform, 16, 0, 0);
}
static __device__ __forceinline__ int kc_swz(int row) { return (4 - ((row >> 2) & 3)) & 3; }      // G = {0, 3, 2, 1}

// per-lane element offsets (without the k-tile term) of the wave's DMA pieces of one operand tile: BT / 64 pieces per wave
template <int BT, bool MC>
static __device__ __forceinline__ void dma_offsets(uint32_t (&off)[BT / 64], int64_t ld, int idx0, int count, int wave, int lane) {
#pragma unroll
    for (int p = 0; p < BT / 64; ++p) {
        const int j = wave + 4 * p;                     // piece j of the tile: floats [256 j, 256 j + 256) of the LDS image
        if (!MC) {
            const int row = 16 * j + (lane >> 2), slot = lane & 3;
            const int chunk = slot ^ kc_swz(row);
            off[p] = (uint32_t)min(idx0 + row, count - 1) * (uint32_t)ld + (uint32_t)(4 * chunk);
        } else {
            const int f = 256 * j + 4 * lane, k = f / BT, pos = f - k * BT;
            off[p] = (uint32_t)k * (uint32_t)ld + (uint32_t)min(idx0 + pos, count - 4);      // r5: laid down as it lies (see dma_frags)
        }
    }
}
template <int BT>
static __device__ __forceinline__ void dma_issue(const float *__restrict__ P, const uint32_t (&off)[BT / 64], uint32_t tile_off, float *stage, int wave) {
#pragma unroll
    for (int p = 0; p < BT / 64; ++p) glds16(P + (off[p] + tile_off), stage + 256 * (wave + 4 * p));
}
// The T = BT / 32 fragments of a wave's half of an operand tile (w0 = first local index of the half), f[tile][MFMA step s].
// KC image: tile i covers the 16 consecutive indices w0 + 16 i + li, one ds_read_b128 along k per tile (k = 4 kq + s).
// MC image [16 k][BT idx] (r5): the image is index-contiguous, so a 16-byte read along the INDEX serves four TILES at once if tile c of a
// group of four is made of the indices w0 + 64 g + 4 li + c (a permutation of the output's rows / columns that the epilogue undoes for
// free -- it only renames which accumulator an output lives in).  Four ds_read_b128 per k-tile and group instead of sixteen ds_read_b32:
// r4's counter pass showed the index-contiguous products' wavefronts living 40 % longer on exactly those round trips.  With a row pitch
// of BT floats (a multiple of 64) the 16 lanes of every b128 service group ({0-3, 12-15, 20-27}, ...) fall on 16 different bank quads
// without a swizzle.  A half of 6 or 2 tiles ends in a group of two: indices w0 + 64 g + 2 li + c, ds_read_b64.
template <int BT, bool MC>
static __device__ __forceinline__ void dma_frags(float (&f)[BT / 32][4], const float *__restrict__ S, int w0, int li, int kq) {
    constexpr int T = BT / 32;
    if (!MC) {
#pragma unroll
        for (int i = 0; i < T; ++i) {
            const f4 v = *(const f4 *)(S + (w0 + 16 * i + li) * 16 + ((kq ^ kc_swz(li)) << 2));
            f[i][0] = v.x; f[i][1] = v.y; f[i][2] = v.z; f[i][3] = v.w;
        }
    } else {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const float *row = S + (kq * 4 + s) * BT + w0;
#pragma unroll
            for (int g = 0; g < T / 4; ++g) {
                const f4 v = *(const f4 *)(row + 64 * g + 4 * li);
                f[4 * g][s] = v.x; f[4 * g + 1][s] = v.y; f[4 * g + 2][s] = v.z; f[4 * g + 3][s] = v.w;
            }
            if (T % 4 == 2) {
                const f2 v = *(const f2 *)(row + 64 * (T / 4) + 2 * li);
                f[T - 2][s] = v.x; f[T - 1][s] = v.y;
            }
        }
    }
}
// where the outputs of the DMA kernel's accumulators live.  The lane holds, per B tile j, 4 T values on the A side: T chunks of four
// CONSECUTIVE a.  chunk q, element e  ->  (A tile, D row r) and the chunk's first a (relative to the wave's half).
template <int T, bool MC>
struct DmaAMap {
    static __device__ __forceinline__ int tile(int q, int e) { return !MC ? q : (q < 4 * (T / 4) ? 4 * (q / 4) + e : T - 2 + (e & 1)); }
    static __device__ __forceinline__ int row(int q, int e) { return !MC ? e : (q < 4 * (T / 4) ? (q & 3) : 2 * (q - 4 * (T / 4)) + (e >> 1)); }
    static __device__ __forceinline__ int a0(int q, int kq) {
        return !MC ? 16 * q + 4 * kq : (q < 4 * (T / 4) ? 64 * (q / 4) + 16 * kq + 4 * (q & 3) : 64 * (T / 4) + 8 * kq + 4 * (q - 4 * (T / 4)));
    }
};
template <int T, bool MC>
static __device__ __forceinline__ int dma_b_of(int j, int li) {       // the b (relative to the wave's half) of D column li of B tile j
    return !MC ? 16 * j + li : (j < 4 * (T / 4) ? 64 * (j / 4) + 4 * li + (j & 3) : 64 * (T / 4) + 2 * li + (j - 4 * (T / 4)));
}
template <int BA, int BB, bool A_MC, bool B_MC, int EPI, bool ONES>
static __device__ __forceinline__ void dma_epilogue(const GemmArgs &g, const f4 (&acc)[BA / 32][BB / 32], const f4 (&oacc)[BB / 32], int a_base, int b_base,
                                                    int at, int z, int wa, int wb, int li, int kq, bool ones_wave) {
    constexpr int TA = BA / 32, TB = BB / 32;
    typedef DmaAMap<TA, A_MC> AM;
    float *out = g.out + (int64_t)z * g.out_split_stride;
#pragma unroll
    for (int q = 0; q < TA; ++q) {
        const int a = a_base + wa * (BA / 2) + AM::a0(q, kq);
        if (a >= g.a_count) continue;
        const bool full = g.o_vec == 4 && a + 3 < g.a_count;
        f4 bias = {0.f, 0.f, 0.f, 0.f};
        if (EPI == 1 && g.bias) {
            if (full) bias = *(const f4 *)(g.bias + a);
            else {
                bias.x = g.bias[a];
                if (a + 1 < g.a_count) bias.y = g.bias[a + 1];
                if (a + 2 < g.a_count) bias.z = g.bias[a + 2];
                if (a + 3 < g.a_count) bias.w = g.bias[a + 3];
            }
        }
#pragma unroll
        for (int j = 0; j < TB; ++j) {
            const int b = b_base + wb * (BB / 2) + dma_b_of<TB, B_MC>(j, li);
            if (b >= g.b_count) continue;
            f4 v = {acc[AM::tile(q, 0)][j][AM::row(q, 0)], acc[AM::tile(q, 1)][j][AM::row(q, 1)], acc[AM::tile(q, 2)][j][AM::row(q, 2)], acc[AM::tile(q, 3)][j][AM::row(q, 3)]};
            if (EPI == 1) {
                v += bias;
                if (g.act == 1) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = v[r] > 0.f ? v[r] : g.alpha * (expf(v[r]) - 1.f);
                } else if (g.act == 2) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
                }
            }
            if (EPI == 2 && g.yprev && g.act != 0) {
                const float *yp = g.yprev + (int64_t)b * g.ldy + a;
                f4 y = {1.f, 1.f, 1.f, 1.f};
                if (full) y = *(const f4 *)yp;
                else {
                    y.x = yp[0];
                    if (a + 1 < g.a_count) y.y = yp[1];
                    if (a + 2 < g.a_count) y.z = yp[2];
                    if (a + 3 < g.a_count) y.w = yp[3];
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] *= (g.act == 1) ? (y[r] > 0.f ? 1.f : y[r] + g.alpha) : (y[r] > 0.f ? 1.f : 0.f);
            }
            float *o = out + (int64_t)b * g.ldo + a;
            if (full) *(f4 *)o = v;
            else {
                o[0] = v.x;
                if (a + 1 < g.a_count) o[1] = v.y;
                if (a + 2 < g.a_count) o[2] = v.z;
                if (a + 3 < g.a_count) o[3] = v.w;
            }
        }
    }
    if (ONES) {
        if (ones_wave && kq == 0) {       // D row 0 of the ones product: this workgroup's share of sum_k Bop(b, k)
#pragma unroll
            for (int j = 0; j < TB; ++j) {
                const int b = b_base + wb * (BB / 2) + dma_b_of<TB, B_MC>(j, li);
                if (b < g.b_count) g.ones_out[((int64_t)z * g.na + at) * g.ones_split_stride + b] = oacc[j].x;
            }
        }
    }
}

template <int BA, int BB, bool A_MC, bool B_MC, int EPI, bool ONES>
__global__ void __launch_bounds__(256, 2) qa_gemm_dma_kernel(GemmArgs g) {
    static_assert(BA % 64 == 0 && BB % 64 == 0, "every wave issues the same number of DMA pieces per tile");
    constexpr int TA = BA / 32, TB = BB / 32;
    constexpr int STAGE = GEMM_BK * (BA + BB), NSTAGE = 3;
    constexpr int NLOAD = BA / 64 + BB / 64;            // DMA pieces per wave and tile
    __shared__ __attribute__((aligned(16))) float lds[NSTAGE * STAGE];      // ONE LDS object (a second one makes hipcc drain the DMA queue before every ds_read)

    const int total = g.na * g.nb * g.nsplit;
    int id = blockIdx.x;
    {
        const int q = total >> 3, rmd = total & 7, xcd = id & 7, local = id >> 3;
        id = (xcd < rmd) ? xcd * (q + 1) + local : rmd * (q + 1) + (xcd - rmd) * q + local;
    }
    const int at = id % g.na, bt = (id / g.na) % g.nb, z = id / (g.na * g.nb);
    const int a_base = at * BA, b_base = bt * BB;
    const int kbeg = z * g.k_per_split;
    const int kend = min(g.kred, kbeg + g.k_per_split);
    const int T = (kend - kbeg) / GEMM_BK;              // exact: dma_ok()

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wa = wave & 1, wb = wave >> 1, li = lane & 15, kq = lane >> 4;
    const bool ones_wave = ONES && wa == 0;

    f4 acc[TA][TB];
    f4 oacc[TB];
#pragma unroll
    for (int i = 0; i < TA; ++i)
#pragma unroll
        for (int j = 0; j < TB; ++j) acc[i][j] = (f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < TB; ++j) oacc[j] = (f4){0.f, 0.f, 0.f, 0.f};

    uint32_t offa[BA / 64], offb[BB / 64];
    dma_offsets<BA, A_MC>(offa, g.lda, a_base, g.a_count, wave, lane);
    dma_offsets<BB, B_MC>(offb, g.ldb, b_base, g.b_count, wave, lane);
    const uint32_t stepa = A_MC ? (uint32_t)GEMM_BK * (uint32_t)g.lda : (uint32_t)GEMM_BK;
    const uint32_t stepb = B_MC ? (uint32_t)GEMM_BK * (uint32_t)g.ldb : (uint32_t)GEMM_BK;
    const uint32_t basea = A_MC ? (uint32_t)kbeg * (uint32_t)g.lda : (uint32_t)kbeg;
    const uint32_t baseb = B_MC ? (uint32_t)kbeg * (uint32_t)g.ldb : (uint32_t)kbeg;
    auto issue = [&](int t, int stage) {       // past the last tile: the last tile again, into a stage nobody reads any more (keeps the wait counts uniform)
        const uint32_t tt = (uint32_t)min(t, T - 1);
        float *S = lds + stage * STAGE;
        dma_issue<BA>(g.A, offa, basea + tt * stepa, S, wave);
        dma_issue<BB>(g.B, offb, baseb + tt * stepb, S + GEMM_BK * BA, wave);
    };
    if (T > 0) {
        issue(0, 0);
        issue(1, 1);
    }
    int stage = 0;
    for (int t = 0; t < T; ++t) {
        // this wave's pieces of tile t have landed (tile t+1's stay in flight) ...
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLOAD) : "memory");
        // ... and after the barrier so have everybody's; everybody has also finished reading tile t-1, whose stage tile t+2 now takes
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        const float *As = lds + stage * STAGE, *Bs = As + GEMM_BK * BA;
        float af[TA][4], bf[TB][4];
        dma_frags<BA, A_MC>(af, As, wa * (BA / 2), li, kq);
        dma_frags<BB, B_MC>(bf, Bs, wb * (BB / 2), li, kq);
        issue(t + 2, stage == 0 ? 2 : stage - 1);       // (t + 2) % 3
        __builtin_amdgcn_sched_barrier(0);
        const bool do_ones = ones_wave && (t % g.na) == at;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
#pragma unroll
            for (int i = 0; i < TA; ++i)
#pragma unroll
                for (int j = 0; j < TB; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i][s], bf[j][s], acc[i][j], 0, 0, 0);
            if (ONES) {
                if (do_ones) {
#pragma unroll
                    for (int j = 0; j < TB; ++j) oacc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(1.0f, bf[j][s], oacc[j], 0, 0, 0);
                }
            }
        }
        stage = stage == 2 ? 0 : stage + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the two dummy tiles: no DMA may outlive the workgroup's LDS allocation
    dma_epilogue<BA, BB, A_MC, B_MC, EPI, ONES>(g, acc, oacc, a_base, b_base, at, z, wa, wb, li, kq, ones_wave);
}

// gw[i] = sum over the nsplit_w slabs (in order) of slabs[z * stride_w + i], i < n_w;  gb[i] = sum over the nsplit_b rows of
// bslabs[z * stride_b + i], i < n_b: the split-K partials of a weight gradient and of its bias gradient.  Fixed order: bit-reproducible,
// no atomics.  Eight independent loads in flight per thread before the (ordered) additions: a plain accumulate loop is one dependent
// L2 round trip per slab.
__global__ void __launch_bounds__(256) qa_slab_reduce_kernel(const float *__restrict__ slabs, int64_t stride_w, int nsplit_w, int64_t n_w,
                                                             const float *__restrict__ bslabs, int64_t stride_b, int nsplit_b, int64_t n_b,
                                                             float *__restrict__ gw, float *__restrict__ gb, int vec) {
    const int64_t nvec_w = (n_w + 3) >> 2, nvec_b = (n_b + 3) >> 2;
    int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= nvec_w + nvec_b) return;
    const bool is_b = t >= nvec_w;
    const int64_t i = (is_b ? t - nvec_w : t) * 4;
    const int64_t n = is_b ? n_b : n_w, stride = is_b ? stride_b : stride_w;
    const int nsplit = is_b ? nsplit_b : nsplit_w;
    const float *src = (is_b ? bslabs : slabs) + i;
    float *dst = (is_b ? gb : gw) + i;
    if (vec == 4 && i + 3 < n && ((uintptr_t)dst & 15) == 0) {
        f4 s = {0.f, 0.f, 0.f, 0.f};
        int zz = 0;
        for (; zz + 8 <= nsplit; zz += 8) {
            f4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = *(const f4 *)(src + (int64_t)(zz + u) * stride);
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; zz < nsplit; ++zz) s += *(const f4 *)(src + (int64_t)zz * stride);
        *(f4 *)dst = s;
    } else {
        for (int r = 0; r < 4 && i + r < n; ++r) {
            float s = 0.f;
            for (int zz = 0; zz < nsplit; ++zz) s += src[(int64_t)zz * stride + r];
            dst[r] = s;
        }
    }
}

// y[r][o] = act( sum over the nsplit slabs (in order) of slabs[z * stride + r * n + o] + bias[o] ): the second half of a forward layer whose
// reduction dimension was split (few output tiles, very long rows: the depth encoder's 62,400 -> 128 layer)
__global__ void __launch_bounds__(256) qa_slab_bias_act_kernel(const float *__restrict__ slabs, int64_t stride, int nsplit, int64_t rows, int n,
                                                               const float *__restrict__ bias, int act, float alpha, float *__restrict__ y, int64_t ldy) {
    const int n4 = n >> 2;
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= rows * n4) return;
    const int64_t r = t / n4;
    const int o = (int)(t - r * n4) * 4;
    const float *src = slabs + r * n + o;
    f4 s = {0.f, 0.f, 0.f, 0.f};
    int z = 0;
    for (; z + 8 <= nsplit; z += 8) {
        f4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = *(const f4 *)(src + (int64_t)(z + u) * stride);
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; z < nsplit; ++z) s += *(const f4 *)(src + (int64_t)z * stride);
    if (bias) s += *(const f4 *)(bias + o);
#pragma unroll
    for (int k = 0; k < 4; ++k) s[k] = act == 1 ? (s[k] > 0.f ? s[k] : alpha * (expf(s[k]) - 1.f)) : (act == 2 ? fmaxf(s[k], 0.f) : s[k]);
    *(f4 *)(y + r * ldy + o) = s;
}

extern thread_local char qa_err_buf[512];
#define g_gerr qa_err_buf

static inline bool aligned16(const void *p) { return ((uintptr_t)p & 15) == 0; }
// operand element offsets are 32-bit inside the kernels (gemm_load: 64-bit row pointers cost 32 registers): the largest offset an
// operand of `count` rows with leading dimension ld can produce must stay below 2^32 (it used to wrap silently)
static inline bool fits_u32(int64_t count, int64_t ld) { return count > 0 && ld >= 0 && count * ld < ((int64_t)1 << 32); }

// tile configuration: 0 = 128 x 128, 1 = 128 x 64, 2 = 64 x 64, 3 = 64 x 128, 4 = 32 x 128 (A side x B side; 4: window GEMMs with <= 32 outputs);
// 10.. = the LDS-DMA kernel: 10 = 128 x 192, 11 = 128 x 128, 12 = 128 x 64, 13 = 64 x 64, 14 = 64 x 192
static void tile_dims(int cfg, int *ba, int *bb) {
    if (cfg >= 10) {
        static const int dims[5][2] = {{128, 192}, {128, 128}, {128, 64}, {64, 64}, {64, 192}};
        const int c = cfg - 10 > 4 ? 1 : cfg - 10;
        *ba = dims[c][0]; *bb = dims[c][1];
        return;
    }
    *ba = (cfg == 0 || cfg == 1) ? 128 : (cfg == 4 ? 32 : 64);
    *bb = (cfg == 0 || cfg == 3 || cfg == 4) ? 128 : 64;
}
// may this product run on qa_gemm_dma_kernel?  (16-byte operands; for k-contiguous operands the reduction in whole 16-wide tiles per split)
// An index-contiguous (MC) operand is fetched in 16-byte chunks ALONG its index and the last chunk is clamped to count - 4: its index count
// has to be a multiple of 4 (this used to be implied by how each caller computes a_vec / b_vec; now it is checked here).
static bool dma_ok(const GemmArgs &g, bool a_mc, bool b_mc) {
    return g.a_vec == 4 && g.b_vec == 4 && g.a_count >= 4 && g.b_count >= 4 && g.kred % GEMM_BK == 0 && g.k_per_split % GEMM_BK == 0 && g.kred >= GEMM_BK &&
           (!a_mc || g.a_count % 4 == 0) && (!b_mc || g.b_count % 4 == 0);
}
template <bool A_MC, bool B_MC, int EPI, bool ONES>
static void gemm_launch_dma(int cfg, GemmArgs &g, hipStream_t st) {
    int ba, bb; tile_dims(cfg, &ba, &bb);
    g.na = (g.a_count + ba - 1) / ba;
    g.nb = (g.b_count + bb - 1) / bb;
    const dim3 grid((unsigned)(g.na * g.nb * g.nsplit));
    switch (cfg) {
    case 10: hipLaunchKernelGGL((qa_gemm_dma_kernel<128, 192, A_MC, B_MC, EPI, ONES>), grid, dim3(256), 0, st, g); break;
    case 12: hipLaunchKernelGGL((qa_gemm_dma_kernel<128, 64, A_MC, B_MC, EPI, ONES>), grid, dim3(256), 0, st, g); break;
    case 13: hipLaunchKernelGGL((qa_gemm_dma_kernel<64, 64, A_MC, B_MC, EPI, ONES>), grid, dim3(256), 0, st, g); break;
    case 14: hipLaunchKernelGGL((qa_gemm_dma_kernel<64, 192, A_MC, B_MC, EPI, ONES>), grid, dim3(256), 0, st, g); break;
    default: hipLaunchKernelGGL((qa_gemm_dma_kernel<128, 128, A_MC, B_MC, EPI, ONES>), grid, dim3(256), 0, st, g); break;
    }
}

template <bool A_MC, bool B_MC, int AV, int BV, int EPI, bool ONES>
static void gemm_launch_v(int cfg, GemmArgs &g, hipStream_t st) {
    if (cfg < 0 || cfg > 3) cfg = 2;          // 4 (32 x 128) exists for the window GEMMs only (gemm_launch_conv); a forced 4 here used to size the grid for 32 x 128 and launch 64 x 128
    int ba, bb; tile_dims(cfg, &ba, &bb);
    g.na = (g.a_count + ba - 1) / ba;
    g.nb = (g.b_count + bb - 1) / bb;
    const dim3 grid((unsigned)(g.na * g.nb * g.nsplit));
    switch (cfg) {
    case 0: hipLaunchKernelGGL((qa_gemm_kernel<128, 128, A_MC, B_MC, AV, BV, EPI, ONES>), grid, dim3(256), 0, st, g); break;
    case 1: hipLaunchKernelGGL((qa_gemm_kernel<128, 64, A_MC, B_MC, AV, BV, EPI, ONES>), grid, dim3(256), 0, st, g); break;
    case 2: hipLaunchKernelGGL((qa_gemm_kernel<64, 64, A_MC, B_MC, AV, BV, EPI, ONES>), grid, dim3(256), 0, st, g); break;
    default: hipLaunchKernelGGL((qa_gemm_kernel<64, 128, A_MC, B_MC, AV, BV, EPI, ONES>), grid, dim3(256), 0, st, g); break;
    }
}
template <bool A_MC, bool B_MC, int EPI, bool ONES, int CONV>
static void gemm_launch_conv(int cfg, GemmArgs &g, hipStream_t st) {
    int ba, bb; tile_dims(cfg, &ba, &bb);
    g.na = (g.a_count + ba - 1) / ba;
    g.nb = (g.b_count + bb - 1) / bb;
    const dim3 grid((unsigned)(g.na * g.nb * g.nsplit));
    if (cfg == 2) hipLaunchKernelGGL((qa_gemm_kernel<64, 64, A_MC, B_MC, 4, 4, EPI, ONES, 0, CONV>), grid, dim3(256), 0, st, g);
    else if (cfg == 3) hipLaunchKernelGGL((qa_gemm_kernel<64, 128, A_MC, B_MC, 4, 4, EPI, ONES, 0, CONV>), grid, dim3(256), 0, st, g);
    else if constexpr (!A_MC) hipLaunchKernelGGL((qa_gemm_kernel<32, 128, A_MC, B_MC, 4, 4, EPI, ONES, 0, CONV>), grid, dim3(256), 0, st, g);
}
template <bool A_MC, bool B_MC, int EPI, bool ONES>
static void gemm_launch(int cfg, GemmArgs &g, hipStream_t st) {
    if (cfg >= 10 && dma_ok(g, A_MC, B_MC)) { gemm_launch_dma<A_MC, B_MC, EPI, ONES>(cfg, g, st); return; }
    if (cfg >= 10) cfg = cfg == 10 || cfg == 11 ? 0 : (cfg == 12 ? 1 : (cfg == 14 ? 3 : 2));      // not DMA-able: the register-staged kernel's nearest tile
    if (g.a_vec == 4 && g.b_vec == 4) gemm_launch_v<A_MC, B_MC, 4, 4, EPI, ONES>(cfg, g, st);
    else if (g.a_vec == 4) gemm_launch_v<A_MC, B_MC, 4, 1, EPI, ONES>(cfg, g, st);
    else if (g.b_vec == 4) gemm_launch_v<A_MC, B_MC, 1, 4, EPI, ONES>(cfg, g, st);
    else gemm_launch_v<A_MC, B_MC, 1, 1, EPI, ONES>(cfg, g, st);
}

// the largest tile that still gives the chip >= 2 workgroups per CU (256 CUs); small problems take the smallest tile
static int pick_cfg(int64_t a_count, int64_t b_count, int64_t nsplit) {
    static const int order[3] = {0, 1, 2};
    for (int c = 0; c < 3; ++c) {
        int ba, bb; tile_dims(order[c], &ba, &bb);
        if (a_count <= 64 && ba > 64) continue;
        const int64_t wgs = ((a_count + ba - 1) / ba) * ((b_count + bb - 1) / bb) * nsplit;
        if (wgs >= 512) return order[c];
    }
    return 2;
}

static int g_force_cfg = -1;      // tools/gemm_bench.py: time one tile configuration

// QA_GEMM_DMA: 1 = products that qualify (dma_ok) take the LDS-DMA kernel with the tile chosen below, 0 = register-staged kernel only
static int dma_mode() {
    static const int m = [] { const char *e = getenv("QA_GEMM_DMA"); return e ? atoi(e) : QA_GEMM_DMA_DEFAULT; }();      // initialised once, thread-safe (C++11 magic static)
    return m;
}
// LDS-DMA tile for an (a_count x b_count) output computed in nsplit slices: the largest tile that still gives every CU two workgroups
// (128 x 192 is resident twice per CU, the smaller ones three or four times); narrow outputs take the 64-wide tiles
static int pick_cfg_dma(int64_t a_count, int64_t b_count, int64_t nsplit) {
    static const int order[5] = {10, 11, 14, 12, 13};
    for (int c = 0; c < 5; ++c) {
        int ba, bb; tile_dims(order[c], &ba, &bb);
        if (a_count <= 64 && ba > 64) continue;
        if (b_count <= 64 && bb > 64) continue;
        const int64_t wgs = ((a_count + ba - 1) / ba) * ((b_count + bb - 1) / bb) * nsplit;
        if (wgs >= 512) return order[c];
    }
    return 13;
}

extern "C" {

void qa_gemm_force_config(int32_t cfg) { g_force_cfg = cfg; }

int qa_linear_forward(const float *x, int64_t ldx, const float *weight, int64_t ldw, const float *bias, float *y, int64_t ldy, int64_t rows,
                      int32_t in_features, int32_t out_features, int32_t act, float alpha, void *stream) {
    if (!x || !weight || !y || rows <= 0 || rows > INT32_MAX || in_features <= 0 || out_features <= 0 || ldx < in_features || ldw < in_features ||
        ldy < out_features || act < 0 || act > 2) { snprintf(g_gerr, sizeof(g_gerr), "qa_linear_forward: bad argument"); return QA_E_ARG; }
    if (!fits_u32(rows, ldx) || !fits_u32(out_features, ldw)) { snprintf(g_gerr, sizeof(g_gerr), "qa_linear_forward: rows * ldx (or out_features * ldw) >= 2^32 elements"); return QA_E_ARG; }
    GemmArgs g = {};
    g.A = weight; g.lda = ldw; g.a_count = out_features;
    g.B = x; g.ldb = ldx; g.b_count = (int)rows;
    g.kred = in_features; g.k_per_split = (in_features + GEMM_BK - 1) / GEMM_BK * GEMM_BK; g.nsplit = 1;
    g.out = y; g.ldo = ldy; g.bias = bias; g.act = act; g.alpha = alpha;
    g.a_vec = (aligned16(weight) && ldw % 4 == 0 && in_features % 4 == 0) ? 4 : 1;
    g.b_vec = (aligned16(x) && ldx % 4 == 0 && in_features % 4 == 0) ? 4 : 1;
    g.o_vec = (aligned16(y) && ldy % 4 == 0 && (!bias || aligned16(bias))) ? 4 : 1;
    gemm_launch<false, false, 1, false>(g_force_cfg >= 0 ? g_force_cfg : ((dma_mode() && dma_ok(g, false, false)) ? pick_cfg_dma(out_features, rows, 1) : pick_cfg(out_features, rows, 1)),
                                         g, (hipStream_t)stream);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { snprintf(g_gerr, sizeof(g_gerr), "qa_linear_forward: %s", hipGetErrorString(e)); return QA_E_DEVICE; }
    return QA_OK;
}

int qa_linear_backward_input(const float *grad_out, int64_t ldg, const float *weight, int64_t ldw, const float *y_prev, int64_t ldyp,
                             float *grad_in, int64_t ldgi, int64_t rows, int32_t in_features, int32_t out_features, int32_t act_prev, float alpha,
                             void *stream) {
    if (!grad_out || !weight || !grad_in || rows <= 0 || rows > INT32_MAX || in_features <= 0 || out_features <= 0 || ldg < out_features ||
        ldw < in_features || ldgi < in_features || act_prev < 0 || act_prev > 2 || (act_prev != 0 && (!y_prev || ldyp < in_features))) {
        snprintf(g_gerr, sizeof(g_gerr), "qa_linear_backward_input: bad argument"); return QA_E_ARG; }
    if (!fits_u32(rows, ldg) || !fits_u32(out_features, ldw)) { snprintf(g_gerr, sizeof(g_gerr), "qa_linear_backward_input: rows * ldg (or out_features * ldw) >= 2^32 elements"); return QA_E_ARG; }
    GemmArgs g = {};
    g.A = weight; g.lda = ldw; g.a_count = in_features;                 // MC: element (in feature a, out feature k) = W[k * ldw + a]
    g.B = grad_out; g.ldb = ldg; g.b_count = (int)rows;
    g.kred = out_features; g.k_per_split = (out_features + GEMM_BK - 1) / GEMM_BK * GEMM_BK; g.nsplit = 1;
    g.out = grad_in; g.ldo = ldgi; g.yprev = act_prev ? y_prev : nullptr; g.ldy = ldyp; g.act = act_prev; g.alpha = alpha;
    g.a_vec = (aligned16(weight) && ldw % 4 == 0 && in_features % 4 == 0) ? 4 : 1;
    g.b_vec = (aligned16(grad_out) && ldg % 4 == 0 && out_features % 4 == 0) ? 4 : 1;
    g.o_vec = (aligned16(grad_in) && ldgi % 4 == 0 && (!g.yprev || (aligned16(y_prev) && ldyp % 4 == 0))) ? 4 : 1;
    gemm_launch<true, false, 2, false>(g_force_cfg >= 0 ? g_force_cfg : ((dma_mode() && dma_ok(g, true, false)) ? pick_cfg_dma(in_features, rows, 1) : pick_cfg(in_features, rows, 1)),
                                        g, (hipStream_t)stream);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { snprintf(g_gerr, sizeof(g_gerr), "qa_linear_backward_input: %s", hipGetErrorString(e)); return QA_E_DEVICE; }
    return QA_OK;
}

// split of the sample dimension: enough slabs for ~3 workgroups per CU, slabs of whole k-tiles and at least 256 rows
static void wgrad_plan(int64_t rows, int32_t in_features, int32_t out_features, int *cfg, int *nsplit, int *k_per_split) {
    // measured on MI355X at 24,576 rows (tools/own_gemm_bench.py, profiles/r3_own_gemm_bench.json): the 128 x 128 tile only pays for the
    // widest layers (671 / 800 x 512), 128 x 64 for the 512 x 256 and 256 x 128 ones, 64 x 64 for everything narrower
    int c = (in_features > 256 && out_features > 256) ? 0 : ((in_features >= 256 && out_features >= 128) ? 1 : 2);
    if (dma_mode() && rows % GEMM_BK == 0 && in_features % 4 == 0 && out_features % 4 == 0)        // (alignment of the operands is checked at launch: dma_ok)
        c = (in_features > 256 && out_features > 256) ? 11 : ((in_features >= 128 && out_features >= 64) ? 12 : 13);
    if (g_force_cfg >= 0) c = g_force_cfg;
    int ba, bb; tile_dims(c, &ba, &bb);
    const int64_t tiles = (int64_t)((in_features + ba - 1) / ba) * ((out_features + bb - 1) / bb);
    // resident workgroups: 2 per CU for the 128 x 128 tile (190-200 registers), 4 otherwise; LDS-DMA tiles: 2 (128 x 192: 60 KB of LDS), 3 (48 KB), 4
    const int64_t target = (c == 0 || c == 10) ? 512 : ((c == 11 || c == 14) ? 768 : 1024);
    int64_t s = target / tiles;                         // never a few workgroups more than one resident round: they would run alone
    const int64_t smax = (rows + 255) / 256;
    if (s > smax) s = smax;
    if (s > 48) s = 48;          // the reduction reads every slab: beyond this it costs more than the extra workgroups give
    if (s < 1) s = 1;
    int64_t kps = ((rows + s - 1) / s + GEMM_BK - 1) / GEMM_BK * GEMM_BK;
    s = (rows + kps - 1) / kps;
    *cfg = c; *nsplit = (int)s; *k_per_split = (int)kps;
}
static int64_t pad4(int64_t n) { return (n + 3) & ~(int64_t)3; }

int64_t qa_linear_backward_weight_scratch_bytes(int64_t rows, int32_t in_features, int32_t out_features) {
    if (rows <= 0 || in_features <= 0 || out_features <= 0) return 0;
    int cfg, s, kps; wgrad_plan(rows, in_features, out_features, &cfg, &s, &kps);
    int ba, bb; tile_dims(cfg, &ba, &bb);
    const int64_t na = (in_features + ba - 1) / ba;
    return (int64_t)s * (pad4((int64_t)in_features * out_features) + na * pad4(out_features)) * 4;
}

int qa_linear_backward_weight_layout(int64_t rows, int32_t in_features, int32_t out_features, int64_t layout[5]) {
    if (!layout || rows <= 0 || in_features <= 0 || out_features <= 0) { snprintf(g_gerr, sizeof(g_gerr), "qa_linear_backward_weight_layout: bad argument"); return QA_E_ARG; }
    int cfg, s, kps; wgrad_plan(rows, in_features, out_features, &cfg, &s, &kps);
    int ba, bb; tile_dims(cfg, &ba, &bb);
    const int64_t na = (in_features + ba - 1) / ba;
    layout[0] = s; layout[1] = pad4((int64_t)in_features * out_features); layout[2] = s * na; layout[3] = pad4(out_features); layout[4] = (int64_t)s * layout[1];
    return QA_OK;
}

int qa_linear_backward_weight(const float *grad_out, int64_t ldg, const float *x, int64_t ldx, float *grad_weight, float *grad_bias, int64_t rows,
                              int32_t in_features, int32_t out_features, void *scratch, int64_t scratch_bytes, void *stream) {
    const bool parts_only = !grad_weight && !grad_bias;          // ABI 17: the slabs stay in `scratch` (qa_linear_backward_weight_layout)
    if (!grad_out || !x || (!parts_only && (!grad_weight || !grad_bias)) || !scratch || rows <= 0 || rows > INT32_MAX || in_features <= 0 || out_features <= 0 ||
        ldg < out_features || ldx < in_features) { snprintf(g_gerr, sizeof(g_gerr), "qa_linear_backward_weight: bad argument"); return QA_E_ARG; }
    if (!fits_u32(rows, ldg) || !fits_u32(rows, ldx)) { snprintf(g_gerr, sizeof(g_gerr), "qa_linear_backward_weight: rows * ldg (or rows * ldx) >= 2^32 elements"); return QA_E_ARG; }
    if (scratch_bytes < qa_linear_backward_weight_scratch_bytes(rows, in_features, out_features) || !aligned16(scratch)) {
        snprintf(g_gerr, sizeof(g_gerr), "qa_linear_backward_weight: scratch too small or not 16-byte aligned"); return QA_E_ARG; }
    int cfg, s, kps; wgrad_plan(rows, in_features, out_features, &cfg, &s, &kps);
    int ba, bb; tile_dims(cfg, &ba, &bb);
    const int64_t n_w = (int64_t)in_features * out_features, n_w_pad = pad4(n_w), n_b_pad = pad4(out_features), na = (in_features + ba - 1) / ba;
    float *bslabs = (float *)scratch + (int64_t)s * n_w_pad;
    GemmArgs g = {};
    g.A = x; g.lda = ldx; g.a_count = in_features;                       // MC: element (in feature a, sample k) = x[k * ldx + a]
    g.B = grad_out; g.ldb = ldg; g.b_count = out_features;               // MC: element (out feature b, sample k) = g[k * ldg + b]
    g.kred = (int)rows; g.k_per_split = kps; g.nsplit = s;
    g.out = (float *)scratch; g.ldo = in_features; g.out_split_stride = n_w_pad;
    g.ones_out = bslabs; g.ones_split_stride = n_b_pad;
    g.a_vec = (aligned16(x) && ldx % 4 == 0 && in_features % 4 == 0) ? 4 : 1;
    g.b_vec = (aligned16(grad_out) && ldg % 4 == 0 && out_features % 4 == 0) ? 4 : 1;
    g.o_vec = (in_features % 4 == 0) ? 4 : 1;
    hipStream_t st = (hipStream_t)stream;
    gemm_launch<true, true, 0, true>(cfg, g, st);
    const int64_t nthreads = (n_w + 3) / 4 + (out_features + 3) / 4;
    if (!parts_only)
        hipLaunchKernelGGL(qa_slab_reduce_kernel, dim3((unsigned)((nthreads + 255) / 256)), dim3(256), 0, st, (const float *)scratch, n_w_pad, s, n_w,
                           (const float *)bslabs, n_b_pad, (int)(s * na), (int64_t)out_features, grad_weight, grad_bias, aligned16(grad_weight) ? 4 : 1);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { snprintf(g_gerr, sizeof(g_gerr), "qa_linear_backward_weight: %s", hipGetErrorString(e)); return QA_E_DEVICE; }
    return QA_OK;
}

// the batch's split of the sample dimension: 64 x 64 tiles for every product, ~256 workgroups each (a dozen of them fill the chip together).
// (r6, measured: slabs of ~512 rows for every product instead -- the launch takes what its longest workgroup takes -- moved the 512 x 671 product's
// group 68 -> 59 us and the small products' groups 14 -> 24 us each: 371 -> 379 us per step, config 3 47.5 -> 48.0 ms; this plan stays.  The three
// launches of a 3,072-row PPO step are 4.5 GFLOP in ~97 us.)
static void wgrad_batch_plan(int64_t rows, int32_t in_features, int32_t out_features, int *nsplit, int *k_per_split) {
    const int64_t tiles = (int64_t)((in_features + 63) / 64) * ((out_features + 63) / 64);
    int64_t s = 256 / tiles;
    const int64_t smax = (rows + 255) / 256;
    if (s > smax) s = smax;
    if (s > 16) s = 16;
    if (s < 1) s = 1;
    int64_t kps = ((rows + s - 1) / s + GEMM_BK - 1) / GEMM_BK * GEMM_BK;
    s = (rows + kps - 1) / kps;
    *nsplit = (int)s; *k_per_split = (int)kps;
}
int qa_linear_backward_weight_batch_layout(int64_t rows, int32_t in_features, int32_t out_features, int64_t layout[5]) {
    if (!layout || rows <= 0 || in_features <= 0 || out_features <= 0) { snprintf(g_gerr, sizeof(g_gerr), "qa_linear_backward_weight_batch_layout: bad argument"); return QA_E_ARG; }
    int s, kps; wgrad_batch_plan(rows, in_features, out_features, &s, &kps);
    const int64_t na = (in_features + 63) / 64;
    layout[0] = s; layout[1] = pad4((int64_t)in_features * out_features); layout[2] = s * na; layout[3] = pad4(out_features); layout[4] = (int64_t)s * layout[1];
    return QA_OK;
}
int64_t qa_linear_backward_weight_batch_scratch_bytes(int64_t rows, int32_t in_features, int32_t out_features) {
    int64_t lay[5];
    if (qa_linear_backward_weight_batch_layout(rows, in_features, out_features, lay) != QA_OK) return 0;
    return (lay[4] + lay[2] * lay[3]) * 4;
}

int qa_linear_backward_weight_batch(const qa_wgrad_desc *descs, int32_t count, void *stream) {
    if (!descs || count <= 0 || count > 32) { snprintf(g_gerr, sizeof(g_gerr), "qa_linear_backward_weight_batch: 1..32 products expected"); return QA_E_ARG; }
    hipStream_t st = (hipStream_t)stream;
    static GemmGroup groups[4];          // by (a_vec == 4, b_vec == 4); filled and launched under the lock below
    static std::mutex mu;
    std::lock_guard<std::mutex> lock(mu);
    for (int v = 0; v < 4; ++v) { groups[v].n = 0; groups[v].start[0] = 0; }
    auto flush = [&](int v) {
        GemmGroup &G = groups[v];
        if (G.n == 0) return;
        const dim3 grid((unsigned)G.start[G.n]);
        switch (v) {
        case 3: hipLaunchKernelGGL((qa_wgrad_group_kernel<4, 4>), grid, dim3(256), 0, st, G); break;
        case 2: hipLaunchKernelGGL((qa_wgrad_group_kernel<4, 1>), grid, dim3(256), 0, st, G); break;
        case 1: hipLaunchKernelGGL((qa_wgrad_group_kernel<1, 4>), grid, dim3(256), 0, st, G); break;
        default: hipLaunchKernelGGL((qa_wgrad_group_kernel<1, 1>), grid, dim3(256), 0, st, G); break;
        }
        G.n = 0; G.start[0] = 0;
    };
    float *red_dst[64]; const float *red_src[64]; int64_t red_stride[64]; int32_t red_parts[64], red_numel[64];
    int nred = 0;
    for (int i = 0; i < count; ++i) {
        const qa_wgrad_desc &d = descs[i];
        const bool parts_only = !d.grad_weight && !d.grad_bias;
        if (!d.grad_out || !d.x || (!parts_only && (!d.grad_weight || !d.grad_bias)) || !d.scratch || d.rows <= 0 || d.rows > INT32_MAX || d.in_features <= 0 ||
            d.out_features <= 0 || d.ldg < d.out_features || d.ldx < d.in_features || !fits_u32(d.rows, d.ldg) || !fits_u32(d.rows, d.ldx) || !aligned16(d.scratch) ||
            d.scratch_bytes < qa_linear_backward_weight_batch_scratch_bytes(d.rows, d.in_features, d.out_features)) {
            snprintf(g_gerr, sizeof(g_gerr), "qa_linear_backward_weight_batch: product %d: bad argument", i); return QA_E_ARG; }
        int s, kps; wgrad_batch_plan(d.rows, d.in_features, d.out_features, &s, &kps);
        const int64_t n_w_pad = pad4((int64_t)d.in_features * d.out_features), n_b_pad = pad4(d.out_features), na = (d.in_features + 63) / 64;
        float *bslabs = (float *)d.scratch + (int64_t)s * n_w_pad;
        GemmArgs g = {};
        g.A = d.x; g.lda = d.ldx; g.a_count = d.in_features;
        g.B = d.grad_out; g.ldb = d.ldg; g.b_count = d.out_features;
        g.kred = (int)d.rows; g.k_per_split = kps; g.nsplit = s;
        g.out = (float *)d.scratch; g.ldo = d.in_features; g.out_split_stride = n_w_pad;
        g.ones_out = bslabs; g.ones_split_stride = n_b_pad;
        // x rows with readable padding up to the next multiple of 4 columns (a column slice of wider rows, e.g. the 671 observation columns of 672-wide
        // rows): read in 16-byte pieces all the same; what the extra columns multiply into lies beyond in_features and is never stored
        const bool x_pad = d.in_features % 4 != 0 && d.ldx >= pad4(d.in_features);
        g.a_vec = (aligned16(d.x) && d.ldx % 4 == 0 && (d.in_features % 4 == 0 || x_pad)) ? 4 : 1;
        g.a_ld_count = (g.a_vec == 4 && x_pad) ? (int)pad4(d.in_features) : 0;
        g.b_vec = (aligned16(d.grad_out) && d.ldg % 4 == 0 && d.out_features % 4 == 0) ? 4 : 1;
        g.o_vec = (d.in_features % 4 == 0) ? 4 : 1;
        g.na = (int)na; g.nb = (d.out_features + 63) / 64;
        const int v = (g.a_vec == 4 ? 2 : 0) + (g.b_vec == 4 ? 1 : 0);
        GemmGroup &G = groups[v];
        if (G.n == GEMM_GROUP_MAX) flush(v);
        G.g[G.n] = g; G.start[G.n + 1] = G.start[G.n] + g.na * g.nb * g.nsplit; ++G.n;
        if (!parts_only) {
            if (nred + 2 > 64) { snprintf(g_gerr, sizeof(g_gerr), "qa_linear_backward_weight_batch: too many finished products in one batch"); return QA_E_ARG; }
            red_dst[nred] = d.grad_weight; red_src[nred] = (const float *)d.scratch; red_stride[nred] = n_w_pad; red_parts[nred] = s; red_numel[nred] = d.in_features * d.out_features; ++nred;
            red_dst[nred] = d.grad_bias; red_src[nred] = bslabs; red_stride[nred] = n_b_pad; red_parts[nred] = (int32_t)(s * na); red_numel[nred] = d.out_features; ++nred;
        }
    }
    for (int v = 0; v < 4; ++v) flush(v);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { snprintf(g_gerr, sizeof(g_gerr), "qa_linear_backward_weight_batch: %s", hipGetErrorString(e)); return QA_E_DEVICE; }
    if (nred) return qa_grad_reduce(red_dst, red_src, red_stride, red_parts, red_numel, nred, stream);       // ONE launch finishes every product that asked for it
    return QA_OK;
}

int qa_slab_sum(const float *slabs, int64_t slab_stride, int32_t num_slabs, int64_t n, float *out, void *stream) {
    if (!slabs || !out || num_slabs <= 0 || n <= 0 || slab_stride < n) { snprintf(g_gerr, sizeof(g_gerr), "qa_slab_sum: bad argument"); return QA_E_ARG; }
    const bool v4 = aligned16(slabs) && aligned16(out) && slab_stride % 4 == 0;
    const int64_t nthreads = (n + 3) / 4;
    hipLaunchKernelGGL(qa_slab_reduce_kernel, dim3((unsigned)((nthreads + 255) / 256)), dim3(256), 0, (hipStream_t)stream, slabs, slab_stride, (int)num_slabs, n,
                       (const float *)nullptr, (int64_t)0, 0, (int64_t)0, out, (float *)nullptr, v4 ? 4 : 1);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { snprintf(g_gerr, sizeof(g_gerr), "qa_slab_sum: %s", hipGetErrorString(e)); return QA_E_DEVICE; }
    return QA_OK;
}

}  // extern "C"

// ---- convolutions over channels-last images (the depth student's second convolution) ----
static int ilog2(int v) { int s = 0; while ((1 << s) < v) ++s; return s; }
static bool conv_geom(ConvGeom *c, int64_t n_img, int ih, int iw, int cin, int kh, int kw) {
    const int oh = ih - kh + 1, ow = iw - kw + 1;
    if (n_img <= 0 || cin < 16 || (cin & (cin - 1)) || kh < 1 || kw < 2 || oh < 1 || ow < 2) return false;
    const int64_t opix = (int64_t)oh * ow, ipix = (int64_t)ih * iw;
    if (opix <= 256 || n_img * opix > INT32_MAX || n_img * ipix * cin >= ((int64_t)1 << 32)) return false;
    c->opix = (int)opix; c->ow = ow; c->iw = iw; c->ipix = (int)ipix; c->cin = cin; c->cs = ilog2(cin); c->kw = kw;
    c->m_opix = (uint32_t)((((uint64_t)1 << 40) + opix - 1) / opix);
    c->m_ow = (uint32_t)((((uint64_t)1 << 32) + ow - 1) / ow);
    c->m_kw = (uint32_t)((((uint64_t)1 << 32) + kw - 1) / kw);
    return true;
}
// mode 0: y = act(conv(x, w) + bias);  mode 1: y = conv(x, w) * act'(deriv_of)
static int conv_forward_like(const char *who, int mode, const float *x, const float *w, const float *bias, const float *deriv_of, float *y, int64_t n_img,
                             int ih, int iw, int cin, int kh, int kw, int cout, int act, float alpha, void *stream) {
    GemmArgs g = {};
    if (!x || !w || !y || cout <= 0 || cout % 4 || act < 0 || act > 2 || !conv_geom(&g.cg, n_img, ih, iw, cin, kh, kw) || !aligned16(x) || !aligned16(w) ||
        !aligned16(y) || (bias && !aligned16(bias)) || (deriv_of && !aligned16(deriv_of)) || (mode == 1 && act != 0 && !deriv_of)) {
        snprintf(g_gerr, sizeof(g_gerr), "%s: bad argument (channels-last fp32, 16-byte aligned, in channels a power of two >= 16, out channels a multiple of 4, "
                 "more than 256 output pixels per image)", who); return QA_E_ARG; }
    const int kred = kh * kw * cin;
    g.A = w; g.lda = kred; g.a_count = cout;
    g.B = x; g.ldb = 0; g.b_count = (int)(n_img * g.cg.opix);
    g.kred = kred; g.k_per_split = kred; g.nsplit = 1;
    g.out = y; g.ldo = cout; g.bias = bias; g.yprev = (mode == 1 && act) ? deriv_of : nullptr; g.ldy = cout; g.act = act; g.alpha = alpha;
    g.a_vec = g.b_vec = g.o_vec = 4;
    int cfg = cout <= 32 ? 4 : (((int64_t)((cout + 63) / 64) * ((g.b_count + 127) / 128) >= 512) ? 3 : 2);
    if (g_force_cfg >= 2 && g_force_cfg <= 4) cfg = g_force_cfg;
    if (mode == 0) gemm_launch_conv<false, false, 1, false, 1>(cfg, g, (hipStream_t)stream);
    else gemm_launch_conv<false, false, 2, false, 1>(cfg, g, (hipStream_t)stream);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { snprintf(g_gerr, sizeof(g_gerr), "%s: %s", who, hipGetErrorString(e)); return QA_E_DEVICE; }
    return QA_OK;
}
static void conv_wgrad_plan(int64_t pixels, int kred, int cout, int *nsplit, int *k_per_split) {
    const int64_t tiles = (int64_t)((kred + 63) / 64) * ((cout + 63) / 64);
    int64_t s = 1024 / tiles;
    const int64_t smax = (pixels + 1023) / 1024;
    if (s > smax) s = smax;
    if (s > 192) s = 192;
    if (s < 1) s = 1;
    int64_t kps = ((pixels + s - 1) / s + GEMM_BK - 1) / GEMM_BK * GEMM_BK;
    s = (pixels + kps - 1) / kps;
    *nsplit = (int)s; *k_per_split = (int)kps;
}

extern "C" {

int qa_conv_nhwc_forward(const float *x, const float *weight, const float *bias, float *y, int64_t n_img, int32_t ih, int32_t iw, int32_t cin,
                         int32_t kh, int32_t kw, int32_t cout, int32_t act, float alpha, void *stream) {
    return conv_forward_like("qa_conv_nhwc_forward", 0, x, weight, bias, nullptr, y, n_img, ih, iw, cin, kh, kw, cout, act, alpha, stream);
}

int qa_conv_nhwc_backward_input(const float *grad_padded, const float *weight_flipped, const float *x_act, float *grad_in, int64_t n_img, int32_t ihp,
                                int32_t iwp, int32_t cout, int32_t kh, int32_t kw, int32_t cin, int32_t act_prev, float alpha, void *stream) {
    return conv_forward_like("qa_conv_nhwc_backward_input", 1, grad_padded, weight_flipped, nullptr, x_act, grad_in, n_img, ihp, iwp, cout, kh, kw, cin,
                             act_prev, alpha, stream);
}

int64_t qa_conv_nhwc_backward_weight_scratch_bytes(int64_t n_img, int32_t ih, int32_t iw, int32_t cin, int32_t kh, int32_t kw, int32_t cout) {
    ConvGeom c;
    if (cout <= 0 || cout % 4 || !conv_geom(&c, n_img, ih, iw, cin, kh, kw)) return 0;
    const int kred = kh * kw * cin;
    int s, kps; conv_wgrad_plan(n_img * c.opix, kred, cout, &s, &kps);
    const int64_t na = (kred + 63) / 64;
    return (int64_t)s * (pad4((int64_t)kred * cout) + na * pad4(cout)) * 4;
}

int qa_conv_nhwc_backward_weight(const float *x, const float *grad_out, float *grad_weight, float *grad_bias, int64_t n_img, int32_t ih, int32_t iw,
                                 int32_t cin, int32_t kh, int32_t kw, int32_t cout, void *scratch, int64_t scratch_bytes, void *stream) {
    GemmArgs g = {};
    if (!x || !grad_out || !grad_weight || !grad_bias || !scratch || cout <= 0 || cout % 4 || !conv_geom(&g.cg, n_img, ih, iw, cin, kh, kw) || !aligned16(x) ||
        !aligned16(grad_out) || !aligned16(grad_weight)) { snprintf(g_gerr, sizeof(g_gerr), "qa_conv_nhwc_backward_weight: bad argument"); return QA_E_ARG; }
    if (scratch_bytes < qa_conv_nhwc_backward_weight_scratch_bytes(n_img, ih, iw, cin, kh, kw, cout) || !aligned16(scratch)) {
        snprintf(g_gerr, sizeof(g_gerr), "qa_conv_nhwc_backward_weight: scratch too small or not 16-byte aligned"); return QA_E_ARG; }
    const int kred = kh * kw * cin;
    const int64_t pixels = n_img * g.cg.opix;
    int s, kps; conv_wgrad_plan(pixels, kred, cout, &s, &kps);
    const int64_t n_w = (int64_t)kred * cout, n_w_pad = pad4(n_w), n_b_pad = pad4(cout), na = (kred + 63) / 64;
    float *bslabs = (float *)scratch + (int64_t)s * n_w_pad;
    g.A = x; g.lda = 0; g.a_count = kred;                                  // window matrix, element ((tap, channel) a, pixel k)
    g.B = grad_out; g.ldb = cout; g.b_count = cout;                        // MC: element (out channel b, pixel k) = g[k * cout + b]
    g.kred = (int)pixels; g.k_per_split = kps; g.nsplit = s;
    g.out = (float *)scratch; g.ldo = kred; g.out_split_stride = n_w_pad;
    g.ones_out = bslabs; g.ones_split_stride = n_b_pad;
    g.a_vec = g.b_vec = g.o_vec = 4;
    hipStream_t st = (hipStream_t)stream;
    gemm_launch_conv<true, true, 0, true, 2>(2, g, st);
    const int64_t nthreads = (n_w + 3) / 4 + (cout + 3) / 4;
    hipLaunchKernelGGL(qa_slab_reduce_kernel, dim3((unsigned)((nthreads + 255) / 256)), dim3(256), 0, st, (const float *)scratch, n_w_pad, s, n_w,
                       (const float *)bslabs, n_b_pad, (int)(s * na), (int64_t)cout, grad_weight, grad_bias, 4);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { snprintf(g_gerr, sizeof(g_gerr), "qa_conv_nhwc_backward_weight: %s", hipGetErrorString(e)); return QA_E_DEVICE; }
    return QA_OK;
}

}  // extern "C"

// split of the reduction (input-feature) dimension of a forward layer with few output tiles
static void fwd_split_plan(int64_t rows, int32_t in_features, int32_t out_features, int *nsplit, int *k_per_split) {
    const int64_t tiles = ((rows + 63) / 64) * ((out_features + 63) / 64);
    int64_t s = (1024 + tiles - 1) / tiles;
    const int64_t smax = in_features / 256;
    if (s > smax) s = smax;
    if (s > 64) s = 64;
    if (s < 1) s = 1;
    int64_t kps = ((in_features + s - 1) / s + GEMM_BK - 1) / GEMM_BK * GEMM_BK;
    s = (in_features + kps - 1) / kps;
    *nsplit = (int)s; *k_per_split = (int)kps;
}

extern "C" {

int64_t qa_linear_forward_split_scratch_bytes(int64_t rows, int32_t in_features, int32_t out_features) {
    if (rows <= 0 || in_features <= 0 || out_features <= 0) return 0;
    int s, kps; fwd_split_plan(rows, in_features, out_features, &s, &kps);
    return (int64_t)s * pad4(rows * out_features) * 4;
}

int qa_linear_forward_split(const float *x, int64_t ldx, const float *weight, int64_t ldw, const float *bias, float *y, int64_t ldy, int64_t rows,
                            int32_t in_features, int32_t out_features, int32_t act, float alpha, void *scratch, int64_t scratch_bytes, void *stream) {
    if (!x || !weight || !y || !scratch || rows <= 0 || rows > INT32_MAX || in_features <= 0 || out_features <= 0 || out_features % 4 || ldx < in_features ||
        ldw < in_features || ldy < out_features || ldy % 4 || act < 0 || act > 2 || !aligned16(y) || (bias && !aligned16(bias)) ||
        !fits_u32(rows, ldx) || !fits_u32(out_features, ldw)) {
        snprintf(g_gerr, sizeof(g_gerr), "qa_linear_forward_split: bad argument (out features and ldy multiples of 4, y / bias 16-byte aligned)"); return QA_E_ARG; }
    if (scratch_bytes < qa_linear_forward_split_scratch_bytes(rows, in_features, out_features) || !aligned16(scratch)) {
        snprintf(g_gerr, sizeof(g_gerr), "qa_linear_forward_split: scratch too small or not 16-byte aligned"); return QA_E_ARG; }
    int s, kps; fwd_split_plan(rows, in_features, out_features, &s, &kps);
    const int64_t n_pad = pad4(rows * out_features);
    GemmArgs g = {};
    g.A = weight; g.lda = ldw; g.a_count = out_features;
    g.B = x; g.ldb = ldx; g.b_count = (int)rows;
    g.kred = in_features; g.k_per_split = kps; g.nsplit = s;
    g.out = (float *)scratch; g.ldo = out_features; g.out_split_stride = n_pad;
    g.a_vec = (aligned16(weight) && ldw % 4 == 0 && in_features % 4 == 0) ? 4 : 1;
    g.b_vec = (aligned16(x) && ldx % 4 == 0 && in_features % 4 == 0) ? 4 : 1;
    g.o_vec = 4;
    hipStream_t st = (hipStream_t)stream;
    gemm_launch<false, false, 0, false>(g_force_cfg >= 0 ? g_force_cfg : 2, g, st);
    const int64_t nthreads = rows * (out_features / 4);
    hipLaunchKernelGGL(qa_slab_bias_act_kernel, dim3((unsigned)((nthreads + 255) / 256)), dim3(256), 0, st, (const float *)scratch, n_pad, s, rows,
                       (int)out_features, bias, (int)act, alpha, y, ldy);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { snprintf(g_gerr, sizeof(g_gerr), "qa_linear_forward_split: %s", hipGetErrorString(e)); return QA_E_DEVICE; }
    return QA_OK;
}

}  // extern "C"
