/* Policy inference of the rollout as ONE launch per env step (include/qa_sim.h, "policy inference").
 *
 * What it replaces: SSInfoGAIL.act (bbc/rsl_rl/algorithms/gail.py:176-197) = Estimator.forward (estimator.py:35-36) +
 * ActorCritic.update_distribution / evaluate (actor_critic.py:171-196,222-225): ~9 small GEMMs at M = num_envs, 10 ELU
 * launches, 3 concatenations per env step.  At M = 4096 every one of those is launch-latency sized (5-30 us each).
 *
 * Shape of the work: a chain of fully connected layers over row-independent data.  One workgroup owns a tile of 16 rows
 * and carries it through the WHOLE chain with the activations in LDS; only the weights are streamed (from L2: they are
 * ~3 MB, shared by all workgroups) and only the heads' outputs go back to HBM.
 *
 *   - fp32 MFMA v_mfma_f32_16x16x4f32: 16 rows x 16 output columns per tile, K consumed 4 at a time.
 *   - 8 wavefronts per workgroup (two per SIMD), each owning a contiguous range of output-column tiles of the current
 *     layer (up to 4 tiles = 16 accumulator registers), all sharing the A fragment read from LDS.
 *   - weights are repacked once per rollout (qa_mlp_pack) so that the B fragment of (k-block j, column tile t) is one
 *     contiguous 1 KB run: lane l reads float4 #l.  Within a 16-wide k-block lane (n, kq) holds W[n][16j + 4kq + 0..3];
 *     MFMA step s of the block uses element s of that float4 as "k = kq" -- a permutation of K that A follows, so the
 *     sum is the same.
 *   - 256 workgroups at 4096 envs = one per CU.
 *
 * Roofline: MFMA-bound.  1.488 MFLOP per row (bench.py ROLLOUT_FLOPS_PER_SAMPLE); a 16-row tile on one CU at the fp32
 * dense peak (256 FLOP/cycle/CU) needs 93 k cycles = 39 us.  L2 -> CU weight traffic is 3 MB per workgroup.
 */
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <atomic>

#include "../../include/qa_sim.h"

namespace {

typedef float f4 __attribute__((ext_vector_type(4)));

constexpr int MLP_ROWS = 16;
constexpr int MLP_WAVES = 8;      /* two per SIMD: one issues MFMAs while the other waits for its weight fragments */
/* LDS buffers (floats per row incl. padding; strides = 4 mod 32 so the 16 rows of an A read fall on distinct bank groups) */
constexpr int MLP_NBUF = 4;
constexpr int S0 = QA_MLP_BUF0_COLS + 4, S1 = QA_MLP_BUF1_COLS + 4, S2 = QA_MLP_BUF2_COLS + 4, S3 = QA_MLP_BUF3_COLS + 4;
constexpr int B1 = MLP_ROWS * S0, B2 = B1 + MLP_ROWS * S1, B3 = B2 + MLP_ROWS * S2;
constexpr int MLP_LDS_FLOATS = B3 + MLP_ROWS * S3;
__device__ __forceinline__ int mlp_stride(int b) { return b == 0 ? S0 : b == 1 ? S1 : b == 2 ? S2 : S3; }
__device__ __forceinline__ int mlp_base(int b) { return b == 0 ? 0 : b == 1 ? B1 : b == 2 ? B2 : B3; }

/* column tiles per wavefront for a layer of nt tiles, and the prefetch depth that goes with it */
__host__ __device__ constexpr int mlp_tpw(int nt) { return (nt + MLP_WAVES - 1) / MLP_WAVES <= 1 ? 1 : (nt + MLP_WAVES - 1) / MLP_WAVES <= 2 ? 2 : 4; }
__host__ __device__ constexpr int mlp_pf(int tpw) { return tpw >= 4 ? 4 : 8; }
/* k-blocks of 16 a layer is stored with: rounded up to the prefetch depth so the pipelined loop has no remainder */
__host__ __device__ constexpr int mlp_kb(int k, int n) { return ((k + 15) / 16 + mlp_pf(mlp_tpw((n + 15) / 16)) - 1) / mlp_pf(mlp_tpw((n + 15) / 16)) * mlp_pf(mlp_tpw((n + 15) / 16)); }

struct MlpDevOp {
    int32_t kind, src_buf, src_col, dst_buf, dst_col, k, n, act, out_index, nt, kb, tpw;
    int64_t w_off, b_off;
    int32_t strand, flags;      /* strand: which workgroup of a tile's group runs the op (blockIdx.y), see mlp_strands; flags: QA_MLP_F_* */
    int32_t out_col, aux_index, aux_col, pad_;      /* ABI 17: first column of the global output; the saved activation an act >= 4 reads */
};

struct MlpArgs {
    const float *x;
    int64_t x_stride;
    int32_t rows, x_cols, num_ops, strands;
    const float *packed;
    float *out[QA_MLP_MAX_OUTPUTS];
    int64_t out_stride[QA_MLP_MAX_OUTPUTS];
    MlpDevOp ops[QA_MLP_MAX_OPS];
    /* two-group launch (qa_mlp_forward_groups_kernel): LDS base / row stride (floats) of buffers 0..3 as group g sees them, the LDS size and
     * where the scratch part (zeroed at the start) begins */
    int32_t g_base[2][MLP_NBUF], g_stride[2][MLP_NBUF], g_lds_floats, g_scratch0;
};

struct PackLayer { const float *w, *b; int32_t n, k; int64_t w_off, b_off; int32_t trans, pad_; };
struct PackArgs { PackLayer l[QA_MLP_MAX_OPS]; int32_t num; float *packed; };

/* packed[w_off + ((j * nt + t) * 64 + lane) * 4 + s] = W[16 t + lane % 16][16 j + 4 (lane / 16) + s], zero outside (n, k) */
__global__ void qa_mlp_pack_kernel(PackArgs a) {
    const PackLayer L = a.l[blockIdx.y];
    const int nt = (L.n + 15) / 16, kb = mlp_kb(L.k, L.n);
    const int64_t total = (int64_t)nt * kb * 256;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int s = (int)(i & 3), lane = (int)((i >> 2) & 63);
        const int64_t jt = i >> 8;
        const int t = (int)(jt % nt), j = (int)(jt / nt);
        const int row = 16 * t + (lane & 15), col = 16 * j + 4 * (lane >> 4) + s;
        /* trans (QA_MLP_F_TRANSPOSED): the source is the FORWARD layer's (k, n) row-major matrix, this op multiplies by its transpose */
        a.packed[L.w_off + i] = (row < L.n && col < L.k) ? (L.trans ? L.w[(int64_t)col * L.n + row] : L.w[(int64_t)row * L.k + col]) : 0.f;
    }
    if (blockIdx.x == 0)
        for (int i = threadIdx.x; i < nt * 16; i += blockDim.x) a.packed[L.b_off + i] = (i < L.n && L.b) ? L.b[i] : 0.f;
}

/* ELU(alpha 1).  __expf = v_exp_f32(v * log2 e), 2 instructions, relative error ~2^-22 of exp(v) <= 1: the same size as
 * the rounding noise of the fp32 dot product in front of it; expf() is ~35 instructions x 16 values per lane per wide layer */
__device__ __forceinline__ float elu1(float v) { return v > 0.f ? v : __expf(v) - 1.f; }
/* epilogue of a layer: bias, then the activation (act 1..3) or, on input-gradient chains, the DERIVATIVE of the forward layer's activation taken
 * from its saved output y (act 4: ELU, y > 0 ? 1 : y + 1; 5: ReLU; 6: tanh, 1 - y^2) */
__device__ __forceinline__ float mlp_act(int act, float v, float y) {
    return act == 0 ? v : act == 1 ? elu1(v) : act == 2 ? fmaxf(v, 0.f) : act == 3 ? tanhf(v) :
           act == 4 ? v * (y > 0.f ? 1.f : y + 1.f) : act == 5 ? (y > 0.f ? v : 0.f) : v * (1.f - y * y);
}

/* one layer for this wavefront: TPW column tiles starting at tile t0.
 * The weight fragments are fetched PF-1 k-blocks ahead of their use into a ring of PF register stages (the L2 round trip is
 * several hundred cycles; one k-block of MFMAs is 128 * TPW cycles).  The sched_barriers keep the compiler from sinking the
 * prefetch back down to its use, which it otherwise does to shorten live ranges. */
struct MlpPlace { int src_base, src_stride, dst_base, dst_stride; };      /* where the op's source / destination buffers are in LDS (floats) */
/* EXT: the ABI 17 features (saved copies, derivative epilogues); the rollout's launches compile without them */
template <int TPW, int PF = mlp_pf(TPW), bool EXT = false>
__device__ __forceinline__ void mlp_layer(const MlpDevOp &op, const float *packed, float *lds, const MlpArgs &a, int row0, int wave, int lane, const MlpPlace pl) {
    const int nt = op.nt, kb = op.kb;
    const int t0 = wave * TPW;
    if (t0 >= nt) return;
    const int m = lane & 15, kq = lane >> 4;
    const float *src = lds + pl.src_base + m * pl.src_stride + op.src_col + 4 * kq;
    const f4 *wp = reinterpret_cast<const f4 *>(packed + op.w_off) + lane;
    int toff[TPW];
#pragma unroll
    for (int i = 0; i < TPW; ++i) toff[i] = (t0 + i < nt ? t0 + i : nt - 1) * 64;
    const int jstride = nt * 64;
    f4 acc[TPW];
#pragma unroll
    for (int i = 0; i < TPW; ++i) acc[i] = f4{0.f, 0.f, 0.f, 0.f};
    float bias[TPW];
#pragma unroll
    for (int i = 0; i < TPW; ++i) bias[i] = packed[op.b_off + (toff[i] >> 2) + m];       /* padded to whole tiles by qa_mlp_pack */
    /* (issued before the k-loop: the round trip of these loads hides behind the layer's MFMAs) */
    const int act = op.act;
    const int live = a.rows - (row0 + 4 * kq);              /* rows of this lane's group of 4 that exist */
    float yv[TPW][4];
    if (EXT && act >= 4) {         /* the saved forward activation whose derivative scales the product (rows beyond the batch: the last row's, unused) */
        const int64_t as = a.out_stride[op.aux_index];
        const float *y = a.out[op.aux_index] + op.aux_col + m;
#pragma unroll
        for (int i = 0; i < TPW; ++i) {
            const int col = (t0 + i < nt ? t0 + i : nt - 1) * 16;
            const bool in = col + m < op.n;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = row0 + 4 * kq + r;
                yv[i][r] = in ? y[(int64_t)(row < a.rows ? row : a.rows - 1) * as + col] : 0.f;
            }
        }
    } else {
#pragma unroll
        for (int i = 0; i < TPW; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) yv[i][r] = 0.f;
    }
    f4 bw[PF][TPW], av[PF];
#pragma unroll
    for (int u = 0; u < PF - 1; ++u) {
        const int j = u < kb ? u : kb - 1;
#pragma unroll
        for (int i = 0; i < TPW; ++i) bw[u][i] = wp[j * jstride + toff[i]];
        av[u] = *reinterpret_cast<const f4 *>(src + 16 * j);
        __builtin_amdgcn_sched_barrier(0);          /* same issue order as inside the loop, so the wait counts at the loop head stay exact */
    }
    for (int j0 = 0; j0 < kb; j0 += PF) {          /* kb is a multiple of PF (zero k-blocks appended by qa_mlp_pack) */
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            const int j = j0 + u;
            {
                constexpr int PFm1 = PF - 1;
                const int jp = j + PFm1 < kb ? j + PFm1 : kb - 1;
                const int st = (u + PFm1) % PF;
#ifndef QA_MLP_ABLATE_W          /* profiling ablations (tools/mlp_profile.py): results are wrong with either defined */
#pragma unroll
                for (int i = 0; i < TPW; ++i) bw[st][i] = wp[jp * jstride + toff[i]];
#else
                (void)jp;
#endif
#ifndef QA_MLP_ABLATE_A
                av[st] = *reinterpret_cast<const f4 *>(src + 16 * jp);
#endif
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
                for (int i = 0; i < TPW; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][s4], bw[u][i][s4], acc[i], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    /* D layout: register r of lane l = D[row 4 (l / 16) + r][column l % 16].  Destination resolved once per layer. */
    const bool to_global = op.dst_buf < 0 || (EXT && (op.flags & QA_MLP_F_SAVE));
    const int64_t os = to_global ? a.out_stride[op.out_index] : 0;
    float *g = to_global ? a.out[op.out_index] + (int64_t)(row0 + 4 * kq) * os + op.out_col + m : nullptr;
    const int ds = pl.dst_stride;
    float *d = op.dst_buf >= 0 ? lds + pl.dst_base + (4 * kq) * ds + op.dst_col + m : nullptr;
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
        const int col = (t0 + i) * 16;
        if (t0 + i < nt && col + m < op.n) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float v = mlp_act(EXT ? act : (act > 3 ? 0 : act), acc[i][r] + bias[i], yv[i][r]);
                if (d) d[r * ds + col] = v;
                if (g && r < live) g[r * os + col] = v;
            }
        }
    }
}

/* QA_MLP_GRAD (ABI 17), elementwise over the tile: v = src (+ dst with QA_MLP_F_ADD), times the activation derivative of `act` (4..6; 0: none)
 * taken from the saved output; to dst and, with QA_MLP_F_SAVE, to the global output.  QA_MLP_COPY with QA_MLP_F_SAVE writes its copy out too. */
__device__ __forceinline__ void mlp_elementwise(const MlpDevOp &op, float *lds, const MlpArgs &a, int row0, int tid, int nthreads, const MlpPlace pl) {
    const bool save = (op.flags & QA_MLP_F_SAVE) != 0, add = op.kind == QA_MLP_GRAD && (op.flags & QA_MLP_F_ADD);
    const int act = op.kind == QA_MLP_GRAD ? op.act : 0;
    for (int i = tid; i < MLP_ROWS * op.n; i += nthreads) {
        const int r = i / op.n, c = i - r * op.n;
        const bool live = row0 + r < a.rows;
        float v = op.kind == QA_MLP_LOAD ? (live ? a.out[op.aux_index][(int64_t)(row0 + r) * a.out_stride[op.aux_index] + op.aux_col + c] : 0.f)
                                         : lds[pl.src_base + r * pl.src_stride + op.src_col + c];
        float *d = lds + pl.dst_base + r * pl.dst_stride + op.dst_col + c;
        if (add) v += *d;
        if (act >= 4) v = mlp_act(act, v, live ? a.out[op.aux_index][(int64_t)(row0 + r) * a.out_stride[op.aux_index] + op.aux_col + c] : 0.f);
        *d = v;
        if (save && live) a.out[op.out_index][(int64_t)(row0 + r) * a.out_stride[op.out_index] + op.out_col + c] = v;
    }
}

/* -DQA_MLP_PROF (tools/mlp_profile.py): s_memtime after every op of workgroup 0 into output 3, read as int64[] */
#ifdef QA_MLP_PROF
#define MLP_STAMP(i) do { if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) reinterpret_cast<long long *>(a.out[3])[i] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
#else
#define MLP_STAMP(i) do { } while (0)
#endif

template <bool EXT>
__global__ __launch_bounds__(MLP_WAVES * 64) void qa_mlp_forward_kernel(MlpArgs a) {
    __shared__ __attribute__((aligned(16))) float lds[MLP_LDS_FLOATS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int row0 = blockIdx.x * MLP_ROWS;
    MLP_STAMP(QA_MLP_MAX_OPS + 1);
    /* scratch buffers start at zero (padding columns are read against zero weights and must be finite) */
    for (int i = mlp_base(1) + tid; i < MLP_LDS_FLOATS; i += MLP_WAVES * 64) lds[i] = 0.f;
    {   /* input tile: one wavefront per 4 rows, coalesced along the row; every load of a row is issued before the first store */
        constexpr int PER = (S0 + 63) / 64;
        for (int r = wave; r < MLP_ROWS; r += MLP_WAVES) {
            const bool live = row0 + r < a.rows;
            const float *xr = a.x + (int64_t)(row0 + r) * a.x_stride;
            float v[PER];
#pragma unroll
            for (int q = 0; q < PER; ++q) { const int c = lane + 64 * q; v[q] = (live && c < a.x_cols) ? xr[c] : 0.f; }
#pragma unroll
            for (int q = 0; q < PER; ++q) { const int c = lane + 64 * q; if (c < S0) lds[r * S0 + c] = v[q]; }
        }
    }
    __syncthreads();
    MLP_STAMP(0);
    MlpDevOp op = a.ops[0];
    const int strand = blockIdx.y;                 /* > 0 only when the launch has too few row tiles to fill the chip (mlp_strands) */
    for (int o = 0; o < a.num_ops; ++o) {
        const MlpDevOp nxt = a.ops[o + 1 < a.num_ops ? o + 1 : o];      /* scalar loads of the next descriptor fly during this op */
        if (op.strand != strand) { op = nxt; continue; }                  /* workgroup-uniform: another workgroup of this tile runs it */
        if (op.kind != QA_MLP_LAYER) {
            mlp_elementwise(op, lds, a, row0, tid, MLP_WAVES * 64, MlpPlace{mlp_base(op.src_buf), mlp_stride(op.src_buf), mlp_base(op.dst_buf), mlp_stride(op.dst_buf)});
        } else {
            const MlpPlace pl{mlp_base(op.src_buf), mlp_stride(op.src_buf), op.dst_buf >= 0 ? mlp_base(op.dst_buf) : 0, op.dst_buf >= 0 ? mlp_stride(op.dst_buf) : 0};
            switch (op.tpw) {
                case 1: mlp_layer<1, mlp_pf(1), EXT>(op, a.packed, lds, a, row0, wave, lane, pl); break;
                case 2: mlp_layer<2, mlp_pf(2), EXT>(op, a.packed, lds, a, row0, wave, lane, pl); break;
                default: mlp_layer<4, mlp_pf(4), EXT>(op, a.packed, lds, a, row0, wave, lane, pl); break;
            }
        }
        __syncthreads();
        MLP_STAMP(o + 1);
        op = nxt;
    }
}

/* Many rows (a tile per CU and more): the chain's independent strands side by side INSIDE the workgroup.  At 4096 rows the launch above is one
 * workgroup per CU walking 13 layers one after the other, a workgroup barrier behind each: every layer pays ~3.3 k cycles of cold weight stream +
 * epilogue + barrier whether it is 671 -> 512 or 64 -> 4 (30 % of the 186 k cycles of a tile, profiles/r1_policy_kernel_timing.md), and nothing else
 * runs on the CU meanwhile.  Here waves 0..3 (one per SIMD) run strand 0 and waves 4..7 strand 1 (mlp_strands: critic | estimator -> encoder ->
 * actor), each group with its own LDS scratch buffers and its own barrier (an LDS arrival counter), so that one strand's narrow layers and pipeline
 * fills hide behind the other's MFMA work; with four waves per group a 512-wide layer is 8 column tiles per wave (one A fragment per 32 MFMAs).
 * Per-tile arithmetic is the one-group kernel's (same k order per output), so the outputs are bit-identical. */
constexpr int GRP_WAVES = MLP_WAVES / 2;
__device__ __forceinline__ void group_barrier(int *ctr, int &target, int lane) {
    /* the LDS unit serves a CU's requests in order: once this wave's writes have returned (lgkmcnt 0) its arrival is behind them for every reader */
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (lane == 0) __hip_atomic_fetch_add(ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    target += GRP_WAVES;
    while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < target) __builtin_amdgcn_s_sleep(2);
    asm volatile("" ::: "memory");
}

__global__ __launch_bounds__(MLP_WAVES * 64) void qa_mlp_forward_groups_kernel(MlpArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    __shared__ int s_arrive[2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int row0 = blockIdx.x * MLP_ROWS;
    if (tid < 2) s_arrive[tid] = 0;
    for (int i = a.g_scratch0 + tid; i < a.g_lds_floats; i += MLP_WAVES * 64) lds[i] = 0.f;
    {   /* input tile: one wavefront per row at a time; columns beyond x_cols (row padding, read against zero weights) are zero */
        const int s0 = a.g_stride[0][0];
        for (int r = wave; r < MLP_ROWS; r += MLP_WAVES) {
            const bool live = row0 + r < a.rows;
            const float *xr = a.x + (int64_t)(row0 + r) * a.x_stride;
            for (int c0 = 0; c0 < s0; c0 += 256) {          /* four loads in flight per lane */
                float v[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) { const int c = c0 + lane + 64 * q; v[q] = (live && c < a.x_cols) ? xr[c] : 0.f; }
#pragma unroll
                for (int q = 0; q < 4; ++q) { const int c = c0 + lane + 64 * q; if (c < s0) lds[r * s0 + c] = v[q]; }
            }
        }
    }
    __syncthreads();
    const int grp = wave / GRP_WAVES, gw = wave % GRP_WAVES, gtid = tid % (GRP_WAVES * 64);
    int target = 0;
    for (int o = 0; o < a.num_ops; ++o) {
        if (a.ops[o].strand != grp) continue;                     /* group-uniform */
        const MlpDevOp op = a.ops[o];
        const MlpPlace pl{a.g_base[grp][op.src_buf], a.g_stride[grp][op.src_buf], op.dst_buf >= 0 ? a.g_base[grp][op.dst_buf] : 0,
                          op.dst_buf >= 0 ? a.g_stride[grp][op.dst_buf] : 0};
        if (op.kind != QA_MLP_LAYER) {
            mlp_elementwise(op, lds, a, row0, gtid, GRP_WAVES * 64, pl);
        } else {
            const int per = (op.nt + GRP_WAVES - 1) / GRP_WAVES;      /* column tiles per wave of the group */
            if (per <= 1) mlp_layer<1, 8>(op, a.packed, lds, a, row0, gw, lane, pl);
            else if (per <= 2) mlp_layer<2, 8>(op, a.packed, lds, a, row0, gw, lane, pl);
            else if (per <= 4) mlp_layer<4, 4>(op, a.packed, lds, a, row0, gw, lane, pl);
            else mlp_layer<8, 4>(op, a.packed, lds, a, row0, gw, lane, pl);
        }
        group_barrier(&s_arrive[grp], target, lane);
    }
}

/* Few rows = few 16-row tiles = most CUs idle while each busy one walks the whole chain (80 us whether the launch has 1 row or 4096).
 * The chain is usually several chains that share nothing but the input tile (SSInfoGAIL.act: the critic is 68 % of the flops and feeds nobody),
 * so when tiles * strands still fits the chip each tile is given to `strands` workgroups (blockIdx.y), every one of which stages the input tile
 * and runs only ITS ops, in program order.  Ops are grouped by FLOW dependence through the LDS scratch buffers (an op joins the strand of the
 * last writer of every column it reads; buffer re-use between unrelated chains is no dependence here, each workgroup has its own LDS); the
 * groups are dealt to strands largest first onto the least loaded.  Same arithmetic per op => the outputs are bit-identical to strands = 1. */
int mlp_strands(const qa_mlp_op *ops, int num_ops, int max_strands, int32_t *strand_of) {
    for (int i = 0; i < num_ops; ++i) strand_of[i] = 0;
    if (max_strands < 2) return 1;
    int parent[QA_MLP_MAX_OPS];
    for (int i = 0; i < num_ops; ++i) parent[i] = i;
    auto find = [&](int i) { while (parent[i] != i) { parent[i] = parent[parent[i]]; i = parent[i]; } return i; };
    constexpr int MAXC = QA_MLP_BUF1_COLS > QA_MLP_BUF2_COLS ? (QA_MLP_BUF1_COLS > QA_MLP_BUF3_COLS ? QA_MLP_BUF1_COLS : QA_MLP_BUF3_COLS)
                                                             : (QA_MLP_BUF2_COLS > QA_MLP_BUF3_COLS ? QA_MLP_BUF2_COLS : QA_MLP_BUF3_COLS);
    int last_writer[MLP_NBUF][MAXC + 4];
    for (int b = 0; b < MLP_NBUF; ++b) for (int c = 0; c < MAXC + 4; ++c) last_writer[b][c] = -1;
    for (int i = 0; i < num_ops; ++i) {
        const qa_mlp_op &o = ops[i];
        const int rd = o.kind == QA_MLP_LAYER ? o.k : o.n;
        if (o.src_buf > 0)
            for (int c = o.src_col; c < o.src_col + rd && c < MAXC + 4; ++c)
                if (last_writer[o.src_buf][c] >= 0) parent[find(i)] = find(last_writer[o.src_buf][c]);
        if (o.dst_buf > 0 && o.kind == QA_MLP_GRAD && (o.flags & QA_MLP_F_ADD))       /* dst += ...: a read of what was there */
            for (int c = o.dst_col; c < o.dst_col + o.n && c < MAXC + 4; ++c)
                if (last_writer[o.dst_buf][c] >= 0) parent[find(i)] = find(last_writer[o.dst_buf][c]);
        if (o.dst_buf > 0)
            for (int c = o.dst_col; c < o.dst_col + o.n && c < MAXC + 4; ++c) last_writer[o.dst_buf][c] = i;
    }
    double cost[QA_MLP_MAX_OPS] = {0};              /* per root: MFMA work (copies and the latency of narrow layers as a small constant) */
    for (int i = 0; i < num_ops; ++i) cost[find(i)] += (ops[i].kind == QA_MLP_LAYER ? (double)ops[i].k * ops[i].n : 0.0) + 4096.0;
    int roots[QA_MLP_MAX_OPS], nroots = 0;
    for (int i = 0; i < num_ops; ++i) if (find(i) == i) roots[nroots++] = i;
    if (nroots < 2) return 1;
    for (int i = 1; i < nroots; ++i)                /* insertion sort, largest first (ties: program order) */
        for (int j = i; j > 0 && cost[roots[j]] > cost[roots[j - 1]]; --j) { const int t = roots[j]; roots[j] = roots[j - 1]; roots[j - 1] = t; }
    const int ns = nroots < max_strands ? nroots : max_strands;
    double load[8] = {0};
    int strand_of_root[QA_MLP_MAX_OPS];
    for (int r = 0; r < nroots; ++r) {
        int best = 0;
        for (int g = 1; g < ns; ++g) if (load[g] < load[best]) best = g;
        strand_of_root[roots[r]] = best; load[best] += cost[roots[r]];
    }
    for (int i = 0; i < num_ops; ++i) strand_of[i] = strand_of_root[find(i)];
    return ns;
}

/* LDS geometry of the two-group launch: buffer 0 (the input tile, shared) as wide as the chain reads it; behind it each group's scratch buffers
 * 1..3, each as wide as the group's ops touch them (row stride = 4 mod 32 floats, as in the one-group layout), and enough room behind the last row
 * for the padded k-blocks of every layer (read against zero weights; what lies there is finite: zeroed scratch or activations).  false when
 * the chain does not fit the CU's LDS this way. */
constexpr int MLP_LDS_LIMIT_FLOATS = (160 * 1024 - 256) / 4;
bool mlp_group_geometry(const qa_mlp_op *ops, int num_ops, const int32_t *strand_of, int x_cols, MlpArgs &a) {
    int width[2][MLP_NBUF] = {{0}};
    int w0 = x_cols;
    for (int i = 0; i < num_ops; ++i) {
        const qa_mlp_op &o = ops[i];
        const int g = strand_of[i], rd = o.kind == QA_MLP_LAYER ? o.k : o.n;
        if (g < 0 || g > 1) return false;
        if (o.src_buf == 0) w0 = w0 > o.src_col + rd ? w0 : o.src_col + rd;
        else width[g][o.src_buf] = width[g][o.src_buf] > o.src_col + rd ? width[g][o.src_buf] : o.src_col + rd;
        if (o.dst_buf > 0) width[g][o.dst_buf] = width[g][o.dst_buf] > o.dst_col + o.n ? width[g][o.dst_buf] : o.dst_col + o.n;
    }
    /* first / last op of the group that touches the buffer: a buffer first used after another one's last use is laid over it (the critic's third
     * layer writes where its first layer's output was) */
    int first[2][MLP_NBUF], last[2][MLP_NBUF];
    for (int g = 0; g < 2; ++g) for (int b = 0; b < MLP_NBUF; ++b) { first[g][b] = num_ops; last[g][b] = -1; }
    for (int i = 0; i < num_ops; ++i) {
        const int g = strand_of[i];
        for (int b : {ops[i].src_buf, ops[i].dst_buf})
            if (b > 0) { first[g][b] = first[g][b] < i ? first[g][b] : i; last[g][b] = i; }
    }
    auto stride_of = [](int w) { return (w + 31) / 32 * 32 + 4; };
    int cur = MLP_ROWS * stride_of(w0);
    a.g_scratch0 = cur;
    for (int g = 0; g < 2; ++g) {
        a.g_base[g][0] = 0; a.g_stride[g][0] = stride_of(w0);
        for (int b = 1; b < MLP_NBUF; ++b) {
            a.g_stride[g][b] = width[g][b] > 0 ? stride_of(width[g][b]) : 0;
            int over = 0;
            for (int e = 1; e < b && !over; ++e)
                if (a.g_stride[g][b] > 0 && a.g_stride[g][e] >= a.g_stride[g][b] && last[g][e] < first[g][b]) over = e;
            if (over) { a.g_base[g][b] = a.g_base[g][over]; last[g][over] = last[g][b]; }      /* the region stays taken until b's last use */
            else { a.g_base[g][b] = cur; cur += MLP_ROWS * a.g_stride[g][b]; }
        }
    }
    for (int i = 0; i < num_ops; ++i) {
        const qa_mlp_op &o = ops[i];
        if (o.kind != QA_MLP_LAYER) continue;
        const int g = strand_of[i];
        const int end = a.g_base[g][o.src_buf] + (MLP_ROWS - 1) * a.g_stride[g][o.src_buf] + o.src_col + 16 * mlp_kb(o.k, o.n);
        cur = cur > end ? cur : end;
    }
    a.g_lds_floats = (cur + 3) & ~3;
    return a.g_lds_floats <= MLP_LDS_LIMIT_FLOATS;
}

int lds_base(int b) { return b == 0 ? 0 : b == 1 ? B1 : b == 2 ? B2 : B3; }
int buf_cols(int b) { return b == 0 ? QA_MLP_BUF0_COLS : b == 1 ? QA_MLP_BUF1_COLS : b == 2 ? QA_MLP_BUF2_COLS : QA_MLP_BUF3_COLS; }

}  // namespace

#ifdef QA_MLP_PROF
thread_local char qa_err_buf[512];
#else
extern thread_local char qa_err_buf[512];
#endif
#define g_perr qa_err_buf

extern "C" {

int64_t qa_mlp_packed_floats(const qa_mlp_op *ops, int32_t num_ops) {
    if (!ops || num_ops <= 0) return -1;
    int64_t end = 0;
    for (int i = 0; i < num_ops; ++i) {
        if (ops[i].kind != QA_MLP_LAYER) continue;
        const int64_t nt = (ops[i].n + 15) / 16, kb = mlp_kb(ops[i].k, ops[i].n);
        const int64_t we = ops[i].w_off + nt * kb * 256, be = ops[i].b_off + nt * 16;
        end = we > end ? we : end;
        end = be > end ? be : end;
    }
    return end;
}

static int mlp_check(const qa_mlp_op *ops, int32_t num_ops, const char *who) {
    if (!ops || num_ops <= 0 || num_ops > QA_MLP_MAX_OPS) { snprintf(g_perr, sizeof(g_perr), "%s: 1..%d ops expected", who, QA_MLP_MAX_OPS); return QA_E_ARG; }
    for (int i = 0; i < num_ops; ++i) {
        const qa_mlp_op &o = ops[i];
        const bool layer = o.kind == QA_MLP_LAYER;
        bool ok = (layer || o.kind == QA_MLP_COPY || o.kind == QA_MLP_GRAD || o.kind == QA_MLP_LOAD) && o.src_buf >= 0 && o.src_buf < MLP_NBUF && o.src_col >= 0 && o.n > 0 &&
                  (o.flags & ~(QA_MLP_F_SAVE | QA_MLP_F_TRANSPOSED | QA_MLP_F_ADD)) == 0 && o.out_col >= 0;
        if (ok && (o.flags & QA_MLP_F_SAVE)) ok = o.out_index >= 0 && o.out_index < QA_MLP_MAX_OUTPUTS;
        if (ok && ((layer && o.act >= 4) || (o.kind == QA_MLP_GRAD && o.act != 0)))
            ok = o.act >= 4 && o.act <= 6 && o.aux_index >= 0 && o.aux_index < QA_MLP_MAX_OUTPUTS && o.aux_col >= 0;
        if (ok && o.kind == QA_MLP_LOAD) {        /* global -> scratch buffer: no LDS source, no activation */
            ok = o.aux_index >= 0 && o.aux_index < QA_MLP_MAX_OUTPUTS && o.aux_col >= 0 && o.act == 0 && o.dst_buf > 0 && o.dst_buf < MLP_NBUF && o.dst_col >= 0 &&
                 o.dst_col + o.n <= buf_cols(o.dst_buf);
            if (!ok) { snprintf(g_perr, sizeof(g_perr), "%s: op %d (load) is malformed", who, i); return QA_E_ARG; }
            continue;
        }
        if (ok && layer) {
            const int kpad = mlp_kb(o.k, o.n) * 16;         /* columns the layer reads (beyond k: against zero weights) */
            /* reading past the row's padding lands in the next row / buffer (finite activations, zero weights): allowed while inside LDS */
            ok = o.k > 0 && o.act >= 0 && o.act <= 6 && (o.src_col % 4) == 0 && o.src_col + o.k <= buf_cols(o.src_buf) + 4 &&
                 lds_base(o.src_buf) + (MLP_ROWS - 1) * (buf_cols(o.src_buf) + 4) + o.src_col + kpad <= MLP_LDS_FLOATS && o.w_off >= 0 && o.b_off >= 0 && (o.w_off % 4) == 0 &&
                 (o.n + 15) / 16 <= 4 * MLP_WAVES;
            if (ok && o.dst_buf >= 0) ok = o.dst_buf > 0 && o.dst_buf < MLP_NBUF && o.dst_buf != o.src_buf && o.dst_col >= 0 && o.dst_col + o.n <= buf_cols(o.dst_buf);
            if (ok && o.dst_buf < 0) ok = o.out_index >= 0 && o.out_index < QA_MLP_MAX_OUTPUTS;
        } else if (ok) {
            ok = o.src_col + o.n <= buf_cols(o.src_buf) && o.dst_buf > 0 && o.dst_buf < MLP_NBUF && o.dst_col >= 0 && o.dst_col + o.n <= buf_cols(o.dst_buf) &&
                 (o.dst_buf != o.src_buf || o.dst_col >= o.src_col + o.n || o.src_col >= o.dst_col + o.n);
        }
        if (!ok) { snprintf(g_perr, sizeof(g_perr), "%s: op %d is malformed (kind %d, src %d@%d, dst %d@%d, k %d, n %d)", who, i, o.kind, o.src_buf, o.src_col,
                            o.dst_buf, o.dst_col, o.k, o.n); return QA_E_ARG; }
    }
    return QA_OK;
}

int qa_mlp_strands(const qa_mlp_op *ops, int32_t num_ops, int32_t max_strands, int32_t *strand_of) {
    int rc = mlp_check(ops, num_ops, "qa_mlp_strands");
    if (rc != QA_OK) return rc;
    if (!strand_of || max_strands < 1) { snprintf(g_perr, sizeof(g_perr), "qa_mlp_strands: bad argument"); return QA_E_ARG; }
    return mlp_strands(ops, num_ops, max_strands > 4 ? 4 : max_strands, strand_of);
}

/* Measured (profiles/r5_policy_groups.txt): 87.7 us against the one-group kernel's 85.5 us at 4096 rows, 336.7 against 320.6 at 16,384 -- the strands
 * share the MFMA pipes they were meant to keep busy, and a lone four-wave strand hides its weight stream worse than eight waves do.  The two-group
 * launch therefore ships OFF (default 1); QA_MLP_GROUPS=2 / qa_mlp_set_groups(2) turn it on (the tests do, it is bit-identical). */
static std::atomic<int> g_mlp_groups{-1};           /* -1: not decided yet (QA_MLP_GROUPS in the environment, default 1) */
static int mlp_groups_switch() {
    int v = g_mlp_groups.load(std::memory_order_relaxed);
    if (v < 0) {
        const char *e = getenv("QA_MLP_GROUPS");
        int expected = -1;
        g_mlp_groups.compare_exchange_strong(expected, e ? (atoi(e) >= 2 ? 2 : 1) : 1);       /* a concurrent qa_mlp_set_groups wins */
        v = g_mlp_groups.load(std::memory_order_relaxed);
    }
    return v;
}

int qa_mlp_set_groups(int32_t groups) {
    mlp_groups_switch();
    return g_mlp_groups.exchange(groups >= 2 ? 2 : 1);
}

int qa_mlp_groups(const qa_mlp_op *ops, int32_t num_ops, int32_t x_cols, int32_t *strand_of, int32_t *base, int32_t *stride, int32_t *lds_floats) {
    int rc = mlp_check(ops, num_ops, "qa_mlp_groups");
    if (rc != QA_OK) return rc;
    if (!strand_of || !base || !stride || !lds_floats || x_cols <= 0 || x_cols > QA_MLP_BUF0_COLS) { snprintf(g_perr, sizeof(g_perr), "qa_mlp_groups: bad argument"); return QA_E_ARG; }
    MlpArgs a{};
    if (mlp_strands(ops, num_ops, 2, strand_of) != 2) return 0;
    const bool fits = mlp_group_geometry(ops, num_ops, strand_of, x_cols, a);
    for (int g = 0; g < 2; ++g) for (int b = 0; b < MLP_NBUF; ++b) { base[g * MLP_NBUF + b] = a.g_base[g][b]; stride[g * MLP_NBUF + b] = a.g_stride[g][b]; }
    *lds_floats = a.g_lds_floats;
    return fits ? 1 : 0;
}

int qa_mlp_pack(const qa_mlp_op *ops, int32_t num_ops, const float *const *weights, const float *const *biases, float *packed, int64_t packed_floats,
                void *stream) {
    int rc = mlp_check(ops, num_ops, "qa_mlp_pack");
    if (rc != QA_OK) return rc;
    if (!weights || !biases || !packed || packed_floats < qa_mlp_packed_floats(ops, num_ops)) { snprintf(g_perr, sizeof(g_perr), "qa_mlp_pack: bad argument"); return QA_E_ARG; }
    PackArgs a{};
    a.packed = packed;
    for (int i = 0; i < num_ops; ++i) {
        if (ops[i].kind != QA_MLP_LAYER) continue;
        if (!weights[i]) { snprintf(g_perr, sizeof(g_perr), "qa_mlp_pack: op %d has no weight", i); return QA_E_ARG; }
        a.l[a.num++] = PackLayer{weights[i], biases[i], ops[i].n, ops[i].k, ops[i].w_off, ops[i].b_off, (ops[i].flags & QA_MLP_F_TRANSPOSED) ? 1 : 0, 0};
    }
    if (a.num == 0) return QA_OK;
    hipLaunchKernelGGL(qa_mlp_pack_kernel, dim3(64, a.num), dim3(256), 0, (hipStream_t)stream, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { snprintf(g_perr, sizeof(g_perr), "qa_mlp_pack: %s", hipGetErrorString(e)); return QA_E_DEVICE; }
    return QA_OK;
}

int qa_mlp_forward(const float *x, int64_t x_stride, int32_t rows, int32_t x_cols, const qa_mlp_op *ops, int32_t num_ops, const float *packed,
                   float *const *outs, const int64_t *out_strides, int32_t num_outs, void *stream) {
    int rc = mlp_check(ops, num_ops, "qa_mlp_forward");
    if (rc != QA_OK) return rc;
    if (!x || !packed || rows <= 0 || x_cols <= 0 || x_cols > QA_MLP_BUF0_COLS || x_stride < x_cols || num_outs < 0 || num_outs > QA_MLP_MAX_OUTPUTS ||
        (num_outs > 0 && (!outs || !out_strides))) { snprintf(g_perr, sizeof(g_perr), "qa_mlp_forward: bad argument"); return QA_E_ARG; }
    MlpArgs a{};
    a.x = x; a.x_stride = x_stride; a.rows = rows; a.x_cols = x_cols; a.num_ops = num_ops; a.packed = packed;
    for (int i = 0; i < num_outs; ++i) { a.out[i] = outs[i]; a.out_stride[i] = out_strides[i]; }
    for (int i = 0; i < num_ops; ++i) {
        const qa_mlp_op &o = ops[i];
        MlpDevOp &d = a.ops[i];
        d.kind = o.kind; d.src_buf = o.src_buf; d.src_col = o.src_col; d.dst_buf = o.dst_buf; d.dst_col = o.dst_col; d.k = o.k; d.n = o.n; d.act = o.act;
        d.out_index = o.out_index; d.w_off = o.w_off; d.b_off = o.b_off;
        d.flags = o.flags; d.out_col = o.out_col; d.aux_index = o.aux_index; d.aux_col = o.aux_col;
        if (o.kind == QA_MLP_LAYER) { d.nt = (o.n + 15) / 16; d.kb = mlp_kb(o.k, o.n); d.tpw = mlp_tpw(d.nt); }
        if ((o.kind == QA_MLP_LAYER && o.dst_buf < 0) || (o.flags & QA_MLP_F_SAVE)) {
            if (o.out_index >= num_outs || !outs[o.out_index] || out_strides[o.out_index] < o.out_col + o.n) {
                snprintf(g_perr, sizeof(g_perr), "qa_mlp_forward: op %d writes output %d which is missing or too narrow", i, o.out_index); return QA_E_ARG; }
        }
        if (((o.kind == QA_MLP_LAYER || o.kind == QA_MLP_GRAD) && o.act >= 4) || o.kind == QA_MLP_LOAD) {
            if (o.aux_index >= num_outs || !outs[o.aux_index] || out_strides[o.aux_index] < o.aux_col + o.n) {
                snprintf(g_perr, sizeof(g_perr), "qa_mlp_forward: op %d reads the saved activation %d which is missing or too narrow", i, o.aux_index); return QA_E_ARG; }
        }
    }
    const int tiles = (rows + MLP_ROWS - 1) / MLP_ROWS;
    static const int strands_switch = [] { const char *e = getenv("QA_MLP_STRANDS"); return e ? atoi(e) : 4; }();      /* 1: never split (A/B runs) */
    int max_strands = 256 / tiles;                  /* one workgroup per CU (117 KB of LDS each): split only into CUs that would idle */
    if (max_strands > strands_switch) max_strands = strands_switch;
    if (max_strands > 4) max_strands = 4;
    int32_t strand_of[QA_MLP_MAX_OPS];
    /* a tile per CU or more: nothing to split over CUs -- two strands side by side inside every workgroup instead (qa_mlp_forward_groups_kernel),
     * when the chain has two and their buffers fit the CU's LDS; QA_MLP_GROUPS=1 in the environment / qa_mlp_set_groups(1) keeps the one-group kernel (A/B runs) */
    bool ext = false;           /* any ABI 17 feature in the program: the kernel variant that has them (the two-group launch does not) */
    for (int i = 0; i < num_ops; ++i) ext |= ops[i].flags != 0 || ops[i].kind == QA_MLP_GRAD || ops[i].kind == QA_MLP_LOAD || ops[i].act >= 4 || ops[i].out_col != 0;
    if (!ext && max_strands < 2 && mlp_groups_switch() >= 2 && mlp_strands(ops, num_ops, 2, strand_of) == 2 && mlp_group_geometry(ops, num_ops, strand_of, x_cols, a)) {
        static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void *>(qa_mlp_forward_groups_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                            MLP_LDS_LIMIT_FLOATS * 4);
        if (attr == hipSuccess) {
            a.strands = 2;
            for (int i = 0; i < num_ops; ++i) a.ops[i].strand = strand_of[i];
            hipLaunchKernelGGL(qa_mlp_forward_groups_kernel, dim3(tiles), dim3(MLP_WAVES * 64), (size_t)a.g_lds_floats * 4, (hipStream_t)stream, a);
            hipError_t e = hipGetLastError();
            if (e != hipSuccess) { snprintf(g_perr, sizeof(g_perr), "qa_mlp_forward (groups): %s", hipGetErrorString(e)); return QA_E_DEVICE; }
            return QA_OK;
        }
    }
    a.strands = mlp_strands(ops, num_ops, max_strands, strand_of);
    for (int i = 0; i < num_ops; ++i) a.ops[i].strand = strand_of[i];
    if (ext) hipLaunchKernelGGL(qa_mlp_forward_kernel<true>, dim3(tiles, a.strands), dim3(MLP_WAVES * 64), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(qa_mlp_forward_kernel<false>, dim3(tiles, a.strands), dim3(MLP_WAVES * 64), 0, (hipStream_t)stream, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { snprintf(g_perr, sizeof(g_perr), "qa_mlp_forward: %s", hipGetErrorString(e)); return QA_E_DEVICE; }
    return QA_OK;
}

}  // extern "C"
