// qa_conv.hip -- the image stem of the vision student's depth encoder as hand-written kernels for gfx950.
//
// What it replaces: the first block of `DepthOnlyFCBackbone58x87.image_compression`
// (tsc/rsl_rl/modules/depth_backbone.py:63-75: Conv2d(1, 32, 5) -> MaxPool2d(2, 2) -> ELU -> Conv2d(32, 64, 3) -> ELU) under training,
// forward and backward, for the three passes of an iteration that run it (the 24 per-step forwards of `learn_vision`,
// on_policy_runner.py:278-441; their backward in `update_depth_actor`, algorithms/ppo.py:327-358; BYOL's two augmented views through the
// online and target encoders, modules/byol.py:242-317).  In PyTorch on ROCm these are MIOpen convolutions whose kernel choice is made by a
// per-process search (a fresh machine measured 200-860 ms per iteration depending on what the search picked); here
//   stem forward    conv 5x5 (1 -> 32) + max-pool 2x2 + ELU in ONE launch over the raw image, output channels-last [img][27][41][32]
//                   (+ the pool's argmax as one byte per output) -- VALU: K = 25 is no MFMA shape and the image sits in LDS
//   second conv     qa_conv_nhwc_forward / _backward_input / _backward_weight (qa_gemm.hip): fp32-MFMA GEMMs that read the 3x3 windows in
//                   place from the channels-last tensor (no im2col buffer), bias + ELU / ELU' in the epilogue, bias gradient on the MFMAs
//   elu-backward    qa_elu_backward_pad: g * ELU'(y) written twice -- plain (the weight-gradient operand) and with a zero border of
//                   kernel-1 pixels (so that the input gradient is a VALID correlation with the flipped kernel: same GEMM, no masks)
//   stem backward   weight / bias gradient of the 5x5 convolution through the pool's argmax, fixed-order slabs
// fp32 throughout (the reference's dtype).  The pool keeps PyTorch's tie rule (first maximum in row-major window order).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/qa_sim.h"

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

extern thread_local char qa_err_buf[512];
#define g_cerr qa_err_buf

namespace {

constexpr int STEM_C = 32;          // output channels = half a wavefront: one pooled pixel per 32 lanes, 128-byte channels-last stores
constexpr int STEM_K = 5;
constexpr int STEM_BAND = 9;        // pooled rows per workgroup
constexpr int STEM_ROWS = 2 * STEM_BAND + STEM_K - 1;      // image rows a band reads (22)
constexpr int STEM_MAXW = 128;
constexpr int STEM_WB = STEM_C * STEM_K * STEM_K + STEM_C;      // weight + bias gradient of the stem: 832 floats per slab
constexpr int STEM_BWD_GRID = 1024;     // 4 workgroups per CU (38 KB of LDS each): the pixel loop is two dependent global loads long

struct StemArgs {
    const float *img;        // [n][ih][iw]
    const float *w, *b;      // [32][25], [32]
    float *y;                // [n][ph][pw][32]
    uint8_t *amax;           // [n][ph][pw][32]: 2 * dy + dx of the window's maximum
    const float *gpre;       // backward: gradient at the pooled pre-activation, [n][ph][pw][32]
    float *slabs;            // backward: [grid][832]
    int n, ih, iw, ph, pw, nbands;
    float alpha;
};

// image rows [r0, r0 + STEM_ROWS) of image `im` -> LDS with a row stride of ldw (even: the 2 px-aligned 6-float patches are read as b64)
static __device__ __forceinline__ void stem_stage(const StemArgs &a, float *S, int ldw, int im, int r0, int tid) {
    const float *src = a.img + (int64_t)im * a.ih * a.iw;
    for (int i = tid; i < STEM_ROWS * a.iw; i += 256) {
        const int r = i / a.iw, c = i - r * a.iw;
        S[r * ldw + c] = (r0 + r < a.ih) ? src[(r0 + r) * a.iw + c] : 0.f;
    }
}

__global__ void __launch_bounds__(256) qa_depth_stem_forward_kernel(StemArgs a) {
    __shared__ __attribute__((aligned(16))) float S[STEM_ROWS * STEM_MAXW];
    const int tid = threadIdx.x, c = tid & 31, slot = tid >> 5;
    const int im = blockIdx.x / a.nbands, band = blockIdx.x - im * a.nbands;
    const int ldw = (a.iw + 1) & ~1;
    float w[STEM_K * STEM_K];
#pragma unroll
    for (int t = 0; t < STEM_K * STEM_K; ++t) w[t] = a.w[c * STEM_K * STEM_K + t];
    const float bias = a.b[c];
    stem_stage(a, S, ldw, im, 2 * STEM_BAND * band, tid);
    __syncthreads();
    const int rows = min(STEM_BAND, a.ph - band * STEM_BAND), items = rows * a.pw;
    for (int q = slot; q < items; q += 8) {
        const int py = q / a.pw, px = q - py * a.pw;
        const float *p = S + (2 * py) * ldw + 2 * px;
        float o00 = 0.f, o01 = 0.f, o10 = 0.f, o11 = 0.f;
#pragma unroll
        for (int r = 0; r < STEM_K + 1; ++r) {
            const f2 v0 = *(const f2 *)(p + r * ldw), v1 = *(const f2 *)(p + r * ldw + 2), v2 = *(const f2 *)(p + r * ldw + 4);
            const float v[6] = {v0.x, v0.y, v1.x, v1.y, v2.x, v2.y};
            if (r < STEM_K) {
#pragma unroll
                for (int kx = 0; kx < STEM_K; ++kx) {
                    o00 = fmaf(w[r * STEM_K + kx], v[kx], o00);
                    o01 = fmaf(w[r * STEM_K + kx], v[kx + 1], o01);
                }
            }
            if (r > 0) {
#pragma unroll
                for (int kx = 0; kx < STEM_K; ++kx) {
                    o10 = fmaf(w[(r - 1) * STEM_K + kx], v[kx], o10);
                    o11 = fmaf(w[(r - 1) * STEM_K + kx], v[kx + 1], o11);
                }
            }
        }
        float best = o00 + bias; int arg = 0;
        { const float v = o01 + bias; if (v > best) { best = v; arg = 1; } }
        { const float v = o10 + bias; if (v > best) { best = v; arg = 2; } }
        { const float v = o11 + bias; if (v > best) { best = v; arg = 3; } }
        const float y = best > 0.f ? best : a.alpha * (expf(best) - 1.f);
        const int64_t o = (((int64_t)im * a.ph + band * STEM_BAND + py) * a.pw + px) * STEM_C + c;
        a.y[o] = y;
        a.amax[o] = (uint8_t)arg;
    }
}

// dW[c][ky][kx] = sum over images and pooled pixels of gpre * image[2 py + dy + ky][2 px + dx + kx] with (dy, dx) the pool's argmax;
// db[c] = sum of gpre.  One slab of 832 partial sums per workgroup (fixed work assignment -> reproducible), summed in a fixed order.
__global__ void __launch_bounds__(256) qa_depth_stem_backward_kernel(StemArgs a) {
    __shared__ __attribute__((aligned(16))) float S[STEM_ROWS * STEM_MAXW];
    __shared__ float R[8][STEM_C][STEM_K * STEM_K + 1];
    const int tid = threadIdx.x, c = tid & 31, slot = tid >> 5;
    const int ldw = (a.iw + 1) & ~1;
    float acc[STEM_K * STEM_K], accb = 0.f;
#pragma unroll
    for (int t = 0; t < STEM_K * STEM_K; ++t) acc[t] = 0.f;
    const int total = a.n * a.nbands;
    for (int item = blockIdx.x; item < total; item += gridDim.x) {
        const int im = item / a.nbands, band = item - im * a.nbands;
        __syncthreads();
        stem_stage(a, S, ldw, im, 2 * STEM_BAND * band, tid);
        __syncthreads();
        const int rows = min(STEM_BAND, a.ph - band * STEM_BAND), items = rows * a.pw;
        // the band's pixels are consecutive in memory: pixel q of the band is element (base + q) * 32 + c.  The next pixel's gradient and argmax
        // are loaded before this pixel's 25 LDS reads + FMAs (one wave per SIMD per workgroup: nothing else hides the load)
        const int64_t base = ((int64_t)im * a.ph + band * STEM_BAND) * a.pw;
        float g = 0.f; int arg = 0;
        if (slot < items) { g = a.gpre[(base + slot) * STEM_C + c]; arg = a.amax[(base + slot) * STEM_C + c]; }
        for (int q = slot; q < items; q += 8) {
            const int qn = min(q + 8, items - 1);
            const float gn = a.gpre[(base + qn) * STEM_C + c];
            const int argn = a.amax[(base + qn) * STEM_C + c];
            const int py = q / a.pw, px = q - py * a.pw;
            const float *p = S + (2 * py + (arg >> 1)) * ldw + 2 * px + (arg & 1);
            accb += g;
#pragma unroll
            for (int ky = 0; ky < STEM_K; ++ky)
#pragma unroll
                for (int kx = 0; kx < STEM_K; ++kx) acc[ky * STEM_K + kx] = fmaf(g, p[ky * ldw + kx], acc[ky * STEM_K + kx]);
            g = gn; arg = argn;
        }
    }
#pragma unroll
    for (int t = 0; t < STEM_K * STEM_K; ++t) R[slot][c][t] = acc[t];
    R[slot][c][STEM_K * STEM_K] = accb;
    __syncthreads();
    float *slab = a.slabs + (int64_t)blockIdx.x * STEM_WB;
    for (int i = tid; i < STEM_WB; i += 256) {
        const int cc = i < STEM_C * STEM_K * STEM_K ? i / (STEM_K * STEM_K) : i - STEM_C * STEM_K * STEM_K;
        const int t = i < STEM_C * STEM_K * STEM_K ? i - cc * STEM_K * STEM_K : STEM_K * STEM_K;
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) s += R[k][cc][t];
        slab[i] = s;
    }
}

// out[i] = sum of slabs[z][i] over z: each output by 8 threads that sum an eighth of the slabs each (8 loads in flight), combined in a fixed
// order -- one thread per output over 1,024 slabs would be 1,024 dependent round trips
__global__ void __launch_bounds__(256) qa_stem_slab_reduce_kernel(const float *__restrict__ slabs, int nslab, float *__restrict__ out) {
    __shared__ float part[8][32];
    const int tid = threadIdx.x, ol = tid & 31, ch = tid >> 5, o = blockIdx.x * 32 + ol;
    const int per = (nslab + 7) / 8, z0 = ch * per, z1 = min(nslab, z0 + per);
    float s = 0.f;
    if (o < STEM_WB) {
        int z = z0;
        for (; z + 8 <= z1; z += 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = slabs[(int64_t)(z + u) * STEM_WB + o];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; z < z1; ++z) s += slabs[(int64_t)z * STEM_WB + o];
    }
    part[ch][ol] = s;
    __syncthreads();
    if (ch == 0 && o < STEM_WB) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) t += part[k][ol];
        out[o] = t;
    }
}

struct PadArgs {
    const float *g, *y;      // [n][oh][ow][c]
    float *dy, *dyp;         // [n][oh][ow][c], [n][oh + 2 pad][ow + 2 pad][c]
    int64_t total;           // n * (oh + 2 pad) * (ow + 2 pad) * c / 4
    int oh, ow, c4, pad, act;
    float alpha;
};

__global__ void __launch_bounds__(256) qa_elu_backward_pad_kernel(PadArgs a) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= a.total) return;
    const int pw = a.ow + 2 * a.pad, ph = a.oh + 2 * a.pad;
    const int64_t pix = t / a.c4;
    const int ch = (int)(t - pix * a.c4);
    const int64_t im = pix / (ph * pw);
    const int r = (int)(pix - im * (ph * pw)), yy = r / pw - a.pad, xx = r - (r / pw) * pw - a.pad;
    f4 v = {0.f, 0.f, 0.f, 0.f};
    if (yy >= 0 && yy < a.oh && xx >= 0 && xx < a.ow) {
        const int64_t o = (((im * a.oh + yy) * a.ow + xx) * a.c4 + ch) * 4;
        const f4 g = *(const f4 *)(a.g + o), y = *(const f4 *)(a.y + o);
#pragma unroll
        for (int k = 0; k < 4; ++k)
            v[k] = g[k] * (a.act == 1 ? (y[k] > 0.f ? 1.f : y[k] + a.alpha) : (a.act == 2 ? (y[k] > 0.f ? 1.f : 0.f) : 1.f));
        *(f4 *)(a.dy + o) = v;
    }
    *(f4 *)(a.dyp + t * 4) = v;
}

static inline bool aligned16(const void *p) { return ((uintptr_t)p & 15) == 0; }

static bool stem_args(StemArgs *a, int64_t n_img, int ih, int iw) {
    if (n_img <= 0 || n_img > (1 << 24) || ih < STEM_K + 1 || iw < STEM_K + 1 || iw > STEM_MAXW - 2) return false;
    a->n = (int)n_img; a->ih = ih; a->iw = iw; a->ph = (ih - STEM_K + 1) / 2; a->pw = (iw - STEM_K + 1) / 2;
    a->nbands = (a->ph + STEM_BAND - 1) / STEM_BAND;
    return true;
}

}  // namespace

extern "C" {

int qa_depth_stem_forward(const float *images, const float *weight, const float *bias, float *y, uint8_t *argmax, int64_t n_img, int32_t ih, int32_t iw,
                          float alpha, void *stream) {
    StemArgs a = {};
    if (!images || !weight || !bias || !y || !argmax || !stem_args(&a, n_img, ih, iw)) {
        snprintf(g_cerr, sizeof(g_cerr), "qa_depth_stem_forward: bad argument (images [n][ih][iw] fp32, 6 <= ih, 6 <= iw <= %d)", STEM_MAXW - 2); return QA_E_ARG; }
    a.img = images; a.w = weight; a.b = bias; a.y = y; a.amax = argmax; a.alpha = alpha;
    hipLaunchKernelGGL(qa_depth_stem_forward_kernel, dim3((unsigned)(a.n * a.nbands)), dim3(256), 0, (hipStream_t)stream, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { snprintf(g_cerr, sizeof(g_cerr), "qa_depth_stem_forward: %s", hipGetErrorString(e)); return QA_E_DEVICE; }
    return QA_OK;
}

int64_t qa_depth_stem_backward_scratch_bytes(void) { return (int64_t)STEM_BWD_GRID * STEM_WB * 4; }

int qa_depth_stem_backward(const float *images, const uint8_t *argmax, const float *grad_pre, float *grad_wb, int64_t n_img, int32_t ih, int32_t iw,
                           void *scratch, int64_t scratch_bytes, void *stream) {
    StemArgs a = {};
    if (!images || !argmax || !grad_pre || !grad_wb || !scratch || !stem_args(&a, n_img, ih, iw) || scratch_bytes < qa_depth_stem_backward_scratch_bytes() ||
        !aligned16(scratch) || !aligned16(grad_wb)) { snprintf(g_cerr, sizeof(g_cerr), "qa_depth_stem_backward: bad argument"); return QA_E_ARG; }
    a.img = images; a.amax = (uint8_t *)argmax; a.gpre = grad_pre; a.slabs = (float *)scratch;
    const int grid = (int)((int64_t)a.n * a.nbands < STEM_BWD_GRID ? (int64_t)a.n * a.nbands : STEM_BWD_GRID);
    hipLaunchKernelGGL(qa_depth_stem_backward_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { snprintf(g_cerr, sizeof(g_cerr), "qa_depth_stem_backward: %s", hipGetErrorString(e)); return QA_E_DEVICE; }
    hipLaunchKernelGGL(qa_stem_slab_reduce_kernel, dim3((STEM_WB + 31) / 32), dim3(256), 0, (hipStream_t)stream, (const float *)scratch, grid, grad_wb);
    e = hipGetLastError();
    if (e != hipSuccess) { snprintf(g_cerr, sizeof(g_cerr), "qa_depth_stem_backward: %s", hipGetErrorString(e)); return QA_E_DEVICE; }
    return QA_OK;
}

int qa_elu_backward_pad(const float *grad_out, const float *y, float *grad_pre, float *grad_pre_padded, int64_t n_img, int32_t oh, int32_t ow,
                        int32_t channels, int32_t pad, int32_t act, float alpha, void *stream) {
    if (!grad_out || !y || !grad_pre || !grad_pre_padded || n_img <= 0 || oh <= 0 || ow <= 0 || channels <= 0 || channels % 4 || pad < 0 || act < 0 || act > 2 ||
        !aligned16(grad_out) || !aligned16(y) || !aligned16(grad_pre) || !aligned16(grad_pre_padded)) {
        snprintf(g_cerr, sizeof(g_cerr), "qa_elu_backward_pad: bad argument"); return QA_E_ARG; }
    PadArgs a = {};
    a.g = grad_out; a.y = y; a.dy = grad_pre; a.dyp = grad_pre_padded; a.oh = oh; a.ow = ow; a.c4 = channels / 4; a.pad = pad; a.act = act; a.alpha = alpha;
    a.total = n_img * (int64_t)(oh + 2 * pad) * (ow + 2 * pad) * a.c4;
    hipLaunchKernelGGL(qa_elu_backward_pad_kernel, dim3((unsigned)((a.total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { snprintf(g_cerr, sizeof(g_cerr), "qa_elu_backward_pad: %s", hipGetErrorString(e)); return QA_E_DEVICE; }
    return QA_OK;
}

}  // extern "C"
