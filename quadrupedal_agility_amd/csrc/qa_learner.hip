// Fused learner-side kernels (include/qa_sim.h, "learner kernels").
//
// qa_ppo_loss: the PPO minibatch objective of SSInfoGAIL.update_actor_critic (bbc/rsl_rl/algorithms/gail.py:328-413)
// -- Gaussian log-prob, ratio, clipped surrogate, clipped value loss, action-bound loss, entropy, the KL estimate of
// the adaptive LR schedule -- and its gradient with respect to (mean, std, value), evaluated in ONE pass over the
// minibatch.  All terms are means over the batch, so every derivative is known locally (x 1/B) and no second pass
// is needed.  In eager PyTorch the same arithmetic is ~150 elementwise/reduction launches over (B,12) tensors per
// minibatch (3,000 per PPO iteration at 5 epochs x 4 minibatches).
//
// Layout: one lane per sample; a sample's 12 means/actions/old means/old sigmas are 48 contiguous bytes = 3 x 16-B
// loads.  Batch sums go through a wavefront DPP/shuffle reduction, then one double atomicAdd per wave and term
// (double so that the summation order of the atomics cannot be seen in the fp32 results).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/qa_sim.h"

namespace {

constexpr int PPO_BLOCK = 256;
constexpr int PPO_D = 12;           // actions per sample (num_actions of the Go2 task)
constexpr int PPO_SUMS = 5 + PPO_D; // surrogate, value, bound, entropy, kl, dstd[12]

struct PpoArgs {
    const float *mu, *std, *value, *actions, *old_logp, *old_mu, *old_sigma, *adv, *returns, *target_values;
    float *dmu, *dvalue;
    double *sums;
    int64_t B;
    float clip, c_surr, c_value, c_bound, c_entropy;
    int clipped_value;
};

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__global__ void __launch_bounds__(PPO_BLOCK) qa_ppo_loss_kernel(PpoArgs a) {
    const int64_t i = (int64_t)blockIdx.x * PPO_BLOCK + threadIdx.x;
    const bool live = i < a.B;
    const int64_t r = live ? i : a.B - 1;
    const float invB = 1.0f / (float)a.B;
    float sd[PPO_D];
#pragma unroll
    for (int j = 0; j < PPO_D; ++j) sd[j] = a.std[j];
    float mu[PPO_D], ac[PPO_D], omu[PPO_D], osd[PPO_D];
    {
        const float4 *m4 = (const float4 *)(a.mu + r * PPO_D), *a4 = (const float4 *)(a.actions + r * PPO_D);
        const float4 *om4 = (const float4 *)(a.old_mu + r * PPO_D), *os4 = (const float4 *)(a.old_sigma + r * PPO_D);
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            float4 m = m4[q], x = a4[q], om = om4[q], os = os4[q];
            mu[4 * q] = m.x; mu[4 * q + 1] = m.y; mu[4 * q + 2] = m.z; mu[4 * q + 3] = m.w;
            ac[4 * q] = x.x; ac[4 * q + 1] = x.y; ac[4 * q + 2] = x.z; ac[4 * q + 3] = x.w;
            omu[4 * q] = om.x; omu[4 * q + 1] = om.y; omu[4 * q + 2] = om.z; omu[4 * q + 3] = om.w;
            osd[4 * q] = os.x; osd[4 * q + 1] = os.y; osd[4 * q + 2] = os.z; osd[4 * q + 3] = os.w;
        }
    }
    const float v = a.value[r], adv = a.adv[r], ret = a.returns[r], tv = a.target_values[r], ologp = a.old_logp[r];

    // Normal(mu, std): log-prob, entropy, KL(old || new) estimate of gail.py:366-369
    const float HALF_LOG_2PI = 0.91893853320467274178f;
    float logp = 0.f, ent = 0.f, kl = 0.f, bl = 0.f;
    float dlogp_dmu[PPO_D], dlogp_dsd[PPO_D], dbl_dmu[PPO_D];
#pragma unroll
    for (int j = 0; j < PPO_D; ++j) {
        const float s = sd[j], inv_s = 1.0f / s, d = ac[j] - mu[j], var = s * s, ls = logf(s);
        logp += -(d * d) / (2.0f * var) - ls - HALF_LOG_2PI;
        ent += 0.5f + HALF_LOG_2PI + ls;
        const float dm = omu[j] - mu[j];
        kl += logf(s / osd[j] + 1.0e-5f) + (osd[j] * osd[j] + dm * dm) / (2.0f * var) - 0.5f;
        dlogp_dmu[j] = d / var;
        dlogp_dsd[j] = d * d * inv_s * inv_s * inv_s - inv_s;
        const float lo = fminf(mu[j] + 1.0f, 0.0f), hi = fmaxf(mu[j] - 1.0f, 0.0f);
        bl += lo * lo + hi * hi;
        dbl_dmu[j] = 2.0f * lo + 2.0f * hi;
    }
    // clipped surrogate: max(-A r, -A clamp(r)); inside the clip range both arguments coincide and the gradient is -A
    const float ratio = expf(logp - ologp);
    const float rc = fminf(fmaxf(ratio, 1.0f - a.clip), 1.0f + a.clip);
    const float s1 = -adv * ratio, s2 = -adv * rc;
    const float surr = fmaxf(s1, s2);
    const bool inside = ratio >= 1.0f - a.clip && ratio <= 1.0f + a.clip;
    float dsurr_dratio = s1 > s2 ? -adv : (s1 == s2 ? (inside ? -adv : -0.5f * adv) : 0.0f);
    const float dsurr_dlogp = dsurr_dratio * ratio;
    // value loss
    float vl, dvl_dv;
    if (a.clipped_value) {
        const float dv = v - tv, dvc = fminf(fmaxf(dv, -a.clip), a.clip), vc = tv + dvc;
        const float l1 = (v - ret) * (v - ret), l2 = (vc - ret) * (vc - ret);
        const float pass = (dv >= -a.clip && dv <= a.clip) ? 1.0f : 0.0f;
        vl = fmaxf(l1, l2);
        dvl_dv = l1 > l2 ? 2.0f * (v - ret) : (l1 == l2 ? (v - ret) + (vc - ret) * pass : 2.0f * (vc - ret) * pass);
    } else {
        vl = (ret - v) * (ret - v);
        dvl_dv = 2.0f * (v - ret);
    }
    // gradients of  c_surr mean(surr) + c_value mean(vl) + c_bound mean(bl) - c_entropy mean(ent)
    if (live) {
        float g[PPO_D];
#pragma unroll
        for (int j = 0; j < PPO_D; ++j) g[j] = invB * (a.c_surr * dsurr_dlogp * dlogp_dmu[j] + a.c_bound * dbl_dmu[j]);
        float4 *o4 = (float4 *)(a.dmu + r * PPO_D);
        o4[0] = make_float4(g[0], g[1], g[2], g[3]); o4[1] = make_float4(g[4], g[5], g[6], g[7]); o4[2] = make_float4(g[8], g[9], g[10], g[11]);
        a.dvalue[r] = invB * a.c_value * dvl_dv;
    }
    const float w = live ? 1.0f : 0.0f;
    float part[PPO_SUMS];
    part[0] = w * surr; part[1] = w * vl; part[2] = w * bl; part[3] = w * ent; part[4] = w * kl;
#pragma unroll
    for (int j = 0; j < PPO_D; ++j) part[5 + j] = w * (a.c_surr * dsurr_dlogp * dlogp_dsd[j] - a.c_entropy / sd[j]);
#pragma unroll
    for (int k = 0; k < PPO_SUMS; ++k) {
        float s = wave_sum(part[k]);
        if ((threadIdx.x & 63) == 0) atomicAdd(&a.sums[k], (double)s);
    }
}

// sums -> out[8] = {loss, surrogate, value, bound, entropy, kl, 0, 0} (means) and dstd[12]
__global__ void qa_ppo_finish_kernel(const double *sums, int64_t B, float c_surr, float c_value, float c_bound, float c_entropy,
                                     float *out, float *dstd) {
    const int t = threadIdx.x;
    const double invB = 1.0 / (double)B;
    if (t < 5) out[1 + t] = (float)(sums[t] * invB);
    if (t == 5) out[0] = (float)((c_surr * sums[0] + c_value * sums[1] + c_bound * sums[2] - c_entropy * sums[3]) * invB);
    if (t == 6 || t == 7) out[t] = 0.f;
    if (t >= 8 && t < 8 + PPO_D) dstd[t - 8] = (float)(sums[5 + t - 8] * invB);
}

}  // namespace

extern thread_local char qa_err_buf[512];
#define g_lerr qa_err_buf

extern "C" {

int qa_ppo_loss(const float *mu, const float *std, const float *value, const float *actions, const float *old_logp,
                const float *old_mu, const float *old_sigma, const float *advantages, const float *returns,
                const float *target_values, int64_t B, int32_t num_actions, float clip, float c_surr, float c_value,
                float c_bound, float c_entropy, int32_t clipped_value, float *dmu, float *dstd, float *dvalue, float *out,
                void *scratch, void *stream) {
    if (!mu || !std || !value || !actions || !old_logp || !old_mu || !old_sigma || !advantages || !returns || !target_values ||
        !dmu || !dstd || !dvalue || !out || !scratch || B <= 0) { snprintf(g_lerr, sizeof(g_lerr), "qa_ppo_loss: null pointer or empty batch"); return QA_E_ARG; }
    if (num_actions != PPO_D) { snprintf(g_lerr, sizeof(g_lerr), "qa_ppo_loss: num_actions must be %d", PPO_D); return QA_E_ARG; }
    if ((((uintptr_t)mu | (uintptr_t)actions | (uintptr_t)old_mu | (uintptr_t)old_sigma | (uintptr_t)dmu) & 15) != 0) {
        snprintf(g_lerr, sizeof(g_lerr), "qa_ppo_loss: (B,12) tensors must be 16-byte aligned"); return QA_E_ARG; }
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(scratch, 0, sizeof(double) * PPO_SUMS, st);
    if (e != hipSuccess) { snprintf(g_lerr, sizeof(g_lerr), "qa_ppo_loss: %s", hipGetErrorString(e)); return QA_E_DEVICE; }
    PpoArgs a{mu, std, value, actions, old_logp, old_mu, old_sigma, advantages, returns, target_values, dmu, dvalue, (double *)scratch,
              B, clip, c_surr, c_value, c_bound, c_entropy, clipped_value};
    const int blocks = (int)((B + PPO_BLOCK - 1) / PPO_BLOCK);
    hipLaunchKernelGGL(qa_ppo_loss_kernel, dim3(blocks), dim3(PPO_BLOCK), 0, st, a);
    hipLaunchKernelGGL(qa_ppo_finish_kernel, dim3(1), dim3(64), 0, st, (const double *)scratch, B, c_surr, c_value, c_bound, c_entropy, out, dstd);
    e = hipGetLastError();
    if (e != hipSuccess) { snprintf(g_lerr, sizeof(g_lerr), "qa_ppo_loss: %s", hipGetErrorString(e)); return QA_E_DEVICE; }
    return QA_OK;
}

}  // extern "C"
