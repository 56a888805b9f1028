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
#include "qa_device.h"

namespace {

constexpr int PPO_BLOCK = 64;            // one wavefront per workgroup: its 17 partial sums go to scratch without atomics
constexpr int PPO_D = 12;           // actions per sample (num_actions of the Go2 task)
constexpr int PPO_SUMS = 5 + PPO_D; // surrogate, value, bound, entropy, kl, dstd[12]

struct PpoArgs {
    const float *mu, *std, *value, *actions, *old_logp, *old_mu, *old_sigma, *adv, *returns, *target_values;
    float *dmu, *dvalue;
    float *partial;          // (gridDim.x, PPO_SUMS)
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
        if (threadIdx.x == k) a.partial[(int64_t)blockIdx.x * PPO_SUMS + k] = s;      // lane k keeps sum k
    }
}

// per-wave partial sums -> out[8] = {loss, surrogate, value, bound, entropy, kl, 0, 0} (means) and dstd[12];
// fixed summation order (double), so results do not depend on scheduling
__global__ void __launch_bounds__(256) qa_ppo_finish_kernel(const float *partial, int nblocks, int64_t B, float c_surr, float c_value,
                                                            float c_bound, float c_entropy, float *out, float *dstd) {
    /* all PPO_SUMS sums in one pass: thread = (sum k, lane l of LANES); a lane adds every LANES-th block partial in double,
     * then one thread per sum adds the lanes in a fixed order (was: one full tree reduction per sum, 17 in sequence) */
    constexpr int LANES = 256 / PPO_SUMS;
    __shared__ double s_acc[256];
    __shared__ double s_tot[PPO_SUMS];
    const int t = threadIdx.x;
    const int k = t / LANES, l = t % LANES;
    double acc = 0.0;
    if (k < PPO_SUMS)
        for (int b = l; b < nblocks; b += LANES) acc += (double)partial[(int64_t)b * PPO_SUMS + k];
    s_acc[t] = acc;
    __syncthreads();
    if (k < PPO_SUMS && l == 0) {
        double tot = 0.0;
        for (int q = 0; q < LANES; ++q) tot += s_acc[k * LANES + q];
        s_tot[k] = tot;
    }
    __syncthreads();
    const double invB = 1.0 / (double)B;
    if (t < 5) out[1 + t] = (float)(s_tot[t] * invB);
    if (t == 5) out[0] = (float)((c_surr * s_tot[0] + c_value * s_tot[1] + c_bound * s_tot[2] - c_entropy * s_tot[3]) * invB);
    if (t == 6 || t == 7) out[t] = 0.f;
    if (t >= 8 && t < 8 + PPO_D) dstd[t - 8] = (float)(s_tot[5 + t - 8] * invB);
}

// ---------------------------------------------------------------------------------------------------------------
// qa_hybrid_ppo_loss: the task-level learner's objective (categorical gait head + Gaussian parameter head) and its gradient, one
// thread per sample, as qa_ppo_loss; in eager PyTorch it is ~100 elementwise launches forward and ~150 backward per minibatch.
constexpr int HYB_ND = 3, HYB_NC = 18, HYB_SUMS = 5 + HYB_NC;       // surr_d, surr_c, value, entropy, kl, dstd[18]
struct HybArgs {
    const float *logits, *mean, *std, *value, *actions, *old_logp_d, *old_logp_c, *old_mu, *old_sigma, *adv, *returns, *target_values;
    float *dlogits, *dmean, *dvalue, *partial;
    int64_t B;
    float clip, c_value, c_entropy;
    int clipped_value;
};
__device__ __forceinline__ void clipped_surrogate(float logp, float ologp, float adv, float clip, float &surr, float &dsurr_dlogp) {
    const float ratio = expf(logp - ologp), rc = fminf(fmaxf(ratio, 1.0f - clip), 1.0f + clip);
    const float s1 = -adv * ratio, s2 = -adv * rc;
    surr = fmaxf(s1, s2);
    const bool inside = ratio >= 1.0f - clip && ratio <= 1.0f + clip;
    const float d = s1 > s2 ? -adv : (s1 == s2 ? (inside ? -adv : -0.5f * adv) : 0.0f);
    dsurr_dlogp = d * ratio;
}
__global__ void __launch_bounds__(PPO_BLOCK) qa_hybrid_ppo_loss_kernel(HybArgs a) {
    const int64_t i = (int64_t)blockIdx.x * PPO_BLOCK + threadIdx.x;
    const bool live = i < a.B;
    const int64_t r = live ? i : a.B - 1;
    const float invB = 1.0f / (float)a.B;
    const float HALF_LOG_2PI = 0.91893853320467274178f, EPS = 1.1920928955078125e-07f;
    // ---- categorical head: torch.distributions.Categorical(probs = softmax(logits))
    float z[HYB_ND], pr[HYB_ND], lp[HYB_ND], fl[HYB_ND];
    float zmax = -3.0e38f;
#pragma unroll
    for (int k = 0; k < HYB_ND; ++k) { z[k] = a.logits[r * HYB_ND + k]; zmax = fmaxf(zmax, z[k]); }
    float den = 0.f;
#pragma unroll
    for (int k = 0; k < HYB_ND; ++k) { pr[k] = expf(z[k] - zmax); den += pr[k]; }
    float Hd = 0.f, sum_pf = 0.f;
#pragma unroll
    for (int k = 0; k < HYB_ND; ++k) {
        pr[k] /= den;
        fl[k] = (pr[k] > EPS && pr[k] < 1.0f - EPS) ? 1.0f : 0.0f;              // the clamp of probs_to_logits passes no gradient outside
        lp[k] = logf(fminf(fmaxf(pr[k], EPS), 1.0f - EPS));
        Hd -= pr[k] * lp[k];
        sum_pf += pr[k] * fl[k];
    }
    const float *act = a.actions + r * (1 + HYB_NC);
    int ad = (int)act[0];
    ad = ad < 0 ? 0 : (ad >= HYB_ND ? HYB_ND - 1 : ad);
    const float lpa = ad == 0 ? lp[0] : (ad == 1 ? lp[1] : lp[2]);
    const float fa = ad == 0 ? fl[0] : (ad == 1 ? fl[1] : fl[2]);
    const float adv = a.adv[r];
    float surr_d, dsd_dlogp;
    clipped_surrogate(lpa, a.old_logp_d[r], adv, a.clip, surr_d, dsd_dlogp);
    // ---- Gaussian head
    float logp = 0.f, ent_c = 0.f, kl = 0.f;
    float dlogp_dmu[HYB_NC], dlogp_dsd[HYB_NC], inv_sd[HYB_NC];
#pragma unroll
    for (int j = 0; j < HYB_NC; ++j) {
        const float s = a.std[j], inv_s = 1.0f / s, mu = a.mean[r * HYB_NC + j], d = act[1 + j] - mu, var = s * s, ls = logf(s);
        const float osd = a.old_sigma[r * HYB_NC + j], dm = a.old_mu[r * HYB_NC + j] - mu;
        logp += -(d * d) / (2.0f * var) - ls - HALF_LOG_2PI;
        ent_c += 0.5f + HALF_LOG_2PI + ls;
        kl += logf(s / osd + 1.0e-5f) + (osd * osd + dm * dm) / (2.0f * var) - 0.5f;
        dlogp_dmu[j] = d / var;
        dlogp_dsd[j] = d * d * inv_s * inv_s * inv_s - inv_s;
        inv_sd[j] = inv_s;
    }
    ent_c *= 1.0f / (float)HYB_NC;
    float surr_c, dsc_dlogp;
    clipped_surrogate(logp, a.old_logp_c[r], adv, a.clip, surr_c, dsc_dlogp);
    // ---- value loss
    const float v = a.value[r], ret = a.returns[r], tv = a.target_values[r];
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
    if (live) {
        // d/dz_k of  surr_d(logp_a) - c_entropy H(p):  d logp_a = f_a (delta_ak - p_k);  dH = -p_k lp_k - p_k H_...
        float plp = 0.f;
#pragma unroll
        for (int k = 0; k < HYB_ND; ++k) plp += pr[k] * lp[k];
#pragma unroll
        for (int k = 0; k < HYB_ND; ++k) {
            const float dlogpa = fa * ((k == ad ? 1.0f : 0.0f) - pr[k]);
            const float dH = -pr[k] * lp[k] + pr[k] * plp - pr[k] * fl[k] + pr[k] * sum_pf;
            a.dlogits[r * HYB_ND + k] = invB * (dsd_dlogp * dlogpa - a.c_entropy * dH);
        }
#pragma unroll
        for (int j = 0; j < HYB_NC; ++j) a.dmean[r * HYB_NC + j] = invB * dsc_dlogp * dlogp_dmu[j];
        a.dvalue[r] = invB * a.c_value * dvl_dv;
    }
    const float w = live ? 1.0f : 0.0f;
    float part[HYB_SUMS];
    part[0] = w * surr_d; part[1] = w * surr_c; part[2] = w * vl; part[3] = w * (Hd + ent_c); part[4] = w * kl;
#pragma unroll
    for (int j = 0; j < HYB_NC; ++j) part[5 + j] = w * (dsc_dlogp * dlogp_dsd[j] - a.c_entropy * inv_sd[j] * (1.0f / (float)HYB_NC));
#pragma unroll
    for (int k = 0; k < HYB_SUMS; ++k) {
        float s = wave_sum(part[k]);
        if (threadIdx.x == k) a.partial[(int64_t)blockIdx.x * HYB_SUMS + k] = s;
    }
}
__global__ void __launch_bounds__(256) qa_hybrid_ppo_finish_kernel(const float *partial, int nblocks, int64_t B, float c_value, float c_entropy, float *out,
                                                                   float *dstd) {
    constexpr int LANES = 256 / HYB_SUMS;
    __shared__ double s_acc[256];
    __shared__ double s_tot[HYB_SUMS];
    const int t = threadIdx.x, k = t / LANES, l = t % LANES;
    double acc = 0.0;
    if (k < HYB_SUMS)
        for (int b = l; b < nblocks; b += LANES) acc += (double)partial[(int64_t)b * HYB_SUMS + k];
    s_acc[t] = acc;
    __syncthreads();
    if (k < HYB_SUMS && l == 0) {
        double tot = 0.0;
        for (int q = 0; q < LANES; ++q) tot += s_acc[k * LANES + q];
        s_tot[k] = tot;
    }
    __syncthreads();
    const double invB = 1.0 / (double)B;
    if (t == 0) {
        out[0] = (float)((s_tot[0] + s_tot[1] + c_value * s_tot[2] - c_entropy * s_tot[3]) * invB);
        out[1] = (float)((s_tot[0] + s_tot[1]) * invB); out[2] = (float)(s_tot[2] * invB); out[3] = (float)(s_tot[3] * invB);
        out[4] = (float)(s_tot[4] * invB); out[5] = (float)(s_tot[0] * invB); out[6] = (float)(s_tot[1] * invB); out[7] = 0.f;
    }
    if (t >= 8 && t < 8 + HYB_NC) dstd[t - 8] = (float)(s_tot[5 + t - 8] * invB);
}

// ---------------------------------------------------------------------------------------------------------------
// ELU backward fused with the bias-gradient column sum (the backward of  y = elu(x W^T + b)  up to the two GEMMs):
//   grad_in[r][c] = grad_out[r][c] * (out[r][c] > 0 ? 1 : out[r][c] + alpha);   grad_bias[c] = sum_r grad_in[r][c]
// (`out` is the ELU OUTPUT: for y <= 0, d elu/dx = alpha e^x = y + alpha.)  In eager PyTorch these are two kernels that
// each stream the (rows, cols) tensor: elu_backward, then a column reduction.
// Block = 256 threads = CW columns x (256/CW) row lanes over a tile of ELU_ROWS rows; per-block column partials go to
// scratch and a second, tiny kernel adds them in a fixed order.
constexpr int ELU_ROWS = 64;

template <int CW>
__global__ void __launch_bounds__(256) qa_elu_bwd_bias_kernel(const float *__restrict__ gout, const float *__restrict__ out, float *__restrict__ gin,
                                                              float *__restrict__ partial, int64_t rows, int cols, float alpha) {
    constexpr int RL = 256 / CW;
    __shared__ float s_red[256];
    const int c0 = threadIdx.x % CW, rl = threadIdx.x / CW;
    const int64_t r0 = (int64_t)blockIdx.x * ELU_ROWS;
    const int64_t r1 = r0 + ELU_ROWS < rows ? r0 + ELU_ROWS : rows;
    for (int cb = 0; cb < cols; cb += CW) {
        const int c = cb + c0;
        float acc = 0.f;
        if (c < cols) {
            int64_t r = r0 + rl;
            for (; r + 3 * RL < r1; r += 4 * RL) {          // 4 rows in flight per lane
                const int64_t i0 = r * cols + c, st = (int64_t)RL * cols;
                const float y0 = out[i0], y1 = out[i0 + st], y2 = out[i0 + 2 * st], y3 = out[i0 + 3 * st];
                const float g0 = gout[i0] * (y0 > 0.f ? 1.0f : y0 + alpha), g1 = gout[i0 + st] * (y1 > 0.f ? 1.0f : y1 + alpha);
                const float g2 = gout[i0 + 2 * st] * (y2 > 0.f ? 1.0f : y2 + alpha), g3 = gout[i0 + 3 * st] * (y3 > 0.f ? 1.0f : y3 + alpha);
                gin[i0] = g0; gin[i0 + st] = g1; gin[i0 + 2 * st] = g2; gin[i0 + 3 * st] = g3;
                acc += (g0 + g1) + (g2 + g3);
            }
            for (; r < r1; r += RL) {
                const int64_t i = r * cols + c;
                const float y = out[i], g = gout[i] * (y > 0.f ? 1.0f : y + alpha);
                gin[i] = g;
                acc += g;
            }
        }
        if (RL > 1) {
            s_red[threadIdx.x] = acc;
            __syncthreads();
            if (rl == 0) {
#pragma unroll
                for (int k = 1; k < RL; ++k) acc += s_red[k * CW + c0];
            }
            __syncthreads();
        }
        if (rl == 0 && c < cols) partial[(int64_t)blockIdx.x * cols + c] = acc;
    }
}

__global__ void __launch_bounds__(256) qa_colsum_finish_kernel(const float *__restrict__ partial, int nblocks, int cols, float *__restrict__ gbias) {
    // block = 32 columns x 8 lanes; lane q adds partials q, q+8, ... with 4 independent accumulators, then the 8 lanes
    // are combined through LDS in a fixed order
    __shared__ float s_q[256];
    const int c = blockIdx.x * 32 + (threadIdx.x & 31), q = threadIdx.x >> 5;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (c < cols) {
        int b = q;
        for (; b + 24 < nblocks; b += 32) {
            a0 += partial[(int64_t)b * cols + c]; a1 += partial[(int64_t)(b + 8) * cols + c];
            a2 += partial[(int64_t)(b + 16) * cols + c]; a3 += partial[(int64_t)(b + 24) * cols + c];
        }
        for (; b < nblocks; b += 8) a0 += partial[(int64_t)b * cols + c];
    }
    s_q[threadIdx.x] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (q == 0 && c < cols) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) t += s_q[threadIdx.x + 32 * k];
        gbias[c] = t;
    }
}

// qa_colsum_finish_kernel with the columns split over two destinations: [0, split) -> a, [split, cols) -> b
__global__ void __launch_bounds__(256) qa_colsum_finish2_kernel(const float *__restrict__ partial, int nblocks, int cols, int split, float *__restrict__ a,
                                                                float *__restrict__ b) {
    __shared__ float s_q[256];
    const int c = blockIdx.x * 32 + (threadIdx.x & 31), q = threadIdx.x >> 5;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (c < cols) {
        int blk = q;
        for (; blk + 24 < nblocks; blk += 32) {
            a0 += partial[(int64_t)blk * cols + c]; a1 += partial[(int64_t)(blk + 8) * cols + c];
            a2 += partial[(int64_t)(blk + 16) * cols + c]; a3 += partial[(int64_t)(blk + 24) * cols + c];
        }
        for (; blk < nblocks; blk += 8) a0 += partial[(int64_t)blk * cols + c];
    }
    s_q[threadIdx.x] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (q == 0 && c < cols) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) t += s_q[threadIdx.x + 32 * k];
        if (c < split) a[c] = t; else b[c - split] = t;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Weight + bias gradient of a narrow linear layer (qa_narrow_wgrad): a weighted column sum of x.  One workgroup per slab of rows,
// one thread per input column (k, k + 256, ...); the slab's grad_out rows sit in LDS and are read as broadcasts; every thread keeps
// OMAX accumulators.  x is read once, coalesced, 4 rows in flight.  Partials [slab][out * in + out] -> qa_colsum_finish_kernel.
constexpr int NARROW_MAX_SLAB = 128;
template <int OMAX>
__global__ void __launch_bounds__(256) qa_narrow_wgrad_kernel(const float *__restrict__ gy, const float *__restrict__ x, int64_t rows, int slab_rows,
                                                              int O, int K, float *__restrict__ partial) {
    __shared__ float s_g[NARROW_MAX_SLAB * OMAX];
    const int64_t r0 = (int64_t)blockIdx.x * slab_rows;
    const int n = (int)min((int64_t)slab_rows, rows - r0);
    for (int i = threadIdx.x; i < n * O; i += 256) s_g[(i / O) * OMAX + (i % O)] = gy[r0 * O + i];
    __syncthreads();
    float *out = partial + (int64_t)blockIdx.x * ((int64_t)O * K + O);
    for (int k = threadIdx.x; k < K; k += 256) {
        float acc[OMAX];
#pragma unroll
        for (int o = 0; o < OMAX; ++o) acc[o] = 0.f;
        const float *xp = x + r0 * K + k;
        int r = 0;
        for (; r + 4 <= n; r += 4) {
            const float x0 = xp[(int64_t)r * K], x1 = xp[(int64_t)(r + 1) * K], x2 = xp[(int64_t)(r + 2) * K], x3 = xp[(int64_t)(r + 3) * K];
#pragma unroll
            for (int o = 0; o < OMAX; ++o)
                if (o < O) acc[o] = fmaf(s_g[(r + 3) * OMAX + o], x3, fmaf(s_g[(r + 2) * OMAX + o], x2, fmaf(s_g[(r + 1) * OMAX + o], x1, fmaf(s_g[r * OMAX + o], x0, acc[o]))));
        }
        for (; r < n; ++r) {
            const float x0 = xp[(int64_t)r * K];
#pragma unroll
            for (int o = 0; o < OMAX; ++o) if (o < O) acc[o] = fmaf(s_g[r * OMAX + o], x0, acc[o]);
        }
#pragma unroll
        for (int o = 0; o < OMAX; ++o) if (o < O) out[(int64_t)o * K + k] = acc[o];
    }
    if (threadIdx.x < O) {                    // bias gradient: this slab's column sums of grad_out
        float b = 0.f;
        for (int r = 0; r < n; ++r) b += s_g[r * OMAX + threadIdx.x];
        out[(int64_t)O * K + threadIdx.x] = b;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Running-moment normaliser of the discriminator inputs (bbc/rsl_rl/utils/utils.py:62-103, RunningMeanStd): fold up to
// four (n_i, d) fp32 batches, in order, into (mean, var, count) kept in double on the device; each batch contributes
// its mean and biased variance (two passes over the batch in double, as numpy's mean/var), merged with the parallel-
// variance formula.  In eager PyTorch one such update is ~35 launches of double-precision reductions and elementwise
// ops per batch (3 batches x 80 discriminator steps per iteration).  Columns are independent, so a workgroup owns 16 of them
// (16 columns x 64 row lanes; 7 workgroups for the 98-wide discriminator input: 46 -> ~10 us); the shared sample count is
// advanced by a second one-thread launch, because a workgroup that starts late must still read the old value.
constexpr int NORM_MAX_BATCHES = 4;
struct NormArgs { const float *batch[NORM_MAX_BATCHES]; int64_t n[NORM_MAX_BATCHES]; int k, d; double *mean, *var, *count; };

__global__ void __launch_bounds__(1024) qa_normalizer_update_kernel(NormArgs a) {
    // r6: the batches' passes side by side -- pass 1 of every batch, ONE barrier, the column means; pass 2 of every batch, one barrier, the merges in batch
    // order.  (Before: per batch two passes and four barriers in sequence, 22 us for three 1,228 x 98 batches, at the end of each of the discriminator's 80
    // steps.)  Every sum keeps its lanes, its order and its precision; the Chan merges run in the batches' order: bit-identical.
    __shared__ double s_red[NORM_MAX_BATCHES][1024];
    __shared__ double s_col[NORM_MAX_BATCHES][16], s_m2[NORM_MAX_BATCHES][16];
    const int CW = 16, RL = 1024 / CW;
    const int cl = threadIdx.x % CW, rl = threadIdx.x / CW;
    const int c = blockIdx.x * CW + cl;
    const bool col = c < a.d;
    const int64_t st = (int64_t)RL * a.d;
    for (int b = 0; b < a.k; ++b) {
        const float *x = a.batch[b];
        const int64_t n = a.n[b];
        double acc = 0.0;
        if (col) {       // 8 loads in flight per lane, 4 independent accumulators (a lone dependent chain is latency-bound)
            double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
            int64_t r = rl;
            for (; r + 7 * RL < n; r += 8 * RL) {
                const float *q = x + r * a.d + c;
                const float v0 = q[0], v1 = q[st], v2 = q[2 * st], v3 = q[3 * st], v4 = q[4 * st], v5 = q[5 * st], v6 = q[6 * st], v7 = q[7 * st];
                a0 += (double)v0 + (double)v4; a1 += (double)v1 + (double)v5; a2 += (double)v2 + (double)v6; a3 += (double)v3 + (double)v7;
            }
            for (; r < n; r += RL) a0 += (double)x[r * a.d + c];
            acc = (a0 + a1) + (a2 + a3);
        }
        s_red[b][threadIdx.x] = acc;
    }
    __syncthreads();
    if (rl < a.k) {          // row lane b adds batch b's 64 lane sums, in lane order
        double t = 0.0;
#pragma unroll
        for (int q = 0; q < RL; ++q) t += s_red[rl][q * CW + cl];
        s_col[rl][cl] = t / (double)a.n[rl];
    }
    __syncthreads();
    for (int b = 0; b < a.k; ++b) {
        const float *x = a.batch[b];
        const int64_t n = a.n[b];
        const double bm = s_col[b][cl];
        double acc = 0.0;
        if (col) {
            double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
            int64_t r = rl;
            for (; r + 7 * RL < n; r += 8 * RL) {
                const float *q = x + r * a.d + c;
                const double d0 = (double)q[0] - bm, d1 = (double)q[st] - bm, d2 = (double)q[2 * st] - bm, d3 = (double)q[3 * st] - bm;
                const double d4 = (double)q[4 * st] - bm, d5 = (double)q[5 * st] - bm, d6 = (double)q[6 * st] - bm, d7 = (double)q[7 * st] - bm;
                a0 += d0 * d0 + d4 * d4; a1 += d1 * d1 + d5 * d5; a2 += d2 * d2 + d6 * d6; a3 += d3 * d3 + d7 * d7;
            }
            for (; r < n; r += RL) { const double dlt = (double)x[r * a.d + c] - bm; a0 += dlt * dlt; }
            acc = (a0 + a1) + (a2 + a3);
        }
        s_red[b][threadIdx.x] = acc;
    }
    __syncthreads();
    if (rl < a.k) {
        double m2b = 0.0;
#pragma unroll
        for (int q = 0; q < RL; ++q) m2b += s_red[rl][q * CW + cl];
        s_m2[rl][cl] = m2b;
    }
    __syncthreads();
    if (rl == 0 && col) {
        double run_mean = a.mean[c], run_var = a.var[c], count = *a.count;
        for (int b = 0; b < a.k; ++b) {
            const double nb = (double)a.n[b], bm = s_col[b][cl], bv = s_m2[b][cl] / nb;
            const double delta = bm - run_mean, total = count + nb;
            const double m2 = run_var * count + bv * nb + delta * delta * count * nb / total;
            run_mean += delta * nb / total;
            run_var = m2 / total;
            count += nb;
        }
        a.mean[c] = run_mean; a.var[c] = run_var;
    }
}
__global__ void qa_normalizer_count_kernel(NormArgs a) {
    double count = *a.count;
    for (int b = 0; b < a.k; ++b) count += (double)a.n[b];
    *a.count = count;
}

// y = clamp((x - mean) / sqrt(var + eps), -clip, clip) with mean / std rounded to fp32 first (utils.py:97-103)
__global__ void __launch_bounds__(256) qa_normalizer_apply_kernel(const float *__restrict__ x, float *__restrict__ y, int64_t total, int d,
                                                                  const double *__restrict__ mean, const double *__restrict__ var, float eps, float clip) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int c = (int)(i % d);
    const float m = (float)mean[c], sd = sqrtf((float)(var[c] + (double)eps));
    y[i] = fminf(fmaxf((x[i] - m) / sd, -clip), clip);
}

// ---------------------------------------------------------------------------------------------------------------
// Gradient clipping + Adam for a set of tensors in three launches (torch: clip_grad_norm_ = ~8 launches, then the
// multi-tensor Adam kernel, whose 64k-element chunks give a 0.7 M-parameter model 12 workgroups): (1) per-chunk sums of
// squares, (2) one workgroup: total norm -> clip coefficient, step counter + bias corrections, (3) the update
//   g = coef * grad + wd * p;  m = b1 m + (1-b1) g;  v = b2 v + (1-b2) g^2;  p -= (lr / (1-b1^t)) * m / (sqrt(v)/sqrt(1-b2^t) + eps)
// which is torch.optim.Adam (amsgrad=False, maximize=False, L2 weight decay) on gradients clipped to max_norm.
// Work is a static table of (tensor, start, length) chunks of <= 2048 elements; tensors are addressed through device
// pointer tables (the gradient table is refreshed by the host whenever autograd allocated new gradient tensors).
constexpr int ADAM_CHUNK = 2048;
struct AdamArgs {
    float *const *params; const float *const *grads; float *const *exp_avg; float *const *exp_avg_sq; float *const *steps;
    const int32_t *chunk_tensor; const int32_t *chunk_start; const int32_t *chunk_len; const float *weight_decay;
    const float *lr; float *scratch;      // scratch: [0] coef [1] bc1 [2] sqrt(bc2) [3] total norm, then one partial per chunk
    int num_chunks, num_tensors;
    float beta1, beta2, eps, max_norm;
    int inline_grads;                     // 1: the gradient pointers travel in the kernel arguments (gin), not in a device table
    const float *gin[QA_ADAM_MAX_INLINE];
    // r5 (qa_clip_adam_step_reduce): tensor t's gradient is still in PARTS -- red_parts[t] > 0 slabs red_src[t] + z red_stride[t], z < parts, to be
    // added in order: the split-K slabs of a weight gradient, the per-row-block column sums of a bias gradient.  The first pass of the step
    // (the sums of squares) adds them, writes the gradient where autograd put its tensor, and squares it -- the fixed-order finish kernels
    // (qa_colsum_finish x 12, qa_slab_reduce x 7 per PPO minibatch step, ~5 us each) are launches this step no longer makes.
    int has_reduce;
    const float *red_src[QA_ADAM_MAX_INLINE];
    int64_t red_stride[QA_ADAM_MAX_INLINE];
    int32_t red_parts[QA_ADAM_MAX_INLINE];
    // r6 (qa_clip_adam_pair_step): TWO optimisers in the same three launches -- tensors >= split_tensor / chunks >= split_chunk are the second one's, with its
    // own clipping norm, step counter and learning rate; the KL rule (qa_kl_lr_rule) moves lr2 in the finalize launch, before the update reads it.
    // split_chunk == 0: one optimiser.  The second head (coef, bc1, sqrt(bc2), norm) sits behind the arrival counter: scratch[4 + num_chunks + 1 ..].
    int split_chunk, split_tensor;
    float *lr2; float max_norm2;
    const float *kl; float desired_kl, kl_factor, lr_min, lr_max;
};
__device__ __forceinline__ const float *adam_grad(const AdamArgs &a, int t) { return a.inline_grads ? a.gin[t] : a.grads[t]; }

// one chunk of a gradient that is still in parts: element i = sum_z src[z stride + i], z ascending (bit-reproducible), written to g and
// returned squared-and-summed per thread.  Few parts (split-K slabs, <= QA_REDUCE_WIDE): a thread owns elements tid, tid + 256, ...; many
// parts (column partials of 24,576 / 64 = 384 row blocks): the chunk is <= 32 elements wide (the host's chunk table guarantees it for
// every tensor that can carry such parts) and the 8 thread rows of 32 share the parts like qa_colsum_finish_kernel does.
constexpr int QA_REDUCE_WIDE = 16;
__device__ __forceinline__ float adam_reduce_chunk(const float *__restrict__ src, int64_t stride, int parts, float *__restrict__ g, int n, float *s_q) {
    float acc = 0.f;
    if (parts <= QA_REDUCE_WIDE) {
        for (int i = threadIdx.x; i < n; i += 256) {
            float v[QA_REDUCE_WIDE];
#pragma unroll
            for (int z = 0; z < QA_REDUCE_WIDE; ++z) v[z] = z < parts ? src[(int64_t)z * stride + i] : 0.f;      // all loads in flight, then the ordered sum
            float s = v[0];
#pragma unroll
            for (int z = 1; z < QA_REDUCE_WIDE; ++z) if (z < parts) s += v[z];
            g[i] = s;
            acc = fmaf(s, s, acc);
        }
    } else {
        const int c = threadIdx.x & 31, q = threadIdx.x >> 5;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        if (c < n) {
            int b = q;
            for (; b + 24 < parts; b += 32) {
                a0 += src[(int64_t)b * stride + c]; a1 += src[(int64_t)(b + 8) * stride + c];
                a2 += src[(int64_t)(b + 16) * stride + c]; a3 += src[(int64_t)(b + 24) * stride + c];
            }
            for (; b < parts; b += 8) a0 += src[(int64_t)b * stride + c];
        }
        s_q[threadIdx.x] = (a0 + a1) + (a2 + a3);
        __syncthreads();
        if (q == 0 && c < n) {
            float t = 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) t += s_q[threadIdx.x + 32 * k];
            g[c] = t;
            acc = t * t;
        }
        __syncthreads();
    }
    return acc;
}

__global__ void __launch_bounds__(256) qa_adam_sumsq_kernel(AdamArgs a) {
    __shared__ float s_w[4];
    __shared__ float s_q[256];
    const int c = blockIdx.x, t = a.chunk_tensor[c];
    const float *g = adam_grad(a, t) + a.chunk_start[c];
    const int n = a.chunk_len[c];
    float acc = 0.f;
    if (a.has_reduce && a.red_parts[t] > 0) {
        acc = adam_reduce_chunk(a.red_src[t] + a.chunk_start[c], a.red_stride[t], a.red_parts[t], const_cast<float *>(g), n, s_q);
    } else {
        for (int i = threadIdx.x; i < n; i += 256) { const float v = g[i]; acc = fmaf(v, v, acc); }
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) a.scratch[4 + c] = (s_w[0] + s_w[1]) + (s_w[2] + s_w[3]);
}

__global__ void __launch_bounds__(256) qa_adam_finalize_kernel(AdamArgs a) {
    __shared__ double s_acc[256], s_acc2[256];
    const int split = a.split_chunk > 0 ? a.split_chunk : a.num_chunks;
    double acc = 0.0, acc2 = 0.0;
    if (a.max_norm > 0.f)          /* without clipping the per-chunk sums were not computed (and the norm is reported as 0) */
        for (int c = threadIdx.x; c < split; c += 256) acc += (double)a.scratch[4 + c];
    if (a.split_chunk > 0 && a.max_norm2 > 0.f)
        for (int c = split + threadIdx.x; c < a.num_chunks; c += 256) acc2 += (double)a.scratch[4 + c];
    s_acc[threadIdx.x] = acc; s_acc2[threadIdx.x] = acc2;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) { s_acc[threadIdx.x] += s_acc[threadIdx.x + o]; s_acc2[threadIdx.x] += s_acc2[threadIdx.x + o]; } __syncthreads(); }
    const int nt1 = a.split_chunk > 0 ? a.split_tensor : a.num_tensors;
    const float step = a.steps[0][0] + 1.0f;
    const float step2 = a.split_chunk > 0 ? a.steps[nt1][0] + 1.0f : 0.f;
    __syncthreads();
    for (int t = threadIdx.x; t < a.num_tensors; t += 256) a.steps[t][0] = t < nt1 ? step : step2;
    if (threadIdx.x == 0) {
        const float norm = (float)sqrt(s_acc[0]);
        a.scratch[0] = a.max_norm > 0.f ? fminf(1.0f, a.max_norm / (norm + 1e-6f)) : 1.0f;
        a.scratch[1] = 1.0f - powf(a.beta1, step);
        a.scratch[2] = sqrtf(1.0f - powf(a.beta2, step));
        a.scratch[3] = norm;
        if (a.split_chunk > 0) {
            float *h = a.scratch + 4 + a.num_chunks + 1;
            const float norm2 = (float)sqrt(s_acc2[0]);
            h[0] = a.max_norm2 > 0.f ? fminf(1.0f, a.max_norm2 / (norm2 + 1e-6f)) : 1.0f;
            h[1] = 1.0f - powf(a.beta1, step2);
            h[2] = sqrtf(1.0f - powf(a.beta2, step2));
            h[3] = norm2;
            if (a.kl) {             /* qa_kl_lr_rule_kernel's rule (gail.py:367-379), on the second optimiser's learning rate */
                const float k = a.kl[0], cur = a.lr2[0];
                float out = cur;
                if (k > a.desired_kl * 2.0f) out = fmaxf(a.lr_min, cur / a.kl_factor);
                else if (k < a.desired_kl / 2.0f && k > 0.0f) out = fminf(a.lr_max, cur * a.kl_factor);
                a.lr2[0] = out;
            }
        }
    }
}

/* ---- pair losses: mean row L2 norm of (a - b) (privileged-latent regulariser, gail.py:346-350) and mean squared difference
 * (estimator regression, gail.py:356-358), each with its gradient w.r.t. a in the same pass.  One lane per row. */
constexpr int PAIR_BLOCK = 256;

__global__ void __launch_bounds__(PAIR_BLOCK) qa_pair_loss_kernel(const float *__restrict__ a, const float *__restrict__ b, int64_t rows, int cols,
                                                                 int64_t b_stride, int mode, float *__restrict__ grad_a, float *__restrict__ partial) {
    __shared__ float s_red[PAIR_BLOCK];
    const int64_t r = (int64_t)blockIdx.x * PAIR_BLOCK + threadIdx.x;
    float contrib = 0.f;
    if (r < rows) {
        const float *ar = a + r * cols, *br = b + r * b_stride;
        float *gr = grad_a + r * cols;
        float ss = 0.f;
        for (int c = 0; c < cols; ++c) { const float d = ar[c] - br[c]; ss = fmaf(d, d, ss); }
        if (mode == 0) {
            const float nrm = sqrtf(ss);
            const float sc = nrm > 0.f ? 1.0f / (nrm * (float)rows) : 0.f;          /* subgradient 0 at the origin, as torch.norm's backward */
            for (int c = 0; c < cols; ++c) gr[c] = (ar[c] - br[c]) * sc;
            contrib = nrm;
        } else {
            const float sc = 2.0f / ((float)rows * (float)cols);
            for (int c = 0; c < cols; ++c) gr[c] = (ar[c] - br[c]) * sc;
            contrib = ss;
        }
    }
    s_red[threadIdx.x] = contrib;
    __syncthreads();
    for (int o = PAIR_BLOCK / 2; o > 0; o >>= 1) { if (threadIdx.x < o) s_red[threadIdx.x] += s_red[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0) partial[blockIdx.x] = s_red[0];
}

__global__ void __launch_bounds__(256) qa_pair_finish_kernel(const float *__restrict__ partial, int nb, float scale, float *__restrict__ out) {
    __shared__ double s_acc[256];
    double acc = 0.0;
    for (int i = threadIdx.x; i < nb; i += 256) acc += (double)partial[i];
    s_acc[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) s_acc[threadIdx.x] += s_acc[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0) out[0] = (float)(s_acc[0] * (double)scale);
}

/* r6 (ABI 18): up to QA_PAIR_MAX_JOBS pair losses in ONE launch, finished by the last workgroup to arrive (the fixed-order double sum of
 * qa_pair_finish_kernel over each job's partials), the gradient optionally times a device scalar -- a PPO chain step's regulariser + estimator losses were five
 * launches (two losses, two finishes, the multiply by the regulariser's coefficient). */
constexpr int PAIR_LDS_COLS = 47;        // 256 rows x (cols | 1) floats in LDS: <= 48 KB
struct PairJobsArgs { qa_pair_job j[QA_PAIR_MAX_JOBS]; int32_t first_block[QA_PAIR_MAX_JOBS + 1]; int32_t n; float *partial; unsigned *ticket; };
__global__ void __launch_bounds__(PAIR_BLOCK) qa_pair_losses_kernel(PairJobsArgs p) {
    __shared__ float s_red[PAIR_BLOCK];
    __shared__ double s_acc[PAIR_BLOCK];
    __shared__ int s_last;
    __shared__ float s_d[PAIR_BLOCK * (PAIR_LDS_COLS | 1)];
    int t = 0;
    while (t + 1 < p.n && (int)blockIdx.x >= p.first_block[t + 1]) ++t;
    const float *a = p.j[t].a, *b = p.j[t].b;
    const int64_t rows = p.j[t].rows, b_stride = p.j[t].b_stride;
    const int cols = p.j[t].cols, mode = p.j[t].mode;
    const bool scaled = p.j[t].grad_scale != nullptr;
    const float gs = scaled ? p.j[t].grad_scale[0] : 1.0f;
    const int64_t r0 = (int64_t)((int)blockIdx.x - p.first_block[t]) * PAIR_BLOCK;
    const int64_t r = r0 + threadIdx.x;
    float contrib = 0.f;
    if (cols <= PAIR_LDS_COLS) {
        // the block's rows through LDS: a lane-per-row walk over rows of 29 floats in global memory is 29 uncoalesced round trips twice over (qa_pair_loss:
        // 11.8 us for 3,072 x 29); here the differences arrive in one coalesced sweep, a lane walks ITS row in LDS (an odd row stride: no bank conflict) in the
        // same order with the same operations -- bit-identical to qa_pair_loss -- and the gradient leaves in one coalesced sweep
        const int nrow = (int)min((int64_t)PAIR_BLOCK, rows - r0);
        const int ld = cols | 1;
        for (int i = threadIdx.x; i < nrow * cols; i += PAIR_BLOCK) {
            const int rr = i / cols, c = i - rr * cols;
            s_d[rr * ld + c] = a[(r0 + rr) * cols + c] - b[(r0 + rr) * b_stride + c];
        }
        __syncthreads();
        if (r < rows) {
            float *dr = s_d + threadIdx.x * ld;
            float ss = 0.f;
            for (int c = 0; c < cols; ++c) { const float d = dr[c]; ss = fmaf(d, d, ss); }
            float sc;
            if (mode == 0) {
                const float nrm = sqrtf(ss);
                sc = nrm > 0.f ? 1.0f / (nrm * (float)rows) : 0.f;
                contrib = nrm;
            } else {
                sc = 2.0f / ((float)rows * (float)cols);
                contrib = ss;
            }
            if (scaled) for (int c = 0; c < cols; ++c) dr[c] = (dr[c] * sc) * gs;           // (the separate multiply's rounding)
            else for (int c = 0; c < cols; ++c) dr[c] = dr[c] * sc;
        }
        __syncthreads();
        float *g0 = p.j[t].grad_a + r0 * cols;
        for (int i = threadIdx.x; i < nrow * cols; i += PAIR_BLOCK) { const int rr = i / cols, c = i - rr * cols; g0[i] = s_d[rr * ld + c]; }
    } else if (r < rows) {
        const float *ar = a + r * cols, *br = b + r * b_stride;
        float *gr = p.j[t].grad_a + r * cols;
        float ss = 0.f;
        for (int c = 0; c < cols; ++c) { const float d = ar[c] - br[c]; ss = fmaf(d, d, ss); }
        float sc;
        if (mode == 0) {
            const float nrm = sqrtf(ss);
            sc = nrm > 0.f ? 1.0f / (nrm * (float)rows) : 0.f;
            contrib = nrm;
        } else {
            sc = 2.0f / ((float)rows * (float)cols);
            contrib = ss;
        }
        if (scaled) for (int c = 0; c < cols; ++c) gr[c] = ((ar[c] - br[c]) * sc) * gs;           // (the separate multiply's rounding)
        else for (int c = 0; c < cols; ++c) gr[c] = (ar[c] - br[c]) * sc;
    }
    s_red[threadIdx.x] = contrib;
    __syncthreads();
    for (int o = PAIR_BLOCK / 2; o > 0; o >>= 1) { if (threadIdx.x < o) s_red[threadIdx.x] += s_red[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0) {
        p.partial[blockIdx.x] = s_red[0];
        __threadfence();
        s_last = atomicAdd(p.ticket, 1u) == (unsigned)(p.first_block[p.n] - 1);
    }
    __syncthreads();
    if (!s_last) return;
    __threadfence();
    for (int k = 0; k < p.n; ++k) {
        const int lo = p.first_block[k], nb = p.first_block[k + 1] - lo;
        double acc = 0.0;
        for (int i = threadIdx.x; i < nb; i += 256) acc += (double)__hip_atomic_load(p.partial + lo + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_acc[threadIdx.x] = acc;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) s_acc[threadIdx.x] += s_acc[threadIdx.x + o]; __syncthreads(); }
        if (threadIdx.x == 0) {
            const float scale = p.j[k].mode == QA_PAIR_ROW_L2 ? 1.0f / (float)p.j[k].rows : 1.0f / ((float)p.j[k].rows * (float)p.j[k].cols);
            p.j[k].out[0] = (float)(s_acc[0] * (double)scale);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) *p.ticket = 0u;           // ready for the next launch (replays of a recorded step included)
}

/* ---- minibatch gather: dst_t[r, :] = src_t[idx[r], :] for up to QA_GATHER_MAX row-major fp32 tensors in one launch
 * (RolloutStorage.mini_batch_generator, rollout_storage.py:122-157: nine indexed reads per minibatch) */
constexpr int GATHER_ROWS = 4;
struct GatherArgs {
    const float *src[QA_GATHER_MAX]; float *dst[QA_GATHER_MAX]; int32_t width[QA_GATHER_MAX]; int64_t src_stride[QA_GATHER_MAX];
    const int64_t *idx; const int64_t *idx_block; int64_t rows; int32_t n;
};

__global__ void __launch_bounds__(256) qa_gather_rows_kernel(GatherArgs a) {
    /* one wavefront per row: the row index is wave-uniform, and a lane's loads along the row (11 for the 671-float
     * observation) are all independent and in flight together */
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * GATHER_ROWS + (threadIdx.x >> 6);
    if (r >= a.rows) return;
    const int64_t s = a.idx[(a.idx_block ? a.idx_block[0] * a.rows : 0) + r];
    for (int t = 0; t < a.n; ++t) {
        const float *sp = a.src[t] + s * a.src_stride[t];
        float *dp = a.dst[t] + r * a.width[t];
        const int w = a.width[t];
        if (w <= 64) { if (lane < w) dp[lane] = sp[lane]; continue; }
        int c = lane;
        for (; c + 192 < w; c += 256) {
            const float v0 = sp[c], v1 = sp[c + 64], v2 = sp[c + 128], v3 = sp[c + 192];
            dp[c] = v0; dp[c + 64] = v1; dp[c + 128] = v2; dp[c + 192] = v3;
        }
        for (; c < w; c += 64) dp[c] = sp[c];
    }
}

/* acc[i] += *src[i], i < n <= 16: a step's logged scalars (they live in different kernels' outputs) onto the update's accumulator in one launch -- was a
 * torch.stack (a copy kernel) + an add per minibatch step (gail.py:275-283 sums the same six values on the host) */
struct AccArgs { float *acc; const float *src[QA_ACC_MAX]; int n; };
__global__ void qa_accumulate_scalars_kernel(AccArgs a) {
    const int i = threadIdx.x;
    if (i < a.n) a.acc[i] += a.src[i][0];
}

/* KL-adaptive learning rate (gail.py:367-379) on device scalars: one thread */
__global__ void qa_kl_lr_rule_kernel(const float *kl, float desired_kl, float factor, float lr_min, float lr_max, float *lr) {
    const float k = kl[0], cur = lr[0];
    float out = cur;
    if (k > desired_kl * 2.0f) out = fmaxf(lr_min, cur / factor);
    else if (k < desired_kl / 2.0f && k > 0.0f) out = fminf(lr_max, cur * factor);
    lr[0] = out;
}

/* extras["episode"] (legged_robot.py:229-240) from the step's EPISODE_STATS bin: one wavefront */
__global__ void qa_episode_means_kernel(const float *__restrict__ stats, const int64_t *__restrict__ step_dev, int64_t step, int num_terms,
                                        float max_len_s, float *__restrict__ means, float *__restrict__ snapshot) {
    const int i = threadIdx.x;
    if (i >= num_terms) return;
    const int64_t st = step_dev ? step_dev[0] : step;
    const float *b = stats + 16 * (int)((st - 1) & 1);
    const float cnt = b[14];
    const float m = cnt > 0.f ? b[i] / fmaxf(cnt, 1.0f) / max_len_s : means[i];
    means[i] = m;
    snapshot[i] = m;
}

// SELF: no clipping, so nothing has to be known about the whole gradient before the update -- the step count is read (old value) by every
// workgroup, the bias corrections are recomputed per workgroup (the finalize kernel's expressions), and the LAST workgroup to arrive writes
// the incremented counters and the scratch head: one launch instead of finalize + update (the discriminator's three optimisers, 80 steps
// per iteration of a launch-latency-bound chain).
template <bool SELF>
__global__ void __launch_bounds__(256) qa_adam_update_kernel(AdamArgs a) {
    const int c = blockIdx.x, t = a.chunk_tensor[c], s0 = a.chunk_start[c], n = a.chunk_len[c];
    float *p = a.params[t] + s0, *m = a.exp_avg[t] + s0, *v = a.exp_avg_sq[t] + s0;
    const float *g = adam_grad(a, t) + s0;
    const float step = SELF ? a.steps[0][0] + 1.0f : 0.f;
    const bool second = !SELF && a.split_chunk > 0 && c >= a.split_chunk;           // (qa_clip_adam_pair_step)
    const float *head = second ? a.scratch + 4 + a.num_chunks + 1 : a.scratch;
    const float coef = SELF ? 1.0f : head[0], bc1 = SELF ? 1.0f - powf(a.beta1, step) : head[1],
                bc2s = SELF ? sqrtf(1.0f - powf(a.beta2, step)) : head[2], wd = a.weight_decay[t];
    const float step_size = (second ? a.lr2[0] : a.lr[0]) / bc1;
    for (int i = threadIdx.x; i < n; i += 256) {
        const float pi = p[i];
        const float gi = fmaf(wd, pi, g[i] * coef);
        const float mi = a.beta1 * m[i] + (1.0f - a.beta1) * gi;
        const float vi = a.beta2 * v[i] + (1.0f - a.beta2) * gi * gi;
        m[i] = mi; v[i] = vi;
        p[i] = pi - step_size * mi / (sqrtf(vi) / bc2s + a.eps);
    }
    if (SELF) {
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned *ticket = (unsigned *)(a.scratch + 4 + a.num_chunks);
            __threadfence();
            if (atomicAdd(ticket, 1u) == (unsigned)(a.num_chunks - 1)) {      // every workgroup has read the old step count by now
                for (int k = 0; k < a.num_tensors; ++k) a.steps[k][0] = step;
                a.scratch[0] = 1.0f; a.scratch[1] = bc1; a.scratch[2] = bc2s; a.scratch[3] = 0.f;
                *ticket = 0u;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Rollout bookkeeping around the env step (rows a10, a12: gail.py:176-212, rollout_storage.py:60-74).
// qa_rollout_act: after the actor / critic GEMMs -- sample a = mu + std * eps, log-prob, and write the step's rows of
// the rollout storage (actions, mu, sigma, log-prob, value) plus the action buffer the env reads: one launch for what
// is ~18 in eager PyTorch (expand std, randn, mul, add, the log_prob chain, five copies).  eps is either supplied
// (tests) or drawn from the engine's Philox stream RS_ACT_NOISE keyed by (seed; env, step).
// qa_rollout_post: after the env step -- reward scaling + time-out bootstrap (r += gamma V time_out), done flags, and
// the runner's per-episode logging sums, another ~12 launches.
constexpr int RS_ACT_NOISE = 20;

__device__ __forceinline__ void rollout_act_one(const int e, const float *__restrict__ mean, const float *__restrict__ std, const float *__restrict__ value,
                                                             const float *__restrict__ noise, uint64_t seed, const int64_t *step_ptr, int64_t step_host, int N, int env0,
                                                             float *__restrict__ actions, float *__restrict__ st_actions, float *__restrict__ st_mu,
                                                             float *__restrict__ st_sigma, float *__restrict__ st_logp, float *__restrict__ st_values) {
    const int64_t step = step_ptr ? *step_ptr : step_host;
    const float HALF_LOG_2PI = 0.91893853320467274178f;
    float eps[12];
    if (noise) {
#pragma unroll
        for (int j = 0; j < 12; ++j) eps[j] = noise[(int64_t)e * 12 + j];
    } else {
#pragma unroll
        for (int b = 0; b < 3; ++b) {          // Box-Muller on the 4 uniforms of a Philox block: 4 normals
            F4 u = rng4(seed, (uint32_t)(e + env0), step, RS_ACT_NOISE, b);
            const float r0 = sqrtf(-2.0f * logf(fmaxf(u.v[0], 1e-7f))), r1 = sqrtf(-2.0f * logf(fmaxf(u.v[2], 1e-7f)));
            float s0, c0, s1, c1;
            sincosf(6.28318530717958647692f * u.v[1], &s0, &c0);
            sincosf(6.28318530717958647692f * u.v[3], &s1, &c1);
            eps[4 * b] = r0 * c0; eps[4 * b + 1] = r0 * s0; eps[4 * b + 2] = r1 * c1; eps[4 * b + 3] = r1 * s1;
        }
    }
    float logp = 0.f;
#pragma unroll
    for (int j = 0; j < 12; ++j) {
        const float m = mean[(int64_t)e * 12 + j], s = std[j], a = m + s * eps[j], d = a - m;
        logp += -(d * d) / (2.0f * s * s) - logf(s) - HALF_LOG_2PI;
        const int64_t i = (int64_t)e * 12 + j;
        actions[i] = a; st_actions[i] = a; st_mu[i] = m; st_sigma[i] = s;
    }
    st_logp[e] = logp;
    st_values[e] = value[e];
}
__global__ void __launch_bounds__(256) qa_rollout_act_kernel(const float *__restrict__ mean, const float *__restrict__ std, const float *__restrict__ value,
                                                             const float *__restrict__ noise, uint64_t seed, const int64_t *step_ptr, int64_t step_host, int N, int env0,
                                                             float *__restrict__ actions, float *__restrict__ st_actions, float *__restrict__ st_mu,
                                                             float *__restrict__ st_sigma, float *__restrict__ st_logp, float *__restrict__ st_values) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= N) return;
    rollout_act_one(e, mean, std, value, noise, seed, step_ptr, step_host, N, env0, actions, st_actions, st_mu, st_sigma, st_logp, st_values);
}
// r6 (qa_rollout_act_store): the same sampling with the step's OBSERVATION rows copied into the storage by the same launch (RolloutStorage.add_transitions,
// rollout_storage.py:60-74: was a strided copy launch of its own, 5-7 us per env step).  ACT_EPB envs per 256-thread workgroup: the first ACT_EPB threads sample,
// all 256 copy the workgroup's rows, 8 loads in flight per thread (3 rounds for 8 x 671 floats).
constexpr int ACT_EPB = 8;
__global__ void __launch_bounds__(256) qa_rollout_act_store_kernel(const float *__restrict__ mean, const float *__restrict__ std, const float *__restrict__ value,
                                                                   const float *__restrict__ noise, uint64_t seed, const int64_t *step_ptr, int64_t step_host, int N, int env0,
                                                                   float *__restrict__ actions, float *__restrict__ st_actions, float *__restrict__ st_mu,
                                                                   float *__restrict__ st_sigma, float *__restrict__ st_logp, float *__restrict__ st_values,
                                                                   const float *__restrict__ obs, int64_t obs_stride, int width, float *__restrict__ st_obs, int64_t st_stride) {
    const int e0 = blockIdx.x * ACT_EPB, ne = min(ACT_EPB, N - e0);
    const int total = ne * width;
    for (int i0 = threadIdx.x; i0 < total; i0 += 256 * 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = min(i0 + 256 * u, total - 1);
            const int le = i / width, c = i - le * width;
            v[u] = obs[(int64_t)(e0 + le) * obs_stride + c];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = i0 + 256 * u;
            if (i < total) { const int le = i / width, c = i - le * width; st_obs[(int64_t)(e0 + le) * st_stride + c] = v[u]; }
        }
    }
    const int e = e0 + (int)threadIdx.x;
    if ((int)threadIdx.x < ne)
        rollout_act_one(e, mean, std, value, noise, seed, step_ptr, step_host, N, env0, actions, st_actions, st_mu, st_sigma, st_logp, st_values);
}

constexpr int RS_TSC_PUSH = 22, RS_TSC_START = 23;

/* the task-level env step's torch glue as kernels (include/qa_sim.h: qa_tsc_push / qa_tsc_start_pose / qa_tsc_reset_where); one thread per env */
__global__ void __launch_bounds__(64) qa_tsc_push_kernel(float *__restrict__ root, int64_t n, int64_t *step_dev, int *ticket, int interval, float vmax, uint64_t seed, int env0) {
    const int64_t e = (int64_t)blockIdx.x * 64 + threadIdx.x;
    const int64_t s = *step_dev + 1;                    // every workgroup reads the old counter; the last one to arrive writes the new one
    if (e < n && interval > 0 && s % interval == 0) {
        const F4 u = rng4(seed, (uint32_t)(e + env0), s, RS_TSC_PUSH, 0);
        root[e * 13 + 7] = (u.v[0] * 2.0f - 1.0f) * vmax;
        root[e * 13 + 8] = (u.v[1] * 2.0f - 1.0f) * vmax;
    }
    if (threadIdx.x == 0) {
        __threadfence();
        if (atomicAdd(ticket, 1) == (int)gridDim.x - 1) { *ticket = 0; *step_dev = s; __threadfence(); }
    }
}

struct StartPoseArgs {
    const uint8_t *flags; int64_t *cur_obst; const float *goals, *angs; int64_t n; int slots, nobst, gpo, randomize; float yaw0; uint64_t seed;
    const int64_t *step_dev; int env0; float *xy, *yaw; int64_t *start_goal;
};
__global__ void __launch_bounds__(64) qa_tsc_start_pose_kernel(StartPoseArgs a) {
    const int64_t e = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (e >= a.n) return;
    int64_t sg = 0;
    float yaw = a.yaw0;
    if (a.randomize) {
        int64_t ob = a.cur_obst[e];
        if (a.flags[e]) {
            const F4 u = rng4(a.seed, (uint32_t)(e + a.env0), *a.step_dev, RS_TSC_START, 0);
            int d = (int)(u.v[0] * (float)a.nobst);
            ob = d >= a.nobst ? a.nobst - 1 : d;
            a.cur_obst[e] = ob;
        }
        sg = ob * a.gpo;
        yaw = a.angs[e * a.nobst + ob];
    }
    const int64_t g = sg < 0 ? 0 : (sg >= a.slots ? a.slots - 1 : sg);
    a.xy[2 * e] = a.goals[(e * a.slots + g) * 3]; a.xy[2 * e + 1] = a.goals[(e * a.slots + g) * 3 + 1];
    a.yaw[e] = yaw;
    a.start_goal[e] = sg;
}

struct ResetWhereArgs {
    const uint8_t *flags, *any_reset; const int64_t *start_goal; int64_t *cur_goal; float *timer, *sums; int terms; int64_t *ep_len; const float *goals; int slots;
    float *cur_goals, *next_goals, *obst; float rest; const int64_t *cur_obst, *order; int64_t n;
};
__global__ void __launch_bounds__(64) qa_tsc_reset_where_kernel(ResetWhereArgs a) {
    const int64_t e = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (e >= a.n) return;
    const bool f = a.flags[e] != 0;
    int64_t g = a.cur_goal[e];
    if (f) {
        g = a.start_goal[e];
        a.cur_goal[e] = g; a.timer[e] = 0.f; a.ep_len[e] = 0;
        for (int k = 0; k < a.terms; ++k) a.sums[(int64_t)k * a.n + e] = 0.f;
    }
    const int64_t g0 = g < 0 ? 0 : (g >= a.slots ? a.slots - 1 : g), g1 = g + 1 < 0 ? 0 : (g + 1 >= a.slots ? a.slots - 1 : g + 1);
#pragma unroll
    for (int k = 0; k < 3; ++k) { a.cur_goals[e * 3 + k] = a.goals[(e * a.slots + g0) * 3 + k]; a.next_goals[e * 3 + k] = a.goals[(e * a.slots + g1) * 3 + k]; }
    if (a.obst) {
        float *st = a.obst + e * 12;                   // (3 obstacles, 4 floats): [position, velocity, ...]
        if (f) st[0] = (a.order && a.cur_obst[e] > a.order[e]) ? -a.rest : a.rest;
        if (a.any_reset[0]) { st[1] = 0.f; st[5] = 0.f; st[9] = 0.f; }
    }
}

constexpr int RS_ACT_CHOICE = 21;
struct HybridActArgs {
    const float *logits, *mean, *std, *value;
    uint64_t seed; const int64_t *step_ptr; int64_t step_host;
    int N, env0, nd, nc;
    float *actions, *st_actions, *st_mu, *st_sigma, *st_logp_d, *st_logp_c, *st_values, *hist;
    const float *hist_in;
    int hist_len;
    int epb;        // envs per 128-thread workgroup (r6): the first epb threads sample, all 128 roll the workgroup's envs' history
};
/* one thread per env: softmax over <= 16 logits, inverse-CDF choice, <= 32 Gaussian parameters, the storage rows and the action-history roll.
 * r6: few envs per workgroup.  With 128 envs per workgroup the out-of-place roll is 19 rounds of (8 loads, 8 stores) per thread whatever N is -- 19 dependent
 * round trips, ~28 of the launch's 32 us at 1024 AND at 8192 envs; the host picks epb so that the launch has >= ~256 workgroups and the roll is 1-5 rounds. */
__global__ void __launch_bounds__(128) qa_rollout_act_hybrid_kernel(HybridActArgs a) {
    const int e = blockIdx.x * a.epb + threadIdx.x;
    const int64_t step = a.step_ptr ? *a.step_ptr : a.step_host;
    const int nd = a.nd, nc = a.nc, w = 1 + nc;
    const bool mine = (int)threadIdx.x < a.epb && e < a.N;
    if (mine) {
    const float HALF_LOG_2PI = 0.91893853320467274178f, EPS = 1.1920928955078125e-07f;
    const float *lg = a.logits + (int64_t)e * nd;
    float mx = lg[0];
    for (int i = 1; i < nd; ++i) mx = fmaxf(mx, lg[i]);
    float p[16], sum = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) { p[i] = i < nd ? expf(lg[i] - mx) : 0.f; sum += p[i]; }
    const float inv = 1.0f / sum;
    const F4 uc = rng4(a.seed, (uint32_t)(e + a.env0), step, RS_ACT_CHOICE, 0);
    const float u = uc.v[0];
    int choice = nd - 1;
    float cdf = 0.f, pa = 0.f;
    bool found = false;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        if (i < nd) {
            const float pi = p[i] * inv;
            cdf += pi;
            if (!found && (u < cdf || i == nd - 1)) { choice = i; pa = pi; found = true; }
        }
    }
    const float logp_d = logf(fminf(fmaxf(pa, EPS), 1.0f - EPS));
    float *act = a.actions + (int64_t)e * w, *sa = a.st_actions + (int64_t)e * w;
    act[0] = (float)choice; sa[0] = (float)choice;
    float logp_c = 0.f;
    for (int b = 0; 4 * b < nc; ++b) {          // Box-Muller on the 4 uniforms of a Philox block: 4 normals (as qa_rollout_act)
        const F4 q = rng4(a.seed, (uint32_t)(e + a.env0), step, RS_ACT_NOISE, b);
        const float r0 = sqrtf(-2.0f * logf(fmaxf(q.v[0], 1e-7f))), r1 = sqrtf(-2.0f * logf(fmaxf(q.v[2], 1e-7f)));
        float s0, c0, s1, c1;
        sincosf(6.28318530717958647692f * q.v[1], &s0, &c0);
        sincosf(6.28318530717958647692f * q.v[3], &s1, &c1);
        const float eps4[4] = {r0 * c0, r0 * s0, r1 * c1, r1 * s1};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int j = 4 * b + k;
            if (j < nc) {
                const float m = a.mean[(int64_t)e * nc + j], s = a.std[j], v = m + s * eps4[k], d = v - m;
                logp_c += -(d * d) / (2.0f * s * s) - logf(s) - HALF_LOG_2PI;
                act[1 + j] = v; sa[1 + j] = v;
                a.st_mu[(int64_t)e * nc + j] = m; a.st_sigma[(int64_t)e * nc + j] = s;
            }
        }
    }
    a.st_logp_d[e] = logp_d; a.st_logp_c[e] = logp_c; a.st_values[e] = a.value[e];
    }
    if (a.hist) {
        // roll by one slot (oldest first), newest slot = this step's action.  Out of place when the caller gives a second buffer (hist_in != hist):
        // the workgroup's 128 envs are one contiguous run of both buffers, copied with coalesced accesses by all threads; in place (hist_in == hist)
        // every thread moves its own env's block front to back.
        __syncthreads();                       // the block's `actions` rows are written
        const int e0 = blockIdx.x * a.epb, ne = min(a.epb, a.N - e0), per = a.hist_len * w;
        if (a.hist_in != a.hist) {
            const float *__restrict__ src = a.hist_in + (int64_t)e0 * per;
            const float *__restrict__ newest = a.actions + (int64_t)e0 * w;
            float *__restrict__ dst = a.hist + (int64_t)e0 * per;
            const int total = ne * per;
            // eight independent loads in flight per thread before the first store (a plain loop waits for every load: 152 dependent round trips)
            for (int i0 = threadIdx.x; i0 < total; i0 += 128 * 8) {
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int i = min(i0 + 128 * u, total - 1);
                    const int le = i / per, o = i - le * per;
                    v[u] = o < per - w ? src[i + w] : newest[le * w + (o - (per - w))];
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) { const int i = i0 + 128 * u; if (i < total) dst[i] = v[u]; }
            }
        } else if (mine) {
            float *h = a.hist + (int64_t)e * per;
            for (int o = 0; o < per - w; ++o) h[o] = h[o + w];
            for (int j = 0; j < w; ++j) h[per - w + j] = a.actions[(int64_t)e * w + j];
        }
    }
}

__global__ void __launch_bounds__(256) qa_rollout_post_kernel(const float *__restrict__ rew, const int64_t *__restrict__ reset, const uint8_t *__restrict__ time_out,
                                                              const float *__restrict__ values, float reward_coef, float gamma, int N, float *__restrict__ st_rewards,
                                                              uint8_t *__restrict__ st_dones, float *__restrict__ cur, float *__restrict__ fin_vals, uint8_t *__restrict__ fin_mask) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= N) return;
    const float r_t = rew[e], r = reward_coef * r_t;
    const bool done = reset[e] > 0;
    st_rewards[e] = r + gamma * values[e] * (time_out[e] ? 1.0f : 0.0f);        // bootstrap on time-outs (gail.py:203-205)
    st_dones[e] = done ? 1 : 0;
    if (cur) {      // running episode sums [total, i, us, ss, t, length] (on_policy_runner.py:187-206), logged before the reset clears them
        const float add[6] = {r, 0.f, 0.f, 0.f, r_t, 1.0f};
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const float c = cur[(int64_t)k * N + e] + add[k];
            fin_vals[(int64_t)k * N + e] = c;
            cur[(int64_t)k * N + e] = done ? 0.f : c;
        }
        fin_mask[e] = done ? 1 : 0;
    }
}

struct PostAmpArgs {
    const float *rew; const int64_t *reset; const uint8_t *time_out; const float *values, *d, *eps, *logits, *obs;
    int64_t obs_stride; int num_obs, dim_c, N;
    float c_i, c_us, c_ss, c_t, dt, gamma;
    float *st_rewards; uint8_t *st_dones; float *cur, *fin_vals; uint8_t *fin_mask;
};

__global__ void __launch_bounds__(256) qa_rollout_post_amp_kernel(PostAmpArgs a) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= a.N) return;
    const float *o = a.obs + (int64_t)e * a.obs_stride;
    const float label_eps = o[a.num_obs - a.dim_c - 1];
    int label = 0;                              /* argmax: first maximum, as torch.argmax */
    for (int k = 1; k < a.dim_c; ++k) if (o[a.num_obs - a.dim_c + k] > o[a.num_obs - a.dim_c + label]) label = k;
    float p[8], m = a.logits[(int64_t)e * a.dim_c];
    for (int k = 1; k < a.dim_c; ++k) m = fmaxf(m, a.logits[(int64_t)e * a.dim_c + k]);
    float z = 0.f;
    for (int k = 0; k < a.dim_c; ++k) { p[k] = expf(a.logits[(int64_t)e * a.dim_c + k] - m); z += p[k]; }
    float pm = 0.f;
    for (int k = 0; k < a.dim_c; ++k) { p[k] = fmaxf(p[k] / z, 1e-20f); pm = fmaxf(pm, p[k]); }
    float lse = 0.f;
    for (int k = 0; k < a.dim_c; ++k) lse += expf(p[k] - pm);
    lse = pm + logf(lse);
    const float dd = a.d[e] - 1.0f;
    const float r_i = fmaxf(1.0f - 0.25f * dd * dd, 0.f) * a.dt;
    const float r_us = -fabsf(a.eps[e] - label_eps) * a.dt;
    const float r_ss = (p[label] - lse) * a.dt;
    const float r_t = a.rew[e];
    const float total = a.c_i * r_i + a.c_us * r_us + a.c_ss * r_ss + a.c_t * r_t;
    const bool done = a.reset[e] > 0;
    a.st_rewards[e] = total + a.gamma * a.values[e] * (a.time_out[e] ? 1.0f : 0.0f);
    a.st_dones[e] = done ? 1 : 0;
    if (a.cur) {
        const float add[6] = {total, r_i, r_us, r_ss, r_t, 1.0f};
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const float c = a.cur[(int64_t)k * a.N + e] + add[k];
            a.fin_vals[(int64_t)k * a.N + e] = c;
            a.cur[(int64_t)k * a.N + e] = done ? 0.f : c;
        }
        a.fin_mask[e] = done ? 1 : 0;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// The head losses of the SS-InfoGAIL discriminator step (bbc/rsl_rl/algorithms/gail.py:452-520) and their gradient with
// respect to the three heads' outputs, one pass over the batch [labelled expert | policy | unlabelled expert]:
//   ss   = mean_lb  CE(log_softmax(c), label)                 (cross-entropy applied to the already-softmaxed c, as the reference)
//   info = mean_ulb -sum_j c_j log(c_j + 1e-20)
//   disc = 1/2 (mean_pi (d + 1)^2 + mean_ulb (d - 1)^2)        (MSELoss variant)
//   us   = mean_pi |eps - eps_label|
//   loss = c_ss ss + c_info info + c_disc disc + c_us us       (gradient penalty, logit and weight regularisers stay in autograd)
// plus the four accuracies the runner logs and mean_ulb c (the prior EMA input).  Eager PyTorch: ~90 launches on
// (3,684 x <=5) tensors per step, 80 steps per iteration.
constexpr int DISC_BLOCK = 256;
constexpr int DISC_SUMS = 13;         // ss, info, disc_pi, disc_exp, us, acc_lb, acc_pi, acc_exp, acc_ulb (counts), pred_mean[4] + the 5th below
struct DiscArgs {
    const float *d, *eps, *c; const int64_t *label; const float *pol_eps, *pol_c; const float *info_coef;
    float *gd, *geps, *gc; float *partial;
    int b_lb, b_pi, b_ulb;
    float c_ss, c_disc, c_us;
    int from_logits;        // r6 (qa_disc_loss_logits): `c` holds the class LOGITS; the softmax (discriminator.py:66) and its backward happen here, gc = d loss / d logits
};

__global__ void __launch_bounds__(DISC_BLOCK) qa_disc_loss_kernel(DiscArgs a) {
    __shared__ float s_w[DISC_BLOCK / 64][DISC_SUMS + 1];
    const int i = blockIdx.x * DISC_BLOCK + threadIdx.x, B = a.b_lb + a.b_pi + a.b_ulb;
    float part[DISC_SUMS + 1];
#pragma unroll
    for (int k = 0; k <= DISC_SUMS; ++k) part[k] = 0.f;
    if (i < B) {
        const float c_info = a.info_coef[0];
        float c[5], craw[5];
        bool pass[5];               /* torch.clamp(c, 1e-20) of Discriminator.forward applied here: clamped entries pass no gradient */
#pragma unroll
        for (int j = 0; j < 5; ++j) craw[j] = a.c[(int64_t)i * 5 + j];
        if (a.from_logits) {        /* softmax over the 5 class logits: max, exp, sum, divide (torch.softmax's steps) */
            float zm = craw[0];
#pragma unroll
            for (int j = 1; j < 5; ++j) zm = fmaxf(zm, craw[j]);
            float zs = 0.f;
#pragma unroll
            for (int j = 0; j < 5; ++j) { craw[j] = expf(craw[j] - zm); zs += craw[j]; }
#pragma unroll
            for (int j = 0; j < 5; ++j) craw[j] = craw[j] / zs;
        }
#pragma unroll
        for (int j = 0; j < 5; ++j) { pass[j] = craw[j] >= 1e-20f; c[j] = fmaxf(craw[j], 1e-20f); }
        int arg = 0;
#pragma unroll
        for (int j = 1; j < 5; ++j) if (c[j] > c[arg]) arg = j;
        float gc[5] = {0.f, 0.f, 0.f, 0.f, 0.f}, gd = 0.f, ge = 0.f;
        const float d = a.d[i];
        if (i < a.b_lb) {                                   // labelled expert: cross-entropy on log_softmax(c)
            float m = c[0];
#pragma unroll
            for (int j = 1; j < 5; ++j) m = fmaxf(m, c[j]);
            float e[5], se = 0.f;
#pragma unroll
            for (int j = 0; j < 5; ++j) { e[j] = expf(c[j] - m); se += e[j]; }
            const int lab = (int)a.label[i];
            const float lse = m + logf(se), inv = 1.0f / se, w = a.c_ss / (float)a.b_lb;
            float cl = c[0];
#pragma unroll
            for (int j = 1; j < 5; ++j) cl = lab == j ? c[j] : cl;
            part[0] = lse - cl;
#pragma unroll
            for (int j = 0; j < 5; ++j) gc[j] = w * (e[j] * inv - (lab == j ? 1.0f : 0.0f));
            part[5] = arg == lab ? 1.0f : 0.0f;
        } else if (i < a.b_lb + a.b_pi) {                   // policy samples: discriminator target -1, eps regression
            const int r = i - a.b_lb;
            part[2] = (d + 1.0f) * (d + 1.0f);
            gd = a.c_disc * (d + 1.0f) / (float)a.b_pi;
            const float de = a.eps[i] - a.pol_eps[r];
            part[4] = fabsf(de);
            ge = a.c_us * (de > 0.f ? 1.0f : (de < 0.f ? -1.0f : 0.0f)) / (float)a.b_pi;
            part[6] = d < 0.f ? 1.0f : 0.0f;
            int pa = 0;
#pragma unroll
            for (int j = 1; j < 5; ++j) if (a.pol_c[(int64_t)r * 5 + j] > a.pol_c[(int64_t)r * 5 + pa]) pa = j;
            part[8] = arg == pa ? 1.0f : 0.0f;
        } else {                                            // unlabelled expert: target +1, information maximisation, prior mean
            part[3] = (d - 1.0f) * (d - 1.0f);
            gd = a.c_disc * (d - 1.0f) / (float)a.b_ulb;
            const float w = c_info / (float)a.b_ulb;
            float h = 0.f;
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                const float l = logf(c[j] + 1e-20f);
                h -= c[j] * l;
                gc[j] = -w * (l + c[j] / (c[j] + 1e-20f));
                part[9 + j] = c[j];
            }
            part[1] = h;
            part[7] = d > 0.f ? 1.0f : 0.0f;
        }
        a.gd[i] = gd; a.geps[i] = ge;
#pragma unroll
        for (int j = 0; j < 5; ++j) gc[j] = pass[j] ? gc[j] : 0.f;
        if (a.from_logits) {        /* softmax backward: d loss / d z_j = p_j (g_j - sum_k g_k p_k), p the UNclamped softmax output */
            float dot = 0.f;
#pragma unroll
            for (int j = 0; j < 5; ++j) dot += gc[j] * craw[j];
#pragma unroll
            for (int j = 0; j < 5; ++j) gc[j] = craw[j] * (gc[j] - dot);
        }
#pragma unroll
        for (int j = 0; j < 5; ++j) a.gc[(int64_t)i * 5 + j] = gc[j];
    }
#pragma unroll
    for (int k = 0; k <= DISC_SUMS; ++k) {
        const float s = wave_sum(part[k]);
        if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6][k] = s;
    }
    __syncthreads();
    if (threadIdx.x <= DISC_SUMS) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < DISC_BLOCK / 64; ++w) t += s_w[w][threadIdx.x];
        a.partial[(int64_t)blockIdx.x * (DISC_SUMS + 1) + threadIdx.x] = t;
    }
}

// out[16] = {loss, ss, info, disc, us, acc_lb, acc_pi, acc_exp, acc_ulb, pred_mean[5], 0, 0}
__global__ void qa_disc_finish_kernel(const float *partial, int nblocks, int b_lb, int b_pi, int b_ulb, float c_ss, const float *info_coef,
                                      float c_disc, float c_us, float *out) {
    __shared__ double s_tot[DISC_SUMS + 1];
    const int t = threadIdx.x;
    if (t <= DISC_SUMS) {
        double acc = 0.0;
        for (int b = 0; b < nblocks; ++b) acc += (double)partial[(int64_t)b * (DISC_SUMS + 1) + t];
        s_tot[t] = acc;
    }
    __syncthreads();
    if (t == 0) {
        const double ss = s_tot[0] / b_lb, info = s_tot[1] / b_ulb, disc = 0.5 * (s_tot[2] / b_pi + s_tot[3] / b_ulb), us = s_tot[4] / b_pi;
        out[0] = (float)(c_ss * ss + (double)info_coef[0] * info + c_disc * disc + c_us * us);
        out[1] = (float)ss; out[2] = (float)info; out[3] = (float)disc; out[4] = (float)us;
        out[5] = (float)(s_tot[5] / b_lb); out[6] = (float)(s_tot[6] / b_pi); out[7] = (float)(s_tot[7] / b_ulb); out[8] = (float)(s_tot[8] / b_pi);
        for (int j = 0; j < 5; ++j) out[9 + j] = (float)(s_tot[9 + j] / b_ulb);
        out[14] = 0.f; out[15] = 0.f;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Tail of a discriminator step: the logged values that are sums of squares -- gradient penalty sum g^2 / rows (gail.py:486-492), logit
// regulariser |W_out|^2 and weight decay sum |W_i|^2 (:497-504) -- assembled with the head statistics of qa_disc_loss into the 11 values
// update_ss_info_gail returns, added to the update's accumulator, and the device-side step counter of the recorded step bumped.  Eager
// PyTorch: pow + sum + div, _foreach_norm (2 launches) + stack + pow + sum + index, stack of 11 scalars, add, counter add = 13 launches
// of a step that is a chain of ~75 launch-latency-sized kernels.  16 workgroups per tensor write partial sums; the last one to arrive
// (ticket) adds them in index order: fixed order, bit-reproducible.
constexpr int TAIL_MAX = 8, TAIL_WG = 16;
struct TailArgs { const float *t[TAIL_MAX]; int64_t n[TAIL_MAX]; int nt; const float *hs; float inv_rows; float *out, *acc; int64_t *step; float *prior; int prior_dim; float prior_c; float *partial; unsigned *ticket; };

__global__ void __launch_bounds__(256) qa_disc_step_tail_kernel(TailArgs a) {
    __shared__ float red[256];
    __shared__ bool last;
    const int ti = blockIdx.x / TAIL_WG, w = blockIdx.x % TAIL_WG, tid = threadIdx.x;
    const float *p = a.t[ti];
    const int64_t n = a.n[ti];
    const int64_t per = ((n + TAIL_WG - 1) / TAIL_WG + 3) & ~(int64_t)3, lo = w * per, hi = lo + per < n ? lo + per : n;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (((uintptr_t)p & 15) == 0) {
        int64_t i = lo + 4 * tid;
        for (; i + 3 < hi; i += 1024) { const float4 v = *(const float4 *)(p + i); s0 = fmaf(v.x, v.x, s0); s1 = fmaf(v.y, v.y, s1); s2 = fmaf(v.z, v.z, s2); s3 = fmaf(v.w, v.w, s3); }
        for (; i < hi; ++i) s0 = fmaf(p[i], p[i], s0);          // the tensor's last (< 4) elements: one thread
    } else {
        for (int64_t i = lo + tid; i < hi; i += 256) s0 = fmaf(p[i], p[i], s0);
    }
    red[tid] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) { if (tid < k) red[tid] += red[tid + k]; __syncthreads(); }
    if (tid == 0) {
        a.partial[blockIdx.x] = red[0];
        __threadfence();
        last = atomicAdd(a.ticket, 1u) == (unsigned)(a.nt * TAIL_WG - 1);
    }
    __syncthreads();
    if (!last) return;
    __threadfence();
    // r6: the last workgroup's tail spread over its threads -- one thread doing 64 atomic loads, 11 read-modify-writes behind stores it cannot tell apart from them
    // and the prior's 5 was ~20 dependent round trips; the sums keep their order (thread 0 adds the staged partials as before)
    __shared__ float s_part[TAIL_MAX * TAIL_WG];
    __shared__ float s_o[11];
    if (tid < a.nt * TAIL_WG) s_part[tid] = __hip_atomic_load(a.partial + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (tid == 0) {
        double tot[TAIL_MAX];
        for (int t = 0; t < a.nt; ++t) {
            double s = 0.0;
            for (int k = 0; k < TAIL_WG; ++k) s += (double)s_part[t * TAIL_WG + k];
            tot[t] = s;
        }
        double wd = 0.0;
        for (int t = 1; t < a.nt; ++t) wd += tot[t];
        s_o[4] = (float)(tot[0] * (double)a.inv_rows); s_o[5] = (float)tot[a.nt - 1]; s_o[6] = (float)wd;
        if (a.step) a.step[0] += 1;
        *a.ticket = 0u;                 // ready for the next launch (replays of a recorded step included)
    }
    if (tid >= 32 && tid < 32 + 4) s_o[tid - 32] = a.hs[1 + tid - 32];                 // o[0..3] = hs[1..4]
    if (tid >= 40 && tid < 40 + 4) s_o[7 + tid - 40] = a.hs[5 + tid - 40];             // o[7..10] = hs[5..8]
    // the class prior's EMA towards this step's mean class probabilities of the unlabelled batch (gail.py:463-464)
    if (a.prior && tid >= 64 && tid < 64 + a.prior_dim) { const int k = tid - 64; a.prior[k] = fmaf(a.prior_c, a.hs[9 + k], a.prior[k] * (1.0f - a.prior_c)); }
    __syncthreads();
    if (tid < 11) { const float o = s_o[tid]; a.out[tid] = o; if (a.acc) a.acc[tid] += o; }
}

// ---------------------------------------------------------------------------------------------------------------
// Discriminator input preparation (bbc/rsl_rl/algorithms/discriminator.py:77-87 + utils.py:97-103) for up to three
// (rows_i, dim) batches written one under the other into one (sum rows_i, dim) matrix:
//   y = clamp(((x * (task_mask ? w : 1)) * frame_mult - (float)mean) / sqrt((float)(var + eps)), -clip, clip)
// Eager PyTorch: clone, two slice multiplies, a reshape multiply, the normaliser's four ops, per batch, plus the cat.
struct PrepArgs { const float *src[3]; int64_t rows[3]; int k, dim; const float *task_mask, *frame_mult, *task_w; const double *mean, *var; float eps, clip; float *out; };

__global__ void __launch_bounds__(256) qa_disc_prepare_kernel(PrepArgs a) {
    const int64_t total = (a.rows[0] + a.rows[1] + a.rows[2]) * a.dim;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int64_t r = i / a.dim; const int c = (int)(i - r * a.dim);
    const float *src; int64_t rr = r;
    if (rr < a.rows[0]) src = a.src[0];
    else if ((rr -= a.rows[0]) < a.rows[1]) src = a.src[1];
    else { rr -= a.rows[1]; src = a.src[2]; }
    float x = src[rr * a.dim + c];
    if (a.task_w && a.task_mask[c] != 0.f) x *= a.task_w[0];
    x *= a.frame_mult[c];
    if (a.mean) {
        const float m = (float)a.mean[c], sd = sqrtf((float)(a.var[c] + (double)a.eps));
        x = fminf(fmaxf((x - m) / sd, -a.clip), a.clip);
    }
    a.out[i] = x;
}

// The sampling front of a recorded discriminator step: minibatch `*block_dev` of the update's index tables is read straight from the
// replay ring / the two mocap tables INTO the prepared (3 mb, dim) matrix (same arithmetic as qa_disc_prepare_kernel, element for element),
// with the policy rows' latent targets and the labelled rows' classes alongside.  Replaces three row gathers, an index_select and the prepare
// launch of a step that is a chain of launch-latency-sized kernels.
struct SampleArgs {
    qa_disc_sample_io io;
    int dim, c_dim;
    const float *task_mask, *frame_mult, *task_w; const double *mean, *var; float eps, clip; float *out;
    int64_t main_blocks;
};

__global__ void __launch_bounds__(256) qa_disc_sample_prepare_kernel(SampleArgs a) {
    const qa_disc_sample_io &io = a.io;
    const int64_t blk = io.block_dev[0];
    if ((int64_t)blockIdx.x < a.main_blocks) {
        const int64_t total = (io.rows[0] + io.rows[1] + io.rows[2]) * a.dim;
        const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
        if (i >= total) return;
        const int64_t r = i / a.dim; const int c = (int)(i - r * a.dim);
        int b = 0; int64_t rr = r;
        if (rr >= io.rows[0]) { rr -= io.rows[0]; b = 1; if (rr >= io.rows[1]) { rr -= io.rows[1]; b = 2; } }
        const int64_t row = io.index[b][blk * io.rows[b] + rr];
        float x = io.src[b][row * a.dim + c];
        if (a.task_w && a.task_mask[c] != 0.f) x *= a.task_w[0];
        x *= a.frame_mult[c];
        if (a.mean) {
            const float m = (float)a.mean[c], sd = sqrtf((float)(a.var[c] + (double)a.eps));
            x = fminf(fmaxf((x - m) / sd, -a.clip), a.clip);
        }
        a.out[i] = x;
        return;
    }
    // the small rows: one thread per policy sample / labelled sample
    const int64_t j = ((int64_t)blockIdx.x - a.main_blocks) * 256 + threadIdx.x;
    if (j < io.rows[1] && io.eps_src) {
        const int64_t row = io.index[1][blk * io.rows[1] + j];
        io.eps_out[j] = io.eps_src[row];
        for (int k = 0; k < a.c_dim; ++k) io.c_out[j * a.c_dim + k] = io.c_src[row * a.c_dim + k];
    }
    if (j < io.rows[0] && io.label_src) io.label_out[j] = io.label_src[blk * io.rows[0] + j];
}

}  // namespace

extern thread_local char qa_err_buf[512];
#define g_lerr qa_err_buf

extern "C" {

int64_t qa_ppo_loss_scratch_bytes(int64_t B) { return B <= 0 ? -1 : (int64_t)sizeof(float) * PPO_SUMS * ((B + PPO_BLOCK - 1) / PPO_BLOCK); }

int qa_ppo_loss(const float *mu, const float *std, const float *value, const float *actions, const float *old_logp,
                const float *old_mu, const float *old_sigma, const float *advantages, const float *returns,
                const float *target_values, int64_t B, int32_t num_actions, float clip, float c_surr, float c_value,
                float c_bound, float c_entropy, int32_t clipped_value, float *dmu, float *dstd, float *dvalue, float *out,
                void *scratch, int64_t scratch_bytes, void *stream) {
    if (!mu || !std || !value || !actions || !old_logp || !old_mu || !old_sigma || !advantages || !returns || !target_values ||
        !dmu || !dstd || !dvalue || !out || !scratch || B <= 0) { snprintf(g_lerr, sizeof(g_lerr), "qa_ppo_loss: null pointer or empty batch"); return QA_E_ARG; }
    if (num_actions != PPO_D) { snprintf(g_lerr, sizeof(g_lerr), "qa_ppo_loss: num_actions must be %d", PPO_D); return QA_E_ARG; }
    if ((((uintptr_t)mu | (uintptr_t)actions | (uintptr_t)old_mu | (uintptr_t)old_sigma | (uintptr_t)dmu) & 15) != 0) {
        snprintf(g_lerr, sizeof(g_lerr), "qa_ppo_loss: (B,12) tensors must be 16-byte aligned"); return QA_E_ARG; }
    hipStream_t st = (hipStream_t)stream;
    const int blocks = (int)((B + PPO_BLOCK - 1) / PPO_BLOCK);
    if (scratch_bytes < qa_ppo_loss_scratch_bytes(B)) { snprintf(g_lerr, sizeof(g_lerr), "qa_ppo_loss: scratch too small"); return QA_E_ARG; }
    PpoArgs a{mu, std, value, actions, old_logp, old_mu, old_sigma, advantages, returns, target_values, dmu, dvalue, (float *)scratch,
              B, clip, c_surr, c_value, c_bound, c_entropy, clipped_value};
    hipLaunchKernelGGL(qa_ppo_loss_kernel, dim3(blocks), dim3(PPO_BLOCK), 0, st, a);
    hipLaunchKernelGGL(qa_ppo_finish_kernel, dim3(1), dim3(256), 0, st, (const float *)scratch, blocks, B, c_surr, c_value, c_bound, c_entropy, out, dstd);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { snprintf(g_lerr, sizeof(g_lerr), "qa_ppo_loss: %s", hipGetErrorString(e)); return QA_E_DEVICE; }
    return QA_OK;
}

int64_t qa_hybrid_ppo_loss_scratch_bytes(int64_t B) { return B <= 0 ? -1 : (int64_t)sizeof(float) * HYB_SUMS * ((B + PPO_BLOCK - 1) / PPO_BLOCK); }

int qa_hybrid_ppo_loss(const float *logits, const float *mean, const float *std, const float *value, const float *actions, const float *old_logp_d,
                       const float *old_logp_c, const float *old_mu, const float *old_sigma, const float *advantages, const float *returns,
                       const float *target_values, int64_t B, int32_t num_d, int32_t num_c, float clip, float c_value, float c_entropy,
                       int32_t clipped_value, float *dlogits, float *dmean, float *dstd, float *dvalue, float *out, void *scratch,
                       int64_t scratch_bytes, void *stream) {
    if (!logits || !mean || !std || !value || !actions || !old_logp_d || !old_logp_c || !old_mu || !old_sigma || !advantages || !returns ||
        !target_values || !dlogits || !dmean || !dstd || !dvalue || !out || !scratch || B <= 0) {
        snprintf(g_lerr, sizeof(g_lerr), "qa_hybrid_ppo_loss: null pointer or empty batch"); return QA_E_ARG; }
    if (num_d != HYB_ND || num_c != HYB_NC) { snprintf(g_lerr, sizeof(g_lerr), "qa_hybrid_ppo_loss: built for num_d = %d, num_c = %d", HYB_ND, HYB_NC); return QA_E_ARG; }
    if (scratch_bytes < qa_hybrid_ppo_loss_scratch_bytes(B)) { snprintf(g_lerr, sizeof(g_lerr), "qa_hybrid_ppo_loss: scratch too small"); return QA_E_ARG; }
    hipStream_t st = (hipStream_t)stream;
    const int blocks = (int)((B + PPO_BLOCK - 1) / PPO_BLOCK);
    HybArgs a{logits, mean, std, value, actions, old_logp_d, old_logp_c, old_mu, old_sigma, advantages, returns, target_values, dlogits, dmean, dvalue,
              (float *)scratch, B, clip, c_value, c_entropy, clipped_value};
    hipLaunchKernelGGL(qa_hybrid_ppo_loss_kernel, dim3(blocks), dim3(PPO_BLOCK), 0, st, a);
    hipLaunchKernelGGL(qa_hybrid_ppo_finish_kernel, dim3(1), dim3(256), 0, st, (const float *)scratch, blocks, B, c_value, c_entropy, out, dstd);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { snprintf(g_lerr, sizeof(g_lerr), "qa_hybrid_ppo_loss: %s", hipGetErrorString(e)); return QA_E_DEVICE; }
    return QA_OK;
}

int64_t qa_elu_backward_bias_scratch_bytes(int64_t rows, int32_t cols) {
    return (rows <= 0 || cols <= 0) ? -1 : (int64_t)sizeof(float) * cols * ((rows + ELU_ROWS - 1) / ELU_ROWS);
}

int qa_elu_backward_bias(const float *grad_out, const float *out, float *grad_in, float *grad_bias, int64_t rows, int32_t cols,
                         float alpha, void *scratch, int64_t scratch_bytes, void *stream) {
    // grad_bias == NULL (r5): the column sums stay in parts -- scratch holds ceil(rows / 64) rows of `cols` partial sums, to be added in row
    // order by whoever consumes the bias gradient (qa_clip_adam_step_reduce / qa_grad_reduce); the finish launch is not made
    if (!grad_out || !out || !grad_in || !scratch || rows <= 0 || cols <= 0) { snprintf(g_lerr, sizeof(g_lerr), "qa_elu_backward_bias: bad argument"); return QA_E_ARG; }
    if (scratch_bytes < qa_elu_backward_bias_scratch_bytes(rows, cols)) { snprintf(g_lerr, sizeof(g_lerr), "qa_elu_backward_bias: scratch too small"); return QA_E_ARG; }
    hipStream_t st = (hipStream_t)stream;
    const int nb = (int)((rows + ELU_ROWS - 1) / ELU_ROWS);
    float *partial = (float *)scratch;
    if (cols > 128) hipLaunchKernelGGL(qa_elu_bwd_bias_kernel<256>, dim3(nb), dim3(256), 0, st, grad_out, out, grad_in, partial, rows, (int)cols, alpha);
    else if (cols > 64) hipLaunchKernelGGL(qa_elu_bwd_bias_kernel<128>, dim3(nb), dim3(256), 0, st, grad_out, out, grad_in, partial, rows, (int)cols, alpha);
    else if (cols > 32) hipLaunchKernelGGL(qa_elu_bwd_bias_kernel<64>, dim3(nb), dim3(256), 0, st, grad_out, out, grad_in, partial, rows, (int)cols, alpha);
    else hipLaunchKernelGGL(qa_elu_bwd_bias_kernel<32>, dim3(nb), dim3(256), 0, st, grad_out, out, grad_in, partial, rows, (int)cols, alpha);
    if (grad_bias) hipLaunchKernelGGL(qa_colsum_finish_kernel, dim3((cols + 31) / 32), dim3(256), 0, st, (const float *)partial, nb, (int)cols, grad_bias);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { snprintf(g_lerr, sizeof(g_lerr), "qa_elu_backward_bias: %s", hipGetErrorString(e)); return QA_E_DEVICE; }
    return QA_OK;
}

static int narrow_slabs(int64_t rows, int *slab_rows) {
    int sr = (int)((rows + 255) / 256);                     // ~256 workgroups: one per CU
    sr = sr < 8 ? 8 : (sr > NARROW_MAX_SLAB ? NARROW_MAX_SLAB : sr);
    *slab_rows = sr;
    return (int)((rows + sr - 1) / sr);
}
int64_t qa_narrow_wgrad_scratch_bytes(int64_t rows, int32_t out_features, int32_t in_features) {
    int sr; const int nb = narrow_slabs(rows, &sr);
    return (int64_t)nb * ((int64_t)out_features * in_features + out_features) * 4;
}
int qa_narrow_wgrad(const float *grad_out, const float *x, int64_t rows, int32_t O, int32_t K, float *grad_weight, float *grad_bias,
                    void *scratch, int64_t scratch_bytes, void *stream) {
    if (!grad_out || !x || !grad_weight || !grad_bias || !scratch || rows <= 0 || O <= 0 || O > QA_NARROW_MAX_OUT || K <= 0) {
        snprintf(g_lerr, sizeof(g_lerr), "qa_narrow_wgrad: bad argument (1 <= out_features <= %d)", QA_NARROW_MAX_OUT); return QA_E_ARG; }
    if (scratch_bytes < qa_narrow_wgrad_scratch_bytes(rows, O, K)) { snprintf(g_lerr, sizeof(g_lerr), "qa_narrow_wgrad: scratch too small"); return QA_E_ARG; }
    hipStream_t st = (hipStream_t)stream;
    int sr; const int nb = narrow_slabs(rows, &sr);
    float *partial = (float *)scratch;
    if (O <= 1) hipLaunchKernelGGL(qa_narrow_wgrad_kernel<1>, dim3(nb), dim3(256), 0, st, grad_out, x, rows, sr, (int)O, (int)K, partial);
    else if (O <= 4) hipLaunchKernelGGL(qa_narrow_wgrad_kernel<4>, dim3(nb), dim3(256), 0, st, grad_out, x, rows, sr, (int)O, (int)K, partial);
    else if (O <= 16) hipLaunchKernelGGL(qa_narrow_wgrad_kernel<16>, dim3(nb), dim3(256), 0, st, grad_out, x, rows, sr, (int)O, (int)K, partial);
    else hipLaunchKernelGGL(qa_narrow_wgrad_kernel<32>, dim3(nb), dim3(256), 0, st, grad_out, x, rows, sr, (int)O, (int)K, partial);
    // [out * in | out] per slab -> grad_weight, grad_bias: two fixed-order column-sum finishes over the same partials
    const int cols = O * K + O;
    hipLaunchKernelGGL(qa_colsum_finish2_kernel, dim3((cols + 31) / 32), dim3(256), 0, st, (const float *)partial, nb, cols, O * K, grad_weight, grad_bias);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { snprintf(g_lerr, sizeof(g_lerr), "qa_narrow_wgrad: %s", hipGetErrorString(e)); return QA_E_DEVICE; }
    return QA_OK;
}

int qa_normalizer_update(const float *const *batches, const int64_t *rows, int32_t num_batches, int32_t dim,
                         double *mean, double *var, double *count, void *stream) {
    if (!batches || !rows || !mean || !var || !count || num_batches <= 0 || num_batches > NORM_MAX_BATCHES || dim <= 0 || dim > 128) {
        snprintf(g_lerr, sizeof(g_lerr), "qa_normalizer_update: bad argument (1..4 batches, dim <= 128)"); return QA_E_ARG; }
    NormArgs a{};
    for (int b = 0; b < num_batches; ++b) {
        if (!batches[b] || rows[b] <= 0) { snprintf(g_lerr, sizeof(g_lerr), "qa_normalizer_update: empty batch"); return QA_E_ARG; }
        a.batch[b] = batches[b]; a.n[b] = rows[b];
    }
    a.k = num_batches; a.d = dim; a.mean = mean; a.var = var; a.count = count;
    hipLaunchKernelGGL(qa_normalizer_update_kernel, dim3((dim + 15) / 16), dim3(1024), 0, (hipStream_t)stream, a);
    hipLaunchKernelGGL(qa_normalizer_count_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { snprintf(g_lerr, sizeof(g_lerr), "qa_normalizer_update: %s", hipGetErrorString(e)); return QA_E_DEVICE; }
    return QA_OK;
}

int qa_normalizer_apply(const float *x, float *y, int64_t rows, int32_t dim, const double *mean, const double *var,
                        float epsilon, float clip, void *stream) {
    if (!x || !y || !mean || !var || rows <= 0 || dim <= 0) { snprintf(g_lerr, sizeof(g_lerr), "qa_normalizer_apply: bad argument"); return QA_E_ARG; }
    const int64_t total = rows * dim;
    hipLaunchKernelGGL(qa_normalizer_apply_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, y, total, (int)dim, mean, var, epsilon, clip);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { snprintf(g_lerr, sizeof(g_lerr), "qa_normalizer_apply: %s", hipGetErrorString(e)); return QA_E_DEVICE; }
    return QA_OK;
}

static int clip_adam_launch(float *const *params, const float *const *grads_dev, const float *const *grads_host, float *const *exp_avg,
                            float *const *exp_avg_sq, float *const *steps, int32_t num_tensors, const int32_t *chunk_tensor, const int32_t *chunk_start,
                            const int32_t *chunk_len, int32_t num_chunks, const float *weight_decay, const float *lr, float beta1, float beta2, float eps,
                            float max_norm, float *scratch, int64_t scratch_floats, void *stream, const char *who,
                            const float *const *red_src = nullptr, const int64_t *red_stride = nullptr, const int32_t *red_parts = nullptr, const qa_adam_pair *pair = nullptr) {
    if (!params || (!grads_dev && !grads_host) || !exp_avg || !exp_avg_sq || !steps || !chunk_tensor || !chunk_start || !chunk_len || !weight_decay || !lr ||
        !scratch || num_tensors <= 0 || num_chunks <= 0 || scratch_floats < 4 + (int64_t)num_chunks || (grads_host && num_tensors > QA_ADAM_MAX_INLINE)) {
        snprintf(g_lerr, sizeof(g_lerr), "%s: bad argument", who); return QA_E_ARG; }
    AdamArgs a{};
    a.params = params; a.grads = grads_dev; a.exp_avg = exp_avg; a.exp_avg_sq = exp_avg_sq; a.steps = steps; a.chunk_tensor = chunk_tensor; a.chunk_start = chunk_start;
    a.chunk_len = chunk_len; a.weight_decay = weight_decay; a.lr = lr; a.scratch = scratch; a.num_chunks = num_chunks; a.num_tensors = num_tensors;
    a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.max_norm = max_norm; a.inline_grads = grads_host ? 1 : 0;
    if (pair) {
        if (!grads_host || pair->split_tensor <= 0 || pair->split_tensor >= num_tensors || pair->split_chunk <= 0 || pair->split_chunk >= num_chunks || !pair->lr2 ||
            !(max_norm > 0.f) || !(pair->max_norm2 > 0.f) || scratch_floats < 4 + (int64_t)num_chunks + 1 + 4) {
            snprintf(g_lerr, sizeof(g_lerr), "%s: two clipping optimisers (max_norm > 0 both), the host pointer form, 0 < split < count and scratch >= num_chunks + 9 floats expected", who);
            return QA_E_ARG; }
        a.split_chunk = pair->split_chunk; a.split_tensor = pair->split_tensor; a.lr2 = pair->lr2; a.max_norm2 = pair->max_norm2;
        a.kl = pair->kl; a.desired_kl = pair->desired_kl; a.kl_factor = pair->kl_factor; a.lr_min = pair->lr_min; a.lr_max = pair->lr_max;
    }
    if (grads_host) for (int t = 0; t < num_tensors; ++t) a.gin[t] = grads_host[t];
    if (red_src) {
        if (!grads_host || !(max_norm > 0.f) || !red_parts || !red_stride) {
            snprintf(g_lerr, sizeof(g_lerr), "%s: gradients in parts need the host pointer form and a clipping step (max_norm > 0): the parts are added by the sums-of-squares pass", who);
            return QA_E_ARG; }
        for (int t = 0; t < num_tensors; ++t) {
            if (red_parts[t] < 0 || (red_parts[t] > 0 && (!red_src[t] || red_stride[t] <= 0))) { snprintf(g_lerr, sizeof(g_lerr), "%s: bad parts record of tensor %d", who, t); return QA_E_ARG; }
            a.red_src[t] = red_src[t]; a.red_stride[t] = red_stride[t]; a.red_parts[t] = red_parts[t];
            if (red_parts[t] > 0) a.has_reduce = 1;
        }
    }
    hipStream_t st = (hipStream_t)stream;
    if (!(max_norm > 0.f) && scratch_floats >= 5 + (int64_t)num_chunks) {
        // no clipping and room for the arrival counter (scratch[4 + num_chunks], zero at the first call: the kernel leaves it at zero)
        hipLaunchKernelGGL(qa_adam_update_kernel<true>, dim3(num_chunks), dim3(256), 0, st, a);
    } else {
        if (max_norm > 0.f) hipLaunchKernelGGL(qa_adam_sumsq_kernel, dim3(num_chunks), dim3(256), 0, st, a);
        hipLaunchKernelGGL(qa_adam_finalize_kernel, dim3(1), dim3(256), 0, st, a);
        hipLaunchKernelGGL(qa_adam_update_kernel<false>, dim3(num_chunks), dim3(256), 0, st, a);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { snprintf(g_lerr, sizeof(g_lerr), "%s: %s", who, hipGetErrorString(e)); return QA_E_DEVICE; }
    return QA_OK;
}

int qa_clip_adam_step(float *const *params, const float *const *grads, float *const *exp_avg, float *const *exp_avg_sq,
                      float *const *steps, int32_t num_tensors, const int32_t *chunk_tensor, const int32_t *chunk_start,
                      const int32_t *chunk_len, int32_t num_chunks, const float *weight_decay, const float *lr, float beta1,
                      float beta2, float eps, float max_norm, float *scratch, int64_t scratch_floats, void *stream) {
    return clip_adam_launch(params, grads, nullptr, exp_avg, exp_avg_sq, steps, num_tensors, chunk_tensor, chunk_start, chunk_len, num_chunks, weight_decay, lr,
                            beta1, beta2, eps, max_norm, scratch, scratch_floats, stream, "qa_clip_adam_step");
}

int qa_clip_adam_step_hostgrads(float *const *params, const float *const *grads_host, float *const *exp_avg, float *const *exp_avg_sq,
                                float *const *steps, int32_t num_tensors, const int32_t *chunk_tensor, const int32_t *chunk_start,
                                const int32_t *chunk_len, int32_t num_chunks, const float *weight_decay, const float *lr, float beta1,
                                float beta2, float eps, float max_norm, float *scratch, int64_t scratch_floats, void *stream) {
    return clip_adam_launch(params, nullptr, grads_host, exp_avg, exp_avg_sq, steps, num_tensors, chunk_tensor, chunk_start, chunk_len, num_chunks, weight_decay,
                            lr, beta1, beta2, eps, max_norm, scratch, scratch_floats, stream, "qa_clip_adam_step_hostgrads");
}

int qa_clip_adam_step_reduce(float *const *params, const float *const *grads_host, float *const *exp_avg, float *const *exp_avg_sq,
                             float *const *steps, int32_t num_tensors, const int32_t *chunk_tensor, const int32_t *chunk_start,
                             const int32_t *chunk_len, int32_t num_chunks, const float *weight_decay, const float *lr, float beta1,
                             float beta2, float eps, float max_norm, float *scratch, int64_t scratch_floats,
                             const float *const *red_src_host, const int64_t *red_stride_host, const int32_t *red_parts_host, void *stream) {
    if (!red_src_host || !red_stride_host || !red_parts_host) { snprintf(g_lerr, sizeof(g_lerr), "qa_clip_adam_step_reduce: bad argument"); return QA_E_ARG; }
    return clip_adam_launch(params, nullptr, grads_host, exp_avg, exp_avg_sq, steps, num_tensors, chunk_tensor, chunk_start, chunk_len, num_chunks, weight_decay,
                            lr, beta1, beta2, eps, max_norm, scratch, scratch_floats, stream, "qa_clip_adam_step_reduce", red_src_host, red_stride_host, red_parts_host);
}

int qa_clip_adam_pair_step(float *const *params, const float *const *grads_host, float *const *exp_avg, float *const *exp_avg_sq,
                           float *const *steps, int32_t num_tensors, const int32_t *chunk_tensor, const int32_t *chunk_start,
                           const int32_t *chunk_len, int32_t num_chunks, const float *weight_decay, const float *lr, float beta1,
                           float beta2, float eps, float max_norm, float *scratch, int64_t scratch_floats,
                           const float *const *red_src_host, const int64_t *red_stride_host, const int32_t *red_parts_host, const qa_adam_pair *pair, void *stream) {
    if (!pair) { snprintf(g_lerr, sizeof(g_lerr), "qa_clip_adam_pair_step: bad argument"); return QA_E_ARG; }
    return clip_adam_launch(params, nullptr, grads_host, exp_avg, exp_avg_sq, steps, num_tensors, chunk_tensor, chunk_start, chunk_len, num_chunks, weight_decay,
                            lr, beta1, beta2, eps, max_norm, scratch, scratch_floats, stream, "qa_clip_adam_pair_step", red_src_host, red_stride_host, red_parts_host, pair);
}

// dst[t][i] = sum_z src[t][z stride[t] + i], z < parts[t] ascending, for a handful of tensors in ONE launch: the gradients-in-parts of a step
// whose optimiser is not qa_clip_adam_step_reduce (the data-parallel step packs its gradients into a bucket first)
struct ReduceArgs { float *dst[QA_ADAM_MAX_INLINE]; const float *src[QA_ADAM_MAX_INLINE]; int64_t stride[QA_ADAM_MAX_INLINE]; int32_t parts[QA_ADAM_MAX_INLINE];
                    int32_t numel[QA_ADAM_MAX_INLINE]; int32_t first_block[QA_ADAM_MAX_INLINE + 1]; int32_t n; };
__global__ void __launch_bounds__(256) qa_grad_reduce_kernel(ReduceArgs a) {
    __shared__ float s_q[256];
    int t = 0;
    while (t + 1 < a.n && (int)blockIdx.x >= a.first_block[t + 1]) ++t;
    const int wide = a.parts[t] > QA_REDUCE_WIDE;
    const int per = wide ? 32 : ADAM_CHUNK;
    const int s0 = ((int)blockIdx.x - a.first_block[t]) * per;
    const int n = min(per, a.numel[t] - s0);
    if (n > 0) (void)adam_reduce_chunk(a.src[t] + s0, a.stride[t], a.parts[t], a.dst[t] + s0, n, s_q);
}
int qa_grad_reduce(float *const *dst_host, const float *const *src_host, const int64_t *stride_host, const int32_t *parts_host, const int32_t *numel_host,
                   int32_t num_tensors, void *stream) {
    if (!dst_host || !src_host || !stride_host || !parts_host || !numel_host || num_tensors <= 0 || num_tensors > QA_ADAM_MAX_INLINE) {
        snprintf(g_lerr, sizeof(g_lerr), "qa_grad_reduce: bad argument"); return QA_E_ARG; }
    ReduceArgs a = {};
    a.n = num_tensors;
    int blocks = 0;
    for (int t = 0; t < num_tensors; ++t) {
        if (!dst_host[t] || !src_host[t] || stride_host[t] <= 0 || parts_host[t] <= 0 || numel_host[t] <= 0) { snprintf(g_lerr, sizeof(g_lerr), "qa_grad_reduce: bad record %d", t); return QA_E_ARG; }
        a.dst[t] = dst_host[t]; a.src[t] = src_host[t]; a.stride[t] = stride_host[t]; a.parts[t] = parts_host[t]; a.numel[t] = numel_host[t];
        a.first_block[t] = blocks;
        const int per = parts_host[t] > QA_REDUCE_WIDE ? 32 : ADAM_CHUNK;
        blocks += (numel_host[t] + per - 1) / per;
    }
    a.first_block[num_tensors] = blocks;
    hipLaunchKernelGGL(qa_grad_reduce_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { snprintf(g_lerr, sizeof(g_lerr), "qa_grad_reduce: %s", hipGetErrorString(e)); return QA_E_DEVICE; }
    return QA_OK;
}

// r6 (ABI 18): the optimiser half of a discriminator step in ONE launch -- a tensor's gradient put together from its parts (the head losses' product, the
// gradient penalty's product times alpha2, the weight regulariser reg * W) and then the tensor's Adam states applied in order: the trunk's parameters sit in
// all three of the reference's optimisers (gail.py:107-132, 518-520).  Replaces per step: qa_grad_reduce, two multi-tensor adds, one add, three
// qa_adam_update<true> launches.  Element i of a chunk belongs to thread i % 256 in both adam_reduce_chunk branches and in the update loop.
constexpr int STACK_CHUNK = 512;      // two elements per thread: ~450 workgroups for the discriminator's 190 k parameters (2048-element chunks: 47 us; 512: 38, 29.5 once the step counters
                                      // are bumped by one thread each; 256 with the moments loaded up front: 32.6)
struct StackArgs { qa_adam_stack_tensor t[QA_ADAM_STACK_MAX_TENSORS]; int32_t first_block[QA_ADAM_STACK_MAX_TENSORS + 1]; int32_t n, blocks; float beta1, beta2, eps; unsigned *ticket; };
__global__ void __launch_bounds__(256) qa_adam_stack_kernel(StackArgs a) {
    __shared__ float s_q[256];
    int t = 0;
    while (t + 1 < a.n && (int)blockIdx.x >= a.first_block[t + 1]) ++t;
    const int parts1 = a.t[t].parts1, parts2 = a.t[t].parts2, ns = a.t[t].num_states;
    const int per = (parts1 > QA_REDUCE_WIDE || parts2 > QA_REDUCE_WIDE) ? 32 : STACK_CHUNK;
    const int s0 = ((int)blockIdx.x - a.first_block[t]) * per;
    const int n = min(per, a.t[t].numel - s0);
    float *p = a.t[t].param + s0, *g = a.t[t].grad + s0, *tmp = a.t[t].tmp + s0;
    const float alpha2 = a.t[t].alpha2, reg = a.t[t].reg;
    float *m[QA_ADAM_STACK_MAX_STATES], *v[QA_ADAM_STACK_MAX_STATES], step_size[QA_ADAM_STACK_MAX_STATES], bc2s[QA_ADAM_STACK_MAX_STATES], wd[QA_ADAM_STACK_MAX_STATES];
#pragma unroll
    for (int s = 0; s < QA_ADAM_STACK_MAX_STATES; ++s) {
        const bool on = s < ns;
        m[s] = on ? a.t[t].state[s].exp_avg + s0 : nullptr; v[s] = on ? a.t[t].state[s].exp_avg_sq + s0 : nullptr;
        const float step = on ? a.t[t].state[s].step[0] + 1.0f : 1.0f;
        step_size[s] = on ? a.t[t].state[s].lr[0] / (1.0f - powf(a.beta1, step)) : 0.f;
        bc2s[s] = sqrtf(1.0f - powf(a.beta2, step)); wd[s] = a.t[t].state[s].weight_decay;
    }
    if (parts1 > 0) (void)adam_reduce_chunk(a.t[t].src1 + s0, a.t[t].stride1, parts1, g, n, s_q);
    if (parts2 > 0) (void)adam_reduce_chunk(a.t[t].src2 + s0, a.t[t].stride2, parts2, tmp, n, s_q);
    const bool rewrite = parts1 > 0 || parts2 > 0 || reg != 0.f;
    for (int i = threadIdx.x; i < n; i += 256) {
        float pi = p[i], gi = g[i];
        if (parts2 > 0) gi = fmaf(alpha2, tmp[i], gi);
        gi = fmaf(reg, pi, gi);
        if (rewrite) g[i] = gi;
#pragma unroll
        for (int s = 0; s < QA_ADAM_STACK_MAX_STATES; ++s) {
            if (s < ns) {
                const float ge = fmaf(wd[s], pi, gi);
                const float mi = a.beta1 * m[s][i] + (1.0f - a.beta1) * ge;
                const float vi = a.beta2 * v[s][i] + (1.0f - a.beta2) * ge * ge;
                m[s][i] = mi; v[s][i] = vi;
                pi = pi - step_size[s] * mi / (sqrtf(vi) / bc2s[s] + a.eps);
            }
        }
        p[i] = pi;
    }
    __syncthreads();
    __shared__ int s_last;
    if (threadIdx.x == 0) {
        __threadfence();
        s_last = atomicAdd(a.ticket, 1u) == (unsigned)(a.blocks - 1);          // every workgroup has read the old step counts by now
    }
    __syncthreads();
    if (!s_last) return;
    // one thread per (tensor, state): the counters are up to 48 separate device words, and one thread bumping them one after the other was 18 dependent
    // round trips -- most of this launch's 38 us
    const int k = threadIdx.x / QA_ADAM_STACK_MAX_STATES, st = threadIdx.x % QA_ADAM_STACK_MAX_STATES;
    if (k < a.n && st < a.t[k].num_states) a.t[k].state[st].step[0] += 1.0f;
    if (threadIdx.x == 0) *a.ticket = 0u;
}
int qa_adam_stack_step(const qa_adam_stack_tensor *tensors_host, int32_t count, float beta1, float beta2, float eps, uint32_t *ticket, void *stream) {
    if (!tensors_host || count <= 0 || count > QA_ADAM_STACK_MAX_TENSORS || !ticket) { snprintf(g_lerr, sizeof(g_lerr), "qa_adam_stack_step: 1..%d tensors and a ticket word expected", QA_ADAM_STACK_MAX_TENSORS); return QA_E_ARG; }
    StackArgs a = {};
    int blocks = 0;
    for (int t = 0; t < count; ++t) {
        const qa_adam_stack_tensor &T = tensors_host[t];
        bool ok = T.param && T.grad && T.numel > 0 && T.num_states >= 1 && T.num_states <= QA_ADAM_STACK_MAX_STATES && T.parts1 >= 0 && T.parts2 >= 0 &&
                  (T.parts1 == 0 || (T.src1 && T.stride1 >= T.numel)) && (T.parts2 == 0 || (T.src2 && T.tmp && T.stride2 >= T.numel));
        for (int s = 0; ok && s < T.num_states; ++s) ok = T.state[s].exp_avg && T.state[s].exp_avg_sq && T.state[s].step && T.state[s].lr;
        if (!ok) { snprintf(g_lerr, sizeof(g_lerr), "qa_adam_stack_step: bad record %d", t); return QA_E_ARG; }
        a.t[t] = T;
        if (!a.t[t].tmp) a.t[t].tmp = T.grad;          // never read (parts2 == 0)
        a.first_block[t] = blocks;
        const int per = (T.parts1 > QA_REDUCE_WIDE || T.parts2 > QA_REDUCE_WIDE) ? 32 : STACK_CHUNK;
        blocks += (T.numel + per - 1) / per;
    }
    a.first_block[count] = blocks; a.n = count; a.blocks = blocks; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.ticket = ticket;
    hipLaunchKernelGGL(qa_adam_stack_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { snprintf(g_lerr, sizeof(g_lerr), "qa_adam_stack_step: %s", hipGetErrorString(e)); return QA_E_DEVICE; }
    return QA_OK;
}

int64_t qa_pair_loss_scratch_bytes(int64_t rows) { return rows <= 0 ? -1 : (int64_t)sizeof(float) * ((rows + PAIR_BLOCK - 1) / PAIR_BLOCK); }

int qa_pair_loss(const float *a, const float *b, int64_t rows, int32_t cols, int64_t b_stride, int32_t mode, float *grad_a, float *out,
                 void *scratch, int64_t scratch_bytes, void *stream) {
    if (!a || !b || !grad_a || !out || !scratch || rows <= 0 || cols <= 0 || b_stride < cols || (mode != QA_PAIR_ROW_L2 && mode != QA_PAIR_MSE) ||
        scratch_bytes < qa_pair_loss_scratch_bytes(rows)) { snprintf(g_lerr, sizeof(g_lerr), "qa_pair_loss: bad argument"); return QA_E_ARG; }
    const int nb = (int)((rows + PAIR_BLOCK - 1) / PAIR_BLOCK);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(qa_pair_loss_kernel, dim3(nb), dim3(PAIR_BLOCK), 0, st, a, b, rows, (int)cols, b_stride, (int)mode, grad_a, (float *)scratch);
    const float scale = mode == QA_PAIR_ROW_L2 ? 1.0f / (float)rows : 1.0f / ((float)rows * (float)cols);
    hipLaunchKernelGGL(qa_pair_finish_kernel, dim3(1), dim3(256), 0, st, (const float *)scratch, nb, scale, out);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { snprintf(g_lerr, sizeof(g_lerr), "qa_pair_loss: %s", hipGetErrorString(e)); return QA_E_DEVICE; }
    return QA_OK;
}

static int pair_jobs_blocks(const qa_pair_job *jobs, int32_t count, int32_t *first_block) {
    int blocks = 0;
    for (int t = 0; t < count; ++t) {
        const qa_pair_job &J = jobs[t];
        if (!J.a || !J.b || !J.grad_a || !J.out || J.rows <= 0 || J.cols <= 0 || J.b_stride < J.cols || (J.mode != QA_PAIR_ROW_L2 && J.mode != QA_PAIR_MSE)) return -1;
        if (first_block) first_block[t] = blocks;
        blocks += (int)((J.rows + PAIR_BLOCK - 1) / PAIR_BLOCK);
    }
    if (first_block) first_block[count] = blocks;
    return blocks;
}
int64_t qa_pair_losses_scratch_bytes(const qa_pair_job *jobs, int32_t count) {
    if (!jobs || count <= 0 || count > QA_PAIR_MAX_JOBS) return -1;
    const int blocks = pair_jobs_blocks(jobs, count, nullptr);
    return blocks < 0 ? -1 : (int64_t)sizeof(float) * (blocks + 1);
}
int qa_pair_losses(const qa_pair_job *jobs, int32_t count, void *scratch, int64_t scratch_bytes, void *stream) {
    if (!jobs || count <= 0 || count > QA_PAIR_MAX_JOBS || !scratch) { snprintf(g_lerr, sizeof(g_lerr), "qa_pair_losses: 1..%d jobs and a scratch buffer expected", QA_PAIR_MAX_JOBS); return QA_E_ARG; }
    PairJobsArgs p{};
    const int blocks = pair_jobs_blocks(jobs, count, p.first_block);
    if (blocks < 0 || scratch_bytes < (int64_t)sizeof(float) * (blocks + 1)) { snprintf(g_lerr, sizeof(g_lerr), "qa_pair_losses: bad job or scratch too small"); return QA_E_ARG; }
    for (int t = 0; t < count; ++t) p.j[t] = jobs[t];
    p.n = count; p.ticket = (unsigned *)scratch; p.partial = (float *)scratch + 1;
    hipLaunchKernelGGL(qa_pair_losses_kernel, dim3(blocks), dim3(PAIR_BLOCK), 0, (hipStream_t)stream, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { snprintf(g_lerr, sizeof(g_lerr), "qa_pair_losses: %s", hipGetErrorString(e)); return QA_E_DEVICE; }
    return QA_OK;
}

int qa_accumulate_scalars(float *acc, const float *const *src_host, int32_t count, void *stream) {
    if (!acc || !src_host || count <= 0 || count > QA_ACC_MAX) { snprintf(g_lerr, sizeof(g_lerr), "qa_accumulate_scalars: 1..%d scalars expected", QA_ACC_MAX); return QA_E_ARG; }
    AccArgs a{};
    a.acc = acc; a.n = count;
    for (int i = 0; i < count; ++i) { if (!src_host[i]) { snprintf(g_lerr, sizeof(g_lerr), "qa_accumulate_scalars: null source %d", i); return QA_E_ARG; } a.src[i] = src_host[i]; }
    hipLaunchKernelGGL(qa_accumulate_scalars_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { snprintf(g_lerr, sizeof(g_lerr), "qa_accumulate_scalars: %s", hipGetErrorString(e)); return QA_E_DEVICE; }
    return QA_OK;
}

int qa_gather_rows(const int64_t *idx, const int64_t *idx_block, int64_t rows, int32_t num_tensors, const float *const *src, const int64_t *src_strides,
                   const int32_t *widths, float *const *dst, void *stream) {
    if (!idx || !src || !src_strides || !widths || !dst || rows <= 0 || num_tensors <= 0 || num_tensors > QA_GATHER_MAX) {
        snprintf(g_lerr, sizeof(g_lerr), "qa_gather_rows: bad argument (1..%d tensors)", QA_GATHER_MAX); return QA_E_ARG; }
    GatherArgs a{};
    for (int t = 0; t < num_tensors; ++t) {
        if (!src[t] || !dst[t] || widths[t] <= 0 || src_strides[t] < widths[t]) { snprintf(g_lerr, sizeof(g_lerr), "qa_gather_rows: tensor %d is malformed", t); return QA_E_ARG; }
        a.src[t] = src[t]; a.dst[t] = dst[t]; a.width[t] = widths[t]; a.src_stride[t] = src_strides[t];
    }
    a.idx = idx; a.idx_block = idx_block; a.rows = rows; a.n = num_tensors;
    hipLaunchKernelGGL(qa_gather_rows_kernel, dim3((unsigned)((rows + GATHER_ROWS - 1) / GATHER_ROWS)), dim3(256), 0, (hipStream_t)stream, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { snprintf(g_lerr, sizeof(g_lerr), "qa_gather_rows: %s", hipGetErrorString(e)); return QA_E_DEVICE; }
    return QA_OK;
}

int qa_kl_lr_rule(const float *kl, float desired_kl, float factor, float lr_min, float lr_max, float *lr, void *stream) {
    if (!kl || !lr || !(desired_kl > 0.f) || !(factor > 1.f) || !(lr_min > 0.f) || !(lr_max >= lr_min)) {
        snprintf(g_lerr, sizeof(g_lerr), "qa_kl_lr_rule: bad argument"); return QA_E_ARG; }
    hipLaunchKernelGGL(qa_kl_lr_rule_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, kl, desired_kl, factor, lr_min, lr_max, lr);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { snprintf(g_lerr, sizeof(g_lerr), "qa_kl_lr_rule: %s", hipGetErrorString(e)); return QA_E_DEVICE; }
    return QA_OK;
}

int qa_episode_means(const float *episode_stats, const int64_t *step_dev, int64_t step, int32_t num_terms, float max_episode_length_s, float *means,
                     float *snapshot, void *stream) {
    if (!episode_stats || !means || !snapshot || num_terms <= 0 || num_terms > 14 || !(max_episode_length_s > 0.f)) {
        snprintf(g_lerr, sizeof(g_lerr), "qa_episode_means: bad argument"); return QA_E_ARG; }
    hipLaunchKernelGGL(qa_episode_means_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, episode_stats, step_dev, step, (int)num_terms,
                       max_episode_length_s, means, snapshot);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { snprintf(g_lerr, sizeof(g_lerr), "qa_episode_means: %s", hipGetErrorString(e)); return QA_E_DEVICE; }
    return QA_OK;
}

int qa_rollout_act(const float *mean, const float *std, const float *value, const float *noise, uint64_t seed, const int64_t *step_dev,
                   int64_t step, int32_t num_envs, int32_t env_id_offset, float *actions, float *st_actions, float *st_mu, float *st_sigma, float *st_logp,
                   float *st_values, void *stream) {
    if (!mean || !std || !value || !actions || !st_actions || !st_mu || !st_sigma || !st_logp || !st_values || num_envs <= 0) {
        snprintf(g_lerr, sizeof(g_lerr), "qa_rollout_act: bad argument"); return QA_E_ARG; }
    hipLaunchKernelGGL(qa_rollout_act_kernel, dim3((num_envs + 255) / 256), dim3(256), 0, (hipStream_t)stream, mean, std, value, noise, seed, step_dev, step,
                       (int)num_envs, (int)env_id_offset, actions, st_actions, st_mu, st_sigma, st_logp, st_values);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { snprintf(g_lerr, sizeof(g_lerr), "qa_rollout_act: %s", hipGetErrorString(e)); return QA_E_DEVICE; }
    return QA_OK;
}

int qa_rollout_act_store(const float *mean, const float *std, const float *value, const float *noise, uint64_t seed, const int64_t *step_dev,
                         int64_t step, int32_t num_envs, int32_t env_id_offset, float *actions, float *st_actions, float *st_mu, float *st_sigma, float *st_logp,
                         float *st_values, const float *obs, int64_t obs_stride, int32_t obs_width, float *st_obs, int64_t st_obs_stride, void *stream) {
    if (!mean || !std || !value || !actions || !st_actions || !st_mu || !st_sigma || !st_logp || !st_values || num_envs <= 0 || !obs || !st_obs || obs_width <= 0 ||
        obs_stride < obs_width || st_obs_stride < obs_width || (int64_t)ACT_EPB * obs_width > INT32_MAX) {
        snprintf(g_lerr, sizeof(g_lerr), "qa_rollout_act_store: bad argument"); return QA_E_ARG; }
    hipLaunchKernelGGL(qa_rollout_act_store_kernel, dim3((num_envs + ACT_EPB - 1) / ACT_EPB), dim3(256), 0, (hipStream_t)stream, mean, std, value, noise, seed, step_dev, step,
                       (int)num_envs, (int)env_id_offset, actions, st_actions, st_mu, st_sigma, st_logp, st_values, obs, obs_stride, (int)obs_width, st_obs, st_obs_stride);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { snprintf(g_lerr, sizeof(g_lerr), "qa_rollout_act_store: %s", hipGetErrorString(e)); return QA_E_DEVICE; }
    return QA_OK;
}

int qa_tsc_push(float *root_states, int64_t num_envs, int64_t *step_dev, int32_t *ticket, int32_t push_interval, float max_push_vel_xy, uint64_t seed,
                int32_t env_id_offset, void *stream) {
    if (!root_states || !step_dev || !ticket || num_envs <= 0) { snprintf(g_lerr, sizeof(g_lerr), "qa_tsc_push: bad argument"); return QA_E_ARG; }
    hipLaunchKernelGGL(qa_tsc_push_kernel, dim3((unsigned)((num_envs + 63) / 64)), dim3(64), 0, (hipStream_t)stream, root_states, num_envs, step_dev, (int *)ticket,
                       (int)push_interval, max_push_vel_xy, seed, (int)env_id_offset);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { snprintf(g_lerr, sizeof(g_lerr), "qa_tsc_push: %s", hipGetErrorString(e)); return QA_E_DEVICE; }
    return QA_OK;
}

int qa_tsc_start_pose(const uint8_t *flags, int64_t *cur_obst_idx, const float *env_goals, const float *obst_angs, int64_t num_envs, int32_t num_goal_slots,
                      int32_t num_obstacles, int32_t goals_per_obstacle, int32_t randomize_start, float frame_yaw0, uint64_t seed, const int64_t *step_dev,
                      int32_t env_id_offset, float *start_xy, float *start_yaw, int64_t *start_goal, void *stream) {
    if (!flags || !env_goals || !start_xy || !start_yaw || !start_goal || num_envs <= 0 || num_goal_slots <= 0 ||
        (randomize_start && (!cur_obst_idx || !obst_angs || !step_dev || num_obstacles <= 0 || goals_per_obstacle <= 0))) {
        snprintf(g_lerr, sizeof(g_lerr), "qa_tsc_start_pose: bad argument"); return QA_E_ARG; }
    StartPoseArgs a{flags, cur_obst_idx, env_goals, obst_angs, num_envs, (int)num_goal_slots, (int)num_obstacles, (int)goals_per_obstacle, (int)(randomize_start != 0),
                    frame_yaw0, seed, step_dev, (int)env_id_offset, start_xy, start_yaw, start_goal};
    hipLaunchKernelGGL(qa_tsc_start_pose_kernel, dim3((unsigned)((num_envs + 63) / 64)), dim3(64), 0, (hipStream_t)stream, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { snprintf(g_lerr, sizeof(g_lerr), "qa_tsc_start_pose: %s", hipGetErrorString(e)); return QA_E_DEVICE; }
    return QA_OK;
}

int qa_tsc_reset_where(const uint8_t *flags, const uint8_t *any_reset, const int64_t *start_goal, int64_t *cur_goal_idx, float *reach_goal_timer,
                       float *episode_sums, int32_t num_terms, int64_t *episode_length, const float *env_goals, int32_t num_goal_slots, float *cur_goals,
                       float *next_goals, float *obst_state, float seesaw_rest, const int64_t *cur_obst_idx, const int64_t *seesaw_order, int64_t num_envs,
                       void *stream) {
    if (!flags || !start_goal || !cur_goal_idx || !reach_goal_timer || !episode_sums || !episode_length || !env_goals || !cur_goals || !next_goals ||
        num_envs <= 0 || num_terms <= 0 || num_goal_slots <= 0 || (obst_state && (!any_reset || (seesaw_order && !cur_obst_idx)))) {
        snprintf(g_lerr, sizeof(g_lerr), "qa_tsc_reset_where: bad argument"); return QA_E_ARG; }
    ResetWhereArgs a{flags, any_reset, start_goal, cur_goal_idx, reach_goal_timer, episode_sums, (int)num_terms, episode_length, env_goals, (int)num_goal_slots,
                     cur_goals, next_goals, obst_state, seesaw_rest, cur_obst_idx, seesaw_order, num_envs};
    hipLaunchKernelGGL(qa_tsc_reset_where_kernel, dim3((unsigned)((num_envs + 63) / 64)), dim3(64), 0, (hipStream_t)stream, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { snprintf(g_lerr, sizeof(g_lerr), "qa_tsc_reset_where: %s", hipGetErrorString(e)); return QA_E_DEVICE; }
    return QA_OK;
}

int qa_rollout_act_hybrid(const float *logits, const float *mean, const float *std, const float *value, uint64_t seed, const int64_t *step_dev, int64_t step,
                          int32_t num_envs, int32_t env_id_offset, int32_t nd, int32_t nc_all, float *actions, float *st_actions, float *st_mu, float *st_sigma,
                          float *st_logp_d, float *st_logp_c, float *st_values, const float *action_history_in, float *action_history, int32_t hist_len,
                          void *stream) {
    if (!logits || !mean || !std || !value || !actions || !st_actions || !st_mu || !st_sigma || !st_logp_d || !st_logp_c || !st_values || num_envs <= 0 ||
        nd <= 0 || nd > 16 || nc_all <= 0 || nc_all > 32 || (action_history && hist_len <= 0) || (action_history_in && !action_history)) {
        snprintf(g_lerr, sizeof(g_lerr), "qa_rollout_act_hybrid: bad argument (nd <= 16, nc_all <= 32)"); return QA_E_ARG; }
    int epb = 128;
    while (epb > 4 && (num_envs + epb - 1) / epb < 256) epb >>= 1;           // >= ~256 workgroups, 4..128 envs each
    HybridActArgs a{logits, mean, std, value, seed, step_dev, step, (int)num_envs, (int)env_id_offset, (int)nd, (int)nc_all, actions, st_actions, st_mu, st_sigma,
                    st_logp_d, st_logp_c, st_values, action_history, action_history_in ? action_history_in : action_history, (int)hist_len, epb};
    hipLaunchKernelGGL(qa_rollout_act_hybrid_kernel, dim3((num_envs + epb - 1) / epb), dim3(128), 0, (hipStream_t)stream, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { snprintf(g_lerr, sizeof(g_lerr), "qa_rollout_act_hybrid: %s", hipGetErrorString(e)); return QA_E_DEVICE; }
    return QA_OK;
}

int qa_rollout_post(const float *rew, const int64_t *reset, const uint8_t *time_out, const float *values, float reward_coef, float gamma,
                    int32_t num_envs, float *st_rewards, uint8_t *st_dones, float *cur, float *fin_vals, uint8_t *fin_mask, void *stream) {
    if (!rew || !reset || !time_out || !values || !st_rewards || !st_dones || num_envs <= 0 || (cur && (!fin_vals || !fin_mask))) {
        snprintf(g_lerr, sizeof(g_lerr), "qa_rollout_post: bad argument"); return QA_E_ARG; }
    hipLaunchKernelGGL(qa_rollout_post_kernel, dim3((num_envs + 255) / 256), dim3(256), 0, (hipStream_t)stream, rew, reset, time_out, values, reward_coef,
                       gamma, (int)num_envs, st_rewards, st_dones, cur, fin_vals, fin_mask);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { snprintf(g_lerr, sizeof(g_lerr), "qa_rollout_post: %s", hipGetErrorString(e)); return QA_E_DEVICE; }
    return QA_OK;
}

int qa_rollout_post_amp(const float *rew, const int64_t *reset, const uint8_t *time_out, const float *values, const float *d, const float *eps,
                        const float *logits, int32_t dim_c, const float *obs, int64_t obs_stride, int32_t num_obs, float c_i, float c_us,
                        float c_ss, float c_t, float dt, float gamma, int32_t num_envs, float *st_rewards, uint8_t *st_dones, float *cur,
                        float *fin_vals, uint8_t *fin_mask, void *stream) {
    if (!rew || !reset || !time_out || !values || !d || !eps || !logits || !obs || !st_rewards || !st_dones || num_envs <= 0 || dim_c <= 0 || dim_c > 8 ||
        num_obs < dim_c + 1 || obs_stride < num_obs || (cur && (!fin_vals || !fin_mask))) {
        snprintf(g_lerr, sizeof(g_lerr), "qa_rollout_post_amp: bad argument"); return QA_E_ARG; }
    PostAmpArgs a{rew, reset, time_out, values, d, eps, logits, obs, obs_stride, (int)num_obs, (int)dim_c, (int)num_envs, c_i, c_us, c_ss, c_t, dt, gamma,
                  st_rewards, st_dones, cur, fin_vals, fin_mask};
    hipLaunchKernelGGL(qa_rollout_post_amp_kernel, dim3((num_envs + 255) / 256), dim3(256), 0, (hipStream_t)stream, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { snprintf(g_lerr, sizeof(g_lerr), "qa_rollout_post_amp: %s", hipGetErrorString(e)); return QA_E_DEVICE; }
    return QA_OK;
}

int64_t qa_disc_loss_scratch_bytes(int64_t rows) { return rows <= 0 ? -1 : (int64_t)sizeof(float) * (DISC_SUMS + 1) * ((rows + DISC_BLOCK - 1) / DISC_BLOCK); }

static int disc_loss_launch(const char *who, int from_logits, const float *d, const float *eps, const float *c, const int64_t *label_lb, const float *policy_eps,
                            const float *policy_c, int32_t b_lb, int32_t b_pi, int32_t b_ulb, float c_ss, const float *info_coef_dev, float c_disc, float c_us,
                            float *grad_d, float *grad_eps, float *grad_c, float *out, void *scratch, int64_t scratch_bytes, void *stream) {
    const int64_t B = (int64_t)b_lb + b_pi + b_ulb;
    if (!d || !eps || !c || !label_lb || !policy_eps || !policy_c || !info_coef_dev || !grad_d || !grad_eps || !grad_c || !out || !scratch ||
        b_lb <= 0 || b_pi <= 0 || b_ulb <= 0 || scratch_bytes < qa_disc_loss_scratch_bytes(B)) {
        snprintf(g_lerr, sizeof(g_lerr), "%s: bad argument", who); return QA_E_ARG; }
    DiscArgs a{d, eps, c, label_lb, policy_eps, policy_c, info_coef_dev, grad_d, grad_eps, grad_c, (float *)scratch, b_lb, b_pi, b_ulb, c_ss, c_disc, c_us, from_logits};
    const int blocks = (int)((B + DISC_BLOCK - 1) / DISC_BLOCK);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(qa_disc_loss_kernel, dim3(blocks), dim3(DISC_BLOCK), 0, st, a);
    hipLaunchKernelGGL(qa_disc_finish_kernel, dim3(1), dim3(64), 0, st, (const float *)scratch, blocks, b_lb, b_pi, b_ulb, c_ss, info_coef_dev, c_disc, c_us, out);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { snprintf(g_lerr, sizeof(g_lerr), "%s: %s", who, hipGetErrorString(e)); return QA_E_DEVICE; }
    return QA_OK;
}
int qa_disc_loss(const float *d, const float *eps, const float *c, const int64_t *label_lb, const float *policy_eps, const float *policy_c,
                 int32_t b_lb, int32_t b_pi, int32_t b_ulb, float c_ss, const float *info_coef_dev, float c_disc, float c_us,
                 float *grad_d, float *grad_eps, float *grad_c, float *out, void *scratch, int64_t scratch_bytes, void *stream) {
    return disc_loss_launch("qa_disc_loss", 0, d, eps, c, label_lb, policy_eps, policy_c, b_lb, b_pi, b_ulb, c_ss, info_coef_dev, c_disc, c_us, grad_d, grad_eps, grad_c,
                            out, scratch, scratch_bytes, stream);
}
int qa_disc_loss_logits(const float *d, const float *eps, const float *logits, const int64_t *label_lb, const float *policy_eps, const float *policy_c,
                        int32_t b_lb, int32_t b_pi, int32_t b_ulb, float c_ss, const float *info_coef_dev, float c_disc, float c_us,
                        float *grad_d, float *grad_eps, float *grad_logits, float *out, void *scratch, int64_t scratch_bytes, void *stream) {
    return disc_loss_launch("qa_disc_loss_logits", 1, d, eps, logits, label_lb, policy_eps, policy_c, b_lb, b_pi, b_ulb, c_ss, info_coef_dev, c_disc, c_us, grad_d, grad_eps,
                            grad_logits, out, scratch, scratch_bytes, stream);
}

int qa_disc_sample_prepare(const qa_disc_sample_io *io, int32_t dim, int32_t c_dim, const float *task_mask, const float *frame_mult,
                           const float *task_weight_dev, const double *mean, const double *var, float epsilon, float clip, float *out, void *stream) {
    if (!io || !task_mask || !frame_mult || !out || dim <= 0 || c_dim < 0 || ((mean == nullptr) != (var == nullptr)) || !io->block_dev ||
        (io->eps_src && (!io->c_src || !io->eps_out || !io->c_out || c_dim <= 0)) || (io->label_src && !io->label_out)) {
        snprintf(g_lerr, sizeof(g_lerr), "qa_disc_sample_prepare: bad argument"); return QA_E_ARG; }
    int64_t total = 0, small = 0;
    for (int b = 0; b < 3; ++b) {
        if (!io->src[b] || !io->index[b] || io->rows[b] <= 0) { snprintf(g_lerr, sizeof(g_lerr), "qa_disc_sample_prepare: empty batch"); return QA_E_ARG; }
        total += io->rows[b] * dim;
    }
    small = io->rows[0] > io->rows[1] ? io->rows[0] : io->rows[1];
    SampleArgs a{};
    a.io = *io; a.dim = dim; a.c_dim = c_dim; a.task_mask = task_mask; a.frame_mult = frame_mult; a.task_w = task_weight_dev; a.mean = mean; a.var = var;
    a.eps = epsilon; a.clip = clip; a.out = out; a.main_blocks = (total + 255) / 256;
    const int64_t extra = (io->eps_src || io->label_src) ? (small + 255) / 256 : 0;
    hipLaunchKernelGGL(qa_disc_sample_prepare_kernel, dim3((unsigned)(a.main_blocks + extra)), dim3(256), 0, (hipStream_t)stream, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { snprintf(g_lerr, sizeof(g_lerr), "qa_disc_sample_prepare: %s", hipGetErrorString(e)); return QA_E_DEVICE; }
    return QA_OK;
}

int64_t qa_disc_step_tail_scratch_bytes(void) { return (int64_t)(TAIL_MAX * TAIL_WG + 4) * 4; }

int qa_disc_step_tail(const float *head_stats, const float *input_grad, int64_t grad_rows, int32_t grad_cols, const float *const *weights,
                      const int64_t *weight_counts, int32_t num_weights, float *out, float *acc, int64_t *step_counter, float *prior, int32_t prior_dim,
                      float prior_soft_coef, void *scratch, int64_t scratch_bytes, void *stream) {
    if (!head_stats || !input_grad || grad_rows <= 0 || grad_cols <= 0 || !weights || !weight_counts || num_weights <= 0 || num_weights > TAIL_MAX - 1 ||
        !out || !scratch || (prior && (prior_dim <= 0 || prior_dim > 5)) || scratch_bytes < qa_disc_step_tail_scratch_bytes() || ((uintptr_t)scratch & 15)) {
        snprintf(g_lerr, sizeof(g_lerr), "qa_disc_step_tail: bad argument (1..%d weight tensors, scratch >= qa_disc_step_tail_scratch_bytes(), zeroed once)", TAIL_MAX - 1);
        return QA_E_ARG; }
    TailArgs a{};
    a.t[0] = input_grad; a.n[0] = grad_rows * grad_cols;
    for (int i = 0; i < num_weights; ++i) {
        if (!weights[i] || weight_counts[i] <= 0) { snprintf(g_lerr, sizeof(g_lerr), "qa_disc_step_tail: empty weight tensor"); return QA_E_ARG; }
        a.t[1 + i] = weights[i]; a.n[1 + i] = weight_counts[i];
    }
    a.nt = 1 + num_weights; a.hs = head_stats; a.inv_rows = 1.0f / (float)grad_rows; a.out = out; a.acc = acc; a.step = step_counter; a.prior = prior; a.prior_dim = prior_dim; a.prior_c = prior_soft_coef;
    a.partial = (float *)scratch; a.ticket = (unsigned *)((float *)scratch + TAIL_MAX * TAIL_WG);
    hipLaunchKernelGGL(qa_disc_step_tail_kernel, dim3((unsigned)(a.nt * TAIL_WG)), dim3(256), 0, (hipStream_t)stream, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { snprintf(g_lerr, sizeof(g_lerr), "qa_disc_step_tail: %s", hipGetErrorString(e)); return QA_E_DEVICE; }
    return QA_OK;
}

int qa_disc_prepare(const float *const *batches, const int64_t *rows, int32_t num_batches, int32_t dim, const float *task_mask,
                    const float *frame_mult, const float *task_weight_dev, const double *mean, const double *var, float epsilon, float clip,
                    float *out, void *stream) {
    if (!batches || !rows || !task_mask || !frame_mult || !out || num_batches <= 0 || num_batches > 3 || dim <= 0 || ((mean == nullptr) != (var == nullptr))) {
        snprintf(g_lerr, sizeof(g_lerr), "qa_disc_prepare: bad argument"); return QA_E_ARG; }
    PrepArgs a{};
    int64_t total = 0;
    for (int b = 0; b < num_batches; ++b) {
        if (!batches[b] || rows[b] <= 0) { snprintf(g_lerr, sizeof(g_lerr), "qa_disc_prepare: empty batch"); return QA_E_ARG; }
        a.src[b] = batches[b]; a.rows[b] = rows[b]; total += rows[b] * dim;
    }
    a.k = num_batches; a.dim = dim; a.task_mask = task_mask; a.frame_mult = frame_mult; a.task_w = task_weight_dev; a.mean = mean; a.var = var;
    a.eps = epsilon; a.clip = clip; a.out = out;
    hipLaunchKernelGGL(qa_disc_prepare_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { snprintf(g_lerr, sizeof(g_lerr), "qa_disc_prepare: %s", hipGetErrorString(e)); return QA_E_DEVICE; }
    return QA_OK;
}

}  // extern "C"
