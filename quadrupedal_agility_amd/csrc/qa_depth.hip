// Depth camera of the vision student for gfx950 (contract: qa_tsc_depth_update in include/qa_sim.h): the
// reference renders a 106 x 60 depth image per env with Isaac Gym's camera sensor and post-processes it per env in a Python loop
// (tsc/legged_gym/envs/base/legged_robot.py:154-200); here ONE launch ray-casts the cropped 58 x 87 pixels of every env against
// the course's height field + ceiling field, applies clip / normalise / noise and pushes the image into the env's ring.
//
// One thread per pixel, 256-thread workgroups, grid (ceil(5046 / 256), N): 256 camera envs = 5,120 workgroups, 20 waves per CU.
// The maps (480 x 600 int16 for a 4-env course, ~0.6 MB; 2 x 18 MB at 256 envs) are read through L2: neighbouring pixels walk
// neighbouring cells, so a wavefront's loads of one march step fall into a handful of cache lines.  Cost = march steps x
// (8 map loads + ~40 VALU); it is latency/L2-bound, not HBM-bound: algorithmic HBM bytes are the image ring (2 x 20 KB per env).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/qa_sim.h"
#include "qa_device.h"

extern thread_local char qa_err_buf[512];

namespace {

struct DepthArgs {
    qa_tsc_depth_cfg c;
    qa_tsc_depth_io io;
    float tan_h, tan_v, inv_hs;
    int wc, hc;
};

// floor / ceiling height over (x, y): two triangles per cell split along the (i,j)-(i+1,j+1) diagonal (the collision terrain's
// triangulation, qa_physics.h ground_query); indices clamped at the map's edge
__device__ __forceinline__ float surface(const int16_t *__restrict__ m, int rows, int cols, float border, float inv_hs, float vs, float x, float y, bool &exists) {
    const float fx = (x + border) * inv_hs, fy = (y + border) * inv_hs;
    const int ix = min(max((int)floorf(fx), 0), rows - 2), iy = min(max((int)floorf(fy), 0), cols - 2);
    const float u = fminf(fmaxf(fx - (float)ix, 0.f), 1.f), v = fminf(fmaxf(fy - (float)iy, 0.f), 1.f);
    const int16_t *g = m + (int64_t)ix * cols + iy;
    const int s00 = g[0], s01 = g[1], s10 = g[cols], s11 = g[cols + 1];
    const bool lower = u >= v;
    const int sm = lower ? s10 : s01;
    exists = s00 != QA_NO_CEILING && s11 != QA_NO_CEILING && sm != QA_NO_CEILING;
    const float h00 = vs * (float)s00, h11 = vs * (float)s11, hm = vs * (float)sm;
    // lower: h00 + u (h10 - h00) + v (h11 - h10); upper: h00 + v (h01 - h00) + u (h11 - h01)
    const float a = lower ? u : v, b = lower ? v : u;
    return h00 + a * (hm - h00) + b * (h11 - hm);
}

__global__ void __launch_bounds__(256) qa_tsc_depth_kernel(DepthArgs a) {
    const qa_tsc_depth_cfg &c = a.c;
    const int64_t e = blockIdx.y;
    const int p = blockIdx.x * 256 + threadIdx.x, npix = a.wc * a.hc;
    if (p >= npix) return;
    const float *rs = a.io.root_states + e * 13;
    const float qx = rs[3], qy = rs[4], qz = rs[5], qw = rs[6];
    const float R[9] = {1 - 2 * (qy * qy + qz * qz), 2 * (qx * qy - qz * qw), 2 * (qx * qz + qy * qw),
                        2 * (qx * qy + qz * qw), 1 - 2 * (qx * qx + qz * qz), 2 * (qy * qz - qx * qw),
                        2 * (qx * qz - qy * qw), 2 * (qy * qz + qx * qw), 1 - 2 * (qx * qx + qy * qy)};
    float sa, ca; sincosf(a.io.camera_pitch[e], &sa, &ca);
    float o[3], d[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) o[i] = rs[i] + R[3 * i] * c.position[0] + R[3 * i + 1] * c.position[1] + R[3 * i + 2] * c.position[2];
    const int i = p / a.wc, j = p - i * a.wc;
    const float sx = (((float)(j + c.crop_left) + 0.5f) / (float)c.width * 2.0f - 1.0f) * a.tan_h;
    const float sy = (((float)(i + c.crop_top) + 0.5f) / (float)c.height * 2.0f - 1.0f) * a.tan_v;
    const float dt3[3] = {ca - sy * sa, -sx, -sa - sy * ca};              // fwd - sx left - sy up, trunk frame
#pragma unroll
    for (int k = 0; k < 3; ++k) d[k] = R[3 * k] * dt3[0] + R[3 * k + 1] * dt3[1] + R[3 * k + 2] * dt3[2];
    const float far = c.far_clip, near = c.near_clip, vs = c.vertical_scale, border = c.border_size;
    const float dxy = sqrtf(d[0] * d[0] + d[1] * d[1]), dt = 0.5f * c.horizontal_scale / fmaxf(dxy, 0.5f);
    const int nsteps = (int)ceilf(far / dt);
    const int16_t *hm = a.io.height_samples, *cm = a.io.ceiling_samples;
    bool ex_prev = false, ex = false, dummy;
    float t_prev = 0.f, hit = far;
    float g_prev = o[2] - surface(hm, c.map_rows, c.map_cols, border, a.inv_hs, vs, o[0], o[1], dummy);
    float h_prev = cm ? o[2] - surface(cm, c.map_rows, c.map_cols, border, a.inv_hs, vs, o[0], o[1], ex_prev) : 0.f;
    if (g_prev < 0.f) hit = 0.f;
    else for (int k = 1; k <= nsteps; ++k) {
        const float t = fminf((float)k * dt, far);
        const float x = o[0] + t * d[0], y = o[1] + t * d[1], z = o[2] + t * d[2];
        const float g = z - surface(hm, c.map_rows, c.map_cols, border, a.inv_hs, vs, x, y, dummy);
        float h = 0.f, best = 1e30f;
        if (g < 0.f) {                      // the floor was crossed in (t_prev, t]: QA_TSC_DEPTH_BISECT halvings, then linear interpolation
            float lo = t_prev, hi = t, flo = g_prev, fhi = g;
            for (int b = 0; b < QA_TSC_DEPTH_BISECT; ++b) {
                const float tm = 0.5f * (lo + hi);
                const float fm = o[2] + tm * d[2] - surface(hm, c.map_rows, c.map_cols, border, a.inv_hs, vs, o[0] + tm * d[0], o[1] + tm * d[1], dummy);
                if (fm < 0.f) { hi = tm; fhi = fm; } else { lo = tm; flo = fm; }
            }
            best = lo + (hi - lo) * flo / (flo - fhi);
        }
        if (cm) {
            h = z - surface(cm, c.map_rows, c.map_cols, border, a.inv_hs, vs, x, y, ex);
            if (ex && ex_prev && ((h_prev < 0.f) != (h < 0.f))) {
                float lo = t_prev, hi = t, flo = h_prev, fhi = h;
                for (int b = 0; b < QA_TSC_DEPTH_BISECT; ++b) {
                    const float tm = 0.5f * (lo + hi); bool exm;
                    const float fm = o[2] + tm * d[2] - surface(cm, c.map_rows, c.map_cols, border, a.inv_hs, vs, o[0] + tm * d[0], o[1] + tm * d[1], exm);
                    if (!exm) break;                                     // a hole in the shell inside the bracket: keep the bracket
                    if ((fm < 0.f) == (flo < 0.f)) { lo = tm; flo = fm; } else { hi = tm; fhi = fm; }
                }
                best = fminf(best, lo + (hi - lo) * flo / (flo - fhi));
            }
        }
        if (best < 1e29f) { hit = best; break; }
        t_prev = t; g_prev = g; h_prev = h; ex_prev = ex;
    }
    const float dd = fminf(fmaxf(hit, near), far);
    float v = (dd - near) / (far - near) - 0.5f;
    // noise (process_depth_image :166-168), Philox stream QA_TSC_DEPTH_STREAM keyed by (seed; global env id, step)
    const uint32_t env = (uint32_t)(e + c.env_id_offset), s_lo = (uint32_t)c.step, s_hi = (uint32_t)((uint64_t)c.step >> 32);
    const U4 r0 = philox(c.seed, env, s_lo, (uint32_t)(QA_TSC_DEPTH_STREAM * 256), s_hi);
    const U4 rp = philox(c.seed, env, s_lo, (uint32_t)(QA_TSC_DEPTH_STREAM * 256 + 1 + (p >> 2)), s_hi);
    const float amp = c.depth_noise * ((float)(r0.v[0] >> 8) * (1.0f / 16777216.0f));
    const float offs = c.depth_noise * 2.0f * ((float)(r0.v[1] >> 8) * (1.0f / 16777216.0f) - 0.5f);
    const uint32_t rpx = (p & 3) == 0 ? rp.v[0] : ((p & 3) == 1 ? rp.v[1] : ((p & 3) == 2 ? rp.v[2] : rp.v[3]));
    v += offs + amp * 2.0f * ((float)(rpx >> 8) * (1.0f / 16777216.0f) - 0.5f);
    float *buf = a.io.depth_buffer + e * c.buffer_len * (int64_t)npix + p;
    if (a.io.episode_length[e] <= 1) {
        for (int s = 0; s < c.buffer_len; ++s) buf[(int64_t)s * npix] = v;
    } else {
        for (int s = 0; s + 1 < c.buffer_len; ++s) buf[(int64_t)s * npix] = buf[(int64_t)(s + 1) * npix];
        buf[(int64_t)(c.buffer_len - 1) * npix] = v;
    }
}

}  // namespace

extern "C" int qa_tsc_depth_update(const qa_tsc_depth_cfg *cfg, const qa_tsc_depth_io *io, void *stream) {
    if (!cfg || !io) { snprintf(qa_err_buf, sizeof(qa_err_buf), "qa_tsc_depth_update: null argument"); return QA_E_ARG; }
    if (!io->root_states || !io->camera_pitch || !io->height_samples || !io->episode_length || !io->depth_buffer) {
        snprintf(qa_err_buf, sizeof(qa_err_buf), "qa_tsc_depth_update: a required io pointer is null"); return QA_E_ARG;
    }
    DepthArgs a; a.c = *cfg; a.io = *io;
    a.wc = cfg->width - cfg->crop_left - cfg->crop_right; a.hc = cfg->height - cfg->crop_top - cfg->crop_bottom;
    if (cfg->num_envs <= 0 || cfg->num_envs > 65535 || a.wc <= 0 || a.hc <= 0 || cfg->buffer_len < 1 || cfg->map_rows < 2 || cfg->map_cols < 2 || !(cfg->horizontal_scale > 0.f) ||
        !(cfg->far_clip > cfg->near_clip) || !(cfg->horizontal_fov_deg > 0.f && cfg->horizontal_fov_deg < 180.f)) {
        snprintf(qa_err_buf, sizeof(qa_err_buf), "qa_tsc_depth_update: inconsistent configuration"); return QA_E_ARG;
    }
    a.tan_h = (float)tan((double)cfg->horizontal_fov_deg * 3.14159265358979323846 / 360.0);
    a.tan_v = (float)(tan((double)cfg->horizontal_fov_deg * 3.14159265358979323846 / 360.0) * cfg->height / cfg->width);
    a.inv_hs = 1.0f / cfg->horizontal_scale;
    const int npix = a.wc * a.hc;
    hipLaunchKernelGGL(qa_tsc_depth_kernel, dim3((unsigned)((npix + 255) / 256), (unsigned)cfg->num_envs), dim3(256), 0, (hipStream_t)stream, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { snprintf(qa_err_buf, sizeof(qa_err_buf), "qa_tsc_depth_update: %s", hipGetErrorString(e)); return QA_E_DEVICE; }
    return QA_OK;
}
