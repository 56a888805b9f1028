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
    int wc, hc, tiles;
};

// cell of (x, y) in the maps' grid: clamped fine cell index and the position inside it (the collision terrain's lookup,
// qa_physics.h ground_query)
struct Cell { int ix, iy; float u, v; };
__device__ __forceinline__ Cell cell_of(float x, float y, float border, float inv_hs, int rows, int cols) {
    const float fx = (x + border) * inv_hs, fy = (y + border) * inv_hs;
    Cell c;
    c.ix = min(max((int)floorf(fx), 0), rows - 2); c.iy = min(max((int)floorf(fy), 0), cols - 2);
    c.u = fminf(fmaxf(fx - (float)c.ix, 0.f), 1.f); c.v = fminf(fmaxf(fy - (float)c.iy, 0.f), 1.f);
    return c;
}
// floor / ceiling height in a cell: two triangles split along the (i,j)-(i+1,j+1) diagonal; a ceiling triangle with a QA_NO_CEILING
// corner does not exist
__device__ __forceinline__ float tri_height(const int16_t *__restrict__ m, int cols, float vs, const Cell &c, bool &exists) {
    const int16_t *g = m + (int64_t)c.ix * cols + c.iy;
    const int s00 = g[0], s01 = g[1], s10 = g[cols], s11 = g[cols + 1];
    const bool lower = c.u >= c.v;
    const int sm = lower ? s10 : s01;
    exists = s00 != QA_NO_CEILING && s11 != QA_NO_CEILING && sm != QA_NO_CEILING;
    const float h00 = vs * (float)s00, h11 = vs * (float)s11, hm = vs * (float)sm;
    const float a = lower ? c.u : c.v, b = lower ? c.v : c.u;       // lower: h00 + u (h10 - h00) + v (h11 - h10); upper: h00 + v (h01 - h00) + u (h11 - h01)
    return h00 + a * (hm - h00) + b * (h11 - hm);
}

__global__ void __launch_bounds__(256) qa_tsc_depth_kernel(DepthArgs a) {
    const qa_tsc_depth_cfg &c = a.c;
    // workgroup id -> (env, pixel tile).  Consecutive workgroup ids go round-robin over the 8 XCDs, each with its own L2: id = 8 s + x
    // runs on XCD x, so env = 8 (s / tiles) + x keeps all tiles of one env -- which walk the same ~100 KB of the two maps -- on ONE L2
    const int npix = a.wc * a.hc, tiles = a.tiles;
    const int x = blockIdx.x & 7, sl = blockIdx.x >> 3;
    const int64_t e = (int64_t)(sl / tiles) * 8 + x;
    if (e >= c.num_envs) return;
    const int p = (sl % tiles) * 256 + threadIdx.x;
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
    const float far = c.far_clip, near = c.near_clip, vs = c.vertical_scale, border = c.border_size, hs = c.horizontal_scale;
    const float dxy = sqrtf(d[0] * d[0] + d[1] * d[1]), dt = 0.5f * hs / fmaxf(dxy, 0.5f);
    const int nsteps = (int)ceilf(far / dt), rows = c.map_rows, cols = c.map_cols;
    const int16_t *hm = a.io.height_samples, *cm = a.io.ceiling_samples;
    // empty-space skipping: per block of 2^S x 2^S cells the highest floor sample and the lowest ceiling sample (apron of one sample).
    // While the ray's height over the samples it would take inside a block stays clear of both, none of them can report a hit:
    // evaluate only the last one (it seeds the next step's bracket) and jump.  Conservative by construction -- the image is the
    // full march's (the oracle's) image.
    const int16_t *cfl = a.io.coarse_floor_max, *cce = a.io.coarse_ceiling_min;
    const int S = c.coarse_log2, ccols = ((cols - 2) >> S) + 1, crows = ((rows - 2) >> S) + 1;
    const float block = (float)(1 << S) * hs;
    const float idx = d[0] != 0.f ? 1.0f / d[0] : 0.f, idy = d[1] != 0.f ? 1.0f / d[1] : 0.f;
    int seen_cx = -1, seen_cy = -1;                                     // the block that last refused a skip: not asked again while inside it
    int blk_cx = -1, blk_cy = -1; bool blk_roof = true;                 // the block of the current sample: has it any ceiling sample at all?
    bool ex_prev = false, ex = false, dummy;
    float t_prev = 0.f, hit = far;
    Cell c0 = cell_of(o[0], o[1], border, a.inv_hs, rows, cols);
    float g_prev = o[2] - tri_height(hm, cols, vs, c0, dummy);
    float h_prev = cm ? o[2] - tri_height(cm, cols, vs, c0, ex_prev) : 0.f;
    if (g_prev < 0.f) hit = 0.f;
    else for (int k = 1; k <= nsteps; ++k) {
        const float t = fminf((float)k * dt, far);
        const float x = o[0] + t * d[0], y = o[1] + t * d[1], z = o[2] + t * d[2];
        const Cell ck = cell_of(x, y, border, a.inv_hs, rows, cols);
        const float g = z - tri_height(hm, cols, vs, ck, dummy);
        float h = 0.f, best = 1e30f;
        if (g < 0.f) {                      // the floor was crossed in (t_prev, t]: QA_TSC_DEPTH_BISECT halvings, then linear interpolation
            float lo = t_prev, hi = t, flo = g_prev, fhi = g;
            for (int b = 0; b < QA_TSC_DEPTH_BISECT; ++b) {
                const float tm = 0.5f * (lo + hi);
                const Cell cb = cell_of(o[0] + tm * d[0], o[1] + tm * d[1], border, a.inv_hs, rows, cols);
                const float fm = o[2] + tm * d[2] - tri_height(hm, cols, vs, cb, dummy);
                if (fm < 0.f) { hi = tm; fhi = fm; } else { lo = tm; flo = fm; }
            }
            best = lo + (hi - lo) * flo / (flo - fhi);
        }
        const int cx = ck.ix >> S, cy = ck.iy >> S;
        if (cce && (cx != blk_cx || cy != blk_cy)) { blk_cx = cx; blk_cy = cy; blk_roof = cce[cx * ccols + cy] != QA_NO_CEILING; }
        ex = false;
        if (cm && blk_roof) {               // a block without a single ceiling sample has no ceiling triangle: ex stays false
            h = z - tri_height(cm, cols, vs, ck, ex);
            if (ex && ex_prev && ((h_prev < 0.f) != (h < 0.f))) {
                float lo = t_prev, hi = t, flo = h_prev, fhi = h;
                for (int b = 0; b < QA_TSC_DEPTH_BISECT; ++b) {
                    const float tm = 0.5f * (lo + hi); bool exm;
                    const Cell cb = cell_of(o[0] + tm * d[0], o[1] + tm * d[1], border, a.inv_hs, rows, cols);
                    const float fm = o[2] + tm * d[2] - tri_height(cm, cols, vs, cb, exm);
                    if (!exm) break;                                     // a hole in the shell inside the bracket: keep the bracket
                    if ((fm < 0.f) == (flo < 0.f)) { lo = tm; flo = fm; } else { hi = tm; fhi = fm; }
                }
                best = fminf(best, lo + (hi - lo) * flo / (flo - fhi));
            }
        }
        if (best < 1e29f) { hit = best; break; }
        t_prev = t; g_prev = g; h_prev = h; ex_prev = ex;
        if (cfl && (cx != seen_cx || cy != seen_cy)) {
            seen_cx = cx; seen_cy = cy;
            // where the ray leaves this block's footprint (the outermost blocks extend outwards: lookups clamp to the map's edge)
            const float x0 = (float)(cx << S) * hs - border, y0 = (float)(cy << S) * hs - border;
            const float bx = d[0] > 0.f ? (cx == crows - 1 ? 1e30f : x0 + block) : (cx == 0 ? -1e30f : x0);
            const float by = d[1] > 0.f ? (cy == ccols - 1 ? 1e30f : y0 + block) : (cy == 0 ? -1e30f : y0);
            const float tx = d[0] != 0.f ? (bx - o[0]) * idx : 1e30f, ty = d[1] != 0.f ? (by - o[1]) * idy : 1e30f;
            const float t_exit = fminf(fminf(tx, ty), far);
            const int kl = min((int)floorf((t_exit - 1e-4f) / dt), nsteps - 1);   // the last lattice sample safely inside (the final, clamped one is always marched)
            if (kl > k + 1) {
                const float tl = (float)kl * dt, zl = o[2] + tl * d[2];
                const float hmax = vs * (float)cfl[cx * ccols + cy];
                const float cmin = (cm && cce) ? vs * (float)cce[cx * ccols + cy] : 1e30f;
                if (fminf(z, zl) > hmax + 1e-3f && fmaxf(z, zl) < cmin - 1e-3f) {
                    const Cell cl = cell_of(o[0] + tl * d[0], o[1] + tl * d[1], border, a.inv_hs, rows, cols);
                    if ((cl.ix >> S) == cx && (cl.iy >> S) == cy) {          // rounding never moves it out, but the jump is only taken if it did not
                        t_prev = tl; g_prev = zl - tri_height(hm, cols, vs, cl, dummy);
                        h_prev = cm ? zl - tri_height(cm, cols, vs, cl, ex_prev) : 0.f;
                        k = kl; seen_cx = -1;                                // the block after this one is asked afresh
                    }
                }
            }
        }
    }
    const float dd = fminf(fmaxf(hit, near), far);
    float v = (dd - near) / (far - near) - 0.5f;
    // noise (process_depth_image :166-168), Philox stream QA_TSC_DEPTH_STREAM keyed by (seed; global env id, step)
    const int64_t step = a.io.step_dev ? *a.io.step_dev : c.step;
    const uint32_t env = (uint32_t)(e + c.env_id_offset), s_lo = (uint32_t)step, s_hi = (uint32_t)((uint64_t)step >> 32);
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
    if (cfg->num_envs <= 0 || cfg->num_envs > (1 << 22) || a.wc <= 0 || a.hc <= 0 || cfg->buffer_len < 1 || cfg->map_rows < 2 || cfg->map_cols < 2 || !(cfg->horizontal_scale > 0.f) ||
        cfg->coarse_log2 < 0 || cfg->coarse_log2 > 6 || !(cfg->far_clip > cfg->near_clip) || !(cfg->horizontal_fov_deg > 0.f && cfg->horizontal_fov_deg < 180.f)) {
        snprintf(qa_err_buf, sizeof(qa_err_buf), "qa_tsc_depth_update: inconsistent configuration"); return QA_E_ARG;
    }
    a.tan_h = (float)tan((double)cfg->horizontal_fov_deg * 3.14159265358979323846 / 360.0);
    a.tan_v = (float)(tan((double)cfg->horizontal_fov_deg * 3.14159265358979323846 / 360.0) * cfg->height / cfg->width);
    a.inv_hs = 1.0f / cfg->horizontal_scale;
    const int npix = a.wc * a.hc;
    a.tiles = (npix + 255) / 256;
    const int64_t groups = (cfg->num_envs + 7) / 8 * 8 * a.tiles;          // envs rounded up to the 8 XCDs
    hipLaunchKernelGGL(qa_tsc_depth_kernel, dim3((unsigned)groups), dim3(256), 0, (hipStream_t)stream, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { snprintf(qa_err_buf, sizeof(qa_err_buf), "qa_tsc_depth_update: %s", hipGetErrorString(e)); return QA_E_DEVICE; }
    return QA_OK;
}
