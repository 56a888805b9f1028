// Dev tool (r5): what one wavefront alone on a SIMD can issue on gfx950, and what it costs to share the SIMD / the CU.
//   hipcc --offload-arch=gfx950 -O3 tools/valu_rate.hip -o tools/_prof/valu_rate && tools/_prof/valu_rate
// Kernels: CH independent chains of v_fma_f32 (PK = 0) or v_pk_fma_f32 (PK = 1), each chain dependent on itself; single-wave workgroups.
// Grids: 256 (one wave per CU), 1024 (one per SIMD), 2048 (two per SIMD), 4096.  Printed: shader cycles per instruction of wave 0 and the
// wall time of the launch.  The env-step kernel runs one wave per CU at 4096 envs and is bound by exactly this number.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <algorithm>
typedef float f2 __attribute__((ext_vector_type(2)));
constexpr int ITERS = 2048;

template <int CH, int PK>
__global__ __launch_bounds__(64) void k(float *out, long long *ticks, float seed) {
    const int lane = threadIdx.x;
    long long t0, t1;
    float r = 0.f;
    if (PK == 0) {
        float a[CH];
        const float b = seed + lane * 1e-9f, c = 1e-7f;
#pragma unroll
        for (int i = 0; i < CH; ++i) a[i] = seed * (i + 1);
        t0 = __builtin_amdgcn_s_memtime();
        for (int it = 0; it < ITERS; ++it) {
#pragma unroll
            for (int i = 0; i < CH; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
        }
        t1 = __builtin_amdgcn_s_memtime();
#pragma unroll
        for (int i = 0; i < CH; ++i) r += a[i];
    } else {
        f2 a[CH];
        const f2 b = f2{seed + lane * 1e-9f, seed}, c = f2{1e-7f, 2e-7f};
#pragma unroll
        for (int i = 0; i < CH; ++i) a[i] = f2{seed * (i + 1), seed};
        t0 = __builtin_amdgcn_s_memtime();
        for (int it = 0; it < ITERS; ++it) {
#pragma unroll
            for (int i = 0; i < CH; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
        }
        t1 = __builtin_amdgcn_s_memtime();
#pragma unroll
        for (int i = 0; i < CH; ++i) r += a[i].x + a[i].y;
    }
    out[blockIdx.x * 64 + lane] = r;
    if (lane == 0) ticks[blockIdx.x] = t1 - t0;
}

// DPP quad reduction chain and LDS round trip, the other two things the PGS sweep is made of
template <int MODE>   // 0: v_add_f32 dpp quad_perm chain (dependent), 1: ds_read_b32 dependent (address from the value read), 2: 8 independent ds_read_b32 then use
__global__ __launch_bounds__(64) void k2(float *out, long long *ticks, float seed) {
    __shared__ float lds[64 * 64];
    const int lane = threadIdx.x;
    for (int i = lane; i < 64 * 64; i += 64) lds[i] = 0.f;
    __builtin_amdgcn_s_waitcnt(0);
    float a = seed + lane;
    long long t0 = __builtin_amdgcn_s_memtime();
    if (MODE == 0) {
        for (int it = 0; it < ITERS; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("s_nop 1\n v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a));
        }
    } else if (MODE == 1) {
        int idx = lane;
        for (int it = 0; it < ITERS; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) { float v = lds[idx]; idx = lane + (int)v; a += v; }
        }
    } else {
        for (int it = 0; it < ITERS; ++it) {
            float v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = lds[lane + 64 * ((i + it) & 63)];
#pragma unroll
            for (int i = 0; i < 8; ++i) a += v[i];
            asm volatile("" : "+v"(a));
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 64 + lane] = a;
    if (lane == 0) ticks[blockIdx.x] = t1 - t0;
}

template <typename F> void run(const char *name, F kern, int per_iter, float *out, long long *ticks, int blocks) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(kern, dim3(blocks), dim3(64), 0, 0, out, ticks, 1.0f);
    hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(kern, dim3(blocks), dim3(64), 0, 0, out, ticks, 1.0f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(blocks); hipMemcpy(h.data(), ticks, blocks * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    const double n = (double)ITERS * per_iter;
    printf("%-44s waves %5d: %8.1f us/launch | cycles per instruction: min %.2f median %.2f max %.2f\n", name, blocks, ms * 1e3 / 10,
           h[0] / n, h[blocks / 2] / n, h[blocks - 1] / n);
}

int main() {
    float *out; long long *ticks;
    hipMalloc(&out, 8192 * 64 * 4); hipMalloc(&ticks, 8192 * 8);
    for (int blocks : {256, 1024, 2048, 4096}) {
        run("v_fma_f32, 1 dependent chain", k<1, 0>, 1, out, ticks, blocks);
        run("v_fma_f32, 2 chains", k<2, 0>, 2, out, ticks, blocks);
        run("v_fma_f32, 4 chains", k<4, 0>, 4, out, ticks, blocks);
        run("v_fma_f32, 8 chains", k<8, 0>, 8, out, ticks, blocks);
        run("v_pk_fma_f32, 1 dependent chain", k<1, 1>, 1, out, ticks, blocks);
        run("v_pk_fma_f32, 2 chains", k<2, 1>, 2, out, ticks, blocks);
        run("v_pk_fma_f32, 4 chains", k<4, 1>, 4, out, ticks, blocks);
        run("v_pk_fma_f32, 8 chains", k<8, 1>, 8, out, ticks, blocks);
        run("v_add_f32_dpp quad_perm, dependent (+s_nop 1)", k2<0>, 8, out, ticks, blocks);
        run("ds_read_b32, dependent round trip", k2<1>, 8, out, ticks, blocks);
        run("ds_read_b32 x8 independent + 8 adds (per read)", k2<2>, 8, out, ticks, blocks);
    }
    return 0;
}
