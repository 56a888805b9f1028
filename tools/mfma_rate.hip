// Dev tool: what bounds the fused-MLP inner loop on gfx950?  fp32 MFMA issue alone / + LDS A reads / + weight stream from L2.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_rate.hip -o tools/_prof/mfma_rate && tools/_prof/mfma_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int TPW = 8, ITERS = 44 * 8;     // 8x the critic's first layer

template <int MODE>   // 0 mfma only, 1 + ds_read, 2 + global stream (lockstep), 3 + global stream (per-WG phase shift)
__global__ __launch_bounds__(256) void k(const float *w, float *out, long long *ticks, int nblk16) {
    __shared__ __attribute__((aligned(16))) float lds[16 * 676];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 16 * 676; i += 256) lds[i] = 0.001f * (i % 97);
    __syncthreads();
    f4 acc[TPW];
    for (int i = 0; i < TPW; ++i) acc[i] = f4{0, 0, 0, 0};
    const f4 *wp = reinterpret_cast<const f4 *>(w) + lane;
    const float *src = lds + (lane & 15) * 676 + 4 * (lane >> 4);
    f4 bw[2][TPW], av[2];
    const int phase = MODE == 3 ? (blockIdx.x * 7) % nblk16 : 0;
    for (int i = 0; i < TPW; ++i) bw[0][i] = MODE >= 2 ? wp[((phase % nblk16) * 32 + wave * TPW + i) * 64] : f4{1.f, 2.f, 3.f, 4.f};
    av[0] = MODE >= 1 ? *reinterpret_cast<const f4 *>(src) : f4{1.f, 1.f, 1.f, 1.f};
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int j0 = 0; j0 < ITERS; j0 += 2) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int j = j0 + u + 1;
            if (MODE >= 2) {
#pragma unroll
                for (int i = 0; i < TPW; ++i) bw[u ^ 1][i] = wp[(((j + phase) % nblk16) * 32 + wave * TPW + i) * 64];
            } else {
#pragma unroll
                for (int i = 0; i < TPW; ++i) bw[u ^ 1][i] = bw[u][i];
            }
            av[u ^ 1] = MODE >= 1 ? *reinterpret_cast<const f4 *>(src + 16 * (j % 42)) : av[u];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int i = 0; i < TPW; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][s], bw[u][i][s], acc[i], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    float r = 0;
    for (int i = 0; i < TPW; ++i) r += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = r;
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}

template <int MODE> void run(const char *name, const float *w, float *out, long long *ticks, int blocks) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int nblk16 = 42;
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, w, out, ticks, nblk16);
    hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, w, out, ticks, nblk16);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(blocks); hipMemcpy(h.data(), ticks, blocks * 8, hipMemcpyDeviceToHost);
    double us = ms * 1e3 / 20, flops = (double)blocks * 4 * ITERS * 32 * 2048;
    printf("%-34s blocks %4d: %8.1f us/launch, %6.1f TFLOP/s, ticks per k-block (WG 0) %.0f, per MFMA %.1f\n", name, blocks, us, flops / us / 1e6,
           (double)h[0] / ITERS, (double)h[0] / ITERS / 32);
}

int main() {
    float *w, *out; long long *ticks;
    hipMalloc(&w, 42 * 32 * 1024 + 4096); hipMemset(w, 0, 42 * 32 * 1024 + 4096);
    hipMalloc(&out, 1024 * 256 * 4); hipMalloc(&ticks, 1024 * 8);
    for (int blocks : {256, 512}) {
        run<0>("mfma only", w, out, ticks, blocks);
        run<1>("mfma + LDS A read", w, out, ticks, blocks);
        run<2>("mfma + LDS + weight stream", w, out, ticks, blocks);
        run<3>("mfma + LDS + weights, WGs dephased", w, out, ticks, blocks);
    }
    return 0;
}
