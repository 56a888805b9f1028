// tools/gemm_probe.hip -- phase ablation of the learner GEMM kernel (csrc/qa_gemm.hip) without torch:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/gemm_probe.hip -o gpurun_out/gemm_probe && ./gpurun_out/gemm_probe
// Times the forward product 24576 x K -> N with: the full kernel, no global loads / LDS writes after the first tile (1),
// no MFMAs (2), no LDS fragment reads (3); operands uniform [-1, 1).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
thread_local char qa_err_buf[512];
#include "../quadrupedal_agility_amd/csrc/qa_gemm.hip"

template <int BA, int BB, int AV, int BV, int ABL>
static float run(GemmArgs g, int reps) {
    g.na = (g.a_count + BA - 1) / BA; g.nb = (g.b_count + BB - 1) / BB;
    dim3 grid(g.na * g.nb * g.nsplit);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((qa_gemm_kernel<BA, BB, false, false, AV, BV, 1, false, ABL>), grid, dim3(256), 0, 0, g);
    hipEventRecord(a, 0);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((qa_gemm_kernel<BA, BB, false, false, AV, BV, 1, false, ABL>), grid, dim3(256), 0, 0, g);
    hipEventRecord(b, 0); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms * 1e3f / reps;
}

int main(int argc, char **argv) {
    const int M = 24576;
    const bool pmc = argc > 1;          // `gemm_probe pmc`: the full 128 x 128 vec4 kernel only (for a counter pass)
    const int shapes[][2] = {{671, 512}, {672, 512}, {512, 256}, {256, 128}};
    for (auto &sh : shapes) {
        const int K = sh[0], N = sh[1];
        std::vector<float> hx((size_t)M * K), hw((size_t)N * K), hb(N);
        for (auto &v : hx) v = rand() / (float)RAND_MAX * 2 - 1;
        for (auto &v : hw) v = (rand() / (float)RAND_MAX * 2 - 1) * 0.05f;
        for (auto &v : hb) v = 0.01f;
        float *x, *w, *b, *y;
        hipMalloc(&x, hx.size() * 4); hipMalloc(&w, hw.size() * 4); hipMalloc(&b, N * 4); hipMalloc(&y, (size_t)M * N * 4);
        hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice); hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(b, hb.data(), N * 4, hipMemcpyHostToDevice);
        GemmArgs g = {};
        g.A = w; g.lda = K; g.a_count = N; g.B = x; g.ldb = K; g.b_count = M; g.kred = K; g.k_per_split = (K + 15) / 16 * 16; g.nsplit = 1;
        g.out = y; g.ldo = N; g.bias = b; g.act = 1; g.alpha = 1.f; g.o_vec = 4;
        const double fl = 2.0 * M * K * N;
        const bool v4 = K % 4 == 0;
        float t[4][3];
        if (pmc) { if (K == 672) printf("%f us\n", run<128, 128, 4, 4, 0>(g, 20)); continue; }
        if (v4) {
            t[0][0] = run<128, 128, 4, 4, 0>(g, 20); t[1][0] = run<128, 128, 4, 4, 1>(g, 20); t[2][0] = run<128, 128, 4, 4, 2>(g, 20); t[3][0] = run<128, 128, 4, 4, 3>(g, 20);
            t[0][1] = run<128, 64, 4, 4, 0>(g, 20); t[1][1] = run<128, 64, 4, 4, 1>(g, 20); t[2][1] = run<128, 64, 4, 4, 2>(g, 20); t[3][1] = run<128, 64, 4, 4, 3>(g, 20);
            t[0][2] = run<64, 64, 4, 4, 0>(g, 20); t[1][2] = run<64, 64, 4, 4, 1>(g, 20); t[2][2] = run<64, 64, 4, 4, 2>(g, 20); t[3][2] = run<64, 64, 4, 4, 3>(g, 20);
        } else {
            t[0][0] = run<128, 128, 1, 1, 0>(g, 20); t[1][0] = run<128, 128, 1, 1, 1>(g, 20); t[2][0] = run<128, 128, 1, 1, 2>(g, 20); t[3][0] = run<128, 128, 1, 1, 3>(g, 20);
            t[0][1] = run<128, 64, 1, 1, 0>(g, 20); t[1][1] = run<128, 64, 1, 1, 1>(g, 20); t[2][1] = run<128, 64, 1, 1, 2>(g, 20); t[3][1] = run<128, 64, 1, 1, 3>(g, 20);
            t[0][2] = run<64, 64, 1, 1, 0>(g, 20); t[1][2] = run<64, 64, 1, 1, 1>(g, 20); t[2][2] = run<64, 64, 1, 1, 2>(g, 20); t[3][2] = run<64, 64, 1, 1, 3>(g, 20);
        }
        const char *names[4] = {"full", "no loads/LDS writes", "no MFMA", "no LDS frag reads"};
        const char *tiles[3] = {"128x128", "128x64", "64x64"};
        for (int c = 0; c < 3; ++c)
            for (int v = 0; v < 4; ++v)
                printf("K=%d N=%d vec%d tile %s  %-22s %8.1f us  %6.1f TFLOP/s-equivalent\n", K, N, v4 ? 4 : 1, tiles[c], names[v], t[v][c], fl / t[v][c] / 1e6);
        hipFree(x); hipFree(w); hipFree(b); hipFree(y);
    }
    return 0;
}
