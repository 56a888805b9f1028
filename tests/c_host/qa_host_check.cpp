// A host written against include/qa_sim.h alone -- no Python, no torch: the drop-in boundary used the way a C/C++ engine would.
// It drives the HIP library (libqa_sim.so, linked) and the CPU oracle (oracle/libqa_oracle.so, dlopen'ed, prefix qo_) through the
// SAME C ABI, starting every step from the oracle's arena, and reports how many envs agree.  TEST INFRASTRUCTURE (it loads the
// oracle); built and run by tests/test_c_host.py on a GPU box:
//   hipcc --offload-arch=gfx950 -O2 tests/c_host/qa_host_check.cpp -Iinclude -Lquadrupedal_agility_amd/csrc -lqa_sim -ldl -o qa_host_check
//   LD_LIBRARY_PATH=quadrupedal_agility_amd/csrc ./qa_host_check cfg.bin oracle/libqa_oracle.so 20
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "qa_sim.h"

#define CK(x) do { int rc_ = (x); if (rc_ != 0) { fprintf(stderr, "%s -> %d (%s)\n", #x, rc_, qa_last_error()); return 2; } } while (0)
#define HK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s -> %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

template <class F> static F sym(void *lib, const char *name) {
    void *p = dlsym(lib, name);
    if (!p) { fprintf(stderr, "missing %s\n", name); exit(2); }
    return reinterpret_cast<F>(p);
}

int main(int argc, char **argv) {
    if (argc < 4) { fprintf(stderr, "usage: %s cfg.bin liboracle.so steps\n", argv[0]); return 2; }
    qa_config cfg;
    FILE *f = fopen(argv[1], "rb");
    if (!f || fread(&cfg, sizeof(cfg), 1, f) != 1) { fprintf(stderr, "cannot read %s (%zu bytes expected)\n", argv[1], sizeof(cfg)); return 2; }
    fclose(f);
    const int steps = atoi(argv[3]);
    if (cfg.abi_version != qa_abi_version()) { fprintf(stderr, "ABI %d vs library %d\n", cfg.abi_version, qa_abi_version()); return 2; }
    void *ol = dlopen(argv[2], RTLD_NOW);
    if (!ol) { fprintf(stderr, "dlopen: %s\n", dlerror()); return 2; }
    auto qo_create = sym<int (*)(const qa_config *, void *, int64_t, void *, qa_sim **)>(ol, "qo_create");
    auto qo_reset_all = sym<int (*)(qa_sim *, int64_t, void *)>(ol, "qo_reset_all");
    auto qo_env_step = sym<int (*)(qa_sim *, const float *, int32_t, int64_t, void *)>(ol, "qo_env_step");

    const int64_t bytes = qa_arena_bytes(&cfg);
    const int N = cfg.num_envs;
    void *d_arena = nullptr; float *d_act = nullptr;
    HK(hipMalloc(&d_arena, bytes)); HK(hipMalloc((void **)&d_act, sizeof(float) * N * 12));
    const size_t padded = ((size_t)bytes + 255) / 256 * 256;
    char *h_arena = (char *)aligned_alloc(256, padded);         // arenas are 256-byte aligned (QA_E_ARENA otherwise)
    std::vector<char> back(bytes);
    hipStream_t st; HK(hipStreamCreate(&st));
    qa_sim *hs = nullptr, *os = nullptr;
    CK(qa_create(&cfg, d_arena, bytes, st, &hs));
    if (qo_create(&cfg, h_arena, bytes, nullptr, &os) != 0) { fprintf(stderr, "qo_create failed\n"); return 2; }
    int64_t off_obs, off_rew, shape[3]; int32_t nd, dt;
    CK(qa_tensor_info(&cfg, QA_T_OBS, &off_obs, shape, &nd, &dt));
    const int obs_w = (int)shape[1];
    CK(qa_tensor_info(&cfg, QA_T_REW, &off_rew, shape, &nd, &dt));
    if (qo_reset_all(os, 0, nullptr) != 0) return 2;
    std::vector<float> act((size_t)N * 12);
    int64_t ok_envs = 0, total = 0; double worst = 0.0;
    for (int s = 0; s < steps; ++s) {
        for (int e = 0; e < N; ++e) for (int j = 0; j < 12; ++j) act[(size_t)e * 12 + j] = 0.8f * sinf(0.37f * (float)(e + 1) + 1.3f * (float)j + 0.21f * (float)s);
        HK(hipMemcpyAsync(d_arena, h_arena, bytes, hipMemcpyHostToDevice, st));       // same start for both engines
        HK(hipMemcpyAsync(d_act, act.data(), sizeof(float) * N * 12, hipMemcpyHostToDevice, st));
        CK(qa_env_step(hs, d_act, 0, s, st));
        HK(hipMemcpyAsync(back.data(), d_arena, bytes, hipMemcpyDeviceToHost, st));
        if (qo_env_step(os, act.data(), 0, s, nullptr) != 0) return 2;
        HK(hipStreamSynchronize(st));
        const float *oo = (const float *)(h_arena + off_obs), *ho = (const float *)(back.data() + off_obs);
        const float *orw = (const float *)(h_arena + off_rew), *hr = (const float *)(back.data() + off_rew);
        for (int e = 0; e < N; ++e) {
            bool good = fabsf(orw[e] - hr[e]) <= 2e-4f + 1e-3f * fabsf(orw[e]);
            for (int c = 0; c < obs_w && good; ++c) {
                const float d = fabsf(oo[(size_t)e * obs_w + c] - ho[(size_t)e * obs_w + c]);
                if (d > 3e-3f + 1e-3f * fabsf(oo[(size_t)e * obs_w + c])) good = false;
                if (good && d > worst) worst = d;
            }
            ok_envs += good; ++total;
        }
    }
    printf("c_host: %lld of %lld env-steps agree (%.4f), worst accepted |diff| %.2e, %d envs x %d steps\n", (long long)ok_envs, (long long)total,
           (double)ok_envs / (double)total, worst, N, steps);
    CK(qa_destroy(hs));
    return (double)ok_envs / (double)total > 0.97 ? 0 : 1;
}
