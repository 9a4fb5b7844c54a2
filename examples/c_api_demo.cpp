// Torch-free client of the libevac C ABI (include/evac.h): raw hipMalloc buffers, no Python.
//   hipcc -O2 --offload-arch=gfx950 examples/c_api_demo.cpp -Iinclude -Levacuation_amd -levac -Wl,-rpath,$PWD/evacuation_amd -o examples/c_api_demo
//   ./examples/c_api_demo [num_envs] [n_ped] [steps] [seed]
// Prints one line: env-steps/s and an FNV-1a checksum of the packed rollout slab, which
// tests/test_gpu_c_client.py compares with the same rollout driven from Python.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "evac.h"

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 2; } } while (0)
#define EVAC_OK_(x, h) do { int rc_ = (x); if (rc_ != EVAC_OK) { std::fprintf(stderr, "libevac error %d: %s (line %d)\n", rc_, evac_last_error(h), __LINE__); return 3; } } while (0)

int main(int argc, char** argv) {
    const int E = argc > 1 ? std::atoi(argv[1]) : 4096;
    const int N = argc > 2 ? std::atoi(argv[2]) : 60;
    const int T = argc > 3 ? std::atoi(argv[3]) : 100;
    const uint64_t seed = argc > 4 ? std::strtoull(argv[4], nullptr, 0) : 0x5EED0001ull;

    evac_config_t cfg{};                       // EnvConfig / EnvWrappersConfig defaults of the reference
    cfg.number_of_pedestrians = N;
    cfg.width = cfg.height = 1.0f;
    cfg.step_size = 0.01f;
    cfg.noise_coef = 0.2f;
    cfg.eps = 1e-8f;
    cfg.enslaving_degree = 1.0f;
    cfg.is_new_exiting_reward = 1;
    cfg.is_new_followers_reward = 1;
    cfg.init_reward_each_step = -1.0f;
    cfg.max_timesteps = 2000;
    cfg.positions = EVAC_POS_GRAV;
    cfg.statuses = EVAC_STAT_NO;
    cfg.type = EVAC_TYPE_DICT;
    cfg.alpha = 3.0f;

    evac_handle_t h = nullptr;
    EVAC_OK_(evac_create(&cfg, E, 0, seed, 0, &h), nullptr);
    const int64_t D = evac_obs_dim(h);

    float *ped, *agent, *acc, *obs, *slab;
    uint8_t* status;
    int32_t* clock;
    HIP_OK(hipMalloc(&ped, sizeof(float) * 4 * (size_t)E * N));
    HIP_OK(hipMalloc(&status, (size_t)E * N));
    HIP_OK(hipMalloc(&agent, sizeof(float) * 4 * E));
    HIP_OK(hipMalloc(&clock, sizeof(int32_t) * 4 * E));
    HIP_OK(hipMalloc(&acc, sizeof(float) * 4 * E));
    HIP_OK(hipMalloc(&obs, sizeof(float) * D * E));
    const size_t slab_floats = (size_t)T * E * (D + 3);
    HIP_OK(hipMalloc(&slab, sizeof(float) * slab_floats));
    HIP_OK(hipMemset(ped, 0, sizeof(float) * 4 * (size_t)E * N));
    HIP_OK(hipMemset(status, 0, (size_t)E * N));
    HIP_OK(hipMemset(agent, 0, sizeof(float) * 4 * E));
    HIP_OK(hipMemset(clock, 0, sizeof(int32_t) * 4 * E));
    HIP_OK(hipMemset(acc, 0, sizeof(float) * 4 * E));

    hipStream_t stream;
    HIP_OK(hipStreamCreate(&stream));
    EVAC_OK_(evac_bind_state(h, ped, status, agent, clock, acc), h);
    // optional workspace (load schedule of large batches, exchange areas of the team kernels): zero-initialised, 256-byte
    // aligned (hipMalloc aligns to 256 B); results are bit-identical with and without it
    void* workspace = nullptr;
    const int64_t ws_bytes = evac_workspace_bytes(h);
    HIP_OK(hipMalloc(&workspace, (size_t)ws_bytes));
    HIP_OK(hipMemset(workspace, 0, (size_t)ws_bytes));
    EVAC_OK_(evac_bind_workspace(h, workspace, ws_bytes), h);
    EVAC_OK_(evac_reset(h, nullptr, nullptr, obs, stream), h);
    EVAC_OK_(evac_rollout(h, T, nullptr, nullptr, slab, nullptr, 0, nullptr, nullptr, stream), h);      // the checked launch
    HIP_OK(hipStreamSynchronize(stream));
    std::vector<float> host(slab_floats);
    HIP_OK(hipMemcpy(host.data(), slab, sizeof(float) * slab_floats, hipMemcpyDeviceToHost));
    uint64_t fnv = 1469598103934665603ull;
    const unsigned char* bytes = reinterpret_cast<const unsigned char*>(host.data());
    for (size_t i = 0; i < slab_floats * sizeof(float); ++i) { fnv ^= bytes[i]; fnv *= 1099511628211ull; }

    const int reps = 10;                                                                         // timing
    const auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < reps; ++r) EVAC_OK_(evac_rollout(h, T, nullptr, nullptr, slab, nullptr, 0, nullptr, nullptr, stream), h);
    HIP_OK(hipStreamSynchronize(stream));
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    std::printf("{\"client\": \"c_api_demo\", \"num_envs\": %d, \"n_ped\": %d, \"steps\": %d, \"obs_dim\": %lld, "
                "\"env_steps_per_s\": %.4e, \"slab_fnv1a\": \"%016llx\"}\n",
                E, N, T, (long long)D, (double)E * T * reps / dt, (unsigned long long)fnv);
    evac_destroy(h);
    void* bufs[] = {ped, status, agent, clock, acc, obs, slab, workspace};
    for (void* b : bufs) (void)hipFree(b);
    return 0;
}
