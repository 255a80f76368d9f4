// abi_demo.cpp -- a plain C++ consumer of the C ABI (include/so100_sim.h): no Python, no PyTorch.
// Build:  hipcc -O2 -o abi_demo examples/abi_demo.cpp -Iinclude -Lso100_mujoco_rl_amd -lso100sim -Wl,-rpath,$PWD/so100_mujoco_rl_amd
// Run:    ./abi_demo [num_envs] [steps]     (prints a checksum of the final observations)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "so100_sim.h"

#define CHECK(x) do { if ((x) != hipSuccess) { std::fprintf(stderr, "HIP error at %s:%d\n", __FILE__, __LINE__); return 2; } } while (0)

int main(int argc, char** argv) {
    const int n = argc > 1 ? std::atoi(argv[1]) : 4096, steps = argc > 2 ? std::atoi(argv[2]) : 100;
    so100_config cfg = {};
    cfg.env_kind = SO100_ENV01; cfg.num_envs = n; cfg.device = 0; cfg.flags = SO100_F_REFERENCE;
    cfg.solver_iters = 3; cfg.contact_iters = 4; cfg.frame_skip = 16; cfg.max_episode_steps = 4000; cfg.seed = 42;
    so100_sim* sim = nullptr;
    if (so100_create(&cfg, &sim) != 0) { std::fprintf(stderr, "so100_create: %s\n", so100_last_error()); return 1; }
    const int od = so100_obs_dim(cfg.env_kind);
    float *act, *obs, *rew; uint8_t *done, *trunc;
    CHECK(hipMalloc(&act, sizeof(float)*6*n)); CHECK(hipMalloc(&obs, sizeof(float)*od*n)); CHECK(hipMalloc(&rew, sizeof(float)*n));
    CHECK(hipMalloc(&done, n)); CHECK(hipMalloc(&trunc, n));
    std::vector<float> h_act(6*(size_t)n);
    for (size_t i = 0; i < h_act.size(); i++) h_act[i] = (float)((i*2654435761u) % 2001) / 1000.0f - 1.0f;   // fixed pseudo-random actions
    CHECK(hipMemcpy(act, h_act.data(), sizeof(float)*h_act.size(), hipMemcpyHostToDevice));
    hipStream_t st; CHECK(hipStreamCreate(&st));
    if (so100_reset(sim, nullptr, nullptr, obs, st) != 0) { std::fprintf(stderr, "so100_reset: %s\n", so100_last_error()); return 1; }
    so100_step_io io = {};
    io.act_dev = act; io.obs_dev = obs; io.rew_dev = rew; io.done_dev = done; io.trunc_dev = trunc;
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    CHECK(hipEventRecord(e0, st));
    for (int t = 0; t < steps; t++)
        if (so100_step(sim, &io, st) != 0) { std::fprintf(stderr, "so100_step: %s\n", so100_last_error()); return 1; }
    CHECK(hipEventRecord(e1, st)); CHECK(hipStreamSynchronize(st));
    float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<float> h_obs((size_t)od*n), h_rew(n);
    CHECK(hipMemcpy(h_obs.data(), obs, sizeof(float)*h_obs.size(), hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(h_rew.data(), rew, sizeof(float)*n, hipMemcpyDeviceToHost));
    double cs = 0, rs = 0; for (float v : h_obs) cs += v; for (float v : h_rew) rs += v;
    std::printf("envs %d steps %d  %.1f us/step  %.2f M env-steps/s  obs_checksum %.6f  reward_sum %.6f\n", n, steps, ms*1e3/steps, n*(double)steps/ms/1e3, cs, rs);
    so100_destroy(sim);
    return 0;
}
