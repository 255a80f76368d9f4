// so100_policy.hpp -- fused rollout-side actor-critic forward (SB3 "MlpPolicy" for a Box action space).
//
// This is a CALLER-side helper of the hot path (the rollout collector), not part of the
// simulator: it replaces the ~28 small PyTorch kernels SB3's `policy.forward(obs)` + `clip` + rollout-buffer writes
// launch per vectorised step (main.py:56-64 -> stable_baselines3 OnPolicyAlgorithm.collect_rollouts) with ONE launch.
//   pi tower: obs -> Linear(od,64) tanh -> Linear(64,64) tanh -> Linear(64,6) = mean;  log_std state independent
//   vf tower: obs -> Linear(od,64) tanh -> Linear(64,64) tanh -> Linear(64,1) = value
//   action = mean + exp(log_std) * eps, eps ~ N(0,1) (Philox4x32-10 + Box-Muller, or injected), log_prob, clip to [-1,1]
// Mapping: workgroup = 4 waves = 64 envs; lane = env; wave w owns hidden units [16w, 16w+16) of BOTH towers, so every
// weight is wave-uniform (s_load -> SGPR operand of v_fmac).  Hidden activations go through LDS as [unit][env]
// (lane-contiguous, conflict-free).  MFMA is not used: fp32 MFMA runs at the VALU rate on gfx950 and the tile would
// need a lane<->matrix transpose; at 10.5 kMAC per env the kernel is launch/latency-bound anyway.
#pragma once
#include <hip/hip_runtime.h>
#include "so100_task.hpp"

namespace so100 {

struct PolicyWeights {                 // device pointers, PyTorch nn.Linear layout weight[out][in], row-major
    const float *pi_w0, *pi_b0, *pi_w1, *pi_b1, *mu_w, *mu_b, *log_std;
    const float *vf_w0, *vf_b0, *vf_w1, *vf_b1, *v_w, *v_b;
};
struct PolicyIO {
    const float* obs;                  // [N][OD]
    const float* noise;                // [N][6] standard normals, nullable (then Philox)
    float* act_env;                    // [N][6] clipped to [-1,1]  -> so100_step
    float* act_raw;                    // [N][6] unclipped (what SB3 stores), nullable
    float* value;                      // [N], nullable
    float* logp;                       // [N], nullable
    float* rollout_row;                // [N][OD+10]: obs | raw action | reward | done | value | logp ; nullable
};

__device__ __forceinline__ float fast_tanh(float x) {
    // tanh(x) = 1 - 2/(exp(2x)+1); exp via v_exp_f32 (2^x), rcp with one Newton step: abs error < 2e-7
    const float e = __builtin_amdgcn_exp2f(x * 2.885390081777927f);          // exp(2x)
    return 1.0f - 2.0f*trcp(e + 1.0f);
}

template <int OD>
__global__ void __launch_bounds__(256) so100_policy_forward_kernel(int n, PolicyWeights w, PolicyIO io, uint32_t seed_lo, uint32_t seed_hi,
                                                            uint32_t env_id_offset, uint32_t step_counter) {
    __shared__ float h1[2][64][64];        // [tower][unit][env]
    __shared__ float h2[2][64][64];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);        // wave-uniform => scalar weight loads
    const int env = blockIdx.x*64 + lane;
    const bool live = env < n;
    float ob[OD];
#pragma unroll
    for (int k = 0; k < OD; k++) ob[k] = live ? io.obs[(size_t)env*OD + k] : 0.0f;
    const int u0 = wave*16;
    // ---- layer 1
#pragma unroll 4
    for (int j = 0; j < 16; j++) {
        const int u = u0 + j;
        float ap = w.pi_b0[u], av = w.vf_b0[u];
#pragma unroll
        for (int k = 0; k < OD; k++) { ap = __builtin_fmaf(w.pi_w0[u*OD + k], ob[k], ap); av = __builtin_fmaf(w.vf_w0[u*OD + k], ob[k], av); }
        h1[0][u][lane] = fast_tanh(ap); h1[1][u][lane] = fast_tanh(av);
    }
    __syncthreads();
    // ---- layer 2
    float accp[16], accv[16];
#pragma unroll
    for (int j = 0; j < 16; j++) { accp[j] = w.pi_b1[u0 + j]; accv[j] = w.vf_b1[u0 + j]; }
#pragma unroll 2
    for (int k = 0; k < 64; k++) {
        const float xp = h1[0][k][lane], xv = h1[1][k][lane];
#pragma unroll
        for (int j = 0; j < 16; j++) {
            accp[j] = __builtin_fmaf(w.pi_w1[(u0 + j)*64 + k], xp, accp[j]);
            accv[j] = __builtin_fmaf(w.vf_w1[(u0 + j)*64 + k], xv, accv[j]);
        }
    }
#pragma unroll
    for (int j = 0; j < 16; j++) { h2[0][u0 + j][lane] = fast_tanh(accp[j]); h2[1][u0 + j][lane] = fast_tanh(accv[j]); }
    __syncthreads();
    // ---- heads: wave 0 -> mean / sample / log-prob, wave 1 -> value
    if (wave == 0) {
        float mean[6];
#pragma unroll
        for (int a = 0; a < 6; a++) mean[a] = w.mu_b[a];
#pragma unroll 4
        for (int k = 0; k < 64; k++) {
            const float x = h2[0][k][lane];
#pragma unroll
            for (int a = 0; a < 6; a++) mean[a] = __builtin_fmaf(w.mu_w[a*64 + k], x, mean[a]);
        }
        float eps[8];
        if (io.noise) {
#pragma unroll
            for (int a = 0; a < 6; a++) eps[a] = live ? io.noise[(size_t)env*6 + a] : 0.0f;
        } else {
#pragma unroll
            for (int b = 0; b < 2; b++) {
                uint32_t r[4];
                philox4x32(env_id_offset + (uint32_t)env, step_counter, 16u + (uint32_t)b, 0x504F4Cu, seed_lo, seed_hi, r);
#pragma unroll
                for (int i = 0; i < 2; i++) {                                  // Box-Muller on (0,1] x [0,1)
                    const float u1 = ((float)(r[2*i] >> 8) + 1.0f) * (1.0f/16777216.0f);
                    const float u2 = (float)(r[2*i + 1] >> 8) * (1.0f/16777216.0f);
                    const float rad = __builtin_sqrtf(-2.0f*__builtin_logf(u1));
                    float s, c; tsincos<float>(6.283185307179586f*u2 - 3.141592653589793f, s, c);
                    eps[4*b + 2*i] = rad*c; eps[4*b + 2*i + 1] = rad*s;
                }
            }
        }
        float lp = 0.0f;
        if (live) {
#pragma unroll
            for (int a = 0; a < 6; a++) {
                const float ls = w.log_std[a];
                const float act = __builtin_fmaf(__builtin_expf(ls), eps[a], mean[a]);
                lp += -0.5f*eps[a]*eps[a] - ls - 0.9189385332046727f;
                io.act_env[(size_t)env*6 + a] = tclamp(act, -1.0f, 1.0f);
                if (io.act_raw) io.act_raw[(size_t)env*6 + a] = act;
                if (io.rollout_row) io.rollout_row[(size_t)env*(OD + 10) + OD + a] = act;
            }
            if (io.logp) io.logp[env] = lp;
            if (io.rollout_row) {
                io.rollout_row[(size_t)env*(OD + 10) + OD + 9] = lp;
#pragma unroll
                for (int k = 0; k < OD; k++) io.rollout_row[(size_t)env*(OD + 10) + k] = ob[k];
            }
        }
    } else if (wave == 1) {
        float v = w.v_b[0];
#pragma unroll 8
        for (int k = 0; k < 64; k++) v = __builtin_fmaf(w.v_w[k], h2[1][k][lane], v);
        if (live) {
            if (io.value) io.value[env] = v;
            if (io.rollout_row) io.rollout_row[(size_t)env*(OD + 10) + OD + 8] = v;
        }
    }
}

}  // namespace so100
