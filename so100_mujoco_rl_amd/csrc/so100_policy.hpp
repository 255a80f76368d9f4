// so100_policy.hpp -- fused rollout-side actor-critic forward (SB3 "MlpPolicy" for a Box action space).
//
// This is a CALLER-side helper of the hot path (the rollout collector), not part of the
// simulator: it replaces the ~28 small PyTorch kernels SB3's `policy.forward(obs)` + `clip` + rollout-buffer writes
// launch per vectorised step (main.py:56-64 -> stable_baselines3 OnPolicyAlgorithm.collect_rollouts) with ONE launch.
//   pi tower: obs -> Linear(od,64) tanh -> Linear(64,64) tanh -> Linear(64,6) = mean;  log_std state independent
//   vf tower: obs -> Linear(od,64) tanh -> Linear(64,64) tanh -> Linear(64,1) = value
//   action = mean + exp(log_std) * eps, eps ~ N(0,1) (Philox4x32-10 + Box-Muller, or injected), log_prob, clip to [-1,1]
// The kernels live in so100_rollout.hpp (so100_policy_forward_mfma, and the policy phase of so100_rollout_fused): the two
// 64-wide hidden layers run on the matrix cores (exact-fp32 v_mfma_f32_32x32x2_f32), heads / sampling on the VALU.
// This header holds what they share: the weight / IO structs, the tanh and the Philox + Box-Muller noise.
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
    // tanh(x) = 1 - 2/(exp(2x)+1); exp via v_exp_f32 (2^x), rcp with one Newton step: abs error < 2e-7.
    // exp(2x) overflows to +inf for x > ~44 and rcp(inf) = 0 would make the Newton step of trcp 0*inf = NaN, so the
    // exponential is capped (1e30: 1 - 2/(1e30+1) rounds to exactly 1.0f): a saturated unit returns +-1, never NaN.
    const float e = __builtin_fminf(__builtin_amdgcn_exp2f(x * 2.885390081777927f), 1.0e30f);          // exp(2x)
    return 1.0f - 2.0f*trcp(e + 1.0f);
}

// ---------------------------------------------------------------------------------------------------------------
// Policy noise shared by the stand-alone policy kernel and the persistent rollout kernel
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void policy_noise(uint32_t env_gid, uint32_t step_counter, uint32_t seed_lo, uint32_t seed_hi, float eps[8]) {
#pragma unroll
    for (int b = 0; b < 2; b++) {
        uint32_t r[4];
        philox4x32(env_gid, step_counter, 16u + (uint32_t)b, 0x504F4Cu, seed_lo, seed_hi, r);
#pragma unroll
        for (int i = 0; i < 2; i++) {                                  // Box-Muller on (0,1] x [0,1)
            const float u1 = ((float)(r[2*i] >> 8) + 1.0f) * (1.0f/16777216.0f);
            const float u2 = (float)(r[2*i + 1] >> 8) * (1.0f/16777216.0f);
            const float rad = __builtin_sqrtf(-2.0f*__builtin_logf(u1));
            float sn, cs; tsincos<float>(6.283185307179586f*u2 - 3.141592653589793f, sn, cs);
            eps[4*b + 2*i] = rad*cs; eps[4*b + 2*i + 1] = rad*sn;
        }
    }
}

}  // namespace so100
