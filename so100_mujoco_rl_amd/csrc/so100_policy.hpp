// so100_policy.hpp -- fused rollout-side actor-critic forward (SB3 "MlpPolicy" for a Box action space).
//
// This is a CALLER-side helper of the hot path (the rollout collector), not part of the
// simulator: it replaces the ~28 small PyTorch kernels SB3's `policy.forward(obs)` + `clip` + rollout-buffer writes
// launch per vectorised step (main.py:56-64 -> stable_baselines3 OnPolicyAlgorithm.collect_rollouts) with ONE launch.
//   pi tower: obs -> Linear(od,64) tanh -> Linear(64,64) tanh -> Linear(64,6) = mean;  log_std state independent
//   vf tower: obs -> Linear(od,64) tanh -> Linear(64,64) tanh -> Linear(64,1) = value
//   action = mean + exp(log_std) * eps, eps ~ N(0,1) (Philox4x32-10 + Box-Muller, or injected), log_prob, clip to [-1,1]
// Mapping: workgroup = NW waves = 64 envs; lane = env; wave w owns hidden units [w*64/NW, (w+1)*64/NW) of BOTH towers, so every
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

__device__ __forceinline__ void policy_noise(uint32_t env_gid, uint32_t step_counter, uint32_t seed_lo, uint32_t seed_hi, float eps[8]);

template <int OD, int NW>
__global__ void __launch_bounds__(64*NW) so100_policy_forward_kernel(int n, PolicyWeights w, PolicyIO io, uint32_t seed_lo, uint32_t seed_hi,
                                                            uint32_t env_id_offset, uint32_t step_counter) {
    constexpr int UPW = 64/NW;                 // hidden units per wave
    constexpr int ODP = (OD + 3) & ~3;         // layer-1 rows padded to a multiple of 4 floats (float4 broadcast reads)
    // per-wave weight slice, staged with coalesced vector loads and read back as LDS broadcasts.  (Reading the
    // weights as scalar loads straight from global memory serialises on scalar-cache misses: 16-19 us per launch.)
    constexpr int W2 = 0, W1 = W2 + 2*UPW*64, B1 = W1 + 2*UPW*ODP, B2 = B1 + 2*UPW, WSZ = (B2 + 2*UPW + 3) & ~3;
    __shared__ __attribute__((aligned(16))) float wl[NW][WSZ];
    __shared__ float h1[2][64][64];            // [tower][unit][env]
    __shared__ float h2[2][64][64];
    __shared__ __attribute__((aligned(16))) float hd[6*64 + 64 + 16];      // mu_w | v_w | mu_b(6) log_std(6) v_b(1)
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int env = blockIdx.x*64 + lane;
    const bool live = env < n;
    const int u0 = wave*UPW;
    float* my = wl[wave];
    if (wave == NW - 1) {                      // head weights: staged in the same (only) global-load phase
        for (int i = lane; i < 6*64; i += 64) hd[i] = w.mu_w[i];
        hd[6*64 + lane] = w.v_w[lane];
        if (lane < 6) { hd[7*64 + lane] = w.mu_b[lane]; hd[7*64 + 6 + lane] = w.log_std[lane]; }
        if (lane == 0) hd[7*64 + 12] = w.v_b[0];
    }
    float eps_in[6];
#pragma unroll
    for (int a = 0; a < 6; a++) eps_in[a] = (wave == 0 && live && io.noise) ? io.noise[(size_t)env*6 + a] : 0.0f;
    for (int i = lane; i < UPW*64; i += 64) { my[W2 + i] = w.pi_w1[u0*64 + i]; my[W2 + UPW*64 + i] = w.vf_w1[u0*64 + i]; }
    for (int i = lane; i < UPW*ODP; i += 64) {
        const int j = i / ODP, k = i - j*ODP;
        my[W1 + i] = k < OD ? w.pi_w0[(u0 + j)*OD + k] : 0.0f; my[W1 + UPW*ODP + i] = k < OD ? w.vf_w0[(u0 + j)*OD + k] : 0.0f;
    }
    if (lane < UPW) { my[B1 + lane] = w.pi_b0[u0 + lane]; my[B1 + UPW + lane] = w.vf_b0[u0 + lane];
                      my[B2 + lane] = w.pi_b1[u0 + lane]; my[B2 + UPW + lane] = w.vf_b1[u0 + lane]; }
    float ob[ODP];
#pragma unroll
    for (int k = 0; k < ODP; k++) ob[k] = (live && k < OD) ? io.obs[(size_t)env*OD + k] : 0.0f;
    __syncthreads();
    // ---- layer 1
#pragma unroll
    for (int j = 0; j < UPW; j++) {
        float ap = my[B1 + j], av = my[B1 + UPW + j];
#pragma unroll
        for (int k = 0; k < ODP; k += 4) {
            const float4 a = *reinterpret_cast<const float4*>(&my[W1 + j*ODP + k]);
            const float4 b = *reinterpret_cast<const float4*>(&my[W1 + UPW*ODP + j*ODP + k]);
            ap = __builtin_fmaf(a.x, ob[k], ap); ap = __builtin_fmaf(a.y, ob[k+1], ap); ap = __builtin_fmaf(a.z, ob[k+2], ap); ap = __builtin_fmaf(a.w, ob[k+3], ap);
            av = __builtin_fmaf(b.x, ob[k], av); av = __builtin_fmaf(b.y, ob[k+1], av); av = __builtin_fmaf(b.z, ob[k+2], av); av = __builtin_fmaf(b.w, ob[k+3], av);
        }
        h1[0][u0 + j][lane] = fast_tanh(ap); h1[1][u0 + j][lane] = fast_tanh(av);
    }
    __syncthreads();
    // ---- layer 2
    float accp[UPW], accv[UPW];
#pragma unroll
    for (int j = 0; j < UPW; j++) { accp[j] = my[B2 + j]; accv[j] = my[B2 + UPW + j]; }
#pragma unroll 4
    for (int k = 0; k < 64; k += 4) {
        const float xp0 = h1[0][k][lane], xp1 = h1[0][k+1][lane], xp2 = h1[0][k+2][lane], xp3 = h1[0][k+3][lane];
        const float xv0 = h1[1][k][lane], xv1 = h1[1][k+1][lane], xv2 = h1[1][k+2][lane], xv3 = h1[1][k+3][lane];
#pragma unroll
        for (int j = 0; j < UPW; j++) {
            const float4 a = *reinterpret_cast<const float4*>(&my[W2 + j*64 + k]);
            const float4 b = *reinterpret_cast<const float4*>(&my[W2 + UPW*64 + j*64 + k]);
            accp[j] = __builtin_fmaf(a.x, xp0, accp[j]); accp[j] = __builtin_fmaf(a.y, xp1, accp[j]);
            accp[j] = __builtin_fmaf(a.z, xp2, accp[j]); accp[j] = __builtin_fmaf(a.w, xp3, accp[j]);
            accv[j] = __builtin_fmaf(b.x, xv0, accv[j]); accv[j] = __builtin_fmaf(b.y, xv1, accv[j]);
            accv[j] = __builtin_fmaf(b.z, xv2, accv[j]); accv[j] = __builtin_fmaf(b.w, xv3, accv[j]);
        }
    }
#pragma unroll
    for (int j = 0; j < UPW; j++) { h2[0][u0 + j][lane] = fast_tanh(accp[j]); h2[1][u0 + j][lane] = fast_tanh(accv[j]); }
    __syncthreads();
    // ---- heads: wave 0 -> mean / sample / log-prob, wave 1 -> value
    if (wave == 0) {
        float mean[6];
#pragma unroll
        for (int a = 0; a < 6; a++) mean[a] = hd[7*64 + a];
#pragma unroll 4
        for (int k = 0; k < 64; k += 4) {
            const float x0 = h2[0][k][lane], x1 = h2[0][k+1][lane], x2 = h2[0][k+2][lane], x3 = h2[0][k+3][lane];
#pragma unroll
            for (int a = 0; a < 6; a++) {
                const float4 m4 = *reinterpret_cast<const float4*>(&hd[a*64 + k]);
                mean[a] = __builtin_fmaf(m4.x, x0, mean[a]); mean[a] = __builtin_fmaf(m4.y, x1, mean[a]);
                mean[a] = __builtin_fmaf(m4.z, x2, mean[a]); mean[a] = __builtin_fmaf(m4.w, x3, mean[a]);
            }
        }
        float eps[8];
        if (io.noise) {
#pragma unroll
            for (int a = 0; a < 6; a++) eps[a] = eps_in[a];
        } else {
            policy_noise(env_id_offset + (uint32_t)env, step_counter, seed_lo, seed_hi, eps);
        }
        float lp = 0.0f;
        if (live) {
#pragma unroll
            for (int a = 0; a < 6; a++) {
                const float ls = hd[7*64 + 6 + a];
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
        float v = hd[7*64 + 12];
#pragma unroll 8
        for (int k = 0; k < 64; k++) v = __builtin_fmaf(hd[6*64 + k], h2[1][k][lane], v);
        if (live) {
            if (io.value) io.value[env] = v;
            if (io.rollout_row) io.rollout_row[(size_t)env*(OD + 10) + OD + 8] = v;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Sampling head shared by the stand-alone policy kernel's wave 0 and the persistent rollout kernel
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
