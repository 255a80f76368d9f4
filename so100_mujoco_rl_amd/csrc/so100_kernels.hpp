#pragma once
// so100_kernels.hpp -- the HIP kernels (gfx950) of libso100sim.so and their per-env-kind launchers.
//
// Execution model, round 1: ONE LANE PER ENV.  A 64-lane wavefront steps 64 envs; each lane keeps its
// env's whole state (82 words) and every intermediate of the 16 fused substeps in VGPRs, so HBM sees the
// state exactly once in and once out per env step (DESIGN.md "Kernels").  State is struct-of-arrays
// [field][N]: lane i of a wave reads word i of a row => one fully coalesced 256-B request per field.
// Workgroups are one wave (64 threads): at N = 4096 that is 64 workgroups, which the dispatcher deals
// round-robin over the 8 XCDs; there is no inter-workgroup communication of any kind.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
#include <new>
#include "../../include/so100_sim.h"
#include "so100_task.hpp"
#include "so100_policy.hpp"

namespace so100 {
// ---- SoA load / store of one env ----------------------------------------------------------------------
template <int KIND, int FL = -1> __device__ __forceinline__ void load_env_state(const float* __restrict__ S, int n, int env, EnvState& e) {
#define X(name, member, kind, group) \
    if constexpr (uses_group<KIND, FL>(group)) { const float w_ = S[(size_t)SF_##name*n + env]; \
        if constexpr (#kind[0] == 'i') e.member = __float_as_int(w_); else e.member = w_; }
    SO100_STATE_FIELDS(X)
#undef X
}
template <typename M> __device__ __forceinline__ float as_word(M v) {
    if constexpr (sizeof(M) == 4 && !__is_floating_point(M)) return __int_as_float((int)v); else return (float)v;
}
template <int KIND, int FL = -1> __device__ __forceinline__ void store_env_state(float* __restrict__ S, int n, int env, const EnvState& e) {
#define X(name, member, kind, group) \
    if constexpr (uses_group<KIND, FL>(group)) S[(size_t)SF_##name*n + env] = as_word(e.member);
    SO100_STATE_FIELDS(X)
#undef X
}

}  // namespace so100

#include "so100_rollout.hpp"

#ifndef SO100_FREE_WAVES
#define SO100_FREE_WAVES 2      // waves per SIMD asked of the contact-free throughput kernel (3 -> 168 VGPRs + 364 B of scratch: measured slower)
#endif

namespace so100 {

constexpr int WG = 64;      // one wavefront per workgroup

struct StepPtrs {
    float* state; const float* start_tab;
    const float* act; float* obs; float* rew; uint8_t* done; uint8_t* trunc; float* tobs; float* ep_ret; int32_t* ep_len;
    const float* inject;
    float* rollout_row;
};

// K1: one fused env step (reward -> ctrl -> 16 x {FK, CRB, RNE, servo, rows, block-PGS / Newton, Euler} -> obs
//     -> TimeLimit -> auto-reset), SURVEY.md section 8a rows a1-a10.
// FL >= 0: the physics flags are a compile-time constant (dead constraint families are not even compiled in:
// fewer live registers, smaller loop body); FL < 0: flags read from the handle at run time.
// The constraint-free variant fits 256 registers: asking for 2 waves per SIMD keeps the latency hiding that large
// batches need (1 M envs: 2 waves/SIMD 2.1 G env-steps/s, 1 wave/SIMD 1.5 G); the constrained variants and the
// look-at envs (more task state) need > 256.  DENSE = the same kernel held to 256 registers (2 waves per SIMD) with the rest
// in scratch: for the friction / limit / cube-floor variants (flags 7, 11) that is faster once the batch fills the chip twice
// (measured, 262 144 envs: 0.57 -> 0.74 G env-steps/s; at 65 536 envs 0.54 -> 0.42, so KindOps::step picks by batch size;
// the pad-contact variants lose with it: 1.4 KB of scratch per lane already).
constexpr int DENSE_MIN_ENVS = 131072;
template <int KIND, int FL, bool DENSE = false>
__global__ void __launch_bounds__(WG, ((FL == (int)SO100_F_CUBE_PINNED && reach_kind<KIND>()) ? SO100_FREE_WAVES : DENSE ? 2 : 1)) so100_step_fused(SimParams p, StepPtrs io) {
    const int env = blockIdx.x*WG + threadIdx.x;
    if (env >= p.n) return;
    if (FL >= 0) p.flags = (unsigned)FL;
    EnvState e;
    load_env_state<KIND, FL>(io.state, p.n, env, e);
    float a[6];
#pragma unroll
    for (int i = 0; i < 6; i++) a[i] = io.act[(size_t)env*6 + i];
    constexpr int OD = obs_dim<KIND>();
    float obs[OD], tobs[OD];
    const float* inj = io.inject ? io.inject + (size_t)env*SO100_NINJECT : nullptr;
    const StepResult r = env_step_vec<KIND>(e, a, p, p.env_id_offset + (uint32_t)env, inj, io.start_tab, obs, tobs);
    store_env_state<KIND, FL>(io.state, p.n, env, e);
#pragma unroll
    for (int i = 0; i < OD; i++) io.obs[(size_t)env*OD + i] = obs[i];
    io.rew[env] = r.reward;
    if (io.rollout_row) {
        io.rollout_row[(size_t)env*(OD + 10) + OD + 6] = r.reward;
        io.rollout_row[(size_t)env*(OD + 10) + OD + 7] = r.done ? (r.trunc_only ? 2.0f : 1.0f) : 0.0f;
    }
    io.done[env] = r.done ? 1 : 0;
    io.trunc[env] = r.trunc_only ? 1 : 0;
    if (r.done) {
        if (io.tobs) {
#pragma unroll
            for (int i = 0; i < OD; i++) io.tobs[(size_t)env*OD + i] = tobs[i];
        }
        if (io.ep_ret) io.ep_ret[env] = r.ep_return;
        if (io.ep_len) io.ep_len[env] = r.ep_length;
    }
}

// K1-mw: the same fused env step with a workgroup of 4 waves per 64 envs -- the physics of env = lane is split over the waves
// exactly as in the persistent rollout kernel (physics_phase_mw: RNEA on wave 1, cube on wave 2, CRBA / solve on wave 0).
// One env step then costs ~2/3 of the single-wave kernel's latency; it occupies 4 SIMDs per 64 envs, so it is the step
// kernel for batches that do not fill the chip (launch_step picks it for N <= 16384) and so100_step_fused stays the
// throughput kernel for large batches.
template <int KIND, int FL>
__global__ void __launch_bounds__(256) so100_step_mw(SimParams p, StepPtrs io) {
    constexpr bool PADS = FL < 0 || (FL & (int)F_ANY_CONTACT) != 0;
    constexpr bool LINKS = FL < 0 || (FL & (int)F_ANY_LINKS) != 0;
    __shared__ float xq[PADS ? 24 : 18][64];
    __shared__ float xc[24][64];
    __shared__ float xb[6][64];
    __shared__ float cbuf[PADS ? MAXC*CF*64 : 1];                 // pad contact records [record][field][lane]
    __shared__ float xa[PADS ? 15 : 1][64];
    __shared__ float xk[PADS ? 12 : 1][64];
    __shared__ float xm[PADS ? 21 : 1][64];
    __shared__ unsigned char pbuf[PADS ? MAXC*64 : 1];
    __shared__ float xw[PADS ? 36 : 1][64];                       // joint axes / screw terms, contact wave's detection -> its solve
    if (FL >= 0) p.flags = (unsigned)FL;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int env = blockIdx.x*p.epw + lane;
    const bool live = lane < p.epw && env < p.n;
    constexpr int OD = obs_dim<KIND>();
    EnvState e; StepCtx ctx{}; float u[8] = {}; float cstale[3] = {};
    const float* inj = (io.inject && live) ? io.inject + (size_t)env*SO100_NINJECT : nullptr;
    if (wave == 0) {
        if (live) load_env_state<KIND, FL>(io.state, p.n, env, e); else idle_lane_state(e);
        float a[6];
#pragma unroll
        for (int i = 0; i < 6; i++) a[i] = live ? io.act[(size_t)env*6 + i] : 0.0f;
        draw8(p, p.env_id_offset + (uint32_t)env, (uint32_t)e.rngc, 0, inj, u);
        e.rngc++;
        env_step_pre<KIND>(e, a, u, p, ctx);
    }
    Arm<float> A; Prof prof_;
    const PhaseLds lds{ xq, xc, xb, cbuf, xa, xk, xm, pbuf, PADS ? xw : nullptr };
    ContactMemo memo;                                             // (reset by physics_phase_mw: the memory lives for the env step's substeps)
    physics_phase_mw<PADS, LINKS>(p, wave, lane, e, ctx.ctrl, cstale, A, lds, memo, prof_, [](int) {}, [&]() {
        ctx = StepCtx{};
#pragma unroll
        for (int k = 0; k < 8; k++) u[k] = 0.0f;
    });
    if (wave != 0 || !live) return;
    e.nsub += p.frame_skip;
    TaskPoses<float> P;
    task_poses<float>(A.s, A.c, !reach_kind<KIND>(), P);
    float obs[OD], tobs[OD]; bool term;
    const float reward = env_step_post<KIND>(e, ctx, u, P, cstale, obs, term);
    const StepResult r = env_step_finish<KIND>(e, reward, term, p, p.env_id_offset + (uint32_t)env, inj, io.start_tab, obs, tobs);
    if constexpr (PADS) e.cload &= 0xFFFF;                         // (the per-launch count of physics_phase_mw is the persistent kernel's business)
    store_env_state<KIND, FL>(io.state, p.n, env, e);
#pragma unroll
    for (int i = 0; i < OD; i++) io.obs[(size_t)env*OD + i] = obs[i];
    io.rew[env] = r.reward;
    if (io.rollout_row) {
        io.rollout_row[(size_t)env*(OD + 10) + OD + 6] = r.reward;
        io.rollout_row[(size_t)env*(OD + 10) + OD + 7] = r.done ? (r.trunc_only ? 2.0f : 1.0f) : 0.0f;
    }
    io.done[env] = r.done ? 1 : 0;
    io.trunc[env] = r.trunc_only ? 1 : 0;
    if (r.done) {
        if (io.tobs) {
#pragma unroll
            for (int i = 0; i < OD; i++) io.tobs[(size_t)env*OD + i] = tobs[i];
        }
        if (io.ep_ret) io.ep_ret[env] = r.ep_return;
        if (io.ep_len) io.ep_len[env] = r.ep_length;
    }
}

// K2: masked reset (MujocoEnv.reset -> mj_resetData -> reset_model), SURVEY.md section 8a row a6
template <int KIND>
__global__ void __launch_bounds__(WG) so100_reset_masked(SimParams p, float* state, const float* start_tab,
                                                         const uint8_t* mask, const float* inject, float* obs_out) {
    const int env = blockIdx.x*WG + threadIdx.x;
    if (env >= p.n) return;
    if (mask && !mask[env]) return;
    EnvState e;
    load_env_state<KIND>(state, p.n, env, e);
    float u[8];
    const float* inj = inject ? inject + (size_t)env*SO100_NINJECT : nullptr;
    draw8(p, p.env_id_offset + (uint32_t)env, (uint32_t)e.rngc, 1, inj, u);
    e.rngc++;
    constexpr int OD = obs_dim<KIND>();
    float obs[OD];
    env_reset<KIND>(e, u, start_tab, obs);
    store_env_state<KIND>(state, p.n, env, e);
    if (obs_out) {
#pragma unroll
        for (int i = 0; i < OD; i++) obs_out[(size_t)env*OD + i] = obs[i];
    }
}

// K0: EnvNN.__init__ for every env (fresh handle)
template <int KIND>
__global__ void __launch_bounds__(WG) so100_init_state(int n, float* state) {
    const int env = blockIdx.x*WG + threadIdx.x;
    if (env >= n) return;
    for (int f = 0; f < SF_COUNT; f++) state[(size_t)f*n + env] = 0.0f;
    EnvState e;
    env_init<KIND>(e);
    store_env_state<KIND>(state, n, env, e);
}


// Which step kernel for which batch (so100_create fills SimParams::mw_max from this; SO100_MW_MAX_ENVS in the environment overrides it
// for measurements).  The 4-wave kernel so100_step_mw splits one env step over 4 waves: lowest latency, and it wins while the batch
// leaves SIMDs idle (256 CUs x 64 envs = 16384).  Beyond that the one-wave kernel so100_step_fused has the higher throughput -- for the
// contact variants too, although they carry 1.0-1.5 KB of scratch per lane there and so100_step_mw<K, 23 | 55> none: measured at 32 768 ...
// 262 144 envs (profiles/r03_large_batch_dispatch.txt) the 4-wave kernel ties at 32 768 and loses 2x from 65 536 on, for every flag set.
// One threshold for all; the two names are kept so that a future contact kernel can move its own.
constexpr int MW_MAX_ENVS = 16384;
constexpr int MW_MAX_ENVS_PADS = 16384;
inline int mw_max_envs_for(unsigned flags) { return (flags & (SO100_F_PADS_FLOOR | SO100_F_PADS_CUBE | SO100_F_LINKS_FLOOR | SO100_F_LINKS_CUBE)) ? MW_MAX_ENVS_PADS : MW_MAX_ENVS; }
inline dim3 grid_for(int n) { return dim3((unsigned)((n + WG - 1)/WG)); }

struct RolloutPtrs { float* obs; float* rew; uint8_t* done; uint8_t* trunc; float* tobs; float* ep_ret; int32_t* ep_len; };

// Launchers of one env kind.  Every (kernel, flags) instantiation of a kind is compiled in its own translation unit
// (so100_kind.hip with -DSO100_KIND=k: the six kinds build in parallel); so100_sim.hip only declares them.
template <int KIND> struct KindOps {
    static hipError_t step(const SimParams& prm, const StepPtrs& io, hipStream_t st);
    static hipError_t reset(const SimParams& prm, float* state, const float* start_tab, const uint8_t* mask, const float* inject, float* obs, hipStream_t st);
    static hipError_t init(int n, float* state);
    static hipError_t rollout(const SimParams& prm, float* state, const float* start_tab, const RolloutPtrs& io, const PolicyWeights& pw, const RolloutArgs& ra, hipStream_t st);
#ifdef SO100_ROLLOUT_PROF
    static int prof_read(long long* out48);       // the cycle counters live in the kind's own code object
    static int prof_read_wg(long long* wg4096, int* env32768);
    static int prof_read_hist(unsigned long long* hist48, int reset);
#endif
};
#ifdef SO100_ROLLOUT_PROF
template <int KIND> int KindOps<KIND>::prof_read(long long* out48) {
    return hipMemcpyFromSymbol(out48, HIP_SYMBOL(so100_prof), sizeof(long long)*48) == hipSuccess ? 0 : -1;
}
template <int KIND> int KindOps<KIND>::prof_read_hist(unsigned long long* h, int reset) {
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(so100_prof_hist), sizeof(unsigned long long)*48) != hipSuccess) return -1;
    if (reset) { unsigned long long z[48] = {}; if (hipMemcpyToSymbol(HIP_SYMBOL(so100_prof_hist), z, sizeof z) != hipSuccess) return -1; }
    return 0;
}
template <int KIND> int KindOps<KIND>::prof_read_wg(long long* wg, int* env) {
    if (hipMemcpyFromSymbol(wg, HIP_SYMBOL(so100_prof_wg), sizeof(long long)*4096) != hipSuccess) return -1;
    return hipMemcpyFromSymbol(env, HIP_SYMBOL(so100_prof_env), sizeof(int)*32768) == hipSuccess ? 0 : -1;
}
#endif

template <int KIND> hipError_t KindOps<KIND>::step(const SimParams& prm, const StepPtrs& io, hipStream_t st) {
    const bool mw = prm.n <= prm.mw_max;
    const dim3 g = mw ? dim3((unsigned)((prm.n + prm.epw - 1)/prm.epw)) : grid_for(prm.n), b(mw ? 256 : WG);
#define SO100_STEP(FLV) do { if (mw) hipLaunchKernelGGL((so100_step_mw<KIND, FLV>), g, b, 0, st, prm, io); \
                             else    hipLaunchKernelGGL((so100_step_fused<KIND, FLV>), g, b, 0, st, prm, io); } while (0)
#define SO100_STEP_ROWS(FLV) do { if (!mw && prm.n >= DENSE_MIN_ENVS) hipLaunchKernelGGL((so100_step_fused<KIND, FLV, true>), g, b, 0, st, prm, io); \
                                  else SO100_STEP(FLV); } while (0)
    switch (prm.flags) {
    case SO100_F_CUBE_PINNED: SO100_STEP(SO100_F_CUBE_PINNED); break;
    case SO100_F_FRICTIONLOSS | SO100_F_LIMITS | SO100_F_CUBE_PINNED: SO100_STEP_ROWS(SO100_F_FRICTIONLOSS | SO100_F_LIMITS | SO100_F_CUBE_PINNED); break;
    case SO100_F_NOPADS: SO100_STEP_ROWS(SO100_F_NOPADS); break;
    case SO100_F_REFERENCE: SO100_STEP(SO100_F_REFERENCE); break;
    case SO100_F_CONTACT5: if constexpr (reach_kind<KIND>()) { SO100_STEP(SO100_F_CONTACT5); } else { SO100_STEP(-1); } break;
    default: SO100_STEP(-1); break;
    }
#undef SO100_STEP_ROWS
#undef SO100_STEP
    return hipGetLastError();
}
template <int KIND> hipError_t KindOps<KIND>::reset(const SimParams& prm, float* state, const float* start_tab, const uint8_t* mask, const float* inject, float* obs, hipStream_t st) {
    hipLaunchKernelGGL(so100_reset_masked<KIND>, grid_for(prm.n), dim3(WG), 0, st, prm, state, start_tab, mask, inject, obs);
    return hipGetLastError();
}
template <int KIND> hipError_t KindOps<KIND>::init(int n, float* state) {
    hipLaunchKernelGGL(so100_init_state<KIND>, grid_for(n), dim3(WG), 0, nullptr, n, state);
    const hipError_t e = hipGetLastError();
    return e != hipSuccess ? e : hipDeviceSynchronize();
}
template <int KIND> hipError_t KindOps<KIND>::rollout(const SimParams& prm, float* state, const float* start_tab, const RolloutPtrs& io, const PolicyWeights& pw, const RolloutArgs& ra, hipStream_t st) {
    const dim3 grid((unsigned)((prm.n + prm.epw - 1)/prm.epw));
#define SO100_RL(FLV) hipLaunchKernelGGL((so100_rollout_fused<KIND, FLV, 4>), grid, dim3(256), 0, st, prm, state, start_tab, \
        io.obs, io.rew, io.done, io.trunc, io.tobs, io.ep_ret, io.ep_len, pw, ra)
    if (prm.flags == SO100_F_CUBE_PINNED) {
        if (prm.epw <= 32) hipLaunchKernelGGL((so100_rollout_fused<KIND, SO100_F_CUBE_PINNED, 4, 32>), grid, dim3(256), 0, st, prm, state, start_tab,
                                              io.obs, io.rew, io.done, io.trunc, io.tobs, io.ep_ret, io.ep_len, pw, ra);
        else SO100_RL(SO100_F_CUBE_PINNED);
    }
    else if (prm.flags == SO100_F_NOPADS) SO100_RL(SO100_F_NOPADS);
    else if (prm.flags == SO100_F_REFERENCE) SO100_RL(SO100_F_REFERENCE);
    else if (prm.flags == SO100_F_CONTACT5 && reach_kind<KIND>()) { if constexpr (reach_kind<KIND>()) SO100_RL(SO100_F_CONTACT5); }
    else SO100_RL(-1);
#undef SO100_RL
    return hipGetLastError();
}

}  // namespace so100
