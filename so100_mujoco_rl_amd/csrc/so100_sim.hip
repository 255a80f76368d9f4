// so100_sim.hip -- HIP kernels (gfx950) and the C ABI of libso100sim.so (include/so100_sim.h).
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

namespace {

using namespace so100;

thread_local char g_err[512] = "";
int fail(int code, const char* fmt, const char* a = "", long b = 0) {
    snprintf(g_err, sizeof g_err, fmt, a, b);
    return code;
}
#define HIP_TRY(expr, code) do { hipError_t e_ = (expr); if (e_ != hipSuccess) \
    return fail(code, "%s (HIP error %ld)", hipGetErrorString(e_), (long)e_); } while (0)

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
// batches need (1 M envs: 2 waves/SIMD 2.3 G env-steps/s, 1 wave/SIMD 1.5 G); the constrained variants and the
// look-at envs (more task state) need > 256 and would spill.
template <int KIND, int FL>
__global__ void __launch_bounds__(WG, ((FL == (int)SO100_F_CUBE_PINNED && reach_kind<KIND>()) ? 2 : 1)) so100_step_fused(SimParams p, StepPtrs io) {
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
    __shared__ float xq[18][64];
    __shared__ float xc[24][64];
    __shared__ float xb[6][64];
    constexpr bool PADS = FL < 0 || (FL & (int)(F_PADS_FLOOR | F_PADS_CUBE)) != 0;
    __shared__ float cbuf[PADS ? MAXC*CF*64 : 1];                 // pad contact records [record][field][lane]
    __shared__ float xa[PADS ? 8 : 1][64];
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
    physics_phase_mw<PADS>(p, wave, lane, e, ctx.ctrl, cstale, A, xq, xc, xb, cbuf, xa, prof_, [](int) {});
    if (wave != 0 || !live) return;
    e.nsub += p.frame_skip;
    TaskPoses<float> P;
    task_poses<float>(A.s, A.c, !reach_kind<KIND>(), P);
    float obs[OD], tobs[OD]; bool term;
    const float reward = env_step_post<KIND>(e, ctx, u, P, cstale, obs, term);
    const StepResult r = env_step_finish<KIND>(e, reward, term, p, p.env_id_offset + (uint32_t)env, inj, io.start_tab, obs, tobs);
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

const char* const kFieldNames[] = {
#define X(name, member, kind, group) #name,
    SO100_STATE_FIELDS(X)
#undef X
};

#include "so100_start_positions.inc"

}  // namespace

struct so100_sim {
    so100_config cfg;
    SimParams prm;
    float* state = nullptr;        // [SF_COUNT][N]
    float* start_tab = nullptr;    // [36][6]
};

namespace {
struct DeviceGuard {
    int prev = -1; bool ok = true;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) { ok = false; return; }
        if (prev != dev && hipSetDevice(dev) != hipSuccess) ok = false;
        target = dev;
    }
    ~DeviceGuard() { if (ok && prev != target) (void)hipSetDevice(prev); }
    int target = -1;
};
inline dim3 grid_for(int n) { return dim3((unsigned)((n + WG - 1)/WG)); }

constexpr int MW_MAX_ENVS = 16384;   // up to here the 4-wave step kernel wins (256 CUs x 64 envs); beyond, the chip is full anyway

template <int KIND> int launch_step(so100_sim* s, const StepPtrs& io, hipStream_t st) {
    const bool mw = s->prm.n <= MW_MAX_ENVS;
    const dim3 g = mw ? dim3((unsigned)((s->prm.n + s->prm.epw - 1)/s->prm.epw)) : grid_for(s->prm.n), b(mw ? 256 : WG);
#define SO100_STEP(FLV) do { if (mw) hipLaunchKernelGGL((so100_step_mw<KIND, FLV>), g, b, 0, st, s->prm, io); \
                             else    hipLaunchKernelGGL((so100_step_fused<KIND, FLV>), g, b, 0, st, s->prm, io); } while (0)
    switch (s->prm.flags) {
    case SO100_F_CUBE_PINNED: SO100_STEP(SO100_F_CUBE_PINNED); break;
    case SO100_F_FRICTIONLOSS | SO100_F_LIMITS | SO100_F_CUBE_PINNED: SO100_STEP(SO100_F_FRICTIONLOSS | SO100_F_LIMITS | SO100_F_CUBE_PINNED); break;
    case SO100_F_NOPADS: SO100_STEP(SO100_F_NOPADS); break;
    case SO100_F_REFERENCE: SO100_STEP(SO100_F_REFERENCE); break;
    case SO100_F_CONTACT5: if constexpr (reach_kind<KIND>()) { SO100_STEP(SO100_F_CONTACT5); } else { SO100_STEP(-1); } break;
    default: SO100_STEP(-1); break;
    }
#undef SO100_STEP
    HIP_TRY(hipGetLastError(), SO100_E_LAUNCH);
    return 0;
}
template <int KIND> int launch_reset(so100_sim* s, const uint8_t* mask, const float* inject, float* obs, hipStream_t st) {
    hipLaunchKernelGGL(so100_reset_masked<KIND>, grid_for(s->prm.n), dim3(WG), 0, st, s->prm, s->state, s->start_tab, mask, inject, obs);
    HIP_TRY(hipGetLastError(), SO100_E_LAUNCH);
    return 0;
}
template <int KIND> int launch_init(so100_sim* s) {
    hipLaunchKernelGGL(so100_init_state<KIND>, grid_for(s->prm.n), dim3(WG), 0, nullptr, s->prm.n, s->state);
    HIP_TRY(hipGetLastError(), SO100_E_LAUNCH);
    HIP_TRY(hipDeviceSynchronize(), SO100_E_LAUNCH);
    return 0;
}
#define DISPATCH_KIND(kind, call) \
    ((kind) == 1 ? call<1> : (kind) == 2 ? call<2> : (kind) == 3 ? call<3> : (kind) == 4 ? call<4> : (kind) == 5 ? call<5> : call<6>)
}  // namespace

extern "C" {

int so100_abi_version(void) { return SO100_ABI_VERSION; }
int so100_obs_dim(int32_t kind) { return (kind == 1 || kind == 2 || kind == 6) ? 15 : (kind >= 3 && kind <= 5) ? 8 : -1; }
int so100_num_state_fields(void) { return SF_COUNT; }
int so100_state_field_index(const char* name) {
    if (!name) return -1;
    for (int i = 0; i < SF_COUNT; i++) if (strcmp(kFieldNames[i], name) == 0) return i;
    return -1;
}
const char* so100_state_field_name(int32_t field) { return (field >= 0 && field < SF_COUNT) ? kFieldNames[field] : nullptr; }
const char* so100_last_error(void) { return g_err; }
#ifdef SO100_ROLLOUT_PROF
int so100_prof_read(long long* out32) { return hipMemcpyFromSymbol(out32, HIP_SYMBOL(so100::so100_prof), sizeof(long long)*32) == hipSuccess ? 0 : -1; }
#endif

int so100_create(const so100_config* cfg, so100_sim** out) {
    if (!cfg || !out) return fail(SO100_E_INVALID, "so100_create: null argument%s");
    *out = nullptr;
    if (cfg->env_kind < 1 || cfg->env_kind > 6) return fail(SO100_E_INVALID, "so100_create: env_kind must be 1..6%s");
    if (cfg->num_envs < 1) return fail(SO100_E_INVALID, "so100_create: num_envs must be >= 1%s");
    if (cfg->solver_iters < 1 || cfg->solver_iters > 64) return fail(SO100_E_INVALID, "so100_create: solver_iters must be in 1..64%s");
    if (cfg->contact_iters < 1 || cfg->contact_iters > 64) return fail(SO100_E_INVALID, "so100_create: contact_iters must be in 1..64%s");
    if (cfg->frame_skip < 1 || cfg->frame_skip > 1024) return fail(SO100_E_INVALID, "so100_create: frame_skip must be in 1..1024%s");
    if (cfg->max_episode_steps < 0) return fail(SO100_E_INVALID, "so100_create: max_episode_steps must be >= 0%s");
    if (cfg->flags & ~(SO100_F_FRICTIONLOSS | SO100_F_LIMITS | SO100_F_FLOOR | SO100_F_CUBE_PINNED | SO100_F_PADS_FLOOR | SO100_F_PADS_CUBE))
        return fail(SO100_E_INVALID, "so100_create: unknown flag bits%s");
    if ((cfg->flags & SO100_F_PADS_CUBE) && (cfg->flags & SO100_F_CUBE_PINNED))
        return fail(SO100_E_INVALID, "so100_create: SO100_F_PADS_CUBE needs a dynamic cube (not SO100_F_CUBE_PINNED)%s");
    if ((cfg->flags & SO100_F_FLOOR) && (cfg->flags & SO100_F_CUBE_PINNED))
        return fail(SO100_E_INVALID, "so100_create: SO100_F_FLOOR and SO100_F_CUBE_PINNED are mutually exclusive%s");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
        return fail(SO100_E_NODEVICE, "so100_create: no HIP device available (this library has no CPU fallback)%s");
    if (cfg->device < 0 || cfg->device >= ndev) return fail(SO100_E_INVALID, "so100_create: device ordinal out of range%s");
    DeviceGuard g(cfg->device);
    if (!g.ok) return fail(SO100_E_NODEVICE, "so100_create: cannot select the device%s");
    so100_sim* s = new (std::nothrow) so100_sim();
    if (!s) return fail(SO100_E_NOMEM, "so100_create: out of host memory%s");
    s->cfg = *cfg;
    s->prm.n = cfg->num_envs; s->prm.flags = cfg->flags; s->prm.solver_iters = cfg->solver_iters;
    s->prm.contact_iters = cfg->contact_iters; s->prm.frame_skip = cfg->frame_skip;
    s->prm.max_episode_steps = cfg->max_episode_steps;
    s->prm.seed_lo = (uint32_t)cfg->seed; s->prm.seed_hi = (uint32_t)(cfg->seed >> 32);
    s->prm.env_id_offset = cfg->env_id_offset;
    {   // envs per workgroup of the multi-wave kernels.  Their step time is set by the slowest lane of a wave (a cube in a contact
        // transient, a pad hitting the floor: data-dependent Newton iterations), so a batch that leaves CUs idle is spread thinner:
        // 16 or 32 envs per 4-wave workgroup while that still fits one workgroup per CU.  Variants without a data-dependent solve
        // (cube pinned) gain nothing and keep 64.
        hipDeviceProp_t prop;
        int cus = 256;
        if (hipGetDeviceProperties(&prop, cfg->device) == hipSuccess && prop.multiProcessorCount > 0) cus = prop.multiProcessorCount;
        int epw = 64;
        if (cfg->flags & (SO100_F_FLOOR | SO100_F_PADS_FLOOR | SO100_F_PADS_CUBE))
            while (epw > 16 && (cfg->num_envs + epw/2 - 1)/(epw/2) <= cus) epw /= 2;
        s->prm.epw = epw;
    }
    const size_t bytes = (size_t)SF_COUNT*(size_t)cfg->num_envs*sizeof(float);
    if (hipMalloc(&s->state, bytes) != hipSuccess) { delete s; return fail(SO100_E_NOMEM, "so100_create: hipMalloc of %s%ld bytes failed", "", (long)bytes); }
    float tab[36*6];
    for (int i = 0; i < 36; i++) for (int j = 0; j < 6; j++) tab[6*i + j] = (float)SO100_VALID_START_POSITIONS[i][j];
    if (hipMalloc(&s->start_tab, sizeof tab) != hipSuccess || hipMemcpy(s->start_tab, tab, sizeof tab, hipMemcpyHostToDevice) != hipSuccess) {
        (void)hipFree(s->state); if (s->start_tab) (void)hipFree(s->start_tab); delete s;
        return fail(SO100_E_NOMEM, "so100_create: start table upload failed%s");
    }
    const int rc = DISPATCH_KIND(cfg->env_kind, launch_init)(s);
    if (rc != 0) { (void)hipFree(s->state); (void)hipFree(s->start_tab); delete s; return rc; }
    *out = s;
    return 0;
}

void so100_destroy(so100_sim* s) {
    if (!s) return;
    DeviceGuard g(s->cfg.device);
    if (s->state) (void)hipFree(s->state);
    if (s->start_tab) (void)hipFree(s->start_tab);
    delete s;
}

int so100_reset(so100_sim* s, const uint8_t* mask_dev, const float* inject_dev, float* obs_dev, void* stream) {
    if (!s) return fail(SO100_E_INVALID, "so100_reset: null handle%s");
    DeviceGuard g(s->cfg.device);
    if (!g.ok) return fail(SO100_E_NODEVICE, "so100_reset: cannot select the device%s");
    return DISPATCH_KIND(s->cfg.env_kind, launch_reset)(s, mask_dev, inject_dev, obs_dev, (hipStream_t)stream);
}

int so100_step(so100_sim* s, const so100_step_io* io, void* stream) {
    if (!s || !io) return fail(SO100_E_INVALID, "so100_step: null argument%s");
    if (!io->act_dev || !io->obs_dev || !io->rew_dev || !io->done_dev || !io->trunc_dev)
        return fail(SO100_E_INVALID, "so100_step: act/obs/rew/done/trunc pointers are required%s");
    DeviceGuard g(s->cfg.device);
    if (!g.ok) return fail(SO100_E_NODEVICE, "so100_step: cannot select the device%s");
    StepPtrs p;
    p.state = s->state; p.start_tab = s->start_tab;
    p.act = io->act_dev; p.obs = io->obs_dev; p.rew = io->rew_dev; p.done = io->done_dev; p.trunc = io->trunc_dev;
    p.tobs = io->terminal_obs_dev; p.ep_ret = io->ep_return_dev; p.ep_len = io->ep_length_dev; p.inject = io->inject_dev; p.rollout_row = io->rollout_row_dev;
    return DISPATCH_KIND(s->cfg.env_kind, launch_step)(s, p, (hipStream_t)stream);
}

int so100_policy_forward(so100_sim* s, const so100_policy_weights* w, const so100_policy_io* io, uint32_t step_counter, void* stream) {
    if (!s || !w || !io) return fail(SO100_E_INVALID, "so100_policy_forward: null argument%s");
    if (!io->obs_dev || !io->act_env_dev) return fail(SO100_E_INVALID, "so100_policy_forward: obs and act_env pointers are required%s");
    const float* const* wp = reinterpret_cast<const float* const*>(w);
    for (int i = 0; i < 13; i++) if (!wp[i]) return fail(SO100_E_INVALID, "so100_policy_forward: null weight pointer%s");
    DeviceGuard g(s->cfg.device);
    if (!g.ok) return fail(SO100_E_NODEVICE, "so100_policy_forward: cannot select the device%s");
    PolicyWeights pw; static_assert(sizeof(PolicyWeights) == sizeof(so100_policy_weights), "layout");
    memcpy(&pw, w, sizeof pw);
    PolicyIO pio; pio.obs = io->obs_dev; pio.noise = io->noise_dev; pio.act_env = io->act_env_dev; pio.act_raw = io->act_raw_dev;
    pio.value = io->value_dev; pio.logp = io->logp_dev; pio.rollout_row = io->rollout_row_dev;
    // the matrix-core kernel, grid-stride over tiles of 64 envs, two workgroups per CU resident
    const int ntiles = (s->prm.n + 63)/64;
    const dim3 grid((unsigned)(ntiles < 512 ? ntiles : 512));
    if (so100_obs_dim(s->cfg.env_kind) == 15)
        hipLaunchKernelGGL((so100_policy_forward_mfma<15>), grid, dim3(256), 0, (hipStream_t)stream, s->prm.n, pw, pio, s->prm.seed_lo, s->prm.seed_hi, s->prm.env_id_offset, step_counter);
    else
        hipLaunchKernelGGL((so100_policy_forward_mfma<8>), grid, dim3(256), 0, (hipStream_t)stream, s->prm.n, pw, pio, s->prm.seed_lo, s->prm.seed_hi, s->prm.env_id_offset, step_counter);
    HIP_TRY(hipGetLastError(), SO100_E_LAUNCH);
    return 0;
}

int so100_rollout(so100_sim* s, const so100_policy_weights* w, const so100_rollout_io* io, int32_t T, uint32_t step_counter0, void* stream) {
    if (!s || !w || !io) return fail(SO100_E_INVALID, "so100_rollout: null argument%s");
    if (T < 1) return fail(SO100_E_INVALID, "so100_rollout: T must be >= 1%s");
    if (!io->rollout_dev || !io->obs_dev || !io->rew_dev || !io->done_dev || !io->trunc_dev)
        return fail(SO100_E_INVALID, "so100_rollout: rollout/obs/rew/done/trunc pointers are required%s");
    const float* const* wp = reinterpret_cast<const float* const*>(w);
    for (int i = 0; i < 13; i++) if (!wp[i]) return fail(SO100_E_INVALID, "so100_rollout: null weight pointer%s");
    DeviceGuard g(s->cfg.device);
    if (!g.ok) return fail(SO100_E_NODEVICE, "so100_rollout: cannot select the device%s");
    PolicyWeights pw; memcpy(&pw, w, sizeof pw);
    RolloutArgs ra; ra.buf = io->rollout_dev; ra.T = T; ra.step_counter0 = step_counter0; ra.obs_in = io->obs_dev; ra.tobs_chunk = io->terminal_obs_chunk_dev;
    const dim3 grid((unsigned)((s->prm.n + s->prm.epw - 1)/s->prm.epw));
    hipStream_t st = (hipStream_t)stream;
#define SO100_RL(KIND, FLV, NW) hipLaunchKernelGGL((so100_rollout_fused<KIND, FLV, NW>), grid, dim3(64*NW), 0, st, s->prm, s->state, s->start_tab, \
        io->obs_dev, io->rew_dev, io->done_dev, io->trunc_dev, io->terminal_obs_dev, io->ep_return_dev, io->ep_length_dev, pw, ra)
#define SO100_RL_FL(KIND, NW) do { if (s->prm.flags == SO100_F_CUBE_PINNED) SO100_RL(KIND, SO100_F_CUBE_PINNED, NW); \
        else if (s->prm.flags == SO100_F_NOPADS) SO100_RL(KIND, SO100_F_NOPADS, NW); \
        else if (s->prm.flags == SO100_F_REFERENCE) SO100_RL(KIND, SO100_F_REFERENCE, NW); \
        else if (s->prm.flags == SO100_F_CONTACT5 && (KIND <= 2 || KIND == 6)) SO100_RL(KIND <= 2 || KIND == 6 ? KIND : 1, SO100_F_CONTACT5, NW); \
        else SO100_RL(KIND, -1, NW); } while (0)
#define SO100_RL_KIND(NW) switch (s->cfg.env_kind) { case 1: SO100_RL_FL(1, NW); break; case 2: SO100_RL_FL(2, NW); break; \
        case 3: SO100_RL_FL(3, NW); break; case 4: SO100_RL_FL(4, NW); break; case 5: SO100_RL_FL(5, NW); break; default: SO100_RL_FL(6, NW); }
    SO100_RL_KIND(4)
#undef SO100_RL_KIND
#undef SO100_RL_FL
#undef SO100_RL
    HIP_TRY(hipGetLastError(), SO100_E_LAUNCH);
    return 0;
}

int so100_get_state(so100_sim* s, float* qpos_dev, float* qvel_dev, void* stream) {
    if (!s || !qpos_dev || !qvel_dev) return fail(SO100_E_INVALID, "so100_get_state: null argument%s");
    DeviceGuard g(s->cfg.device);
    if (!g.ok) return fail(SO100_E_NODEVICE, "so100_get_state: cannot select the device%s");
    const size_t n = (size_t)s->prm.n;
    HIP_TRY(hipMemcpyAsync(qpos_dev, s->state + (size_t)SF_QPOS0*n, 13*n*sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream), SO100_E_LAUNCH);
    HIP_TRY(hipMemcpyAsync(qvel_dev, s->state + (size_t)SF_QVEL0*n, 12*n*sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream), SO100_E_LAUNCH);
    return 0;
}
int so100_set_state(so100_sim* s, const float* qpos_dev, const float* qvel_dev, void* stream) {
    if (!s || !qpos_dev || !qvel_dev) return fail(SO100_E_INVALID, "so100_set_state: null argument%s");
    DeviceGuard g(s->cfg.device);
    if (!g.ok) return fail(SO100_E_NODEVICE, "so100_set_state: cannot select the device%s");
    const size_t n = (size_t)s->prm.n;
    HIP_TRY(hipMemcpyAsync(s->state + (size_t)SF_QPOS0*n, qpos_dev, 13*n*sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream), SO100_E_LAUNCH);
    HIP_TRY(hipMemcpyAsync(s->state + (size_t)SF_QVEL0*n, qvel_dev, 12*n*sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream), SO100_E_LAUNCH);
    HIP_TRY(hipMemsetAsync(s->state + (size_t)SF_qc0*n, 0, 6*n*sizeof(float), (hipStream_t)stream), SO100_E_LAUNCH);   // new q: drop the compensation
    return 0;
}
int so100_get_field(so100_sim* s, int32_t field, void* out_dev, void* stream) {
    if (!s || !out_dev || field < 0 || field >= SF_COUNT) return fail(SO100_E_INVALID, "so100_get_field: bad argument%s");
    DeviceGuard g(s->cfg.device);
    if (!g.ok) return fail(SO100_E_NODEVICE, "so100_get_field: cannot select the device%s");
    const size_t n = (size_t)s->prm.n;
    HIP_TRY(hipMemcpyAsync(out_dev, s->state + (size_t)field*n, n*sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream), SO100_E_LAUNCH);
    return 0;
}
int so100_set_field(so100_sim* s, int32_t field, const void* in_dev, void* stream) {
    if (!s || !in_dev || field < 0 || field >= SF_COUNT) return fail(SO100_E_INVALID, "so100_set_field: bad argument%s");
    DeviceGuard g(s->cfg.device);
    if (!g.ok) return fail(SO100_E_NODEVICE, "so100_set_field: cannot select the device%s");
    const size_t n = (size_t)s->prm.n;
    HIP_TRY(hipMemcpyAsync(s->state + (size_t)field*n, in_dev, n*sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream), SO100_E_LAUNCH);
    return 0;
}

}  // extern "C"
