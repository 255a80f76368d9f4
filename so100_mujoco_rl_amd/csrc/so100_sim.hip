// so100_sim.hip -- the C ABI of libso100sim.so (include/so100_sim.h).  The kernels live in so100_kernels.hpp; their per-kind
// instantiations are compiled in so100_kind.hip (one object per env kind), this file only dispatches to them.
#include "so100_kernels.hpp"
#include "so100_balance.hpp"

namespace so100 {
extern template struct KindOps<1>; extern template struct KindOps<2>; extern template struct KindOps<3>;
extern template struct KindOps<4>; extern template struct KindOps<5>; extern template struct KindOps<6>;
}

namespace {

using namespace so100;

thread_local char g_err[512] = "";
int fail(int code, const char* fmt, const char* a = "", long b = 0) {
    snprintf(g_err, sizeof g_err, fmt, a, b);
    return code;
}
#define HIP_TRY(expr, code) do { hipError_t e_ = (expr); if (e_ != hipSuccess) \
    return fail(code, "%s (HIP error %ld)", hipGetErrorString(e_), (long)e_); } while (0)

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
    int32_t* slot_env = nullptr;   // [workgroups x epw] lane-slot map of the persistent rollout kernel (pad-contact variants; so100_balance.hpp)
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
#define DISPATCH_KIND(kind, fn) \
    ((kind) == 1 ? KindOps<1>::fn : (kind) == 2 ? KindOps<2>::fn : (kind) == 3 ? KindOps<3>::fn : (kind) == 4 ? KindOps<4>::fn : (kind) == 5 ? KindOps<5>::fn : KindOps<6>::fn)
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
int so100_prof_read(int kind, long long* out48) { return DISPATCH_KIND(kind, prof_read)(out48); }
int so100_prof_read_wg(int kind, long long* wg4096, int* env32768) { return DISPATCH_KIND(kind, prof_read_wg)(wg4096, env32768); }
int so100_prof_read_hist(int kind, unsigned long long* hist48, int reset) { return DISPATCH_KIND(kind, prof_read_hist)(hist48, reset); }
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
    if (cfg->flags & ~(SO100_F_FRICTIONLOSS | SO100_F_LIMITS | SO100_F_FLOOR | SO100_F_CUBE_PINNED | SO100_F_PADS_FLOOR | SO100_F_PADS_CUBE | SO100_F_LINKS_FLOOR | SO100_F_LINKS_CUBE))
        return fail(SO100_E_INVALID, "so100_create: unknown flag bits%s");
    if ((cfg->flags & (SO100_F_PADS_CUBE | SO100_F_LINKS_CUBE)) && (cfg->flags & SO100_F_CUBE_PINNED))
        return fail(SO100_E_INVALID, "so100_create: SO100_F_PADS_CUBE / SO100_F_LINKS_CUBE need a dynamic cube (not SO100_F_CUBE_PINNED)%s");
    if (cfg->envs_per_workgroup != 0 && cfg->envs_per_workgroup != 16 && cfg->envs_per_workgroup != 32 && cfg->envs_per_workgroup != 64)
        return fail(SO100_E_INVALID, "so100_create: envs_per_workgroup must be 0 (automatic), 16, 32 or 64%s");
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
        if (cfg->flags & (SO100_F_FLOOR | SO100_F_PADS_FLOOR | SO100_F_PADS_CUBE | SO100_F_LINKS_FLOOR | SO100_F_LINKS_CUBE))
            while (epw > 16 && (cfg->num_envs + epw/2 - 1)/(epw/2) <= cus) epw /= 2;
        // contact disabled (no data-dependent solve): 32 envs per workgroup while that fits the CUs -- the persistent kernel's policy phase then
        // runs one 32-row MFMA tile per tower instead of two (so100_rollout_fused<K, 8, 4, 32>: half the matrix-core time per step)
        if (cfg->flags == SO100_F_CUBE_PINNED && (cfg->num_envs + 31)/32 <= cus) epw = 32;
        if (cfg->envs_per_workgroup != 0) epw = (int)cfg->envs_per_workgroup;             // the caller pins it (validated above)
        s->prm.epw = epw;
        s->prm.mw_max = mw_max_envs_for(cfg->flags);
        if (const char* ov = getenv("SO100_MW_MAX_ENVS")) { const long v = atol(ov); if (v >= 0) s->prm.mw_max = (int32_t)(v > (1L << 30) ? (1L << 30) : v); }
    }
    const size_t bytes = (size_t)SF_COUNT*(size_t)cfg->num_envs*sizeof(float);
    if (hipMalloc(&s->state, bytes) != hipSuccess) { delete s; return fail(SO100_E_NOMEM, "so100_create: hipMalloc of %s%ld bytes failed", "", (long)bytes); }
    float tab[36*6];
    for (int i = 0; i < 36; i++) for (int j = 0; j < 6; j++) tab[6*i + j] = (float)SO100_VALID_START_POSITIONS[i][j];
    if (hipMalloc(&s->start_tab, sizeof tab) != hipSuccess || hipMemcpy(s->start_tab, tab, sizeof tab, hipMemcpyHostToDevice) != hipSuccess) {
        (void)hipFree(s->state); if (s->start_tab) (void)hipFree(s->start_tab); delete s;
        return fail(SO100_E_NOMEM, "so100_create: start table upload failed%s");
    }
    {   // workgroup load balancing of the persistent rollout kernel (pad-contact variants, batches it serves; SO100_BALANCE=0 turns it off)
        const char* bal = getenv("SO100_BALANCE");
        const bool pads = (cfg->flags & (SO100_F_PADS_FLOOR | SO100_F_PADS_CUBE | SO100_F_LINKS_FLOOR | SO100_F_LINKS_CUBE)) != 0;
        if (pads && cfg->num_envs <= BALANCE_MAX_ENVS && !(bal && atoi(bal) == 0)) {
            const size_t slots = (size_t)((cfg->num_envs + s->prm.epw - 1)/s->prm.epw)*(size_t)s->prm.epw;
            if (hipMalloc(&s->slot_env, slots*sizeof(int32_t)) != hipSuccess) {
                (void)hipFree(s->state); (void)hipFree(s->start_tab); delete s;
                return fail(SO100_E_NOMEM, "so100_create: hipMalloc of the slot map failed%s");
            }
        }
    }
    const hipError_t he = DISPATCH_KIND(cfg->env_kind, init)(s->prm.n, s->state);
    if (he != hipSuccess) { (void)hipFree(s->state); (void)hipFree(s->start_tab); if (s->slot_env) (void)hipFree(s->slot_env); delete s; return fail(SO100_E_LAUNCH, "so100_create: %s (HIP error %ld)", hipGetErrorString(he), (long)he); }
    *out = s;
    return 0;
}

int so100_envs_per_workgroup(const so100_sim* s) { return s ? s->prm.epw : SO100_E_INVALID; }

void so100_destroy(so100_sim* s) {
    if (!s) return;
    DeviceGuard g(s->cfg.device);
    if (s->state) (void)hipFree(s->state);
    if (s->start_tab) (void)hipFree(s->start_tab);
    if (s->slot_env) (void)hipFree(s->slot_env);
    delete s;
}

int so100_reset(so100_sim* s, const uint8_t* mask_dev, const float* inject_dev, float* obs_dev, void* stream) {
    if (!s) return fail(SO100_E_INVALID, "so100_reset: null handle%s");
    DeviceGuard g(s->cfg.device);
    if (!g.ok) return fail(SO100_E_NODEVICE, "so100_reset: cannot select the device%s");
    HIP_TRY(DISPATCH_KIND(s->cfg.env_kind, reset)(s->prm, s->state, s->start_tab, mask_dev, inject_dev, obs_dev, (hipStream_t)stream), SO100_E_LAUNCH);
    return 0;
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
    HIP_TRY(DISPATCH_KIND(s->cfg.env_kind, step)(s->prm, p, (hipStream_t)stream), SO100_E_LAUNCH);
    return 0;
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
    ra.slot_env = s->slot_env;
    if (s->slot_env) {                                       // deal the envs out over the workgroups by their contact load
        const int nwg = (s->prm.n + s->prm.epw - 1)/s->prm.epw;
        hipLaunchKernelGGL(so100_build_slot_map, dim3(1), dim3(BALANCE_THREADS), 0, (hipStream_t)stream, s->prm.n, s->prm.epw, nwg,
                           reinterpret_cast<const int32_t*>(s->state + (size_t)SF_contact_load*(size_t)s->prm.n), s->slot_env);
        HIP_TRY(hipGetLastError(), SO100_E_LAUNCH);
    }
    RolloutPtrs rp{ io->obs_dev, io->rew_dev, io->done_dev, io->trunc_dev, io->terminal_obs_dev, io->ep_return_dev, io->ep_length_dev };
    HIP_TRY(DISPATCH_KIND(s->cfg.env_kind, rollout)(s->prm, s->state, s->start_tab, rp, pw, ra, (hipStream_t)stream), SO100_E_LAUNCH);
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
