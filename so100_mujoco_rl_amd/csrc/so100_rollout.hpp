// so100_rollout.hpp -- persistent rollout kernel: T vectorised steps of {policy forward, sample, env step} in ONE launch.
//
// One workgroup = NW waves = 64 envs.  Each step has two phases:
//   policy phase  (all 4 waves): the two hidden layers of both towers on the matrix cores (fp32 MFMA, weight
//                 fragments resident in VGPRs for the whole launch, activations through LDS); the heads on the VALU:
//                 wave 0 ends with the action of env = lane in registers.
//   physics phase: the same fused env step as so100_step_fused (reward -> ctrl -> 16 substeps -> obs -> TimeLimit ->
//                 auto-reset) for env = lane of wave 0, with each substep's RNEA bias force computed concurrently on
//                 wave 1 and, when the cube is simulated, its free-body / floor-contact substep on wave 2 (two workgroup
//                 barriers per substep; sin/cos, q-dot, the bias and the cube state cross through LDS).
// The env state lives in wave 0's registers across all T steps (loaded once, stored once per launch), the
// observation goes to the next policy phase through LDS, and the only per-step HBM traffic is the rollout-buffer row
// (obs | action | reward | done | value | logp = (obs_dim+10) words per env).  Compared with one policy launch + one
// env launch per step this removes 2T-1 kernel boundaries, the per-step state round trip and the per-launch
// first-touch latencies of the weights.
#pragma once
#include "so100_policy.hpp"

namespace so100 {

struct RolloutArgs {
    float* buf;                  // [T][N][OD+10]
    int32_t T;
    uint32_t step_counter0;      // policy-noise Philox counter of the first step
    const float* obs_in;         // [N][OD] current observation (reset or previous launch)
    float* tobs_chunk;           // [T][N][OD] terminal observation of every episode end inside the chunk (written where done); nullable
    const int32_t* slot_env;     // [workgroups x epw] env of every lane slot, -1 = none (so100_balance.hpp); null = identity (slot s holds env s)
};

typedef float f32x16 __attribute__((ext_vector_type(16)));

// Optional cycle accounting of the persistent kernel's phases (tools/rollout_prof.py builds a side library with
// -DSO100_ROLLOUT_PROF; the product library compiles these to nothing).  Slots: see tools/rollout_prof.py.
#ifdef SO100_ROLLOUT_PROF
__device__ long long so100_prof[48];
__device__ long long so100_prof_wg[1024*4];    // per workgroup: wave 0 total, wave 3 narrowphase, wave 3 contact Newton, wave 0 barrier-3 wait
__device__ int so100_prof_env[16384*2];        // per env: row passes of the contact Newton, substeps with a pad contact (this launch)
// per (workgroup, substep): [0..15] histogram of the slowest lane's gradient + Hessian passes (where any lane solved), [16..32] histogram of the number of
// envs in pad contact, [33..36] passes of all envs by type (full, sign, gradient, line search), [37] sum of the slowest lane's estimated instruction
// count (970 / 140 / 520 / 300 per pass type), [38] (workgroup, substep) pairs with a solve, [39] all pairs, [40] sum over ENVS of that estimate
__device__ unsigned long long so100_prof_hist[48];
struct Prof {
    long long t[12] = {}, c = __builtin_readcyclecounter();
    int work = 0, insub = 0, sub = 0;
    // wave 3, once per substep (all its lanes): `sub` holds this substep's typed pass counts of the lane's env (0 if it did not solve)
    __device__ __forceinline__ void substep_stats(bool solved, bool first_of_env) {
        const int f = sub & 255, s = (sub >> 8) & 255, g = (sub >> 16) & 255, l = (sub >> 24) & 255;
        work += f + s + g + l;
        int cost = solved ? 970*f + 140*s + 520*g + 300*l : 0, fmaxl = solved ? f : 0;
        const unsigned long long m = __ballot(solved && first_of_env);
        int best = cost, bf = fmaxl;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) { const int oc = __shfl_xor(best, d, 64), of = __shfl_xor(bf, d, 64); if (oc > best) { best = oc; bf = of; } }
        if (solved && first_of_env) {
            atomicAdd(&so100_prof_hist[33], (unsigned long long)f); atomicAdd(&so100_prof_hist[34], (unsigned long long)s);
            atomicAdd(&so100_prof_hist[35], (unsigned long long)g); atomicAdd(&so100_prof_hist[36], (unsigned long long)l);
            atomicAdd(&so100_prof_hist[40], (unsigned long long)cost);
        }
        if ((threadIdx.x & 63) == 0) {
            const int ne = __popcll(m);
            atomicAdd(&so100_prof_hist[16 + (ne > 16 ? 16 : ne)], 1ull); atomicAdd(&so100_prof_hist[39], 1ull);
            if (m) { atomicAdd(&so100_prof_hist[bf > 15 ? 15 : bf], 1ull); atomicAdd(&so100_prof_hist[37], (unsigned long long)best); atomicAdd(&so100_prof_hist[38], 1ull); }
        }
        sub = 0;
    }
    __device__ __forceinline__ void mark(int slot) { const long long n = __builtin_readcyclecounter(); t[slot] += n - c; c = n; }
    __device__ __forceinline__ void flush(int base, bool who) const { if (who) for (int i = 0; i < 12; i++) so100_prof[base + i] = t[i]; }
    __device__ __forceinline__ void flush_wg(int wave, int lane, int env, bool live) const {
        if (blockIdx.x < 1024 && lane == 0) {
            long long tot = 0; for (int i = 0; i < 12; i++) tot += t[i];
            if (wave == 0) { so100_prof_wg[blockIdx.x*4 + 0] = tot; so100_prof_wg[blockIdx.x*4 + 3] = t[8]; }
            if (wave == 3) { so100_prof_wg[blockIdx.x*4 + 1] = t[4] + t[10]; so100_prof_wg[blockIdx.x*4 + 2] = t[6] + t[11]; }
        }
        if (wave == 3 && live && env < 16384) { so100_prof_env[2*env] = work; so100_prof_env[2*env + 1] = insub; }
    }
};
#define SO100_PROF_WORK (&prof_.sub)
#define SO100_PROF_INSUB() (prof_.insub++)
#else
struct Prof {
    __device__ __forceinline__ void substep_stats(bool, bool) {}
    __device__ __forceinline__ void mark(int) {}
    __device__ __forceinline__ void flush(int, bool) const {}
    __device__ __forceinline__ void flush_wg(int, int, int, bool) const {}
};
#define SO100_PROF_WORK nullptr
#define SO100_PROF_INSUB() ((void)0)
#endif
#define SO100_PROF_DECL Prof prof_;
#define SO100_PROF(slot) prof_.mark(slot)
#define SO100_PROF_FLUSH(base) prof_.flush((base), blockIdx.x == 0 && lane == 0)

// ---------------------------------------------------------------------------------------------------------------
// The 16 physics substeps of ONE env step for the 64 envs of a workgroup, split over its waves (all waves of the
// workgroup must call this; env state of env = lane lives in wave 0's registers):
//   wave 0: sin/cos, CRBA mass matrix + factorisation, then solve + integrate          wave 1: RNEA bias force
//   wave 2: the cube's free-body / floor-contact substep (when it is simulated)         others: only the barriers
// Two workgroup barriers per substep; sin/cos, q-dot, the bias force and the cube state cross through LDS.  Same
// operations in the same order as physics_substeps(), so results agree with it to the last bit or two.  Used by the
// persistent rollout kernel and by the multi-wave single-step kernel (so100_step_mw).  `after_first_barrier(sub)` lets
// the caller hang work on an idle wave (the rollout kernel pre-draws the next step's policy noise on wave 3).
// On return (wave 0): e.q/v/qc/ff/fl/cube updated, A.s / A.c = sin/cos the LAST substep started from, cstale = the
// cube position it started from (Q1).
// ---------------------------------------------------------------------------------------------------------------
// With the finger-pad contact flags (PADS) wave 3 -- idle otherwise, and with an empty register file: it holds no env state --
// is the CONTACT WAVE of the workgroup.  Per substep, concurrently with RNEA (wave 1) / CRBA + factorisation (wave 0) /
// cube_prepare (wave 2): world-frame kinematics and the pad narrowphase, records into LDS (cbuf, [record][field][lane]) and, if
// any of its lanes has a contact, its own copy of the CRBA mass matrix (the primal solve needs M itself, not its factor).  After
// the mid-substep barrier it solves the lanes that have contacts in the primal (so100_contact.hpp) while wave 0 runs the block
// PGS for all lanes as before; a third barrier later wave 0 takes the contact wave's acceleration for the lanes that had
// contacts, and wave 2 the cube's for lanes whose pad touches the cube (arm + cube solved together), and both integrate.
// (The first version ran the Newton on wave 0 beside the env state: 409 scratch accesses in the kernel, 78 % of the wave's
// cycles waiting on them, 600 us per step in sustained contact.)
// LDS: xq [24][64] = sin q, cos q, v, q; xk [12][64] = ctrl (written once per env step) and the arm's acceleration of the
// previous substep (6-11: the Newton's warm start, republished by wave 0 after every substep); xa [15][64] = arm
// acceleration (0-5) and cube acceleration (6-11) of the contact wave's solve, contact code (12: count | coupled << 8 |
// dropped << 16), solver residual (13), signature of the pad-contact set (14).  During the FIRST half of a substep rows 0-11 of xa and the
// q rows of xq (18-23) carry, from wave 1 to the contact wave, the arm's smooth force tau (xa 0-5), the limit rows' constants clv (xa 6-11) and
// sg / R_limit (xq 18-23): wave 1 forms them behind RNEA (it is the short leg of that half), the contact wave then needs no q, ctrl or bias.
// xm [21][64] = the arm's mass matrix (packed lower, unfactored), published by wave 0 before it factorises it in place.
// xw [36][64] (nullable) = the six joint axes z_k and screw terms o_k x z_k in the world frame (all the solve needs of the world FK), from the
// contact wave's detection to its solve; without it the solve rebuilds the frames from sin / cos (330 instructions per substep in contact).
struct PhaseLds { float (*xq)[64]; float (*xc)[64]; float (*xb)[64]; float* cbuf; float (*xa)[64]; float (*xk)[64]; float (*xm)[64]; unsigned char* pbuf; float (*xw)[64]; };
// the contact wave's active-set memory (so100_contact.hpp: primal_newton) lives for ONE env step, like the one-wave kernel's
// (physics_substeps): zones = the arm rows' zones of the last contact solve (-1: none yet), prev_n = length of the (id | mask) list in L.pbuf
struct ContactMemo { int zones = -1, prev_n = 0; };

// Register pressure.  The kernel is ONE control-flow graph: whatever another wave will read later (wave 0's env state `e`, its
// mass-matrix factor A, wave 2's cube block) is live across the contact wave's Newton as far as the register allocator can
// tell, although those registers hold nothing on wave 3.  After its solve the contact wave therefore overwrites all of it
// with constants (`forget`, the caller's `forget_caller_state` for what lives outside this function): every path from the
// Newton to a use passes that definition, so the old values are dead during the solve and their registers are free.
template <bool PADS, bool LINKS, class Hook, class Forget>
__device__ __forceinline__ void physics_phase_mw(const SimParams& p, int wave, int lane, EnvState& e, float ctrl[6], float cstale[3],
                                                 Arm<float>& A, const PhaseLds& L, ContactMemo& memo, Prof& prof_, Hook after_first_barrier, Forget forget_caller_state) {
    float (*xq)[64] = L.xq; float (*xc)[64] = L.xc; float (*xb)[64] = L.xb; float (*xa)[64] = L.xa; float (*xk)[64] = L.xk; float (*xm)[64] = L.xm;
    float dq[6] = { 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f };
    // The cube is dynamically independent of the arm unless a pad touches it: when it is simulated (not pinned) wave 2 owns it
    // for the substep loop and steps it concurrently (cube/floor Newton ~ 900 instructions per substep in contact).
    const bool cube_live = (p.flags & F_CUBE_PINNED) == 0u;
    const bool pads = PADS && (p.flags & F_ANY_CONTACT) != 0u;
    const bool padcube = PADS && (p.flags & F_ARM_CUBE) != 0u && cube_live;       // pad/cube or link/cube pairs: the contact wave needs the cube's state
    Cube<float> cb; CubePrep<float> cprep;
    float applied[3] = { 0.0f, 0.0f, 0.0f };
    memo = ContactMemo{};                                      // every kernel starts an env step without active-set memory: one definition of the solve
    // Contact wave: with p.epw < 64 envs per workgroup each env gets 64 / p.epw adjacent lanes (so100_contact.hpp: cooperative lanes);
    // el = the env's column in the LDS images, part = this lane's place in the env's group.
    const int np3 = 64/p.epw, sh3 = np3 == 4 ? 2 : np3 == 2 ? 1 : 0;
    const int el = lane >> sh3, part3 = lane & (np3 - 1);
    if (wave == 0) {
        e.res = 0.0f; e.cstat = 0; e.csig = 0;
        if (pads) {
#pragma unroll
            for (int i = 0; i < 6; i++) { xk[i][lane] = ctrl[i]; xk[6 + i][lane] = e.aw[i]; }
        }
    }
    if (cube_live) {
        if (wave == 0) {
#pragma unroll
            for (int i = 0; i < 3; i++) xc[i][lane] = e.cube.pos[i];
#pragma unroll
            for (int i = 0; i < 4; i++) xc[3 + i][lane] = e.cube.quat[i];
#pragma unroll
            for (int i = 0; i < 6; i++) { xc[7 + i][lane] = e.cube.vel[i]; xc[13 + i][lane] = e.cube.warm[i]; }
            xc[19][lane] = (e.bits & B_ANTIGRAV) ? (float)(so100g::CUBE_MASS*so100g::GRAVITY) : 0.0f;
        }
        __syncthreads();
        if (wave == 2) {
#pragma unroll
            for (int i = 0; i < 3; i++) cb.pos[i] = xc[i][lane];
#pragma unroll
            for (int i = 0; i < 4; i++) cb.quat[i] = xc[3 + i][lane];
#pragma unroll
            for (int i = 0; i < 6; i++) { cb.vel[i] = xc[7 + i][lane]; cb.warm[i] = xc[13 + i][lane]; }
            applied[2] = xc[19][lane];
        }
    } else {
        if (wave == 0) { cstale[0] = e.cube.pos[0]; cstale[1] = e.cube.pos[1]; cstale[2] = e.cube.pos[2]; }      // kinematic cube
        if (pads) __syncthreads();                         // xk visible to the contact wave
    }
    if (pads && wave == 3) { xa[12][lane] = 0.0f; xa[14][lane] = 0.0f; }      // (columns >= p.epw are never written again: no contacts there)
    // (Tried: letting wave 2 run the cube's 16 substeps back to back ahead of the arm when no pad can touch it.  A workgroup
    // barrier needs every wave, so the others simply waited for it at the first one: 72 -> 101 us per step.  The cube stays in
    // step with the arm: detection + row set-up in the first half-substep, Newton in the second.)
    const bool cube_in_step = cube_live;
#pragma unroll 1
    for (int sub = 0; sub < p.frame_skip; sub++) {
        if (wave == 0) {
            // sin/cos: exact at the first substep, then rotated by the integration increment (arm_substep does the same)
            if (sub == 0) arm_trig<float>(e.q, A); else arm_trig_update<float>(e.q, dq, A);
#pragma unroll
            for (int i = 0; i < 6; i++) { xq[i][lane] = A.s[i]; xq[6 + i][lane] = A.c[i]; xq[12 + i][lane] = e.v[i]; }
            if (pads) {
#pragma unroll
                for (int i = 0; i < 6; i++) xq[18 + i][lane] = e.q[i];
            }
            SO100_PROF(2);                                 // trig (wave 0)
        }
        if (padcube && wave == 2 && sub > 0) {             // the cube's pose for this substep's narrowphase / coupled solve
#pragma unroll
            for (int i = 0; i < 3; i++) xc[i][lane] = cb.pos[i];
#pragma unroll
            for (int i = 0; i < 4; i++) xc[3 + i][lane] = cb.quat[i];
#pragma unroll
            for (int i = 0; i < 6; i++) { xc[7 + i][lane] = cb.vel[i]; xc[13 + i][lane] = cb.warm[i]; }
        }
        __syncthreads();
        SO100_PROF(3);                                     // barrier 1 wait
        after_first_barrier(sub);
        // ---- first half of the substep: everything that does not need the other waves' results
        WorldFK<float> W3; Arm<float> A3; ContactsLds<float> cs3{ L.cbuf, el, L.pbuf, part3, np3 };     // contact wave only
        cs3.prev_n = memo.prev_n;
        float Rc3[9] = { 1.0f, 0.0f, 0.0f, 0.0f, 1.0f, 0.0f, 0.0f, 0.0f, 1.0f }, cpos3[3] = { 0.0f, 0.0f, 0.0f };
        bool coupled3 = false, any3 = false;
        if (wave == 1) {
            float v1[6], q1[6] = { 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f }, c1[6] = { 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f };
#pragma unroll
            for (int i = 0; i < 6; i++) { A.s[i] = xq[i][lane]; A.c[i] = xq[6 + i][lane]; v1[i] = xq[12 + i][lane]; }
            if (pads) {
#pragma unroll
                for (int i = 0; i < 6; i++) { q1[i] = xq[18 + i][lane]; c1[i] = xk[i][lane]; }
            }
            arm_bias<float>(v1, A);
#pragma unroll
            for (int i = 0; i < 6; i++) xb[i][lane] = A.bias[i];
            if (pads) {                                    // what the contact wave's primal solve needs of q, ctrl and the bias (see "LDS" above)
                float t1[6]; ArmRows<float> r1;
                arm_tau<float>(q1, v1, c1, A, t1);
                arm_row_consts<float>(q1, v1, p.flags, r1);
#pragma unroll
                for (int i = 0; i < 6; i++) { xa[i][lane] = t1[i]; xa[6 + i][lane] = r1.clv[i]; xq[18 + i][lane] = r1.sg[i]*r1.Dl[i]; }
            }
            SO100_PROF(4);                                 // RNEA (wave 1)
        } else if (wave == 0) {
            arm_mass<float>(A);
            if (pads) {                                    // the contact wave's primal solve needs M itself
#pragma unroll
                for (int i = 0; i < 21; i++) xm[i][lane] = A.M[i];
            }
            arm_factor<float>(p.flags, A);          // everything that needs only M happens before the barrier
            SO100_PROF(4);                                 // CRBA + factor (wave 0)
        } else if (wave == 2 && cube_in_step) {
            if (sub == p.frame_skip - 1) { xc[20][lane] = cb.pos[0]; xc[21][lane] = cb.pos[1]; xc[22][lane] = cb.pos[2]; }   // stale xpos (Q1)
            cube_prepare<float>(cb, applied, p.flags, cprep);           // contact detection + row setup ...
            SO100_PROF(4);                                 // cube_prepare (wave 2)
        } else if (wave == 3 && pads) {
            float v3[6];
#pragma unroll
            for (int i = 0; i < 6; i++) { A3.s[i] = xq[i][el]; A3.c[i] = xq[6 + i][el]; v3[i] = xq[12 + i][el]; }
            world_fk<float>(A3.s, A3.c, W3);
            SO100_PROF(10);                                // world FK (wave 3)
            Cube<float> c3b;
#pragma unroll
            for (int i = 0; i < 3; i++) c3b.pos[i] = 0.0f;
#pragma unroll
            for (int i = 0; i < 6; i++) c3b.vel[i] = 0.0f;
            if (padcube) {
#pragma unroll
                for (int i = 0; i < 3; i++) { c3b.pos[i] = xc[i][el]; cpos3[i] = c3b.pos[i]; }
                float qn[4] = { xc[3][el], xc[4][el], xc[5][el], xc[6][el] };
#pragma unroll
                for (int i = 0; i < 6; i++) c3b.vel[i] = xc[7 + i][el];
                quat_normalize(qn); quat_to_mat(qn, Rc3);
            }
            coupled3 = detect_pad_contacts<float>(W3, v3, c3b, Rc3, p.flags, padcube, cs3);
            memo.prev_n = cs3.prev_n;
            if (part3 == 0) xa[12][el] = __int_as_float(cs3.n | (coupled3 ? 256 : 0) | ((cs3.dropped > 0xFFFF ? 0xFFFF : cs3.dropped) << 16));
            any3 = __any(cs3.n > 0);
            if (any3) { const int sig3 = cs3.n > 0 ? contact_signature(cs3) : 0; if (part3 == 0) xa[14][el] = __int_as_float(sig3); }
            if (L.xw && cs3.n > 0 && part3 == 0) {         // the joint frames for the solve after the barrier
#pragma unroll
                for (int i = 0; i < 6; i++)
#pragma unroll
                    for (int k = 0; k < 3; k++) { L.xw[3*i + k][el] = W3.z[i][k]; L.xw[18 + 3*i + k][el] = W3.oz[i][k]; }
            }
            SO100_PROF(4);                                 // FK + narrowphase (wave 3)
        }
        __syncthreads();
        SO100_PROF(5);                                     // barrier 2 wait
        // ---- second half: solves
        float cal[3], caa[3], acc0[6]; ArmRows<float> r0;
        if (wave == 2 && cube_in_step) { cube_solve<float>(cb, p.flags, p.contact_iters, cprep, cal, caa); SO100_PROF(6); }   // Newton behind the arm's solve
        if (wave == 0) {
#pragma unroll
            for (int i = 0; i < 6; i++) A.bias[i] = xb[i][lane];
            float tau[6];
            arm_tau<float>(e.q, e.v, ctrl, A, tau);
            if ((p.flags & (F_FRICTIONLOSS | F_LIMITS)) != 0u) {
                arm_rows<float>(e.q, e.v, tau, e.ff, e.fl, p.flags, A, r0);
                float res;
                arm_pgs<float>(e.ff, e.fl, p.solver_iters, A, r0, acc0, res);
                e.res = tmax(e.res, res);
            } else {
                if (pads) arm_row_consts<float>(e.q, e.v, p.flags, r0);
#pragma unroll
                for (int i = 0; i < 6; i++) acc0[i] = tau[i];
                ldl6_solve<float>(A.M, A.Dinv, acc0);
            }
            SO100_PROF(6);                                 // solve (wave 0)
        }
        if (wave == 3 && pads && any3) {
            if (cs3.n > 0) {
                float v3[6], tau3[6], x3[6], sD3[6], clv3[6], xcube[6] = { 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f }, cwarm[6], ap3[3] = { 0.0f, 0.0f, 0.0f };
                // The joint axes / screw terms: from LDS where the kernel has room for them, else rebuilt from the published sin / cos (330
                // instructions).  Keeping the detection's copy in registers across the barrier is not an option: one register allocation
                // serves the whole kernel, and the 72 floats showed up as scratch traffic in the arm / cube legs.
                if (L.xw) {
#pragma unroll
                    for (int i = 0; i < 6; i++)
#pragma unroll
                        for (int k = 0; k < 3; k++) { W3.z[i][k] = L.xw[3*i + k][el]; W3.oz[i][k] = L.xw[18 + 3*i + k][el]; }
                } else {
#pragma unroll
                    for (int i = 0; i < 6; i++) { A3.s[i] = xq[i][el]; A3.c[i] = xq[6 + i][el]; }
                    world_fk<float>(A3.s, A3.c, W3);
                }
                if (padcube) {
#pragma unroll
                    for (int i = 0; i < 3; i++) cpos3[i] = xc[i][el];
                    float qn[4] = { xc[3][el], xc[4][el], xc[5][el], xc[6][el] };
                    quat_normalize(qn); quat_to_mat(qn, Rc3);
                }
#pragma unroll
                for (int i = 0; i < 21; i++) A3.M[i] = xm[i][el];
#pragma unroll
                for (int i = 0; i < 6; i++) { v3[i] = xq[12 + i][el]; tau3[i] = xa[i][el]; clv3[i] = xa[6 + i][el]; sD3[i] = xq[18 + i][el]; x3[i] = xk[6 + i][el]; cwarm[i] = 0.0f; }
                ArmRows<float> r3;
                arm_row_consts_from<float>(v3, p.flags, sD3, clv3, r3);
                if (coupled3) {
#pragma unroll
                    for (int i = 0; i < 6; i++) cwarm[i] = xc[13 + i][el];
                    ap3[2] = xc[19][el];
                }
                SO100_PROF(11);                                // contact solve set-up (wave 3)
                const float res = contact_solve<LINKS>(tau3, r3, A3.M, W3, cs3, coupled3, cpos3, cwarm, Rc3, ap3, p.contact_iters, x3, xcube, &memo.zones, SO100_PROF_WORK);
                SO100_PROF_INSUB();
                if (part3 == 0) {
#pragma unroll
                    for (int i = 0; i < 6; i++) { xa[i][el] = x3[i]; xa[6 + i][el] = xcube[i]; }
                    xa[13][el] = res;
                }
            }
            // (see "Register pressure" above)
            e = EnvState{}; A = Arm<float>{}; cb = Cube<float>{}; cprep = CubePrep<float>{};
#pragma unroll
            for (int i = 0; i < 6; i++) { dq[i] = 0.0f; ctrl[i] = 0.0f; }
            applied[0] = applied[1] = applied[2] = 0.0f; cstale[0] = cstale[1] = cstale[2] = 0.0f;
            forget_caller_state();
            SO100_PROF(6);                                 // contact Newton (wave 3)
        }
        if (wave == 3 && pads) prof_.substep_stats(cs3.n > 0, part3 == 0);      // (profiling builds only)
        if (pads) __syncthreads();                         // the contact wave's accelerations are in xa
        SO100_PROF(8);                                     // barrier 3 wait
        if (wave == 0) {
            if (pads) {
                const int code = __float_as_int(xa[12][lane]);
                const int nc = code & 255;
                if (nc > 0) {
#pragma unroll
                    for (int i = 0; i < 6; i++) acc0[i] = xa[i][lane];
                    arm_row_forces<float>(r0, acc0, e.ff, e.fl);          // the block PGS's warm-start memory for when the contact is gone
                    e.res = tmax(e.res, xa[13][lane]);
                }
#pragma unroll
                for (int i = 0; i < 6; i++) { e.aw[i] = acc0[i]; xk[6 + i][lane] = acc0[i]; }      // the next substep's Newton warm start (read after its first barrier)
                const int n0 = e.cstat & 255, dr = (e.cstat >> 8) + (code >> 16);
                e.cstat = (nc > n0 ? nc : n0) | ((dr > 0xFFFF ? 0xFFFF : dr) << 8);
                e.csig = nc > 0 ? __float_as_int(xa[14][lane]) : 0;
                e.cload += nc > 0 ? 1 << 16 : 0;               // this launch's contact substeps ride in the upper half until the launch folds them in
            }
            arm_integrate<float>(e.q, e.v, e.qc, acc0, dq);
        }
        if (wave == 2 && cube_in_step) {
            if (padcube && (__float_as_int(xa[12][lane]) & 256) != 0) {      // arm and cube were solved together on the contact wave
#pragma unroll
                for (int i = 0; i < 3; i++) { cal[i] = xa[6 + i][lane]; caa[i] = xa[9 + i][lane]; cb.warm[i] = cal[i] - cprep.a0[i]; cb.warm[3 + i] = caa[i]; }
            }
            cube_integrate<float>(cb, cal, caa);
        }
        SO100_PROF(9);                                     // integrate
    }
    if (cube_live) {
        if (wave == 2) {
#pragma unroll
            for (int i = 0; i < 3; i++) xc[i][lane] = cb.pos[i];
#pragma unroll
            for (int i = 0; i < 4; i++) xc[3 + i][lane] = cb.quat[i];
#pragma unroll
            for (int i = 0; i < 6; i++) { xc[7 + i][lane] = cb.vel[i]; xc[13 + i][lane] = cb.warm[i]; }
        }
        __syncthreads();
        if (wave == 0) {
#pragma unroll
            for (int i = 0; i < 3; i++) { e.cube.pos[i] = xc[i][lane]; cstale[i] = xc[20 + i][lane]; }
#pragma unroll
            for (int i = 0; i < 4; i++) e.cube.quat[i] = xc[3 + i][lane];
#pragma unroll
            for (int i = 0; i < 6; i++) { e.cube.vel[i] = xc[7 + i][lane]; e.cube.warm[i] = xc[13 + i][lane]; }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Stand-alone policy forward on the matrix cores (the policy phase of the rollout kernel as its own launch): each
// workgroup (4 waves) keeps its weight fragments in registers and walks over tiles of 64 envs (grid-stride), so the
// weights are fetched once per workgroup, not once per 64 envs.  Same arithmetic as so100_rollout_fused's policy phase.
// Replaces the VALU kernel so100_policy_forward_kernel as what so100_policy_forward launches (1 M envs: 961 -> see
// profiles/): the VALU kernel re-staged 43 KB of weights per 64 envs and ran the 2 x 64 x (OD + 64) products at VALU rate.
// ---------------------------------------------------------------------------------------------------------------
template <int OD>
__global__ void __launch_bounds__(256, 2) so100_policy_forward_mfma(int n, PolicyWeights w, PolicyIO io, uint32_t seed_lo, uint32_t seed_hi,
                                                                 uint32_t env_id_offset, uint32_t step_counter) {
    constexpr int ODP = (OD + 3) & ~3;
    constexpr int LD = 65;
    __shared__ __attribute__((aligned(16))) float hd[6*64 + 64 + 16];   // mu_w | v_w | mu_b(6) log_std(6) v_b(1)
    __shared__ float oxt[64][ODP + 1];
    __shared__ float h1t[2][64][LD];
    __shared__ float h2t[2][64][LD];
    __shared__ float xmean[6][64];
    __shared__ float xn[6][64];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int tower = wave >> 1, rt = wave & 1;
    const int lj = lane & 31, lh = lane >> 5;
    const float* W1 = tower ? w.vf_w0 : w.pi_w0; const float* W2 = tower ? w.vf_w1 : w.pi_w1;
    const float* Bi1 = tower ? w.vf_b0 : w.pi_b0; const float* Bi2 = tower ? w.vf_b1 : w.pi_b1;
    float bw1[2][ODP/2], bw2[2][32], b1v[2], b2v[2];
#pragma unroll
    for (int ct = 0; ct < 2; ct++) {
        const int unit = 32*ct + lj;
        b1v[ct] = Bi1[unit]; b2v[ct] = Bi2[unit];
#pragma unroll
        for (int s2 = 0; s2 < ODP/2; s2++) { const int k = 2*s2 + lh; bw1[ct][s2] = k < OD ? W1[unit*OD + k] : 0.0f; }
#pragma unroll
        for (int s2 = 0; s2 < 32; s2++) bw2[ct][s2] = W2[unit*64 + 2*s2 + lh];
    }
    if (wave == 3) {
        for (int i = lane; i < 6*64; i += 64) hd[i] = w.mu_w[i];
        hd[6*64 + lane] = w.v_w[lane];
        if (lane < 6) { hd[7*64 + lane] = w.mu_b[lane]; hd[7*64 + 6 + lane] = w.log_std[lane]; }
        if (lane == 0) hd[7*64 + 12] = w.v_b[0];
    }
    for (int i = threadIdx.x; i < 64*(ODP + 1); i += 256) (&oxt[0][0])[i] = 0.0f;     // padding columns stay zero
    const int ntiles = (n + 63)/64;
#pragma unroll 1
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int env = tile*64 + lane;
        const bool live = env < n;
        __syncthreads();                                            // previous tile fully consumed (and the zero fill / hd visible)
        {   // the tile's observations: 64 x OD contiguous floats, loaded by all 256 threads
            const size_t base = (size_t)tile*64*OD;
            const int cnt = (n - tile*64 < 64 ? n - tile*64 : 64)*OD;
            for (int i = threadIdx.x; i < 64*OD; i += 256) oxt[i / OD][i % OD] = i < cnt ? io.obs[base + i] : 0.0f;
        }
        __syncthreads();
        {
            f32x16 acc0, acc1;
#pragma unroll
            for (int r = 0; r < 16; r++) { acc0[r] = b1v[0]; acc1[r] = b1v[1]; }
#pragma unroll
            for (int s2 = 0; s2 < ODP/2; s2++) {
                const float a = oxt[32*rt + lj][2*s2 + lh];
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bw1[0][s2], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bw1[1][s2], acc1, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int row = 32*rt + (r & 3) + 8*(r >> 2) + 4*lh;
                h1t[tower][row][lj] = fast_tanh(acc0[r]); h1t[tower][row][32 + lj] = fast_tanh(acc1[r]);
            }
        }
        __syncthreads();
        {
            f32x16 acc0, acc1;
#pragma unroll
            for (int r = 0; r < 16; r++) { acc0[r] = b2v[0]; acc1[r] = b2v[1]; }
#pragma unroll
            for (int s2 = 0; s2 < 32; s2++) {
                const float a = h1t[tower][32*rt + lj][2*s2 + lh];
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bw2[0][s2], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bw2[1][s2], acc1, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int row = 32*rt + (r & 3) + 8*(r >> 2) + 4*lh;
                h2t[tower][row][lj] = fast_tanh(acc0[r]); h2t[tower][row][32 + lj] = fast_tanh(acc1[r]);
            }
        }
        __syncthreads();
        // heads: waves 0, 1, 3 two action means each; wave 2 the value head and the noise
        if (wave != 2) {
            const int a0 = wave == 0 ? 0 : (wave == 1 ? 2 : 4);
            float m0 = hd[7*64 + a0], m1 = hd[7*64 + a0 + 1];
#pragma unroll 4
            for (int k = 0; k < 64; k += 4) {
                const float x0 = h2t[0][lane][k], x1 = h2t[0][lane][k+1], x2 = h2t[0][lane][k+2], x3 = h2t[0][lane][k+3];
                const float4 w0 = *reinterpret_cast<const float4*>(&hd[a0*64 + k]);
                const float4 w1 = *reinterpret_cast<const float4*>(&hd[(a0 + 1)*64 + k]);
                m0 = __builtin_fmaf(w0.x, x0, m0); m0 = __builtin_fmaf(w0.y, x1, m0); m0 = __builtin_fmaf(w0.z, x2, m0); m0 = __builtin_fmaf(w0.w, x3, m0);
                m1 = __builtin_fmaf(w1.x, x0, m1); m1 = __builtin_fmaf(w1.y, x1, m1); m1 = __builtin_fmaf(w1.z, x2, m1); m1 = __builtin_fmaf(w1.w, x3, m1);
            }
            xmean[a0][lane] = m0; xmean[a0 + 1][lane] = m1;
        } else {
            float v = hd[7*64 + 12];
#pragma unroll 8
            for (int k = 0; k < 64; k++) v = __builtin_fmaf(hd[6*64 + k], h2t[1][lane][k], v);
            if (live) {
                if (io.value) io.value[env] = v;
                if (io.rollout_row) io.rollout_row[(size_t)env*(OD + 10) + OD + 8] = v;
            }
            float eps[8];
            if (io.noise) {
#pragma unroll
                for (int a = 0; a < 6; a++) eps[a] = live ? io.noise[(size_t)env*6 + a] : 0.0f;
            } else {
                policy_noise(env_id_offset + (uint32_t)env, step_counter, seed_lo, seed_hi, eps);
            }
#pragma unroll
            for (int a = 0; a < 6; a++) xn[a][lane] = eps[a];
        }
        __syncthreads();
        if (wave == 0 && live) {
            float lp = 0.0f;
#pragma unroll
            for (int a = 0; a < 6; a++) {
                const float ls = hd[7*64 + 6 + a], e = xn[a][lane];
                const float act = __builtin_fmaf(__builtin_expf(ls), e, xmean[a][lane]);
                lp += -0.5f*e*e - ls - 0.9189385332046727f;
                io.act_env[(size_t)env*6 + a] = tclamp(act, -1.0f, 1.0f);
                if (io.act_raw) io.act_raw[(size_t)env*6 + a] = act;
                if (io.rollout_row) io.rollout_row[(size_t)env*(OD + 10) + OD + a] = act;
            }
            if (io.logp) io.logp[env] = lp;
            if (io.rollout_row) io.rollout_row[(size_t)env*(OD + 10) + OD + 9] = lp;
        } else if (wave == 1 && live && io.rollout_row) {
#pragma unroll
            for (int k = 0; k < OD; k++) io.rollout_row[(size_t)env*(OD + 10) + k] = oxt[lane][k];
        }
    }
}

// Policy hidden layers on the matrix cores: per tower and layer, H[64 envs][64 units] = tanh(X[64][K] W^T + b) is a genuine
// contraction (K = 16 / 64).  v_mfma_f32_32x32x2_f32 is exact fp32 (a k-ordered fmaf chain, so results are bit-compatible
// with the VALU policy kernel).  4 waves = one per SIMD: wave w owns tower w>>1, env rows [32(w&1), +32) and both 32-unit
// column tiles; its B fragments (weights) stay in VGPRs for the whole launch, A fragments (activations) come from LDS in
// [env][unit] layout with a 65-float row stride (conflict-free for the A read, the D write and the per-lane head read).
// ROWS = 32: the workgroup serves at most 32 envs (p.epw <= 32: a batch of <= 8192 envs spread over all 256 CUs).  The policy phase then has ONE
// 32-row tile per tower, so wave w owns (tower w >> 1, COLUMN tile w & 1) instead of (tower, row tile) x both column tiles: 40 MFMAs per wave and
// step instead of 80, half the tanh / LDS traffic, 40 weight registers fewer.  Same k-ordered fmaf chains: results are bit-identical.
template <int KIND, int FL, int NW, int ROWS = 64>
__global__ void __launch_bounds__(64*NW) so100_rollout_fused(SimParams p, float* __restrict__ state, const float* start_tab,
                                                             float* obs_out, float* rew_out, uint8_t* done_out, uint8_t* trunc_out,
                                                             float* tobs_out, float* ep_ret_out, int32_t* ep_len_out,
                                                             PolicyWeights w, RolloutArgs ra) {
    static_assert(NW == 4, "one wave per SIMD: (tower, row tile) per wave");
    static_assert(ROWS == 64 || ROWS == 32, "envs per workgroup the policy phase is tiled for");
    constexpr bool HALF = ROWS == 32;
    constexpr int OD = obs_dim<KIND>();
    constexpr int ODP = (OD + 3) & ~3;                            // K of layer 1, padded with zero weights
    constexpr int LD = 65;                                        // LDS row stride of the [env][unit] activation images
    __shared__ __attribute__((aligned(16))) float hd[6*64 + 64 + 16];   // mu_w | v_w | mu_b(6) log_std(6) v_b(1)
    __shared__ float oxt[64][ODP + 1];                            // observation [env][k]
    // The hidden-activation images and the action means in ONE array: they are dead during the physics phase, when the bytes
    // carry the pad-contact records ([MAXC][CF][64] floats = 61 440 B) and the mass matrix for the contact wave ([21][64]).
    constexpr int HB = 2*2*64*LD;                                 // floats of the two activation images
    __shared__ float pool[HB + 6*64];
    float (*h1t)[64][LD] = reinterpret_cast<float (*)[64][LD]>(pool);
    float (*h2t)[64][LD] = reinterpret_cast<float (*)[64][LD]>(pool + 2*64*LD);
    float (*xmean)[64] = reinterpret_cast<float (*)[64]>(pool + HB);   // action means, two per wave (waves 0, 1, 3 -> wave 0)
    float (*xm)[64] = reinterpret_cast<float (*)[64]>(pool + MAXC*CF*64);
    static_assert(HB + 6*64 >= MAXC*CF*64 + 21*64, "contact records + mass matrix must fit under the policy phase's images");
    // Without pad/cube contacts (compile-time flags) the store never holds more than the MAXPADC pad/floor records: its tail carries the joint
    // frames from the contact wave's detection to its solve.  (Variants with F_PADS_CUBE have no room left: they rebuild the frames.)
    constexpr bool XW = FL >= 0 && (FL & (int)F_PADS_CUBE) == 0 && (FL & (int)F_PADS_FLOOR) != 0;
    static_assert(!XW || MAXPADC*CF*64 + 36*64 <= MAXC*CF*64, "joint frames must fit in the unused tail of the contact store");
    float (*xw)[64] = XW ? reinterpret_cast<float (*)[64]>(pool + MAXPADC*CF*64) : nullptr;
    constexpr bool PADS = FL < 0 || (FL & (int)F_ANY_CONTACT) != 0;
    constexpr bool LINKS = FL < 0 || (FL & (int)F_ANY_LINKS) != 0;
    __shared__ float xa[PADS ? 15 : 1][64];                       // pad contacts: the contact wave's accelerations, contact code, residual, set signature
    __shared__ float xk[PADS ? 12 : 1][64];                       //               ctrl and arm warm start of the env step
    __shared__ float xn[6][64];                                   // next step's policy noise, pre-drawn by wave 3 during the physics phase
    __shared__ float xq[PADS ? 24 : 18][64];                      // physics split: sin q, cos q, v (+ q) of env = lane (wave 0 -> waves 1, 3)
    __shared__ float xc[24][64];                                  //                cube state hand-over (wave 0 <-> wave 2)
    __shared__ float xb[6][64];                                   //                bias force          (wave 1 -> wave 0)
    __shared__ unsigned char pbuf[PADS ? MAXC*64 : 1];            // pad contacts: (feature hash | edge mask) per record of the last solve
    ContactMemo memo;                                             // ... and the rest of the contact wave's active-set memory
    if (FL >= 0) p.flags = (unsigned)FL;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // p.epw lanes of each wave own an env (64, or fewer to spread a small batch over all CUs); WHICH env a slot holds is the launcher's
    // choice (so100_balance.hpp deals contact-prone envs out over the workgroups): results do not depend on it
    const int slot = blockIdx.x*p.epw + lane;
    const int env = lane < p.epw ? (ra.slot_env ? ra.slot_env[slot] : (slot < p.n ? slot : -1)) : -1;
    const bool live = env >= 0;
    const int tower = wave >> 1, rt = wave & 1;
    const int lj = lane & 31, lh = lane >> 5;
    // ---- once per launch: weight fragments -> registers, head weights -> LDS, env state -> registers, obs -> LDS
    const float* W1 = tower ? w.vf_w0 : w.pi_w0; const float* W2 = tower ? w.vf_w1 : w.pi_w1;
    const float* Bi1 = tower ? w.vf_b0 : w.pi_b0; const float* Bi2 = tower ? w.vf_b1 : w.pi_b1;
    // Layer-2 B fragments: 64 VGPRs per lane, live through the whole physics phase.  The contact-disabled variant has the
    // registers to spare; the constrained variants do not (they spilled to scratch), so there the fragments live in 64 KB of
    // LDS [wave][column tile][k step][lane] and are read back right before each MFMA (measured: +3..4 % / -0.5 %).
    constexpr bool W2_LDS = FL != (int)F_CUBE_PINNED;
    static_assert(!HALF || !W2_LDS, "the 32-row tiling exists for the contact-disabled variant");
    constexpr int NCT = HALF ? 1 : 2;                             // column tiles per wave
    float bw1[NCT][ODP/2], bw2[NCT][W2_LDS ? 1 : 32], b1v[NCT], b2v[NCT];
    __shared__ float w2s[W2_LDS ? 4 : 1][2][32][W2_LDS ? 64 : 1];
#pragma unroll
    for (int ct = 0; ct < NCT; ct++) {
        const int unit = 32*(HALF ? rt : ct) + lj;
        b1v[ct] = Bi1[unit]; b2v[ct] = Bi2[unit];
#pragma unroll
        for (int s2 = 0; s2 < ODP/2; s2++) { const int k = 2*s2 + lh; bw1[ct][s2] = k < OD ? W1[unit*OD + k] : 0.0f; }
#pragma unroll
        for (int s2 = 0; s2 < 32; s2++) { if constexpr (W2_LDS) w2s[wave][ct][s2][lane] = W2[unit*64 + 2*s2 + lh]; else bw2[ct][s2] = W2[unit*64 + 2*s2 + lh]; }
    }
    if (wave == NW - 1) {
        for (int i = lane; i < 6*64; i += 64) hd[i] = w.mu_w[i];
        hd[6*64 + lane] = w.v_w[lane];
        if (lane < 6) { hd[7*64 + lane] = w.mu_b[lane]; hd[7*64 + 6 + lane] = w.log_std[lane]; }
        if (lane == 0) hd[7*64 + 12] = w.v_b[0];
    }
    EnvState e;
    if (wave == 0) {
        if (live) load_env_state<KIND, FL>(state, p.n, env, e); else idle_lane_state(e);
#pragma unroll
        for (int k = 0; k < ODP; k++) oxt[lane][k] = (live && k < OD) ? ra.obs_in[(size_t)env*OD + k] : 0.0f;
    }
    if (wave == 3 && ra.T > 0) {
        float eps[8];
        policy_noise(p.env_id_offset + (uint32_t)env, ra.step_counter0, p.seed_lo, p.seed_hi, eps);
#pragma unroll
        for (int a = 0; a < 6; a++) xn[a][lane] = eps[a];
    }
    __syncthreads();
    StepResult last{}; float last_obs[OD];
    StepCtx ctx{}; float ustep[8] = {}; float cstale[3] = {};
    SO100_PROF_DECL
#pragma unroll 1
    for (int t = 0; t < ra.T; t++) {
        SO100_PROF(7);
        // ---- layer 1: K = ODP
        if constexpr (HALF) {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; r++) acc[r] = b1v[0];
#pragma unroll
            for (int s2 = 0; s2 < ODP/2; s2++) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(oxt[lj][2*s2 + lh], bw1[0][s2], acc, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 16; r++) h1t[tower][(r & 3) + 8*(r >> 2) + 4*lh][32*rt + lj] = fast_tanh(acc[r]);
        } else {
            f32x16 acc0, acc1;
#pragma unroll
            for (int r = 0; r < 16; r++) { acc0[r] = b1v[0]; acc1[r] = b1v[1]; }
#pragma unroll
            for (int s2 = 0; s2 < ODP/2; s2++) {
                const float a = oxt[32*rt + lj][2*s2 + lh];
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bw1[0][s2], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bw1[1][s2], acc1, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int row = 32*rt + (r & 3) + 8*(r >> 2) + 4*lh;
                h1t[tower][row][lj] = fast_tanh(acc0[r]); h1t[tower][row][32 + lj] = fast_tanh(acc1[r]);
            }
        }
        __syncthreads();
        // ---- layer 2: K = 64
        if constexpr (HALF) {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; r++) acc[r] = b2v[0];
#pragma unroll
            for (int s2 = 0; s2 < 32; s2++) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(h1t[tower][lj][2*s2 + lh], bw2[0][s2], acc, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 16; r++) h2t[tower][(r & 3) + 8*(r >> 2) + 4*lh][32*rt + lj] = fast_tanh(acc[r]);
        } else {
            f32x16 acc0, acc1;
#pragma unroll
            for (int r = 0; r < 16; r++) { acc0[r] = b2v[0]; acc1[r] = b2v[1]; }
#pragma unroll
            for (int s2 = 0; s2 < 32; s2++) {
                const float a = h1t[tower][32*rt + lj][2*s2 + lh];
                float b0, b1;
                if constexpr (W2_LDS) { b0 = w2s[wave][0][s2][lane]; b1 = w2s[wave][1][s2][lane]; } else { b0 = bw2[0][s2]; b1 = bw2[1][s2]; }
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b0, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b1, acc1, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int row = 32*rt + (r & 3) + 8*(r >> 2) + 4*lh;
                h2t[tower][row][lj] = fast_tanh(acc0[r]); h2t[tower][row][32 + lj] = fast_tanh(acc1[r]);
            }
        }
        __syncthreads();
        SO100_PROF(0);                                             // policy hidden layers (MFMA + tanh + 2 barriers)
        float* row = ra.buf + ((size_t)t*p.n + (size_t)env)*(OD + 10);
        // ---- heads (VALU, lane = env): the 6 x 64 mean head is split two actions per wave over waves 0, 1, 3 and the
        //      value head runs on wave 2; each mean is the same k-ordered fmaf chain as in the stand-alone policy kernel
        if (wave != 2) {
            const int a0 = wave == 0 ? 0 : (wave == 1 ? 2 : 4);
            float m0 = hd[7*64 + a0], m1 = hd[7*64 + a0 + 1];
#pragma unroll 4
            for (int k = 0; k < 64; k += 4) {
                const float x0 = h2t[0][lane][k], x1 = h2t[0][lane][k+1], x2 = h2t[0][lane][k+2], x3 = h2t[0][lane][k+3];
                const float4 w0 = *reinterpret_cast<const float4*>(&hd[a0*64 + k]);
                const float4 w1 = *reinterpret_cast<const float4*>(&hd[(a0 + 1)*64 + k]);
                m0 = __builtin_fmaf(w0.x, x0, m0); m0 = __builtin_fmaf(w0.y, x1, m0); m0 = __builtin_fmaf(w0.z, x2, m0); m0 = __builtin_fmaf(w0.w, x3, m0);
                m1 = __builtin_fmaf(w1.x, x0, m1); m1 = __builtin_fmaf(w1.y, x1, m1); m1 = __builtin_fmaf(w1.z, x2, m1); m1 = __builtin_fmaf(w1.w, x3, m1);
            }
            xmean[a0][lane] = m0; xmean[a0 + 1][lane] = m1;
        } else {
            float v = hd[7*64 + 12];
#pragma unroll 8
            for (int k = 0; k < 64; k++) v = __builtin_fmaf(hd[6*64 + k], h2t[1][lane][k], v);
            if (live) row[OD + 8] = v;
        }
        __syncthreads();
        if (wave == 0) {
            // ---- sample -> clip -> rollout row; then the physics phase.  The noise was drawn by wave 3 a step ago.
            float mean[6], eps[6];
#pragma unroll
            for (int a = 0; a < 6; a++) { mean[a] = xmean[a][lane]; eps[a] = xn[a][lane]; }
            float act[6], lp = 0.0f;
#pragma unroll
            for (int a = 0; a < 6; a++) {
                const float ls = hd[7*64 + 6 + a];
                const float raw = __builtin_fmaf(__builtin_expf(ls), eps[a], mean[a]);
                lp += -0.5f*eps[a]*eps[a] - ls - 0.9189385332046727f;
                act[a] = live ? tclamp(raw, -1.0f, 1.0f) : 0.0f;
                if (live) row[OD + a] = raw;
            }
            if (live) {
#pragma unroll
                for (int k = 0; k < OD; k++) row[k] = oxt[lane][k];
                row[OD + 9] = lp;
            }
            draw8(p, p.env_id_offset + (uint32_t)env, (uint32_t)e.rngc, 0, nullptr, ustep);
            e.rngc++;
            env_step_pre<KIND>(e, act, ustep, p, ctx);
            SO100_PROF(1);                                         // head + noise + row + env_step_pre (wave 0)
        }
        // ---- physics phase, split over the waves: per substep wave 1 computes the RNEA bias force while wave 0
        //      computes the CRBA mass matrix and factorises it; wave 0 then solves, integrates and publishes q, v.
        //      Same operations in the same order as physics_substeps(): contact-free lanes agree with it to the last bit or two,
        //      lanes with pad contacts to the solver tolerance (same warm start and active-set memory, differently rounded inputs).
        {
            Arm<float> A;
                        const PhaseLds lds{ xq, xc, xb, pool, xa, xk, xm, pbuf, xw };
            physics_phase_mw<PADS, LINKS>(p, wave, lane, e, ctx.ctrl, cstale, A, lds, memo, prof_, [&](int sub) {
                if (wave == 3 && sub == 0 && t + 1 < ra.T) {       // wave 0 has consumed xn before this barrier
                    float eps[8];
                    policy_noise(p.env_id_offset + (uint32_t)env, ra.step_counter0 + (uint32_t)(t + 1), p.seed_lo, p.seed_hi, eps);
#pragma unroll
                    for (int a = 0; a < 6; a++) xn[a][lane] = eps[a];
                }
            }, [&]() {                                             // contact wave: wave 0's step-scope state is not ours to keep alive
                ctx = StepCtx{}; last = StepResult{};
#pragma unroll
                for (int k = 0; k < 8; k++) ustep[k] = 0.0f;
#pragma unroll
                for (int k = 0; k < OD; k++) last_obs[k] = 0.0f;
            });
            if (wave == 0) {
                e.nsub += p.frame_skip;
                TaskPoses<float> P;
                task_poses<float>(A.s, A.c, !reach_kind<KIND>(), P);
                float obs[OD], tobs[OD]; bool term;
                const float reward = env_step_post<KIND>(e, ctx, ustep, P, cstale, obs, term);
                const StepResult r = env_step_finish<KIND>(e, reward, term, p, p.env_id_offset + (uint32_t)env, nullptr, start_tab, obs, tobs);
#pragma unroll
                for (int k = 0; k < OD; k++) { oxt[lane][k] = obs[k]; last_obs[k] = obs[k]; }
                last = r;
                if (live) {
                    row[OD + 6] = r.reward; row[OD + 7] = r.done ? (r.trunc_only ? 2.0f : 1.0f) : 0.0f;
                    if (r.done) {
                        if (ra.tobs_chunk) {
#pragma unroll
                            for (int k = 0; k < OD; k++) ra.tobs_chunk[((size_t)t*p.n + (size_t)env)*OD + k] = tobs[k];
                        }
                        if (tobs_out) {
#pragma unroll
                            for (int k = 0; k < OD; k++) tobs_out[(size_t)env*OD + k] = tobs[k];
                        }
                        if (ep_ret_out) ep_ret_out[env] = r.ep_return;
                        if (ep_len_out) ep_len_out[env] = r.ep_length;
                    }
                }
            }
        }
        __syncthreads();
    }
    SO100_PROF_FLUSH(12*wave);
    {
        const int np = 64/p.epw, e3 = blockIdx.x*p.epw + lane/np;      // the contact wave's lanes are grouped per env
        if (wave == 3) prof_.flush_wg(wave, lane, e3, (lane % np) == 0 && e3 < p.n); else prof_.flush_wg(wave, lane, env, live);
    }
    if (wave == 0 && live) {
        if constexpr (PADS) e.cload = ((e.cload & 0xFFFF) + (e.cload >> 16)) >> 1;       // contact load: average of its history and this launch's count
        store_env_state<KIND, FL>(state, p.n, env, e);
        if (ra.T > 0) {
#pragma unroll
            for (int k = 0; k < OD; k++) obs_out[(size_t)env*OD + k] = last_obs[k];
            rew_out[env] = last.reward; done_out[env] = last.done ? 1 : 0; trunc_out[env] = last.trunc_only ? 1 : 0;
        }
    }
}

}  // namespace so100
