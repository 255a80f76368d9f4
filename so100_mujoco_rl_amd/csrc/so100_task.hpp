// so100_task.hpp -- the reference's task layer (reward / obs / ctrl / reset / curriculum / reprojection)
// for one env held in registers, fp32.  Each function cites the reference lines it restates
// ("ref:" = /root/reference/src/so100_mujoco_rl/envs/).  Semantics that parity depends on (SURVEY.md
// section 8a quirks Q1-Q7) are reproduced on purpose:
//   Q1 stale kinematics: poses read by obs/reward are those of the START of the last substep,
//   Q2 the reward of step t is computed from the state left by step t-1,
//   Q3 ctrl is relative to the measured angle (Env01/02) or to the commanded angle (Env03-05),
//   Q4 Env05 step observations scale the image centre by 5, reset observations do not,
//   after reset all poses are zero (mj_resetData) until the first step.
#pragma once
#include "so100_physics.hpp"
#include "so100_cube.hpp"
#include "so100_contact.hpp"
#include <stdint.h>

namespace so100 {

struct SimParams {
    int32_t  n;
    uint32_t flags;
    int32_t  solver_iters, contact_iters, frame_skip, max_episode_steps;
    uint32_t seed_lo, seed_hi, env_id_offset;
    int32_t  epw;                // envs per workgroup of the multi-wave kernels (64, 32 or 16 lanes of each wave in use)
    int32_t  mw_max;             // so100_step: batches up to this size take the 4-wave kernel so100_step_mw, larger ones so100_step_fused
};

// State of a lane that owns no env (tail of the last workgroup, or lanes >= epw): it must not make its wave wait.  An all-zero
// state would: the cube sits 1 cm inside the floor (a full Newton solve every substep) and the stretched arm's pads touch it.
// Arm folded up 28 cm above the floor, cube floating (anti-gravity), everything at rest; its action is forced to zero.

// ---- per-env persistent state ---------------------------------------------------------------------
// bits of EnvState::bits
enum : int { B_HAS_PREV = 1, B_HAVE_BLOCK = 2, B_HAVE_LAST_BLOCK = 4, B_HAVE_CENTER = 8, B_HAVE_ANGVEL = 16,
             B_BLOCK_UPDATED = 32, B_ANTIGRAV = 64, B_BAD_STATE = 128 /* latched: a non-finite state ended an episode */,
             B_BAD_ACTION = 256 /* this step's action was non-finite (consumed by env_step_finish) */ };

struct EnvState {
    float q[6];                 // arm joint angles                      (qpos[0:6])
    Cube<float> cube;           // pos, quat, vel, Newton warm start     (qpos[6:13], qvel[6:12])
    float v[6];                 // arm joint velocities                  (qvel[0:6])
    float ff[6], fl[6];         // friction-loss / limit row forces (warm start)
    float qc[6];                // Kahan compensation of the joint-angle integration
    float ee[3], wrist_z, cx[3];// stale end effector, wrist height, cube xpos (Q1)
    int   nsub, elapsed, bits, rngc;
    float epret; int eplen;
    float bp[3], lbp[3];        // Env02 sampled block positions (ref: env02_v1.py:64-68)
    float cmd[6];               // Env03-05 commanded angles (ref: env_base_02.py:85-86)
    float lc[2]; int lost;      // last detected centre, lost counter (ref: env03_v1.py:152-164)
    float tgt[3], tdt, ttime;   // cube target, dwell, time of last retarget (ref: env03_v1.py:77-93)
    float av[6];                // last "angular velocities" (ref: env_base_01.py:165-178)
    float res;                  // largest constraint-solver residual of the last env step: the acceleration change of the last
                                // block-PGS sweep, or the last Newton step of the contact solve when it ran out of iterations
    float aw[6];                // arm qacc of the previous substep: warm start of the pad-contact Newton (so100_contact.hpp)
    int   cstat;                // pad contacts: most contacts in a substep of the last env step | (dropped over the budget) << 8
    int   csig;                 // signature of the pad-contact SET of the last substep (so100_contact.hpp: contact_signature; 0 = no contact)
    int   cload;                // contact load: running average (half-life one launch) of the substeps per persistent-rollout launch this env spent in
                                // contact -- what so100_balance.hpp deals the envs out over the workgroups by; 0 after a reset
};

SO100_HD void idle_lane_state(EnvState& e) {
    e = EnvState{};
    const float q[6] = { 0.0f, -1.9f, 1.6f, 0.3f, 1.5708f, 0.1f };
#pragma unroll
    for (int i = 0; i < 6; i++) e.q[i] = q[i];
    e.cube.pos[0] = 0.3f; e.cube.pos[1] = 0.3f; e.cube.pos[2] = 0.5f; e.cube.quat[0] = 1.0f;
    e.bits = B_ANTIGRAV;
}

// The [field][N] state matrix.  X(name, member, kind, group): kind f = float, i = int32;
// group 0 = all env kinds, 1 = Env01/02/06 (reach family), 2 = Env02/06 (block memory), 3 = Env03-05 (look-at family),
// 4 = any constraint rows simulated (flags & (friction | limits | pads)), 5 = finger-pad contacts simulated (flags & pads).
#define SO100_STATE_FIELDS(X) \
    X(q0, q[0], f, 0) X(q1, q[1], f, 0) X(q2, q[2], f, 0) X(q3, q[3], f, 0) X(q4, q[4], f, 0) X(q5, q[5], f, 0) \
    X(cube_x, cube.pos[0], f, 0) X(cube_y, cube.pos[1], f, 0) X(cube_z, cube.pos[2], f, 0) \
    X(cube_qw, cube.quat[0], f, 0) X(cube_qx, cube.quat[1], f, 0) X(cube_qy, cube.quat[2], f, 0) X(cube_qz, cube.quat[3], f, 0) \
    X(v0, v[0], f, 0) X(v1, v[1], f, 0) X(v2, v[2], f, 0) X(v3, v[3], f, 0) X(v4, v[4], f, 0) X(v5, v[5], f, 0) \
    X(cube_vx, cube.vel[0], f, 0) X(cube_vy, cube.vel[1], f, 0) X(cube_vz, cube.vel[2], f, 0) \
    X(cube_wx, cube.vel[3], f, 0) X(cube_wy, cube.vel[4], f, 0) X(cube_wz, cube.vel[5], f, 0) \
    X(ff0, ff[0], f, 0) X(ff1, ff[1], f, 0) X(ff2, ff[2], f, 0) X(ff3, ff[3], f, 0) X(ff4, ff[4], f, 0) X(ff5, ff[5], f, 0) \
    X(fl0, fl[0], f, 0) X(fl1, fl[1], f, 0) X(fl2, fl[2], f, 0) X(fl3, fl[3], f, 0) X(fl4, fl[4], f, 0) X(fl5, fl[5], f, 0) \
    X(cw0, cube.warm[0], f, 0) X(cw1, cube.warm[1], f, 0) X(cw2, cube.warm[2], f, 0) \
    X(cw3, cube.warm[3], f, 0) X(cw4, cube.warm[4], f, 0) X(cw5, cube.warm[5], f, 0) \
    X(qc0, qc[0], f, 0) X(qc1, qc[1], f, 0) X(qc2, qc[2], f, 0) X(qc3, qc[3], f, 0) X(qc4, qc[4], f, 0) X(qc5, qc[5], f, 0) \
    X(substeps, nsub, i, 0) X(elapsed_steps, elapsed, i, 0) X(bits, bits, i, 0) X(rng_counter, rngc, i, 0) \
    X(episode_return, epret, f, 0) X(episode_length, eplen, i, 0) \
    X(ee_x, ee[0], f, 1) X(ee_y, ee[1], f, 1) X(ee_z, ee[2], f, 1) X(wrist_z, wrist_z, f, 1) \
    X(cx_x, cx[0], f, 1) X(cx_y, cx[1], f, 1) X(cx_z, cx[2], f, 1) \
    X(bp_x, bp[0], f, 2) X(bp_y, bp[1], f, 2) X(bp_z, bp[2], f, 2) X(lbp_x, lbp[0], f, 2) X(lbp_y, lbp[1], f, 2) X(lbp_z, lbp[2], f, 2) \
    X(cmd0, cmd[0], f, 3) X(cmd1, cmd[1], f, 3) X(cmd2, cmd[2], f, 3) X(cmd3, cmd[3], f, 3) X(cmd4, cmd[4], f, 3) X(cmd5, cmd[5], f, 3) \
    X(lc_x, lc[0], f, 3) X(lc_y, lc[1], f, 3) X(lost_count, lost, i, 3) \
    X(tgt_x, tgt[0], f, 3) X(tgt_y, tgt[1], f, 3) X(tgt_z, tgt[2], f, 3) X(target_dt, tdt, f, 3) X(target_time, ttime, f, 3) \
    X(av0, av[0], f, 3) X(av1, av[1], f, 3) X(av2, av[2], f, 3) X(av3, av[3], f, 3) X(av4, av[4], f, 3) X(av5, av[5], f, 3) \
    X(solver_residual, res, f, 4) \
    X(aw0, aw[0], f, 5) X(aw1, aw[1], f, 5) X(aw2, aw[2], f, 5) X(aw3, aw[3], f, 5) X(aw4, aw[4], f, 5) X(aw5, aw[5], f, 5) \
    X(contact_stat, cstat, i, 5) X(contact_sig, csig, i, 5) X(contact_load, cload, i, 5)

enum StateField : int {
#define X(name, member, kind, group) SF_##name,
    SO100_STATE_FIELDS(X)
#undef X
    SF_COUNT
};
constexpr int SF_QPOS0 = SF_q0, SF_QVEL0 = SF_v0;       // 13 qpos rows then 12 qvel rows, contiguous

template <int KIND> SO100_HD constexpr bool reach_kind() { return KIND <= 2 || KIND == 6; }    // obs 15, reward env_base_01/06
template <int KIND> SO100_HD constexpr bool block_kind() { return KIND == 2 || KIND == 6; }    // block_pos / last_block_pos memory
// FL: the kernel's compile-time physics flags (-1 = decided at run time: every row is kept)
template <int KIND, int FL = -1> SO100_HD constexpr bool uses_group(int g) {
    return g == 0 || (g == 1 && reach_kind<KIND>()) || (g == 2 && block_kind<KIND>()) || (g == 3 && !reach_kind<KIND>())
        || (g == 4 && (FL < 0 || (FL & (int)(F_FRICTIONLOSS | F_LIMITS | F_ANY_CONTACT)) != 0))
        || (g == 5 && (FL < 0 || (FL & (int)F_ANY_CONTACT) != 0));
}
template <int KIND> SO100_HD constexpr int obs_dim() { return reach_kind<KIND>() ? 15 : 8; }

// ---- constants of the task layer --------------------------------------------------------------------
#define SO100_PI_F 3.14159265358979323846f
static constexpr float JOINT_STEP_SCALE = 0.075f;                                  // ref: utils.py:9
static constexpr float REST_POSITION[6]  = { 0.0f, -3.141f, 3.117f, 1.0f, 0.0f, 0.0f };   // ref: utils.py:11
static constexpr float START_POSITION[6] = { 0.0f, -2.04f, 1.19f, 1.5f, -1.58f, 0.5f };   // ref: env03_v1.py:10
static constexpr float SPACE_START[2][3]  = { {-0.05f, -0.4f, 0.01f}, {0.05f, -0.3f, 0.01f} };   // ref: env03_v1.py:13-16
static constexpr float SPACE_END_03[2][3] = { {-0.35f, -0.45f, 0.01f}, {0.35f, -0.25f, 0.01f} }; // ref: env03_v1.py:17-20
static constexpr float SPACE_END_05[2][3] = { {-0.45f, -0.45f, 0.01f}, {0.45f, -0.25f, 0.5f} };  // ref: env05_v1.py:17-20
// VALID_START_POSITIONS (ref: utils.py:13-50, 36 recorded poses) is indexed at run time, so it is passed as a
// pointer to a [36][6] float table (device memory in the kernel; csrc/so100_start_positions.inc on the host).

// ---- Philox4x32-10, same stream as oracle/so100_oracle.c ---------------------------------------------
SO100_HD uint32_t mulhi32(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a * b) >> 32); }
SO100_HD void philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t out[4]) {
#pragma unroll
    for (int r = 0; r < 10; r++) {
        const uint32_t h0 = mulhi32(0xD2511F53u, c0), l0 = 0xD2511F53u * c0;
        const uint32_t h1 = mulhi32(0xCD9E8D57u, c2), l1 = 0xCD9E8D57u * c2;
        const uint32_t n0 = h1 ^ c1 ^ k0, n1 = l1, n2 = h0 ^ c3 ^ k1, n3 = l0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
SO100_HD void draw8(const SimParams& p, uint32_t env_gid, uint32_t counter, int phase, const float* inject, float u[8]) {
    if (inject) {
#pragma unroll
        for (int i = 0; i < 8; i++) u[i] = inject[8*phase + i];
        return;
    }
#pragma unroll
    for (int b = 0; b < 2; b++) {
        uint32_t r[4];
        philox4x32(env_gid, counter, (uint32_t)(2*phase + b), 0u, p.seed_lo, p.seed_hi, r);
#pragma unroll
        for (int i = 0; i < 4; i++) u[4*b + i] = (float)(r[i] >> 8) * (1.0f/16777216.0f);
    }
}

// ---- pure task functions -----------------------------------------------------------------------------
SO100_HD float joint_penalty(float a, float lo, float hi) {                // ref: env_base_01.py:153-163
    const float lt = lo + 0.05f*(hi - lo), ut = hi - 0.05f*(hi - lo);
    float pen = 0.0f;
    if (a < lt) pen -= (lt - a)*10.0f;
    else if (a > ut) pen -= (a - ut)*10.0f;
    return pen;
}
SO100_HD float joint_reward(const float q[6]) {                            // ref: env_base_01.py:144-151
    float r = 0.0f;
#pragma unroll
    for (int i = 0; i < 6; i++) r += joint_penalty(q[i], (float)so100g::JNT_RANGE[i][0], (float)so100g::JNT_RANGE[i][1]);
    return r;
}
SO100_HD float reward_base(const float q[6], const float block[3], const float ee[3], float wrist_z, bool has_prev) {
    // ref: env_base_01.py:180-239
    float reward = 0.0f;
    const float dx = block[0] - ee[0], dy = block[1] - ee[1], dz = block[2] - ee[2];
    const float distance = tsqrt(dx*dx + dy*dy + dz*dz);
    if (block[1] < -0.1f) {
        const float pitch = q[1];
        if (has_prev && pitch < -0.7f*SO100_PI_F) reward += (pitch + 0.7f*SO100_PI_F)*0.7f;
    }
    if (has_prev && ee[2] < 0.02f) reward += (ee[2] - 0.02f)*20.0f;
    if (has_prev && wrist_z < 0.08f) reward += tclamp((wrist_z - 0.08f)*10.0f, -0.8f, 0.8f);
    reward += tmin(-distance + 0.02f, 0.0f)*0.5f;
    reward += joint_reward(q);
    return reward;
}
// pinhole reprojection of the cube into the end-point camera; ref: env_base_02.py:88-127.
// Returns false for "None" (NaN or outside the 1080 x 1920 frame); no z-sign test, like the reference.
SO100_HD bool project(const float cam_pos[3], const float cam_mat[9], const float p[3], int& u_out, int& v_out) {
    const float r0 = p[0] - cam_pos[0], r1 = p[1] - cam_pos[1], r2 = p[2] - cam_pos[2];
    const float x = cam_mat[0]*r0 + cam_mat[3]*r1 + cam_mat[6]*r2;
    const float y = cam_mat[1]*r0 + cam_mat[4]*r1 + cam_mat[7]*r2;
    const float z = cam_mat[2]*r0 + cam_mat[5]*r1 + cam_mat[8]*r2;
    const float f = 554.25625842204073f;                       // 0.5 * 1920 / tan(120 deg / 2)
    const float u = f*x/z + 540.0f, v = f*y/z + 960.0f;
    if (u != u || v != v) return false;
    if (!(tabs(u) < 2.0e9f) || !(tabs(v) < 2.0e9f)) return false;   // int(inf) raises in Python: treated as None
    const int iu = (int)u, iv = (int)v;                        // truncation toward zero, like int()
    if (iu < 0 || iu >= 1080 || iv < 0 || iv >= 1920) return false;
    u_out = 1080 - iu; v_out = 1920 - iv;
    return true;
}

// 8-corner bounding box of the cube in the image; ref: env_base_02.py:129-176 (get_projected_cube_bounding_box, unused
// upstream).  Returns false for None (fewer than two corners project into the frame); centre with YOLO's integer arithmetic
// (env_base_02.py:206-209).  Env03 / Env04 use it in place of render + YOLO.
SO100_HD bool project_bbox_center(const float cam_pos[3], const float cam_mat[9], const float p[3], int& cx, int& cy) {
    const float d = 0.01f;
    int n = 0, x0 = 0, x1 = 0, y0 = 0, y1 = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const float c[3] = { (k & 4) ? p[0] + d : p[0] - d, (k & 2) ? p[1] + d : p[1] - d, (k & 1) ? p[2] + d : p[2] - d };
        int u, v;
        if (project(cam_pos, cam_mat, c, u, v)) {
            if (n == 0) { x0 = x1 = u; y0 = y1 = v; }
            else { x0 = u < x0 ? u : x0; x1 = u > x1 ? u : x1; y0 = v < y0 ? v : y0; y1 = v > y1 ? v : y1; }
            n++;
        }
    }
    cx = (x0 + x1)/2; cy = (y0 + y1)/2;
    return n >= 2;
}

SO100_HD void set_random_block_position(EnvState& e, bool remember, float dlo, const float u[8]) {
    // ref: env01_v1.py:45-52 (dlo 0.18), env02_v1.py:52-68 and env06_v1.py:52-69 (dlo 0.22, remember); u[1] is the discarded draw
    const float dist = dlo + (0.42f - dlo)*u[0];
    const float theta = -0.5f*SO100_PI_F + (-0.25f*SO100_PI_F + (0.5f*SO100_PI_F)*u[2]);
    float s, c; tsincos<float>(theta, s, c);
    const float p[3] = { dist*c, dist*s, 0.0f };
    e.cube.pos[0] = p[0]; e.cube.pos[1] = p[1]; e.cube.pos[2] = p[2];
    if (remember) {
        if (!(e.bits & B_HAVE_LAST_BLOCK)) { e.lbp[0] = p[0]; e.lbp[1] = p[1]; e.lbp[2] = p[2]; e.bits |= B_HAVE_LAST_BLOCK; }
        else { e.lbp[0] = e.bp[0]; e.lbp[1] = e.bp[1]; e.lbp[2] = e.bp[2]; }
        e.bp[0] = p[0]; e.bp[1] = p[1]; e.bp[2] = p[2]; e.bits |= B_HAVE_BLOCK;
    }
}

template <int KIND> SO100_HD void set_initial_values_03(EnvState& e) {     // ref: env03_v1.py:35-57, env04_v1.py:25-46
#pragma unroll
    for (int i = 0; i < 6; i++) e.cmd[i] = START_POSITION[i];
    if (KIND == 4) e.bits |= B_HAVE_CENTER; else e.bits &= ~B_HAVE_CENTER;
    e.lc[0] = e.lc[1] = -1.0f; e.lost = 0;
#pragma unroll
    for (int i = 0; i < 3; i++) e.tgt[i] = (SPACE_START[0][i] + SPACE_START[1][i]) / 2;
    e.tdt = 0.01f; e.ttime = 0.0f;
    e.bits &= ~B_BLOCK_UPDATED;
}

// first-time initialisation of a fresh handle (EnvNN.__init__)
template <int KIND> SO100_HD void env_init(EnvState& e) {
    e = EnvState{};
    e.cube.quat[0] = 1.0f;
    if (!reach_kind<KIND>()) {                                             // ref: env_base_02.py:32,51
        set_initial_values_03<KIND>(e);
        e.cube.pos[0] = e.tgt[0]; e.cube.pos[1] = e.tgt[1]; e.cube.pos[2] = e.tgt[2];
    }
}

// MujocoEnv.reset -> mj_resetData -> reset_model(); u = reset-phase uniforms
template <int KIND> SO100_HD void env_reset(EnvState& e, const float u[8], const float* start_tab, float* obs) {
    // mj_resetData: qpos = qpos0, everything else (velocities, warm starts, applied forces, time, POSES) zero
#pragma unroll
    for (int i = 0; i < 6; i++) { e.q[i] = 0.0f; e.v[i] = 0.0f; e.qc[i] = 0.0f; e.ff[i] = 0.0f; e.fl[i] = 0.0f; e.cube.vel[i] = 0.0f; e.cube.warm[i] = 0.0f; e.aw[i] = 0.0f; }
    e.res = 0.0f; e.cstat = 0; e.csig = 0; e.cload = 0;
    e.cube.pos[0] = e.cube.pos[1] = e.cube.pos[2] = 0.0f;
    e.cube.quat[0] = 1.0f; e.cube.quat[1] = e.cube.quat[2] = e.cube.quat[3] = 0.0f;
    e.ee[0] = e.ee[1] = e.ee[2] = 0.0f; e.wrist_z = 0.0f; e.cx[0] = e.cx[1] = e.cx[2] = 0.0f;
    e.nsub = 0; e.bits &= ~B_ANTIGRAV;
    e.elapsed = 0; e.epret = 0.0f; e.eplen = 0;
    if (KIND == 1) {                                                       // ref: env01_v1.py:39-63
        set_random_block_position(e, false, 0.18f, u);
        int idx = (int)(u[3]*36.0f); idx = idx > 35 ? 35 : idx;
#pragma unroll
        for (int i = 0; i < 5; i++) e.q[i] = start_tab[6*idx + i];                             // Jaw skipped (:58-59)
    } else if (block_kind<KIND>()) {                                       // ref: env02_v1.py:70-81, env06_v1.py:71-82
        set_random_block_position(e, true, 0.22f, u);
#pragma unroll
        for (int i = 0; i < 6; i++) e.q[i] = REST_POSITION[i];
    } else {                                                               // ref: env03_v1.py:203-215
        set_initial_values_03<KIND>(e);
        e.cube.pos[0] = e.tgt[0]; e.cube.pos[1] = e.tgt[1]; e.cube.pos[2] = e.tgt[2];
#pragma unroll
        for (int i = 0; i < 6; i++) e.q[i] = START_POSITION[i];
    }
    if (reach_kind<KIND>()) {                                              // ref: env_base_01.py:241-270, all poses zero (Q1)
#pragma unroll
        for (int i = 0; i < 6; i++) obs[i] = e.q[i];
#pragma unroll
        for (int i = 6; i < 15; i++) obs[i] = 0.0f;
    } else {                                                               // ref: env05_v1.py:32-75: camera pose zero => NaN => None
#pragma unroll
        for (int i = 0; i < 6; i++) obs[i] = e.cmd[i];
        obs[6] = -1.0f; obs[7] = -1.0f;
    }
}

// 16 physics substeps; leaves the stale poses of the LAST substep in P / cube_stale
SO100_HD void physics_substeps(EnvState& e, const float ctrl[6], const SimParams& p, bool want_cam,
                               TaskPoses<float>& P, float cube_stale[3]) {
    Arm<float> A;
    float dq[6] = { 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f };
    const float applied[3] = { 0.0f, 0.0f, (e.bits & B_ANTIGRAV) ? (float)(so100g::CUBE_MASS*so100g::GRAVITY) : 0.0f };
    const bool pads = (p.flags & F_ANY_CONTACT) != 0u;
    e.res = 0.0f; e.cstat = 0; e.csig = 0;
    ContactsPriv<float> cs; int zones = -1;                  // the contact solve's active-set memory lives for this env step
#pragma unroll 1
    for (int s = 0; s < p.frame_skip; s++) {
        cube_stale[0] = e.cube.pos[0]; cube_stale[1] = e.cube.pos[1]; cube_stale[2] = e.cube.pos[2];
        if (pads) {
            int st[4];
            substep_with_pads<float>(e.q, e.v, e.qc, ctrl, e.ff, e.fl, e.aw, e.cube, applied, p.flags, p.solver_iters, p.contact_iters, A, s == 0, dq, &e.res, cs, zones, st);
            const int n = e.cstat & 255, dr = e.cstat >> 8;
            e.cstat = (st[0] > n ? st[0] : n) | ((dr + st[2] > 0xFFFF ? 0xFFFF : dr + st[2]) << 8);
            e.csig = st[3];
        } else {
            arm_substep<float>(e.q, e.v, e.qc, ctrl, e.ff, e.fl, p.flags, p.solver_iters, A, s == 0, dq, &e.res);
            cube_substep<float>(e.cube, applied, p.flags, p.contact_iters);
        }
    }
    e.nsub += p.frame_skip;
    task_poses<float>(A.s, A.c, want_cam, P);      // sin/cos of the angles the last substep STARTED from
}

// One EnvNN.step, in three stages around the 16 physics substeps (the stepwise kernel runs them back to back; the
// persistent rollout kernel runs the substeps on several waves in between):
//   env_step_pre  : reward of the previous state (Q2), ctrl, Env02 reach branch / Env03-05 curriculum
//   physics       : physics_substeps (or its multi-wave version)
//   env_step_post : stale poses -> obs, look-at reward, termination
struct StepCtx {
    float reward;
    float ctrl[6];
    float old[6], ncmd[6], frac, smin[3], smax[3];       // Env03-05
};

template <int KIND> SO100_HD void env_step_pre(EnvState& e, const float a_in[6], const float u[8], const SimParams& p, StepCtx& c) {
    // a non-finite action is an error of the caller: the step runs with a zero action and env_step_finish ends the episode
    float a[6], az = 0.0f;
#pragma unroll
    for (int i = 0; i < 6; i++) az += a_in[i]*0.0f;
    const bool abad = !(az == 0.0f);
    if (abad) e.bits |= B_BAD_ACTION;
#pragma unroll
    for (int i = 0; i < 6; i++) a[i] = abad ? 0.0f : a_in[i];
    if (reach_kind<KIND>()) {
        // ref: env01_v1.py:15-37 / env02_v1.py:18-50 / env06_v1.py:18-50
        c.reward = reward_base(e.q, e.cx, e.ee, e.wrist_z, (e.bits & B_HAS_PREV) != 0);
        e.bits |= B_HAS_PREV;
#pragma unroll
        for (int i = 0; i < 6; i++) c.ctrl[i] = e.q[i] + a[i]*JOINT_STEP_SCALE;
        if (block_kind<KIND>()) {
            const float dx = e.cx[0] - e.ee[0], dy = e.cx[1] - e.ee[1], dz = e.cx[2] - e.ee[2];
            if (tsqrt(dx*dx + dy*dy + dz*dz) < 0.03f) {
                if (KIND == 6) {                                           // gripper term, ref: env_base_06.py:149-162, 253-256
                    const float jn = tmin(tmax((e.q[5] + 0.2f)*(1.0f/2.2f), 0.0f), 1.0f);
                    c.reward += 100.0f/(1.0f + texp(-10.0f*(jn - 0.3f)));
                }
                const float bx = e.bp[0] - e.lbp[0], by = e.bp[1] - e.lbp[1], bz = e.bp[2] - e.lbp[2];
                c.reward += tsqrt(bx*bx + by*by + bz*bz)*20.0f;
                if (KIND == 2) set_random_block_position(e, true, 0.22f, u);   // Env06 keeps the cube (env06_v1.py:36)
            }
        }
    } else {
        // ref: env03_v1.py:124-201 (Env03, Env05) / env04_v1.py:62-160 (Env04)
        const float h = (float)so100g::TIMESTEP;
        const float time = (float)e.nsub*h;
        c.frac = tmin(time/12.0f, 1.0f);
#pragma unroll
        for (int i = 0; i < 3; i++) { c.smin[i] = SPACE_START[0][i]; c.smax[i] = SPACE_START[1][i]; }
        if (KIND != 4) {
#pragma unroll
            for (int i = 0; i < 3; i++) {                                  // _update_block_space :59-68
                const float e0 = KIND == 5 ? SPACE_END_05[0][i] : SPACE_END_03[0][i];
                const float e1 = KIND == 5 ? SPACE_END_05[1][i] : SPACE_END_03[1][i];
                c.smin[i] = SPACE_START[0][i] + c.frac*(e0 - SPACE_START[0][i]);
                c.smax[i] = SPACE_START[1][i] + c.frac*(e1 - SPACE_START[1][i]);
            }
            const float speed = c.frac <= 0.05f ? 0.0f : (c.frac - 0.05f)*2.0f/(1.0f - 0.05f);   // :70-75
            {                                                              // _update_block_target :77-93
                const float tx = e.tgt[0] - e.cube.pos[0], ty = e.tgt[1] - e.cube.pos[1], tz = e.tgt[2] - e.cube.pos[2];
                const float dist_t = tsqrt(tx*tx + ty*ty + tz*tz);
                if (!(time - e.ttime < e.tdt && dist_t > 0.02f)) {
#pragma unroll
                    for (int i = 0; i < 3; i++) e.tgt[i] = c.smin[i] + (c.smax[i] - c.smin[i])*u[i];
                    e.tdt = 1.2f + (5.1f - 1.2f)*u[3];
                    e.ttime = time;
                }
            }
            {                                                              // _update_block_position :95-122
                const float tx = e.tgt[0] - e.cube.pos[0], ty = e.tgt[1] - e.cube.pos[1], tz = e.tgt[2] - e.cube.pos[2];
                const float dist = tsqrt(tx*tx + ty*ty + tz*tz);
                if (dist > 0.0f) {
                    const float sd = tmin(speed*h, dist), inv = trcp(dist);
                    e.cube.pos[0] += tx*inv*sd; e.cube.pos[1] += ty*inv*sd; e.cube.pos[2] += tz*inv*sd;
                    e.cube.vel[0] = e.cube.vel[1] = e.cube.vel[2] = 0.0f;
                    e.bits |= B_ANTIGRAV;                                  // qfrc_applied = -m g, kept until reset
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 6; i++) { c.old[i] = e.cmd[i]; c.ncmd[i] = e.cmd[i] + a[i]*JOINT_STEP_SCALE; c.ctrl[i] = c.ncmd[i]; }
        c.reward = 0.0f;
    }
}

template <int KIND> SO100_HD float env_step_post(EnvState& e, const StepCtx& c, const float u[8], const TaskPoses<float>& P,
                                                  const float cstale[3], float* obs, bool& term) {
    term = false;
    float reward;
    if (reach_kind<KIND>()) {
        reward = c.reward;
        // stale poses -> persistent (read by the next step's reward) and -> obs; ref: env_base_01.py:118-127, 241-270
#pragma unroll
        for (int i = 0; i < 3; i++) { e.ee[i] = P.jaw_pos[i] + P.jaw_mat[3*i + 1]*(-0.1f); e.cx[i] = cstale[i]; }
        e.wrist_z = P.wrist[2];
#pragma unroll
        for (int i = 0; i < 6; i++) obs[i] = e.q[i];
#pragma unroll
        for (int i = 0; i < 3; i++) { obs[6 + i] = e.cx[i] - e.ee[i]; obs[9 + i] = e.cx[i]; obs[12 + i] = e.ee[i]; }
    } else {
        const float h = (float)so100g::TIMESTEP;
        // ref: env05_v1.py:32-75 (Env05: centre reprojection + noise); Env03/04: the bounding box of the projected cube
        // corners (env_base_02.py:129-176) with YOLO's centre arithmetic stands in for render + YOLO (env_base_02.py:178-222)
        float cxn = -1.0f, cyn = -1.0f; int pu, pv;
        if (KIND == 5 ? project(P.cam_pos, P.cam_mat, e.cube.pos, pu, pv) : project_bbox_center(P.cam_pos, P.cam_mat, e.cube.pos, pu, pv)) {
            cxn = (float)pu/1080.0f; cyn = (float)pv/1920.0f;
            if (KIND == 5) { cxn += -0.05f + 0.1f*u[4]; cyn += -0.05f + 0.1f*u[5]; }
        }
#pragma unroll
        for (int i = 0; i < 6; i++) obs[i] = c.old[i];
        obs[6] = cxn; obs[7] = cyn;
        if (cxn == -1.0f && cyn == -1.0f) {                                // :152-164
            if (e.lost > 30) term = true;
            e.lost++;
            if (KIND == 4) { obs[6] = e.lc[0]; obs[7] = e.lc[1]; }
        } else { e.lc[0] = cxn; e.lc[1] = cyn; e.bits |= B_HAVE_CENTER; e.lost = 0; }
        reward = 0.5f;
        if (e.bits & B_HAVE_CENTER) {
            const float fx = 0.5f - e.lc[0], fy = 0.5f - e.lc[1];
            const float dd = tsqrt(fx*fx + fy*fy);
            if (KIND == 4) {                                               // env04_v1.py:108-131
                reward += __builtin_expf(-10.0f*dd);
                reward += -1.0f*dd;
                if (dd < 0.1f && !(e.bits & B_BLOCK_UPDATED)) {
                    e.bits |= B_BLOCK_UPDATED;
#pragma unroll
                    for (int i = 0; i < 3; i++) { e.tgt[i] = c.smin[i] + (c.smax[i] - c.smin[i])*u[i]; e.cube.pos[i] = e.tgt[i]; }
                    reward += 10.0f;
                }
            } else reward += -1.0f*dd;                                     // env03_v1.py:168-176
        }
        reward += joint_reward(c.old);
        if (KIND == 4) {                                                   // env04_v1.py:137-148
            const float wr = tclamp(joint_penalty(c.old[4], START_POSITION[4] - 0.2f, START_POSITION[4] + 0.2f), -0.2f, 0.0f);
            reward += wr*0.5f;
        } else {                                                           // env03_v1.py:182-189, env_base_01.py:165-178
            float pen = 0.0f, av[6];
#pragma unroll
            for (int i = 0; i < 6; i++) av[i] = (c.ncmd[i] - c.old[i])/h;
            if (e.bits & B_HAVE_ANGVEL) {
#pragma unroll
                for (int i = 0; i < 6; i++) pen += tabs(av[i] - e.av[i])*0.0025f;
            }
#pragma unroll
            for (int i = 0; i < 6; i++) e.av[i] = av[i];
            e.bits |= B_HAVE_ANGVEL;
            reward += (-pen)*c.frac;
        }
        obs[6] = 5.0f*obs[6]; obs[7] = 5.0f*obs[7];                        // :195-196 (Q4)
#pragma unroll
        for (int i = 0; i < 6; i++) e.cmd[i] = c.ncmd[i];                  // :198
    }
    return reward;
}

template <int KIND> SO100_HD float env_step(EnvState& e, const float a[6], const float u[8], const SimParams& p,
                                             float* obs, bool& term) {
    StepCtx c;
    env_step_pre<KIND>(e, a, u, p, c);
    TaskPoses<float> P;
    float cstale[3];
    physics_substeps(e, c.ctrl, p, !reach_kind<KIND>(), P, cstale);
    return env_step_post<KIND>(e, c, u, P, cstale, obs, term);
}

// Env.step + gymnasium TimeLimit + SB3 DummyVecEnv auto-reset, for one env
struct StepResult { float reward; bool done, trunc_only; float ep_return; int ep_length; };

template <int KIND> SO100_HD StepResult env_step_finish(EnvState& e, float reward, bool term, const SimParams& p, uint32_t env_gid,
                                                         const float* inject, const float* start_tab, float* obs, float* terminal_obs);

template <int KIND> SO100_HD StepResult env_step_vec(EnvState& e, const float a[6], const SimParams& p, uint32_t env_gid,
                                                      const float* inject, const float* start_tab, float* obs, float* terminal_obs) {
    float u[8];
    draw8(p, env_gid, (uint32_t)e.rngc, 0, inject, u);
    e.rngc++;
    bool term;
    const float reward = env_step<KIND>(e, a, u, p, obs, term);
    return env_step_finish<KIND>(e, reward, term, p, env_gid, inject, start_tab, obs, terminal_obs);
}

// TimeLimit + episode statistics + auto-reset (the tail of env_step_vec; u8 scratch is drawn here for the reset phase)
template <int KIND> SO100_HD StepResult env_step_finish(EnvState& e, float reward, bool term, const SimParams& p, uint32_t env_gid,
                                                         const float* inject, const float* start_tab, float* obs, float* terminal_obs) {
    float u[8];
    StepResult r;
    // Non-finite state guard (the observation is a finite function of a finite state).  MuJoCo answers a NaN / > 1e10 qpos, qvel or qacc with a warning and mj_resetData
    // (mj_checkPos / mj_checkVel / mj_checkAcc); the reference adds nothing, so one NaN action poisons data.ctrl (Env01/02)
    // or the command integrator (Env03-05) for the rest of the episode.  Here that env's episode ends instead:
    // terminated, reward 0, terminal observation 0, B_BAD_STATE latched in `bits`, then the normal auto-reset; the other
    // envs of the batch never see it.
    {
        float z = reward*0.0f, mx = 0.0f;                          // x*0 is NaN for NaN / inf (and is not folded under IEEE rules)
#pragma unroll
        for (int i = 0; i < 6; i++) { z += e.q[i]*0.0f + e.v[i]*0.0f + e.cube.vel[i]*0.0f; mx = tmax(mx, tmax(tabs(e.q[i]), tabs(e.v[i]))); mx = tmax(mx, tabs(e.cube.vel[i])); }
#pragma unroll
        for (int i = 0; i < 3; i++) { z += e.cube.pos[i]*0.0f; mx = tmax(mx, tabs(e.cube.pos[i])); }
#pragma unroll
        for (int i = 0; i < 4; i++) z += e.cube.quat[i]*0.0f;
        if (!(z == 0.0f) || !(mx < 1.0e10f) || (e.bits & B_BAD_ACTION)) {
            term = true; reward = 0.0f; e.bits = (e.bits | B_BAD_STATE) & ~B_BAD_ACTION;
#pragma unroll
            for (int i = 0; i < obs_dim<KIND>(); i++) obs[i] = 0.0f;
            if (uses_group<KIND>(3)) {                             // the one pose memory that survives env_reset (ref: env_base_01.py:170-177)
#pragma unroll
                for (int i = 0; i < 6; i++) e.av[i] = 0.0f;
                e.bits &= ~B_HAVE_ANGVEL;
            }
        }
    }
    r.reward = reward;
    e.elapsed++;
    const bool trunc = p.max_episode_steps > 0 && e.elapsed >= p.max_episode_steps;
    e.epret += r.reward; e.eplen++;
    r.done = term || trunc; r.trunc_only = trunc && !term;
    r.ep_return = e.epret; r.ep_length = e.eplen;
    if (r.done) {
#pragma unroll
        for (int i = 0; i < obs_dim<KIND>(); i++) terminal_obs[i] = obs[i];
        draw8(p, env_gid, (uint32_t)e.rngc, 1, inject, u);
        e.rngc++;
        env_reset<KIND>(e, u, start_tab, obs);
    }
    return r;
}

}  // namespace so100
