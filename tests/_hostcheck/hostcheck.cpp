// Host instantiation (float and double) of the DEVICE physics header, for CPU-side unit tests only.
// Not a product path: libso100sim.so never links this file, and the product has no CPU fallback.
#include "../../so100_mujoco_rl_amd/csrc/so100_physics.hpp"
#include "../../so100_mujoco_rl_amd/csrc/so100_cube.hpp"
#include <cstring>
using namespace so100;

template <typename T> static void dyn(const double* q, const double* v, double* M36, double* bias) {
    T qq[6], vv[6]; for (int i = 0; i < 6; i++) { qq[i] = (T)q[i]; vv[i] = (T)v[i]; }
    Arm<T> A; arm_dynamics(qq, vv, A);
    for (int i = 0; i < 6; i++) { bias[i] = A.bias[i]; for (int j = 0; j < 6; j++) M36[6*i+j] = sym6(A.M, i, j); }
}
template <typename T> static void sub(double* q, double* v, const double* ctrl, double* ff, double* fl, unsigned flags, int iters, int n) {
    T qq[6], vv[6], cc[6], f1[6], f2[6], qc[6] = {0,0,0,0,0,0};
    for (int i = 0; i < 6; i++) { qq[i] = (T)q[i]; vv[i] = (T)v[i]; cc[i] = (T)ctrl[i]; f1[i] = (T)ff[i]; f2[i] = (T)fl[i]; }
    Arm<T> A; T dq[6] = {0,0,0,0,0,0};
    for (int s = 0; s < n; s++) arm_substep(qq, vv, qc, cc, f1, f2, flags, iters, A, (s % 16) == 0, dq);
    for (int i = 0; i < 6; i++) { q[i] = qq[i]; v[i] = vv[i]; ff[i] = f1[i]; fl[i] = f2[i]; }
}
template <typename T> static void poses(const double* q, double* out /*3+3+9+3+9*/) {
    T s[6], c[6]; for (int i = 0; i < 6; i++) tsincos<T>((T)q[i], s[i], c[i]);
    TaskPoses<T> P; task_poses(s, c, true, P);
    int k = 0;
    for (int i = 0; i < 3; i++) out[k++] = P.wrist[i];
    for (int i = 0; i < 3; i++) out[k++] = P.jaw_pos[i];
    for (int i = 0; i < 9; i++) out[k++] = P.jaw_mat[i];
    for (int i = 0; i < 3; i++) out[k++] = P.cam_pos[i];
    for (int i = 0; i < 9; i++) out[k++] = P.cam_mat[i];
}
template <typename T> static void cube(double* pos, double* quat, double* vel, double* warm, const double* applied, unsigned flags, int iters, int n) {
    Cube<T> c;
    for (int i = 0; i < 3; i++) c.pos[i] = (T)pos[i];
    for (int i = 0; i < 4; i++) c.quat[i] = (T)quat[i];
    for (int i = 0; i < 6; i++) c.vel[i] = (T)vel[i];
    for (int i = 0; i < 6; i++) c.warm[i] = (T)warm[i];
    T ap[3] = { (T)applied[0], (T)applied[1], (T)applied[2] };
    for (int s = 0; s < n; s++) cube_substep(c, ap, flags, iters);
    for (int i = 0; i < 3; i++) pos[i] = c.pos[i];
    for (int i = 0; i < 4; i++) quat[i] = c.quat[i];
    for (int i = 0; i < 6; i++) vel[i] = c.vel[i];
    for (int i = 0; i < 6; i++) warm[i] = c.warm[i];
}
extern "C" {
void hc_dyn_d(const double* q, const double* v, double* M, double* b) { dyn<double>(q, v, M, b); }
void hc_dyn_f(const double* q, const double* v, double* M, double* b) { dyn<float>(q, v, M, b); }
void hc_sub_d(double* q, double* v, const double* c, double* ff, double* fl, unsigned fg, int it, int n) { sub<double>(q, v, c, ff, fl, fg, it, n); }
void hc_sub_f(double* q, double* v, const double* c, double* ff, double* fl, unsigned fg, int it, int n) { sub<float>(q, v, c, ff, fl, fg, it, n); }
void hc_poses_d(const double* q, double* out) { poses<double>(q, out); }
void hc_poses_f(const double* q, double* out) { poses<float>(q, out); }
void hc_cube_d(double* p, double* q, double* v, double* w, const double* a, unsigned fg, int it, int n) { cube<double>(p, q, v, w, a, fg, it, n); }
void hc_cube_f(double* p, double* q, double* v, double* w, const double* a, unsigned fg, int it, int n) { cube<float>(p, q, v, w, a, fg, it, n); }
void hc_sincos_f(float x, float* s, float* c) { tsincos<float>(x, *s, *c); }
void hc_dbg_counters(long* out) { out[0] = g_dbg_newton_iters; out[1] = g_dbg_newton_ls; }
}

// ---- finger-pad contacts (so100_contact.hpp): whole substeps of one env, fp64 and fp32 ----------------------------------------
#include "../../so100_mujoco_rl_amd/csrc/so100_contact.hpp"
// state: q6 v6 ff6 fl6 aw6 | cube pos3 quat4 vel6 warm6  (49 doubles); stat[5]: per substep max of [contacts, coupled, dropped], [3] = max residual*1e9,
// [4] = contact-set signature of the last substep
template <typename T> static void csub(double* st, const double* ctrl, const double* applied, unsigned flags, int iters, int citers, int n, int* stat) {
    T q[6], v[6], ff[6], fl[6], aw[6], qc[6] = {0,0,0,0,0,0}, cc[6], ap[3] = { (T)applied[0], (T)applied[1], (T)applied[2] };
    Cube<T> cb;
    for (int i = 0; i < 6; i++) { q[i] = (T)st[i]; v[i] = (T)st[6+i]; ff[i] = (T)st[12+i]; fl[i] = (T)st[18+i]; aw[i] = (T)st[24+i]; cc[i] = (T)ctrl[i]; }
    for (int i = 0; i < 3; i++) cb.pos[i] = (T)st[30+i];
    for (int i = 0; i < 4; i++) cb.quat[i] = (T)st[33+i];
    for (int i = 0; i < 6; i++) { cb.vel[i] = (T)st[37+i]; cb.warm[i] = (T)st[43+i]; }
    Arm<T> A; T dq[6] = {0,0,0,0,0,0}; T res = T(0);
    ContactsPriv<T> cs; int zones = -1;
    stat[0] = stat[1] = stat[2] = stat[3] = stat[4] = 0;
    for (int s = 0; s < n; s++) {
        int sst[4];
        substep_with_pads<T>(q, v, qc, cc, ff, fl, aw, cb, ap, flags, iters, citers, A, (s % 16) == 0, dq, &res, cs, zones, sst);
        for (int k = 0; k < 3; k++) stat[k] = sst[k] > stat[k] ? sst[k] : stat[k];
        stat[4] = sst[3];                                    // signature of the LAST substep's pad-contact set
    }
    stat[3] = (int)(res*1e9 > 2e9 ? 2e9 : res*1e9);
    for (int i = 0; i < 6; i++) { st[i] = q[i]; st[6+i] = v[i]; st[12+i] = ff[i]; st[18+i] = fl[i]; st[24+i] = aw[i]; }
    for (int i = 0; i < 3; i++) st[30+i] = cb.pos[i];
    for (int i = 0; i < 4; i++) st[33+i] = cb.quat[i];
    for (int i = 0; i < 6; i++) { st[37+i] = cb.vel[i]; st[43+i] = cb.warm[i]; }
}
template <typename T> static int bb(const double* cA, const double* RA, const double* hA, const double* cB, const double* RB, const double* hB, double* pos, double* nrm, double* dist) {
    T a[3], ra[9], ha[3], b[3], rb[9], hb[3], n[3];
    for (int i = 0; i < 3; i++) { a[i] = (T)cA[i]; ha[i] = (T)hA[i]; b[i] = (T)cB[i]; hb[i] = (T)hB[i]; }
    for (int i = 0; i < 9; i++) { ra[i] = (T)RA[i]; rb[i] = (T)RB[i]; }
    int k = 0;
    const int cnt = box_box<T>(a, ra, ha, b, rb, hb, n, [&](const T* p, T d) { pos[3*k] = p[0]; pos[3*k+1] = p[1]; pos[3*k+2] = p[2]; dist[k] = d; k++; });
    for (int i = 0; i < 3; i++) nrm[i] = n[i];
    return cnt;
}
template <typename T> static int capbox(const double* a, const double* b, double r, const double* c, const double* R, const double* h, double* pos, double* nrm, double* dist) {
    T a_[3], b_[3], c_[3], R_[9], h_[3], p[3], n[3], d = T(0);
    for (int i = 0; i < 3; i++) { a_[i] = (T)a[i]; b_[i] = (T)b[i]; c_[i] = (T)c[i]; h_[i] = (T)h[i]; }
    for (int i = 0; i < 9; i++) R_[i] = (T)R[i];
    const int k = capsule_box<T>(a_, b_, (T)r, c_, R_, h_, p, n, d);
    if (k) { for (int i = 0; i < 3; i++) { pos[i] = p[i]; nrm[i] = n[i]; } *dist = d; }
    return k;
}
extern "C" {
int hc_capbox_d(const double* a, const double* b, double r, const double* c, const double* R, const double* h, double* pos, double* nrm, double* dist) { return capbox<double>(a, b, r, c, R, h, pos, nrm, dist); }
int hc_capbox_f(const double* a, const double* b, double r, const double* c, const double* R, const double* h, double* pos, double* nrm, double* dist) { return capbox<float>(a, b, r, c, R, h, pos, nrm, dist); }
void hc_csub_d(double* st, const double* ctrl, const double* ap, unsigned fg, int it, int cit, int n, int* stat) { csub<double>(st, ctrl, ap, fg, it, cit, n, stat); }
void hc_csub_f(double* st, const double* ctrl, const double* ap, unsigned fg, int it, int cit, int n, int* stat) { csub<float>(st, ctrl, ap, fg, it, cit, n, stat); }
int hc_boxbox_d(const double* cA, const double* RA, const double* hA, const double* cB, const double* RB, const double* hB, double* pos, double* nrm, double* dist) { return bb<double>(cA, RA, hA, cB, RB, hB, pos, nrm, dist); }
int hc_boxbox_f(const double* cA, const double* RA, const double* hA, const double* cB, const double* RB, const double* hB, double* pos, double* nrm, double* dist) { return bb<float>(cA, RA, hA, cB, RB, hB, pos, nrm, dist); }
void hc_cdbg_trace(int on) { g_dbg_cnewton_trace = on; }
void hc_cdbg_counters(long* out) { out[0] = g_dbg_cnewton_calls; out[1] = g_dbg_cnewton_iters; out[2] = g_dbg_cnewton_ls; }
long hc_cdbg_passes(void) { return g_dbg_cnewton_passes; }
long hc_cdbg_signpasses(void) { return g_dbg_cnewton_signpasses; }
long hc_cdbg_gradpasses(void) { return g_dbg_cnewton_gradpasses; }
long hc_cdbg_lastiter(void) { return g_dbg_cnewton_lastiter; }
void hc_cdbg_hist(long* out64, int reset) { for (int i = 0; i < 64; i++) { out64[i] = (&g_dbg_cnewton_hist[0][0])[i]; if (reset) (&g_dbg_cnewton_hist[0][0])[i] = 0; } }
int hc_contact_id_hash(int id) { return contact_id_hash(id); }
}

// ---- the full task layer (so100_task.hpp) on the host, fp32, one env ------------------------------------------------
#include "../../so100_mujoco_rl_amd/csrc/so100_task.hpp"
namespace { 
#include "../../so100_mujoco_rl_amd/csrc/so100_start_positions.inc"
float g_tab[36*6]; bool g_tab_ok = false;
const float* tab() { if (!g_tab_ok) { for (int i = 0; i < 36; i++) for (int j = 0; j < 6; j++) g_tab[6*i+j] = (float)SO100_VALID_START_POSITIONS[i][j]; g_tab_ok = true; } return g_tab; }
template <int KIND> void env_new(EnvState* e) { env_init<KIND>(*e); }
template <int KIND> void env_rst(EnvState* e, const float* inject, float* obs) {
    SimParams p{}; float u[8]; draw8(p, 0, (uint32_t)e->rngc, 1, inject, u); e->rngc++; env_reset<KIND>(*e, u, tab(), obs);
}
template <int KIND> void env_stp(EnvState* e, const SimParams& p, const float* a, const float* inject, float* obs, float* tobs, float* rew, int* done, int* trunc) {
    StepResult r = env_step_vec<KIND>(*e, a, p, p.env_id_offset, inject, tab(), obs, tobs);
    *rew = r.reward; *done = r.done; *trunc = r.trunc_only;
}
}
#define KSWITCH(kind, fn, ...) switch (kind) { case 1: fn<1>(__VA_ARGS__); break; case 2: fn<2>(__VA_ARGS__); break; case 3: fn<3>(__VA_ARGS__); break; case 4: fn<4>(__VA_ARGS__); break; case 5: fn<5>(__VA_ARGS__); break; default: fn<6>(__VA_ARGS__); }
extern "C" {
void* hc_env_new(int kind) { EnvState* e = new EnvState; KSWITCH(kind, env_new, e); return e; }
void hc_env_free(void* e) { delete (EnvState*)e; }
void hc_env_reset(void* e, int kind, const float* inject, float* obs) { KSWITCH(kind, env_rst, (EnvState*)e, inject, obs); }
void hc_env_step(void* e, int kind, unsigned flags, int iters, int citers, int max_steps, const float* a, const float* inject,
                 float* obs, float* tobs, float* rew, int* done, int* trunc) {
    SimParams p{}; p.n = 1; p.flags = flags; p.solver_iters = iters; p.contact_iters = citers; p.frame_skip = 16; p.max_episode_steps = max_steps;
    KSWITCH(kind, env_stp, (EnvState*)e, p, a, inject, obs, tobs, rew, done, trunc);
}
void hc_env_stats(void* ev, double* out2) { EnvState* e = (EnvState*)ev; out2[0] = e->cstat; out2[1] = e->res; }
void hc_env_qpos(void* ev, double* qpos13, double* qvel12) {
    EnvState* e = (EnvState*)ev;
    for (int i = 0; i < 6; i++) { qpos13[i] = e->q[i]; qvel12[i] = e->v[i]; qvel12[6+i] = e->cube.vel[i]; }
    for (int i = 0; i < 3; i++) qpos13[6+i] = e->cube.pos[i];
    for (int i = 0; i < 4; i++) qpos13[9+i] = e->cube.quat[i];
}
}
