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
    T qq[6], vv[6], cc[6], f1[6], f2[6];
    for (int i = 0; i < 6; i++) { qq[i] = (T)q[i]; vv[i] = (T)v[i]; cc[i] = (T)ctrl[i]; f1[i] = (T)ff[i]; f2[i] = (T)fl[i]; }
    Arm<T> A;
    for (int s = 0; s < n; s++) arm_substep(qq, vv, cc, f1, f2, flags, iters, A);
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
}
