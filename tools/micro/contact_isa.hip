// Compile-only helper: static instruction counts of the pad-contact solve in isolation (tools/contact_count.sh).
#include <hip/hip_runtime.h>
#include "so100_contact.hpp"
using namespace so100;
// the solve of one substep for a lane with pad/floor contacts (6 unknowns), records in LDS as in the multi-wave kernels
__global__ void __launch_bounds__(64) k_solve6(const float* in, float* out, unsigned flags, int iters, int nc) {
    __shared__ float cbuf[MAXC*CF*64];
    const int t = threadIdx.x;
    for (int i = t; i < MAXC*CF*64; i += 64) cbuf[i] = in[i];
    __syncthreads();
    float q[6], v[6], qc[6], ctrl[6], ff[6], fl[6], aw[6], dq[6];
    Arm<float> A;
    for (int i = 0; i < 6; i++) { q[i] = in[t + 64*i]; v[i] = in[t + 64*(6 + i)]; qc[i] = 0; ctrl[i] = in[t + 64*(12 + i)]; ff[i] = in[t + 64*(18 + i)]; fl[i] = in[t + 64*(24 + i)]; aw[i] = in[t + 64*(30 + i)]; A.s[i] = in[t + 64*(36 + i)]; A.c[i] = in[t + 64*(42 + i)]; A.bias[i] = in[t + 64*(48 + i)]; }
    for (int i = 0; i < 21; i++) { A.M[i] = in[t + 64*(60 + i)]; A.Minv[i] = in[t + 64*(90 + i)]; }
    for (int i = 0; i < 6; i++) A.Dinv[i] = in[t + 64*(120 + i)];
    WorldFK<float> W; world_fk<float>(A.s, A.c, W);
    ContactsLds<float> cs{ cbuf, t }; cs.n = nc;
    Cube<float> ct{}; float Rc[9] = {1,0,0,0,1,0,0,0,1}, ap[3] = {0,0,0}, res = 0;
    int zones = in[t + 64*130] > 0.0f ? 5 : -1;
    contact_solve_integrate<float>(q, v, qc, ctrl, ff, fl, aw, flags, iters, A, W, cs, false, ct, Rc, ap, dq, &res, &zones);
    for (int i = 0; i < 6; i++) { out[t + 64*i] = q[i]; out[t + 64*(6 + i)] = v[i]; out[t + 64*(12 + i)] = aw[i]; out[t + 64*(18 + i)] = ff[i] + fl[i] + dq[i]; }
    out[t + 64*24] = res;
}
// detection alone (wave 3's job)
__global__ void __launch_bounds__(64) k_detect(const float* in, float* out, unsigned flags) {
    __shared__ float cbuf[MAXC*CF*64];
    const int t = threadIdx.x;
    float s[6], c[6], v[6];
    for (int i = 0; i < 6; i++) { s[i] = in[t + 64*i]; c[i] = in[t + 64*(6 + i)]; v[i] = in[t + 64*(12 + i)]; }
    WorldFK<float> W; world_fk<float>(s, c, W);
    Cube<float> cb{}; float Rc[9] = {1,0,0,0,1,0,0,0,1};
    ContactsLds<float> cs{ cbuf, t };
    const bool cp = detect_pad_contacts<float>(W, v, cb, Rc, flags, false, cs);
    out[t] = cs.n + (cp ? 100 : 0) + cbuf[t];
}
// one evaluation of gradient + Hessian of the 6-unknown problem (the unit of work of a Newton iteration)
__global__ void __launch_bounds__(64) k_eval6(const float* in, float* out, int nc) {
    __shared__ float cbuf[MAXC*CF*64];
    const int t = threadIdx.x;
    for (int i = t; i < MAXC*CF*64; i += 64) cbuf[i] = in[i];
    __syncthreads();
    float s[6], c[6], x[6], tau[6], Marm[21];
    ArmRows<float> r;
    for (int i = 0; i < 6; i++) { s[i] = in[t + 64*i]; c[i] = in[t + 64*(6 + i)]; x[i] = in[t + 64*(12 + i)]; tau[i] = in[t + 64*(18 + i)];
        r.fmax_[i] = in[t + 64*(24 + i)]; r.Rf[i] = in[t + 64*(30 + i)]; r.cfv[i] = in[t + 64*(36 + i)]; r.sg[i] = in[t + 64*(42 + i)]; r.Rl[i] = in[t + 64*(48 + i)]; r.clv[i] = in[t + 64*(54 + i)]; }
    for (int i = 0; i < 21; i++) Marm[i] = in[t + 64*(60 + i)];
    WorldFK<float> W; world_fk<float>(s, c, W);
    ContactsLds<float> cs{ cbuf, t }; cs.n = nc;
    float Rc[9] = {1,0,0,0,1,0,0,0,1}, z3[3] = {0,0,0};
    int zones = 0;
    PrimalProblem<6, float, ContactsLds<float>> P{ W, cs, Marm, tau, r, Rc, z3, z3, &zones };
    float g[6], H[21];
    P.eval<2>(x, g, H);
    float acc = 0.0f;
    for (int i = 0; i < 6; i++) acc += g[i];
    for (int i = 0; i < 21; i++) acc += H[i];
    out[t] = acc;
}
