// so100_cube.hpp -- the free-joint cube "block_a" and its box/plane contact with the floor.
//
// Reference: scene:29-35 (body, free joint, box geom half size 0.01), scene:39 (floor plane z = 0).
// MuJoCo stages restated (Appendix A.4):
//   mjc_PlaneBox narrowphase (<= 4 corner contacts), pyramidal friction cone (condim 3, mu = 1):
//   4 edge rows per contact with J = n +- mu t_k, aref = -B Jv - K imp dist, R = 2 mu^2 (1-imp)/imp * 2/m,
//   force >= 0 per edge, semi-implicit Euler with mju_quatIntegrate for the orientation.
// Solver: the 16 edge rows of a resting cube are highly redundant (rank <= 6), so scalar PGS on the dual
// needs > 100 sweeps for 1e-5 (measured against the oracle).  This block is therefore solved in the
// PRIMAL, like MuJoCo's default Newton solver: minimise over x = qacc - qacc_smooth (6 numbers)
//     1/2 x'Mx + sum_r 1/(2 R_r) min(0, b_r + J_r x)^2 ,      M = diag(m,m,m,I,I,I),
// by Newton: full step when it does not increase the cost, otherwise an exact bracketed line search along the
// direction (the cost along a ray is a convex piecewise quadratic); a 6x6 LDL^T per iteration, 1-4 iterations.
// It reaches the same optimum as the oracle's PGS-to-convergence (tests/test_hostcheck.py).
// The cube is dynamically decoupled from the arm (no arm-cube contact is modelled: the reference
// excludes block_a against 5 of the 7 arm bodies, scene:44-48, and the remaining mesh geoms are not
// available), so its 6x6 mass matrix is the constant diag(m,m,m,I,I,I) and its rows are solved on their own.
//
// Everything is indexed statically (4 contact slots x 4 edges) so that it stays in registers.
#pragma once
#include "so100_physics.hpp"

namespace so100 {

#if !defined(__HIPCC__)
static long g_dbg_newton_iters = 0, g_dbg_newton_ls = 0;      // host-only instrumentation (tests/_hostcheck)
#endif

template <typename T> struct Cube {
    T pos[3], quat[4], vel[6];     // vel = (linear, world frame; angular, body frame) like MuJoCo qvel
    T warm[6];                     // previous x = qacc - qacc_smooth (Newton warm start)
};

template <typename T> SO100_HD void quat_to_mat(const T q[4], T m[9]) {
    const T w = q[0], x = q[1], y = q[2], z = q[3];
    m[0] = w*w + x*x - y*y - z*z; m[1] = T(2)*(x*y - w*z);       m[2] = T(2)*(x*z + w*y);
    m[3] = T(2)*(x*y + w*z);       m[4] = w*w - x*x + y*y - z*z; m[5] = T(2)*(y*z - w*x);
    m[6] = T(2)*(x*z - w*y);       m[7] = T(2)*(y*z + w*x);       m[8] = w*w - x*x - y*y + z*z;
}
template <typename T> SO100_HD void quat_normalize(T q[4]) {
    const T n = tsqrt(q[0]*q[0] + q[1]*q[1] + q[2]*q[2] + q[3]*q[3]);
    if (n < T(1e-15)) { q[0] = T(1); q[1] = q[2] = q[3] = T(0); return; }
    const T r = trcp(n);
    q[0] *= r; q[1] *= r; q[2] *= r; q[3] *= r;
}

// the four pyramid edge directions n +- mu t1, n +- mu t2 with n = +z, t1 = +y, t2 = -x, mu = 1
// (mju_makeFrame for a +z normal), in the oracle's row order
template <int E> struct EdgeDir;
template <> struct EdgeDir<0> { static constexpr double d[3] = { 0,  1, 1 }; };
template <> struct EdgeDir<1> { static constexpr double d[3] = { 0, -1, 1 }; };
template <> struct EdgeDir<2> { static constexpr double d[3] = { -1, 0, 1 }; };
template <> struct EdgeDir<3> { static constexpr double d[3] = { 1,  0, 1 }; };
// The linear part of every edge row is a constant with two non-zero entries, d = sa e_A + e_z (A = tangential axis,
// sa = +-1): products with it are written out as signed adds (`0.0f * x` is not folded by the compiler under IEEE rules).
template <int E> struct EdgeAxis { static constexpr int A = E < 2 ? 1 : 0; static constexpr bool pos = EdgeDir<E>::d[E < 2 ? 1 : 0] > 0; };
template <int E, typename T> SO100_HD T edge_dot(const T v[3]) { return (EdgeAxis<E>::pos ? v[EdgeAxis<E>::A] : -v[EdgeAxis<E>::A]) + v[2]; }   // d_E . v

template <typename T> struct CubeRows {
    bool act[4];
    T rl[4][3];        // contact point relative to the cube centre, body frame
    T dl[4][3];        // edge directions in the body frame (shared by the 4 contacts)
    T b[4][4], R[4], arinv[4][4];
};

// row (S,E): jar = b + J x,  J = [dir_E ; rl[S] x dl[E]];  accumulates cost / gradient / Hessian
template <int S, int E, typename T>
SO100_HD void cube_row_accum(const CubeRows<T>& r, const T x[6], T& cost, T g[6], T Hm[21], bool want_gh) {
    constexpr int A = EdgeAxis<E>::A; constexpr bool pos = EdgeAxis<E>::pos;
    T ja[3]; cross(r.rl[S], r.dl[E], ja);
    T jar = r.b[S][E] + edge_dot<E>(x);
    jar += ja[0]*x[3]; jar += ja[1]*x[4]; jar += ja[2]*x[5];
    const bool on = r.act[S] && jar < T(0);
    const T D = on ? r.arinv[S][E] : T(0);             // 1/R of the row when active
    cost += T(0.5)*D*jar*jar;
    if (want_gh) {
        const T dj = D*jar, sD = pos ? D : -D;
        g[A] += pos ? dj : -dj; g[2] += dj;
        Hm[SO100_TRI(A, A)] += D; Hm[SO100_TRI(2, A)] += sD; Hm[SO100_TRI(2, 2)] += D;
#pragma unroll
        for (int i = 0; i < 3; i++) {
            const T dji = D*ja[i];
            g[3 + i] += jar*dji;
            Hm[SO100_TRI(3 + i, A)] += pos ? dji : -dji; Hm[SO100_TRI(3 + i, 2)] += dji;
#pragma unroll
            for (int j = 0; j <= i; j++) Hm[SO100_TRI(3 + i, 3 + j)] += dji*ja[j];
        }
    }
}
template <typename T>
SO100_HD T cube_rows_eval(const CubeRows<T>& r, const T x[6], T g[6], T Hm[21], bool want_gh) {
    const T m = T(so100g::CUBE_MASS), I = T(so100g::CUBE_INERTIA);
    T cost = T(0.5)*(m*(x[0]*x[0] + x[1]*x[1] + x[2]*x[2]) + I*(x[3]*x[3] + x[4]*x[4] + x[5]*x[5]));
    if (want_gh) {
#pragma unroll
        for (int i = 0; i < 21; i++) Hm[i] = T(0);
#pragma unroll
        for (int i = 0; i < 3; i++) { g[i] = m*x[i]; g[3+i] = I*x[3+i]; Hm[SO100_TRI(i, i)] = m; Hm[SO100_TRI(3+i, 3+i)] = I; }
    }
#define SO100_ACC(S) cube_row_accum<S, 0>(r, x, cost, g, Hm, want_gh); cube_row_accum<S, 1>(r, x, cost, g, Hm, want_gh); \
                     cube_row_accum<S, 2>(r, x, cost, g, Hm, want_gh); cube_row_accum<S, 3>(r, x, cost, g, Hm, want_gh);
    SO100_ACC(0) SO100_ACC(1) SO100_ACC(2) SO100_ACC(3)
#undef SO100_ACC
    return cost;
}
// gradient only (no cost, no Hessian): the cheap optimality test of the warm start
template <int S, int E, typename T>
SO100_HD void cube_row_grad(const CubeRows<T>& r, const T x[6], T g[6]) {
    constexpr int A = EdgeAxis<E>::A; constexpr bool pos = EdgeAxis<E>::pos;
    T ja[3]; cross(r.rl[S], r.dl[E], ja);
    T jar = r.b[S][E] + edge_dot<E>(x);
    jar += ja[0]*x[3]; jar += ja[1]*x[4]; jar += ja[2]*x[5];
    const T dj = (r.act[S] && jar < T(0)) ? r.arinv[S][E]*jar : T(0);
    g[A] += pos ? dj : -dj; g[2] += dj;
    g[3] += dj*ja[0]; g[4] += dj*ja[1]; g[5] += dj*ja[2];
}
template <typename T>
SO100_HD void cube_rows_grad(const CubeRows<T>& r, const T x[6], T g[6]) {
    const T m = T(so100g::CUBE_MASS), I = T(so100g::CUBE_INERTIA);
#pragma unroll
    for (int i = 0; i < 3; i++) { g[i] = m*x[i]; g[3+i] = I*x[3+i]; }
#define SO100_ACC(S) cube_row_grad<S, 0>(r, x, g); cube_row_grad<S, 1>(r, x, g); cube_row_grad<S, 2>(r, x, g); cube_row_grad<S, 3>(r, x, g);
    SO100_ACC(0) SO100_ACC(1) SO100_ACC(2) SO100_ACC(3)
#undef SO100_ACC
}
// line-search helpers: per row jar(alpha) = j0 + alpha jd
template <int S, int E, typename T>
SO100_HD void cube_row_lin(const CubeRows<T>& r, const T x[6], const T dx[6], T j0[4][4], T jd[4][4]) {
    T ja[3]; cross(r.rl[S], r.dl[E], ja);
    T a = r.b[S][E] + edge_dot<E>(x), b = edge_dot<E>(dx);
#pragma unroll
    for (int i = 0; i < 3; i++) { a += ja[i]*x[3 + i]; b += ja[i]*dx[3 + i]; }
    j0[S][E] = a; jd[S][E] = b;
}
template <typename T>
SO100_HD void cube_phi(const CubeRows<T>& r, const T j0[4][4], const T jd[4][4], T q1, T q2, T alpha, T& d1, T& d2) {
    d1 = q1 + alpha*q2; d2 = q2;                       // quadratic part: (x + alpha dx)'M dx, dx'M dx
#pragma unroll
    for (int s = 0; s < 4; s++)
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const T t = j0[s][e] + alpha*jd[s][e];
            const T D = (r.act[s] && t < T(0)) ? r.arinv[s][e] : T(0);
            d1 += D*t*jd[s][e]; d2 += D*jd[s][e]*jd[s][e];
        }
}
template <int S, int E, typename T>
SO100_HD void cube_row_setup(CubeRows<T>& r, const T a0[3], const T vl[3], const T va[3], T Kimpdist, T rinv) {
    T ja[3]; cross(r.rl[S], r.dl[E], ja);
    const T jv = edge_dot<E>(vl) + dot(ja, va);
    const T ja0 = edge_dot<E>(a0);                                 // angular a0 is zero (isotropic, no torque)
    r.b[S][E] = ja0 + T(so100g::SOLREF_B)*jv + Kimpdist;
    r.arinv[S][E] = rinv;
}

// One cube substep in two halves (the persistent rollout kernel places a workgroup barrier between them so that the
// cube hides behind both phases of the arm substep): cube_prepare = contact detection + row setup (reads the state),
// cube_finish = Newton solve + semi-implicit Euler (updates it).  cube_substep = both.
template <typename T> struct CubePrep { CubeRows<T> r; T a0[3]; };

template <typename T>
SO100_HD void cube_prepare(const Cube<T>& c, const T applied[3], unsigned flags, CubePrep<T>& P) {
    const T im = T(1.0/so100g::CUBE_MASS);
    // qacc_smooth: gravity + applied force (Env03-05 anti-gravity); no gyroscopic term (isotropic inertia)
    P.a0[0] = applied[0]*im; P.a0[1] = applied[1]*im; P.a0[2] = applied[2]*im - T(so100g::GRAVITY);
    const T* a0 = P.a0;
    if (flags & F_FLOOR) {
        T qn[4] = { c.quat[0], c.quat[1], c.quat[2], c.quat[3] };
        quat_normalize(qn);
        T Rm[9]; quat_to_mat(qn, Rm);
        CubeRows<T>& r = P.r;
        // ---- mjc_PlaneBox: corners below the centre and at / below the plane; first four in corner order
        const T hs = T(so100g::CUBE_HALF);
        const T cdist = c.pos[2];
        unsigned mask = 0u;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const T sx = (k & 1) ? hs : -hs, sy = (k & 2) ? hs : -hs, sz = (k & 4) ? hs : -hs;
            const T ld = Rm[6]*sx + Rm[7]*sy + Rm[8]*sz;
            mask |= !(cdist + ld > T(0) || ld > T(0)) ? (1u << k) : 0u;
        }
        T dist[4], rinv[4];
#pragma unroll
        for (int s = 0; s < 4; s++) {                             // slot s = the s-th hit in corner order
            const bool act = mask != 0u;
            const int k = act ? __builtin_ctz(mask) : 0;
            mask &= mask - 1u;
            r.act[s] = act;
            const T sx = (k & 1) ? hs : -hs, sy = (k & 2) ? hs : -hs, sz = (k & 4) ? hs : -hs;
            const T ld = Rm[6]*sx + Rm[7]*sy + Rm[8]*sz;
            dist[s] = act ? cdist + ld : T(0);
            const T v[3] = { act ? sx : T(0), act ? sy : T(0), act ? sz : T(0) };
            // contact point = corner - n dist/2; relative to the centre, in the body frame: v - R^T n dist/2
            const T hd = T(0.5)*dist[s];
            r.rl[s][0] = v[0] - Rm[6]*hd; r.rl[s][1] = v[1] - Rm[7]*hd; r.rl[s][2] = v[2] - Rm[8]*hd;
            const T imp = impedance(tabs(dist[s]));
            // R = 2 mu^2 * (1-imp)/imp * diagApprox, diagApprox = (1 + mu^2)/m, mu = 1
            r.R[s] = T(4.0/so100g::CUBE_MASS)*(T(1) - imp)*trcp(imp);
            rinv[s] = trcp(r.R[s]);
            dist[s] = T(so100g::SOLREF_K)*imp*dist[s];            // K imp dist
        }
        // edge directions in the body frame: dl = R^T dir
#define SO100_DL(E) { const T d0 = T(EdgeDir<E>::d[0]), d1 = T(EdgeDir<E>::d[1]), d2 = T(EdgeDir<E>::d[2]); \
            r.dl[E][0] = Rm[0]*d0 + Rm[3]*d1 + Rm[6]*d2; r.dl[E][1] = Rm[1]*d0 + Rm[4]*d1 + Rm[7]*d2; r.dl[E][2] = Rm[2]*d0 + Rm[5]*d1 + Rm[8]*d2; }
        SO100_DL(0) SO100_DL(1) SO100_DL(2) SO100_DL(3)
#undef SO100_DL
        const T vl[3] = { c.vel[0], c.vel[1], c.vel[2] }, va[3] = { c.vel[3], c.vel[4], c.vel[5] };
#define SO100_SETUP(S) cube_row_setup<S, 0>(r, a0, vl, va, dist[S], rinv[S]); cube_row_setup<S, 1>(r, a0, vl, va, dist[S], rinv[S]); \
                       cube_row_setup<S, 2>(r, a0, vl, va, dist[S], rinv[S]); cube_row_setup<S, 3>(r, a0, vl, va, dist[S], rinv[S]);
        SO100_SETUP(0) SO100_SETUP(1) SO100_SETUP(2) SO100_SETUP(3)
#undef SO100_SETUP
    }
}

// cube_finish = cube_solve (Newton on the floor rows -> linear / angular acceleration) + cube_integrate (semi-implicit Euler)
template <typename T>
SO100_HD void cube_solve(Cube<T>& c, unsigned flags, int iters, const CubePrep<T>& P, T al[3], T aa[3]) {
    const T* a0 = P.a0;
    al[0] = a0[0]; al[1] = a0[1]; al[2] = a0[2]; aa[0] = T(0); aa[1] = T(0); aa[2] = T(0);
    if (flags & F_FLOOR) {
        const CubeRows<T>& r = P.r;
        // Newton on x = qacc - qacc_smooth, warm-started from the previous substep
        T x[6];
#pragma unroll
        for (int i = 0; i < 6; i++) x[i] = r.act[0] ? c.warm[i] : T(0);
        if (r.act[0]) {                                   // slot 0 is filled first: no contact => x = 0
            const T m = T(so100g::CUBE_MASS), I = T(so100g::CUBE_INERTIA);
            for (int it = 0; it < iters; it++) {
                T g[6], Hm[21], Dinv[6], dx[6];
                // Cheap optimality test of the current x (at rest: the warm start) from the gradient alone.  H >= M, so the
                // Newton step dx = -H^-1 g obeys dx'M dx <= g'M^-1 g =: E; with I = (2/3) m half^2 both step measures of the
                // convergence test below are then < sqrt(1.5 E / m).  A resting cube -- the common case: one gradient
                // evaluation instead of gradient + Hessian + LDL^T -- keeps its x; threshold = a quarter of that tolerance.
                {
                    cube_rows_grad(r, x, g);
                    const T E = (g[0]*g[0] + g[1]*g[1] + g[2]*g[2])*T(1.0/so100g::CUBE_MASS)
                              + (g[3]*g[3] + g[4]*g[4] + g[5]*g[5])*T(1.0/so100g::CUBE_INERTIA);
                    const T tq = sizeof(T) == 4 ? T(0.25e-4) : T(0.25e-11);
                    if (E < tq*tq*T(so100g::CUBE_MASS/1.5)) break;
                }
#if !defined(__HIPCC__)
                g_dbg_newton_iters++;
#endif
                const T cost = cube_rows_eval(r, x, g, Hm, true);
#pragma unroll
                for (int i = 0; i < 6; i++) dx[i] = -g[i];
                ldl6(Hm, Dinv);
                ldl6_solve(Hm, Dinv, dx);
                // converged when the Newton step no longer changes the acceleration (fp32: ~1e-6 g): at rest the
                // warm start is already the optimum, so a resting cube costs one gradient/Hessian evaluation
                const T tol = sizeof(T) == 4 ? T(1e-4) : T(1e-11);          // fp32: 1e-5 g, above the ~1e-5 m/s^2 round-off floor
                const T dmax = tmax(tmax(tabs(dx[0]), tabs(dx[1])), tabs(dx[2]));
                const T amax = tmax(tmax(tabs(dx[3]), tabs(dx[4])), tabs(dx[5]))*T(so100g::CUBE_HALF);
                if (tmax(dmax, amax) < tol) {
#pragma unroll
                    for (int i = 0; i < 6; i++) x[i] += dx[i];
                    break;
                }
                // end game: a full Newton step, accepted when it achieves the decrease a quadratic model predicts (near the
                // optimum the exact fp32 line search below dithers for up to 9 evaluations per substep); otherwise fall through
                // Full Newton step first.  It is accepted when it achieves (80 % of) the decrease g.dx/2 that the quadratic
                // model predicts; if the cost at x + dx even MATCHES the model, no row switched on or off along the step, the
                // problem was quadratic and x + dx is its exact minimiser: stop without a confirming iteration (a cube that
                // is settling after a reset would otherwise pay two full evaluations per substep and hold up its wave).
                // Otherwise: exact line search (big active-set changes: impacts, tumbling).
                {
                    T xn[6], gt[6], Ht[21];
#pragma unroll
                    for (int i = 0; i < 6; i++) xn[i] = x[i] + dx[i];
                    const T cn = cube_rows_eval(r, xn, gt, Ht, false);
                    const T gdx = g[0]*dx[0] + g[1]*dx[1] + g[2]*dx[2] + g[3]*dx[3] + g[4]*dx[4] + g[5]*dx[5];
                    if (cn <= cost + T(0.4)*gdx) {
#pragma unroll
                        for (int i = 0; i < 6; i++) x[i] = xn[i];
                        const T model = cost + T(0.5)*gdx;
                        // (the second term is the round-off of the two cost evaluations themselves)
                        if (tabs(cn - model) <= (sizeof(T) == 4 ? T(1e-5) : T(1e-9))*tabs(gdx) + (sizeof(T) == 4 ? T(4e-7) : T(1e-15))*cost) break;
                        continue;
                    }
                }
                // exact line search: phi'(alpha) is increasing and piecewise linear; root by safeguarded Newton
                T j0[4][4], jd[4][4];
#define SO100_LIN(S) cube_row_lin<S, 0>(r, x, dx, j0, jd); cube_row_lin<S, 1>(r, x, dx, j0, jd); \
                     cube_row_lin<S, 2>(r, x, dx, j0, jd); cube_row_lin<S, 3>(r, x, dx, j0, jd);
                SO100_LIN(0) SO100_LIN(1) SO100_LIN(2) SO100_LIN(3)
#undef SO100_LIN
                const T q1 = m*(x[0]*dx[0] + x[1]*dx[1] + x[2]*dx[2]) + I*(x[3]*dx[3] + x[4]*dx[4] + x[5]*dx[5]);
                const T q2 = m*(dx[0]*dx[0] + dx[1]*dx[1] + dx[2]*dx[2]) + I*(dx[3]*dx[3] + dx[4]*dx[4] + dx[5]*dx[5]);
                T lo = T(0), hi = T(-1), alpha = T(1), d1, d2;      // hi < 0: no upper bracket yet
#pragma unroll 1
                for (int ls = 0; ls < 10; ls++) {
                    cube_phi(r, j0, jd, q1, q2, alpha, d1, d2);
#if !defined(__HIPCC__)
                    g_dbg_newton_ls++;
#endif
                    if (d1 < T(0)) lo = alpha; else hi = alpha;
                    T an = d2 > T(0) ? alpha - d1*trcp(d2) : alpha;
                    const bool inside = an > lo && (hi < T(0) || an < hi);
                    if (!inside) an = hi < T(0) ? T(2)*alpha : T(0.5)*(lo + hi);
                    if (tabs(an - alpha) <= (sizeof(T) == 4 ? T(1e-3) : T(1e-9))*tabs(alpha)) { alpha = an; break; }
                    alpha = an;
                }
#pragma unroll
                for (int i = 0; i < 6; i++) x[i] += alpha*dx[i];
            }
        }
#pragma unroll
        for (int i = 0; i < 6; i++) c.warm[i] = x[i];
        const T wl[3] = { x[0], x[1], x[2] }, wa[3] = { x[3], x[4], x[5] };
        al[0] += wl[0]; al[1] += wl[1]; al[2] += wl[2];
        aa[0] = wa[0]; aa[1] = wa[1]; aa[2] = wa[2];
    }
}

template <typename T>
SO100_HD void cube_integrate(Cube<T>& c, const T al[3], const T aa[3]) {
    const T h = T(so100g::TIMESTEP);
    // mj_Euler
#pragma unroll
    for (int i = 0; i < 3; i++) { c.vel[i] += h*al[i]; c.vel[3+i] += h*aa[i]; }
#pragma unroll
    for (int i = 0; i < 3; i++) c.pos[i] += h*c.vel[i];
    {   // mju_quatIntegrate
        T w[3] = { c.vel[3], c.vel[4], c.vel[5] };
        const T nrm = tsqrt(dot(w, w));
        if (nrm < T(1e-15)) { w[0] = T(1); w[1] = T(0); w[2] = T(0); }
        else { const T rn = trcp(nrm); w[0] *= rn; w[1] *= rn; w[2] *= rn; }
        T sn, cs; tsincos<T>(T(0.5)*h*nrm, sn, cs);
        quat_normalize(c.quat);
        const T a = c.quat[0], b = c.quat[1], cc = c.quat[2], d = c.quat[3];
        const T rw = cs, rx = w[0]*sn, ry = w[1]*sn, rz = w[2]*sn;
        c.quat[0] = a*rw - b*rx - cc*ry - d*rz;
        c.quat[1] = a*rx + b*rw + cc*rz - d*ry;
        c.quat[2] = a*ry - b*rz + cc*rw + d*rx;
        c.quat[3] = a*rz + b*ry - cc*rx + d*rw;
    }
}


template <typename T>
SO100_HD void cube_finish(Cube<T>& c, unsigned flags, int iters, const CubePrep<T>& P) {
    T al[3], aa[3];
    cube_solve(c, flags, iters, P, al, aa);
    cube_integrate(c, al, aa);
}

template <typename T>
SO100_HD void cube_substep(Cube<T>& c, const T applied[3], unsigned flags, int iters) {
    if (flags & F_CUBE_PINNED) return;
    CubePrep<T> P;
    cube_prepare(c, applied, flags, P);
    cube_finish(c, flags, iters, P);
}

}  // namespace so100
