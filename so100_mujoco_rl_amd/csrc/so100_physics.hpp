// so100_physics.hpp -- per-env physics of the so100 scene, written for one GPU lane per env.
//
// What it computes is what MuJoCo's mj_step computes for this model:
// forward kinematics, composite-rigid-body mass matrix, RNE bias force, position-servo actuation,
// friction-loss / joint-limit / cube-floor constraint rows, a projected Gauss-Seidel solve of the
// dual, semi-implicit Euler.  HOW it computes it is not MuJoCo's: the arm is a 6-link serial chain, so
// everything runs in LINK coordinates with the joint axis a coordinate axis of the link frame
// (Featherstone RNEA + CRBA), model constants are compile-time literals (so100_model_gen.h), the
// 6x6 system is factorised in registers, and the constraint rows (all +-e_i on the arm) are solved in
// joint space.  World-frame poses are only formed where the task layer reads them.
//
// Templated on the scalar T: float on the device; float/double host instantiations exist ONLY for the
// CPU-side unit tests in tests/_hostcheck (never linked into libso100sim.so).
#pragma once
#if defined(SO100_MODEL_GEN_HEADER)      // a second generated model (tests of `make gen INERTIALS=...`)
#include SO100_MODEL_GEN_HEADER
#else
#include "so100_model_gen.h"
#endif

#ifndef SO100_HD
#if defined(__HIPCC__)
#define SO100_HD __host__ __device__ __forceinline__
#else
#define SO100_HD inline
#endif
#endif

namespace so100 {

// physics option flags (include/so100_sim.h)
enum : unsigned { F_FRICTIONLOSS = 1u, F_LIMITS = 2u, F_FLOOR = 4u, F_CUBE_PINNED = 8u };

template <typename T> SO100_HD T tmin(T a, T b) { return a < b ? a : b; }
template <typename T> SO100_HD T tmax(T a, T b) { return a > b ? a : b; }
template <typename T> SO100_HD T tclamp(T x, T lo, T hi) { return tmin(tmax(x, lo), hi); }
template <typename T> SO100_HD T tabs(T a) { return a < T(0) ? -a : a; }

SO100_HD float  tsqrt(float x)  { return __builtin_sqrtf(x); }
SO100_HD double tsqrt(double x) { return __builtin_sqrt(x); }
SO100_HD float  texp(float x)   { return __builtin_expf(x); }
// reciprocal: on the device one v_rcp_f32 (1 ulp) plus a Newton step instead of the ~10-instruction IEEE division
SO100_HD float trcp(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    const float r = __builtin_amdgcn_rcpf(x);
    return __builtin_fmaf(__builtin_fmaf(-x, r, 1.0f), r, r);
#else
    return 1.0f/x;
#endif
}
SO100_HD double trcp(double x) { return 1.0/x; }
// true if the predicate holds in any active lane of the wavefront (on the host: the one "lane" there is)
SO100_HD bool wave_any(bool b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_ballot_w64(b) != 0ull;
#else
    return b;
#endif
}
SO100_HD float  tfloor(float x)  { return __builtin_floorf(x); }
SO100_HD double tfloor(double x) { return __builtin_floor(x); }

// sin and cos for |x| <= ~8 rad (joint angles are bounded by their limits; |x| < 3.5 in practice).
// Cody-Waite reduction by pi/2 (3-term split) + minimax polynomials on [-pi/4, pi/4] (cephes sinf /
// cosf coefficients; max error < 1 ulp in fp32).  Identical arithmetic on host and device.
template <typename T> SO100_HD void tsincos(T x, T& s, T& c);
template <> SO100_HD void tsincos<float>(float x, float& s, float& c) {
    const float k = tfloor(x * 0.636619772367581343f + 0.5f);          // nearest multiple of pi/2
    float r = __builtin_fmaf(k, -1.5703125f, x);
    r = __builtin_fmaf(k, -4.837512969970703125e-4f, r);
    r = __builtin_fmaf(k, -7.54978995489188e-8f, r);
    const float z = r * r;
    float ps = __builtin_fmaf(z, -1.9515295891e-4f, 8.3321608736e-3f);
    ps = __builtin_fmaf(z, ps, -1.6666654611e-1f);
    ps = __builtin_fmaf(ps * z, r, r);
    float pc = __builtin_fmaf(z, 2.443315711809948e-5f, -1.388731625493765e-3f);
    pc = __builtin_fmaf(z, pc, 4.166664568298827e-2f);
    pc = __builtin_fmaf(pc * z, z, __builtin_fmaf(z, -0.5f, 1.0f));
    const int n = (int)k & 3;
    const float ss = (n & 1) ? pc : ps, cc = (n & 1) ? ps : pc;
    s = (n & 2) ? -ss : ss;
    c = ((n + 1) & 2) ? -cc : cc;
}
template <> SO100_HD void tsincos<double>(double x, double& s, double& c) {
    s = __builtin_sin(x); c = __builtin_cos(x);
}

// ---------------------------------------------------------------------------------------------
// small vector helpers (everything is fully unrolled; arrays live in registers)
// ---------------------------------------------------------------------------------------------
template <typename T> SO100_HD void cross(const T a[3], const T b[3], T r[3]) {
    const T x = a[1]*b[2] - a[2]*b[1], y = a[2]*b[0] - a[0]*b[2], z = a[0]*b[1] - a[1]*b[0];
    r[0] = x; r[1] = y; r[2] = z;
}
template <typename T> SO100_HD T dot(const T a[3], const T b[3]) { return a[0]*b[0] + a[1]*b[1] + a[2]*b[2]; }

// rotation about coordinate axis A by (s,c): y = Rot x   and   y = Rot^T x
template <int A, typename T> SO100_HD void rot_axis(const T x[3], T s, T c, T y[3]) {
    constexpr int B = (A + 1) % 3, C = (A + 2) % 3;
    const T xb = x[B], xc = x[C];
    y[A] = x[A]; y[B] = c*xb - s*xc; y[C] = s*xb + c*xc;
}
template <int A, typename T> SO100_HD void rot_axis_T(const T x[3], T s, T c, T y[3]) {
    constexpr int B = (A + 1) % 3, C = (A + 2) % 3;
    const T xb = x[B], xc = x[C];
    y[A] = x[A]; y[B] = c*xb + s*xc; y[C] = -s*xb + c*xc;
}
// Compile-time-sparse products with model constants.  The link rotations are (nearly) axis-aligned: five exact
// zeros per matrix for links 0-4, and most link offsets have a zero x component.  `0.0f * x` cannot be folded by the
// compiler under IEEE rules, so exact zeros are skipped here with `if constexpr` (x * 1.0f and x * -1.0f fold anyway).
template <int K, int A, int B, int C_, typename T> SO100_HD T cdot3(T x0, T x1, T x2) {     // sum of LINK_C[K][.] * x.
    constexpr double a = so100g::LINK_C[K][A], b = so100g::LINK_C[K][B], c = so100g::LINK_C[K][C_];
    if constexpr (a != 0 && b != 0 && c != 0) return T(a)*x0 + T(b)*x1 + T(c)*x2;
    else if constexpr (a != 0 && b != 0) return T(a)*x0 + T(b)*x1;
    else if constexpr (a != 0 && c != 0) return T(a)*x0 + T(c)*x2;
    else if constexpr (b != 0 && c != 0) return T(b)*x1 + T(c)*x2;
    else if constexpr (a != 0) return T(a)*x0;
    else if constexpr (b != 0) return T(b)*x1;
    else if constexpr (c != 0) return T(c)*x2;
    else return T(0);
}
template <int K, int IA, int IB, typename T> SO100_HD T pdiff(T xa, T xb) {                  // P[IA]*xa - P[IB]*xb
    constexpr double a = so100g::LINK_P[K][IA], b = so100g::LINK_P[K][IB];
    if constexpr (a != 0 && b != 0) return T(a)*xa - T(b)*xb;
    else if constexpr (a != 0) return T(a)*xa;
    else if constexpr (b != 0) return -(T(b)*xb);
    else return T(0);
}
template <int K, int I, typename T> SO100_HD T pmul(T x) {                                      // P[I] * x
    constexpr double a = so100g::LINK_P[K][I];
    if constexpr (a != 0) return T(a)*x; else return T(0);
}
template <int K, int I, int J, typename T> SO100_HD T ppmul(T x) {                              // P[I] P[J] * x
    constexpr double a = so100g::LINK_P[K][I]*so100g::LINK_P[K][J];
    if constexpr (a != 0) return T(a)*x; else return T(0);
}
template <int K, typename T> SO100_HD void pcross(const T v[3], T r[3]) {                    // r = p_K x v
    const T v0 = v[0], v1 = v[1], v2 = v[2];
    r[0] = pdiff<K, 1, 2>(v2, v1); r[1] = pdiff<K, 2, 0>(v0, v2); r[2] = pdiff<K, 0, 1>(v1, v0);
}
template <int K, typename T> SO100_HD void crossp(const T v[3], T r[3]) {                    // r = v x p_K
    const T v0 = v[0], v1 = v[1], v2 = v[2];
    r[0] = pdiff<K, 2, 1>(v1, v2); r[1] = pdiff<K, 0, 2>(v2, v0); r[2] = pdiff<K, 1, 0>(v0, v1);
}
// constant link rotation C_K (child->parent) and its transpose
template <int K, typename T> SO100_HD void cmat(const T x[3], T y[3]) {
    const T x0 = x[0], x1 = x[1], x2 = x[2];
    y[0] = cdot3<K, 0, 1, 2>(x0, x1, x2); y[1] = cdot3<K, 3, 4, 5>(x0, x1, x2); y[2] = cdot3<K, 6, 7, 8>(x0, x1, x2);
}
template <int K, typename T> SO100_HD void cmat_T(const T x[3], T y[3]) {
    const T x0 = x[0], x1 = x[1], x2 = x[2];
    y[0] = cdot3<K, 0, 3, 6>(x0, x1, x2); y[1] = cdot3<K, 1, 4, 7>(x0, x1, x2); y[2] = cdot3<K, 2, 5, 8>(x0, x1, x2);
}
// parent coords -> link K coords (E_K^T) and back (E_K), E_K = C_K Rot(axis_K, q_K)
template <int K, typename T> SO100_HD void to_child(const T x[3], T s, T c, T y[3]) {
    T t[3]; cmat_T<K>(x, t); rot_axis_T<so100g::LINK_AXIS[K]>(t, s, c, y);
}
template <int K, typename T> SO100_HD void to_parent(const T x[3], T s, T c, T y[3]) {
    T t[3]; rot_axis<so100g::LINK_AXIS[K]>(x, s, c, t); cmat<K>(t, y);
}

// symmetric 3x3 stored (xx, yy, zz, xy, xz, yz)
template <typename T> SO100_HD void sym_mul(const T I[6], const T v[3], T r[3]) {
    const T x = I[0]*v[0] + I[3]*v[1] + I[4]*v[2];
    const T y = I[3]*v[0] + I[1]*v[1] + I[5]*v[2];
    const T z = I[4]*v[0] + I[5]*v[1] + I[2]*v[2];
    r[0] = x; r[1] = y; r[2] = z;
}
SO100_HD constexpr int sym_idx(int i, int j) {
    return i == j ? i : ((i + j == 1) ? 3 : ((i + j == 2) ? 4 : 5));
}
// B = C_K A C_K^T with the constant, sparse link rotation C_K
template <int K, typename T> SO100_HD void sym_similarity_c(const T A[6], T Bm[6]) {
    T t[9];                                                   // t = C A
#define SO100_ROW(i) \
    t[3*i]   = cdot3<K, 3*i, 3*i+1, 3*i+2>(A[0], A[3], A[4]); \
    t[3*i+1] = cdot3<K, 3*i, 3*i+1, 3*i+2>(A[3], A[1], A[5]); \
    t[3*i+2] = cdot3<K, 3*i, 3*i+1, 3*i+2>(A[4], A[5], A[2]);
    SO100_ROW(0) SO100_ROW(1) SO100_ROW(2)
#undef SO100_ROW
    Bm[0] = cdot3<K, 0, 1, 2>(t[0], t[1], t[2]);
    Bm[1] = cdot3<K, 3, 4, 5>(t[3], t[4], t[5]);
    Bm[2] = cdot3<K, 6, 7, 8>(t[6], t[7], t[8]);
    Bm[3] = cdot3<K, 3, 4, 5>(t[0], t[1], t[2]);
    Bm[4] = cdot3<K, 6, 7, 8>(t[0], t[1], t[2]);
    Bm[5] = cdot3<K, 6, 7, 8>(t[3], t[4], t[5]);
}
// similarity by a rotation about coordinate axis A: B = Rot A Rot^T
template <int AX, typename T> SO100_HD void sym_rot_axis(const T A[6], T s, T c, T Bm[6]) {
    constexpr int B = (AX + 1) % 3, C = (AX + 2) % 3;
    constexpr int iab = sym_idx(AX, B), iac = sym_idx(AX, C), ibc = sym_idx(B, C);
    const T aab = A[iab], aac = A[iac], abb = A[B], acc = A[C], abc = A[ibc];
    const T cc = c*c, ss = s*s, sc = s*c;
    Bm[AX] = A[AX];
    Bm[iab] = c*aab - s*aac;
    Bm[iac] = s*aab + c*aac;
    Bm[B] = cc*abb - T(2)*sc*abc + ss*acc;
    Bm[C] = ss*abb + T(2)*sc*abc + cc*acc;
    Bm[ibc] = sc*(abb - acc) + (cc - ss)*abc;
}

// ---------------------------------------------------------------------------------------------
// arm dynamics: M (packed lower triangle, armature included) and bias = C(q,v) + g(q)
// ---------------------------------------------------------------------------------------------
template <typename T> struct Arm {
    T s[6], c[6];          // sin/cos of the joint angles
    T M[21];               // M[i(i+1)/2 + j], j <= i   (after arm_factor: the LDL^T factor)
    T bias[6];
    T Dinv[6], Minv[21];   // arm_factor: 1/D of the factor; explicit inverse (constrained variants only)
};

template <int K, typename T> struct LinkFwd {            // RNEA forward step for link K
    SO100_HD static void run(const T s[6], const T c[6], const T v[6], T w[3], T wd[3], T a[3],
                             T f[6][3], T n[6][3]) {
        constexpr int AX = so100g::LINK_AXIS[K];
        constexpr int B = (AX + 1) % 3, C = (AX + 2) % 3;
        const T qd = v[K];
        if constexpr (K == 0) {
            // the base is at rest: w_p = wd_p = 0, so the origin acceleration is the base's (gravity) and
            // w = axis qd, wd = 0 (the generic code below would spend ~40 instructions multiplying zeros)
            T ao[3] = { a[0], a[1], a[2] };
            to_child<K>(ao, s[K], c[K], a);
            w[AX] = qd; w[B] = T(0); w[C] = T(0);
            wd[0] = T(0); wd[1] = T(0); wd[2] = T(0);
        } else {
            // acceleration of this link's origin, in parent coords: a + wd x p + w x (w x p)
            T t1[3], t2[3], ao[3];
            crossp<K>(wd, t1); crossp<K>(w, t2); cross(w, t2, t2);
            ao[0] = a[0] + t1[0] + t2[0]; ao[1] = a[1] + t1[1] + t2[1]; ao[2] = a[2] + t1[2] + t2[2];
            T wc[3], wdc[3];
            to_child<K>(w, s[K], c[K], wc);
            to_child<K>(wd, s[K], c[K], wdc);
            to_child<K>(ao, s[K], c[K], a);
            // wd_k = E^T wd_p + (E^T w_p) x (axis qd);  w_k = E^T w_p + axis qd
            wdc[B] += wc[C]*qd; wdc[C] -= wc[B]*qd;
            wc[AX] += qd;
            w[0] = wc[0]; w[1] = wc[1]; w[2] = wc[2];
            wd[0] = wdc[0]; wd[1] = wdc[1]; wd[2] = wdc[2];
        }
        // spatial force about the link origin:  f = m a + wd x h + w x (w x h),  n = Io wd + w x (Io w) + h x a
        const T h[3] = { T(so100g::LINK_H[K][0]), T(so100g::LINK_H[K][1]), T(so100g::LINK_H[K][2]) };
        const T Io[6] = { T(so100g::LINK_IORG[K][0]), T(so100g::LINK_IORG[K][1]), T(so100g::LINK_IORG[K][2]),
                          T(so100g::LINK_IORG[K][3]), T(so100g::LINK_IORG[K][4]), T(so100g::LINK_IORG[K][5]) };
        const T m = T(so100g::LINK_MASS[K]);
        T u1[3], u2[3], Iw[3];
        cross(wd, h, u1); cross(w, h, u2); cross(w, u2, u2);
        f[K][0] = m*a[0] + u1[0] + u2[0]; f[K][1] = m*a[1] + u1[1] + u2[1]; f[K][2] = m*a[2] + u1[2] + u2[2];
        sym_mul(Io, wd, u1); sym_mul(Io, w, Iw); cross(w, Iw, u2);
        T u3[3]; cross(h, a, u3);
        n[K][0] = u1[0] + u2[0] + u3[0]; n[K][1] = u1[1] + u2[1] + u3[1]; n[K][2] = u1[2] + u2[2] + u3[2];
    }
};

template <int K, typename T> struct LinkBwd {            // RNEA backward step: project, pass to parent
    SO100_HD static void run(const T s[6], const T c[6], T f[6][3], T n[6][3], T bias[6]) {
        bias[K] = n[K][so100g::LINK_AXIS[K]];
        if constexpr (K > 0) {
            T fp[3], np[3], t[3];
            to_parent<K>(f[K], s[K], c[K], fp);
            to_parent<K>(n[K], s[K], c[K], np);
            pcross<K>(fp, t);
#pragma unroll
            for (int i = 0; i < 3; i++) { f[K-1][i] += fp[i]; n[K-1][i] += np[i] + t[i]; }
        }
    }
};

// composite inertia of the subtree rooted at link K, in link-K coords about the link-K origin
template <typename T> struct Composite { T m, h[3], I[6]; };

template <int K, typename T> struct LinkCrb {
    // add link K's composite (already complete) into its parent's, then emit column K of M
    SO100_HD static void run(const T s[6], const T c[6], Composite<T> cmp[6], T M[21]) {
        constexpr int AX = so100g::LINK_AXIS[K];
        // ---- column K: unit acceleration about axis K of the composite body
        {
            constexpr int B = (AX + 1) % 3, C = (AX + 2) % 3;
            T f[3], n[3];
            // f = axis x h,  n = Ic[:,axis]
            f[AX] = T(0); f[B] = -cmp[K].h[C]; f[C] = cmp[K].h[B];
            n[0] = cmp[K].I[sym_idx(0, AX)]; n[1] = cmp[K].I[sym_idx(1, AX)]; n[2] = cmp[K].I[sym_idx(2, AX)];
            M[K*(K+1)/2 + K] = n[AX] + T(so100g::ARMATURE);
            walk<K>(s, c, f, n, M);
        }
        // ---- accumulate into the parent
        if constexpr (K > 0) {
            T hp[3], Ir[6], Ip[6];
            to_parent<K>(cmp[K].h, s[K], c[K], hp);
            sym_rot_axis<AX>(cmp[K].I, s[K], c[K], Ir);
            sym_similarity_c<K>(Ir, Ip);
            const T m = cmp[K].m;
            constexpr double pp_c = so100g::LINK_P[K][0]*so100g::LINK_P[K][0] + so100g::LINK_P[K][1]*so100g::LINK_P[K][1]
                                  + so100g::LINK_P[K][2]*so100g::LINK_P[K][2];
            const T ph = pmul<K, 0>(hp[0]) + pmul<K, 1>(hp[1]) + pmul<K, 2>(hp[2]);
            const T dg = m*T(pp_c) + T(2)*ph;
            Composite<T>& P = cmp[K-1];
            P.I[0] += Ip[0] + dg - ppmul<K, 0, 0>(m) - T(2)*pmul<K, 0>(hp[0]);
            P.I[1] += Ip[1] + dg - ppmul<K, 1, 1>(m) - T(2)*pmul<K, 1>(hp[1]);
            P.I[2] += Ip[2] + dg - ppmul<K, 2, 2>(m) - T(2)*pmul<K, 2>(hp[2]);
            P.I[3] += Ip[3] - ppmul<K, 0, 1>(m) - pmul<K, 0>(hp[1]) - pmul<K, 1>(hp[0]);
            P.I[4] += Ip[4] - ppmul<K, 0, 2>(m) - pmul<K, 0>(hp[2]) - pmul<K, 2>(hp[0]);
            P.I[5] += Ip[5] - ppmul<K, 1, 2>(m) - pmul<K, 1>(hp[2]) - pmul<K, 2>(hp[1]);
            P.h[0] += hp[0] + pmul<K, 0>(m); P.h[1] += hp[1] + pmul<K, 1>(m); P.h[2] += hp[2] + pmul<K, 2>(m);
            P.m += m;
        }
    }
    // carry (f, n) from frame J+1 to frame J for J = K-1 .. 0 and read M[K][J]
    template <int J1> SO100_HD static void walk(const T s[6], const T c[6], T f[3], T n[3], T M[21]) {
        if constexpr (J1 > 0) {
            T fp[3], np[3], t[3];
            to_parent<J1>(f, s[J1], c[J1], fp);
            to_parent<J1>(n, s[J1], c[J1], np);
            pcross<J1>(fp, t);
            f[0] = fp[0]; f[1] = fp[1]; f[2] = fp[2];
            n[0] = np[0] + t[0]; n[1] = np[1] + t[1]; n[2] = np[2] + t[2];
            M[K*(K+1)/2 + (J1-1)] = n[so100g::LINK_AXIS[J1-1]];
            walk<J1-1>(s, c, f, n, M);
        }
    }
};

// stages of arm_dynamics (separately callable: the persistent rollout kernel runs arm_bias and arm_mass on different waves)
template <typename T> SO100_HD void arm_trig(const T q[6], Arm<T>& A) {
#pragma unroll
    for (int k = 0; k < 6; k++) tsincos<T>(q[k], A.s[k], A.c[k]);
}
// sin/cos of q + dq from those of q: a rotation by dq with Taylor series for sin dq / cos dq (|dq| = h |qd| < 0.1 rad:
// truncation < 1e-11).  fp32 only: 66 instructions instead of ~150 for six fresh sin/cos pairs; the fp64 (host-test)
// instantiation recomputes exact trig so that the formulation-equivalence tests keep their 1e-13 tolerances.
// Round-off accumulates over at most frame_skip-1 updates (~2e-7): arm_trig() resynchronises at every env step.
template <typename T> SO100_HD void arm_trig_update(const T q[6], const T dq[6], Arm<T>& A) {
    if constexpr (sizeof(T) == 8) { arm_trig(q, A); }
    else {
#pragma unroll
        for (int k = 0; k < 6; k++) {
            const T d = dq[k], d2 = d*d;
            const T sd = d*(T(1) + d2*(T(-1.0/6.0) + d2*T(1.0/120.0)));
            const T cd = T(1) + d2*(T(-0.5) + d2*(T(1.0/24.0) + d2*T(-1.0/720.0)));
            const T sn = A.s[k]*cd + A.c[k]*sd, cs = A.c[k]*cd - A.s[k]*sd;
            A.s[k] = sn; A.c[k] = cs;
        }
    }
}
template <typename T> SO100_HD void arm_bias(const T v[6], Arm<T>& A) {
    // RNEA (bias): base at rest, gravity folded in as a base acceleration of +g along world z
    T w[3] = { T(0), T(0), T(0) }, wd[3] = { T(0), T(0), T(0) }, a[3] = { T(0), T(0), T(so100g::GRAVITY) };
    T f[6][3], n[6][3];
    LinkFwd<0, T>::run(A.s, A.c, v, w, wd, a, f, n);
    LinkFwd<1, T>::run(A.s, A.c, v, w, wd, a, f, n);
    LinkFwd<2, T>::run(A.s, A.c, v, w, wd, a, f, n);
    LinkFwd<3, T>::run(A.s, A.c, v, w, wd, a, f, n);
    LinkFwd<4, T>::run(A.s, A.c, v, w, wd, a, f, n);
    LinkFwd<5, T>::run(A.s, A.c, v, w, wd, a, f, n);
    LinkBwd<5, T>::run(A.s, A.c, f, n, A.bias);
    LinkBwd<4, T>::run(A.s, A.c, f, n, A.bias);
    LinkBwd<3, T>::run(A.s, A.c, f, n, A.bias);
    LinkBwd<2, T>::run(A.s, A.c, f, n, A.bias);
    LinkBwd<1, T>::run(A.s, A.c, f, n, A.bias);
    LinkBwd<0, T>::run(A.s, A.c, f, n, A.bias);
}
template <typename T> SO100_HD void arm_mass(Arm<T>& A) {
    // CRBA
    Composite<T> cmp[6];
#pragma unroll
    for (int k = 0; k < 6; k++) {
        cmp[k].m = T(so100g::LINK_MASS[k]);
#pragma unroll
        for (int i = 0; i < 3; i++) cmp[k].h[i] = T(so100g::LINK_H[k][i]);
#pragma unroll
        for (int i = 0; i < 6; i++) cmp[k].I[i] = T(so100g::LINK_IORG[k][i]);
    }
    LinkCrb<5, T>::run(A.s, A.c, cmp, A.M);
    LinkCrb<4, T>::run(A.s, A.c, cmp, A.M);
    LinkCrb<3, T>::run(A.s, A.c, cmp, A.M);
    LinkCrb<2, T>::run(A.s, A.c, cmp, A.M);
    LinkCrb<1, T>::run(A.s, A.c, cmp, A.M);
    LinkCrb<0, T>::run(A.s, A.c, cmp, A.M);
}
template <typename T> SO100_HD void arm_dynamics(const T q[6], const T v[6], Arm<T>& A) {
    arm_trig(q, A);
    arm_bias(v, A);
    arm_mass(A);
}

// ---------------------------------------------------------------------------------------------
// 6x6 SPD: LDL^T in place on the packed lower triangle, then the explicit inverse
// ---------------------------------------------------------------------------------------------
#define SO100_TRI(i, j) ((i)*((i)+1)/2 + (j))
template <typename T> SO100_HD void ldl6(T M[21], T Dinv[6]) {     // M <- L (unit lower, strict part), D on the diagonal
#pragma unroll
    for (int j = 0; j < 6; j++) {
        T d = M[SO100_TRI(j, j)];
#pragma unroll
        for (int k = 0; k < j; k++) d -= M[SO100_TRI(j, k)]*M[SO100_TRI(j, k)]*M[SO100_TRI(k, k)];
        M[SO100_TRI(j, j)] = d;
        Dinv[j] = trcp(d);
#pragma unroll
        for (int i = j + 1; i < 6; i++) {
            T t = M[SO100_TRI(i, j)];
#pragma unroll
            for (int k = 0; k < j; k++) t -= M[SO100_TRI(i, k)]*M[SO100_TRI(j, k)]*M[SO100_TRI(k, k)];
            M[SO100_TRI(i, j)] = t*Dinv[j];
        }
    }
}
template <typename T> SO100_HD void ldl6_solve(const T L[21], const T Dinv[6], T x[6]) {
#pragma unroll
    for (int i = 1; i < 6; i++)
#pragma unroll
        for (int k = 0; k < i; k++) x[i] -= L[SO100_TRI(i, k)]*x[k];
#pragma unroll
    for (int i = 0; i < 6; i++) x[i] *= Dinv[i];
#pragma unroll
    for (int i = 4; i >= 0; i--)
#pragma unroll
        for (int k = i + 1; k < 6; k++) x[i] -= L[SO100_TRI(k, i)]*x[k];
}
// Minv (packed lower) = L^-T D^-1 L^-1
template <typename T> SO100_HD void ldl6_inverse(const T L[21], const T Dinv[6], T Minv[21]) {
    T Li[21];                                  // Li = L^-1 (unit lower)
#pragma unroll
    for (int j = 0; j < 6; j++) {
        Li[SO100_TRI(j, j)] = T(1);
#pragma unroll
        for (int i = j + 1; i < 6; i++) {
            T t = -L[SO100_TRI(i, j)];
#pragma unroll
            for (int k = j + 1; k < i; k++) t -= L[SO100_TRI(i, k)]*Li[SO100_TRI(k, j)];
            Li[SO100_TRI(i, j)] = t;
        }
    }
#pragma unroll
    for (int i = 0; i < 6; i++)
#pragma unroll
        for (int j = 0; j <= i; j++) {
            T t = T(0);
#pragma unroll
            for (int k = i; k < 6; k++) t += Li[SO100_TRI(k, i)]*Dinv[k]*Li[SO100_TRI(k, j)];
            Minv[SO100_TRI(i, j)] = t;
        }
}
template <typename T> SO100_HD T sym6(const T A[21], int i, int j) { return i >= j ? A[SO100_TRI(i, j)] : A[SO100_TRI(j, i)]; }

// impedance d(r) of the default solimp (0.9, 0.95, 0.001, 0.5, 2), r = |pos - margin|
template <typename T> SO100_HD T impedance(T r) {
    const T x = r * T(1.0/so100g::SOLIMP_WIDTH);
    const T d0 = T(so100g::SOLIMP_D0), dm = T(so100g::SOLIMP_DMAX);
    if (x >= T(1)) return dm;
    if (x <= T(0)) return d0;
    const T y = x <= T(0.5) ? T(2)*x*x : T(1) - T(2)*(T(1) - x)*(T(1) - x);
    return d0 + y*(dm - d0);
}

// ---------------------------------------------------------------------------------------------
// one arm substep: forward dynamics + constraint solve + semi-implicit Euler
//   q, v        : joint positions / velocities (updated)
//   ctrl        : servo targets
//   qc          : running compensation of the position integration (Kahan): fp32 q += h v loses up to half an
//                 ulp of q per substep, a random walk of ~1e-5 rad over 16000 substeps that Env01's measured-angle
//                 ctrl never pulls back; with the compensation q carries ~48 significant bits across substeps
//   ff, fl      : friction-loss / limit row forces of the previous substep (PGS warm start; updated)
//   iters       : PGS sweeps
// ---------------------------------------------------------------------------------------------
template <typename T>
SO100_HD void arm_factor(unsigned flags, Arm<T>& A) {
    // needs only A.M: LDL^T in place (+ the explicit inverse when constraint rows are simulated)
    ldl6(A.M, A.Dinv);
    if ((flags & (F_FRICTIONLOSS | F_LIMITS | 16u | 32u | 64u)) != 0u) ldl6_inverse(A.M, A.Dinv, A.Minv);      // 16 | 32 | 64: the contact flags (so100_contact.hpp)
}

// The solve of one arm substep in separately callable stages (so100_contact.hpp puts a contact-aware Newton solve between
// arm_rows and arm_integrate for the lanes that have finger-pad contacts):
//   arm_tau       : mj_fwdActuation - bias                      arm_rows : friction-loss / limit rows + a0 = M^-1 tau
//   arm_pgs       : block Gauss-Seidel over the joints -> acc   arm_integrate : semi-implicit Euler (Kahan-compensated)
template <typename T> struct ArmRows {
    T a0[6];                                   // qacc_smooth of the arm
    T bf[6], Rf[6], bl[6], Rl[6], sg[6], fmax_[6];
    T cfv[6], clv[6];                          // the rows' constants without the a0 term: -aref = B Jv (+ K imp dist)
    T Df[6], Dl[6];                            // 1/R of the friction rows (0 without friction loss) and of the ACTIVE limit rows (0 otherwise)
};

template <typename T>
SO100_HD void arm_tau(const T q[6], const T v[6], const T ctrl[6], const Arm<T>& A, T tau[6]) {
    // mj_fwdActuation: position servo kp (u - q) - kv qd, u clamped to ctrlrange, force to forcerange
#pragma unroll
    for (int i = 0; i < 6; i++) {
        const T u = tclamp(ctrl[i], T(-so100g::ACT_CTRL), T(so100g::ACT_CTRL));
        T f = T(so100g::ACT_KP)*u - T(so100g::ACT_KP)*q[i] - T(so100g::ACT_KV[i])*v[i];
        f = tclamp(f, T(-so100g::ACT_FORCE), T(so100g::ACT_FORCE));
        tau[i] = f - A.bias[i];
    }
}

template <typename T>
SO100_HD void arm_rows(const T q[6], const T v[6], const T tau[6], T ff[6], T fl[6], unsigned flags, const Arm<T>& A, ArmRows<T>& r) {
    const T* Minv = A.Minv;
#pragma unroll
    for (int i = 0; i < 6; i++) {
        T t = T(0);
#pragma unroll
        for (int j = 0; j < 6; j++) t += sym6(Minv, i, j)*tau[j];
        r.a0[i] = t;
    }
    // rows: friction (J = e_i, |f| <= frictionloss) then limits (J = sg_i e_i, f >= 0)
    const T Bd = T(so100g::SOLREF_B), Kd = T(so100g::SOLREF_K);
#pragma unroll
    for (int i = 0; i < 6; i++) {
        const bool fr = (flags & F_FRICTIONLOSS) != 0u;
        r.Rf[i] = T((1.0 - so100g::SOLIMP_D0)/so100g::SOLIMP_D0 * so100g::DOF_INVWEIGHT0[i]);
        r.cfv[i] = Bd*v[i];
        r.bf[i] = r.a0[i] + Bd*v[i];                          // J a0 - aref, aref = -B v
        r.fmax_[i] = fr ? T(so100g::FRICTIONLOSS) : T(0);
        if (!fr) ff[i] = T(0);
        const T dlo = q[i] - T(so100g::JNT_RANGE[i][0]), dhi = T(so100g::JNT_RANGE[i][1]) - q[i];
        const bool lo = dlo < T(0), hi = dhi < T(0);
        const bool act = (flags & F_LIMITS) != 0u && (lo || hi);
        const T dist = lo ? dlo : dhi;
        r.sg[i] = lo ? T(1) : T(-1);
        const T imp = impedance(tabs(dist));
        r.Rl[i] = (T(1) - imp)*trcp(imp) * T(so100g::DOF_INVWEIGHT0[i]);
        r.clv[i] = Bd*r.sg[i]*v[i] + Kd*imp*dist;
        r.bl[i] = r.sg[i]*r.a0[i] + Bd*r.sg[i]*v[i] + Kd*imp*dist;
        if (!act) { fl[i] = T(0); r.sg[i] = T(0); }          // inactive row: force pinned at 0
        else fl[i] = tmax(fl[i], T(0));
    }
}

// the rows' constants alone (what a primal solve needs: no a0, no warm-start clamping)
template <typename T>
SO100_HD void arm_row_consts(const T q[6], const T v[6], unsigned flags, ArmRows<T>& r) {
    const T Bd = T(so100g::SOLREF_B), Kd = T(so100g::SOLREF_K);
#pragma unroll
    for (int i = 0; i < 6; i++) {
        r.Rf[i] = T((1.0 - so100g::SOLIMP_D0)/so100g::SOLIMP_D0 * so100g::DOF_INVWEIGHT0[i]);
        r.cfv[i] = Bd*v[i];
        r.fmax_[i] = (flags & F_FRICTIONLOSS) != 0u ? T(so100g::FRICTIONLOSS) : T(0);
        const T dlo = q[i] - T(so100g::JNT_RANGE[i][0]), dhi = T(so100g::JNT_RANGE[i][1]) - q[i];
        const bool lo = dlo < T(0), hi = dhi < T(0);
        const bool act = (flags & F_LIMITS) != 0u && (lo || hi);
        const T dist = lo ? dlo : dhi;
        const T sg = lo ? T(1) : T(-1);
        const T imp = impedance(tabs(dist));
        r.Rl[i] = (T(1) - imp)*trcp(imp) * T(so100g::DOF_INVWEIGHT0[i]);
        r.clv[i] = Bd*sg*v[i] + Kd*imp*dist;
        r.sg[i] = act ? sg : T(0);
        r.Df[i] = (flags & F_FRICTIONLOSS) != 0u ? T(so100g::SOLIMP_D0/((1.0 - so100g::SOLIMP_D0)*so100g::DOF_INVWEIGHT0[i])) : T(0);
        r.Dl[i] = act ? trcp(r.Rl[i]) : T(0);
    }
}
// the same constants for a primal solve on ANOTHER wave, from what the wave that holds q published: sD[i] = sg_i / R_limit,i (0 = limit row
// inactive) and clv (physics_phase_mw: wave 1 prepares them behind RNEA, the contact wave reads 12 words instead of redoing ~150 instructions)
template <typename T>
SO100_HD void arm_row_consts_from(const T v[6], unsigned flags, const T sD[6], const T clv[6], ArmRows<T>& r) {
#pragma unroll
    for (int i = 0; i < 6; i++) {
        r.Rf[i] = T((1.0 - so100g::SOLIMP_D0)/so100g::SOLIMP_D0 * so100g::DOF_INVWEIGHT0[i]);
        r.cfv[i] = T(so100g::SOLREF_B)*v[i];
        r.fmax_[i] = (flags & F_FRICTIONLOSS) != 0u ? T(so100g::FRICTIONLOSS) : T(0);
        r.Df[i] = (flags & F_FRICTIONLOSS) != 0u ? T(so100g::SOLIMP_D0/((1.0 - so100g::SOLIMP_D0)*so100g::DOF_INVWEIGHT0[i])) : T(0);
        r.sg[i] = sD[i] > T(0) ? T(1) : (sD[i] < T(0) ? T(-1) : T(0));
        r.Dl[i] = tabs(sD[i]);
        r.clv[i] = clv[i];
        r.Rl[i] = T(0);                                      // (not used by the primal solve)
    }
}
// friction-loss / limit row forces that go with a given acceleration (the block PGS's warm-start memory after a primal solve)
template <typename T>
SO100_HD void arm_row_forces(const ArmRows<T>& r, const T acc[6], T ff[6], T fl[6]) {
#pragma unroll
    for (int i = 0; i < 6; i++) {
        const T F = r.fmax_[i], R = r.Rf[i], jar = acc[i] + r.cfv[i];
        ff[i] = F > T(0) ? tclamp(-jar*trcp(R), -F, F) : T(0);
        const T jl = r.sg[i]*acc[i] + r.clv[i];
        fl[i] = (r.sg[i] != T(0) && jl < T(0)) ? -jl*trcp(r.Rl[i]) : T(0);
    }
}

template <typename T>
SO100_HD void arm_pgs(T ff[6], T fl[6], int iters, const Arm<T>& A, const ArmRows<T>& r, T acc[6], T& residual) {
    const T* Minv = A.Minv;
    const T* Rf = r.Rf; const T* Rl = r.Rl; const T* sg = r.sg; const T* bf = r.bf; const T* bl = r.bl; const T* fmax_ = r.fmax_;
    // Block Gauss-Seidel over JOINTS: the friction row and the limit row of one joint are collinear
    // (J = e_i and sg_i e_i), so scalar PGS crawls when both are active (rate a^2/((a+Rf)(a+Rl)) ~ 0.84).
    // Each joint's 2-row box QP  min 1/2 [f l] [[a+Rf, sg a],[sg a, a+Rl]] [f l]' + [f l].[cf cl],
    // |f| <= fmax, l >= 0  is solved exactly by enumerating its active sets; the coupling between joints
    // (armature-dominated M => nearly diagonal Minv) then converges in a few sweeps.
    T tq[6];                                              // joint-space constraint torque J^T f
    T rAf[6], rAl[6], rdet[6];                            // sweep-invariant reciprocals of the 2x2 blocks
    // A joint whose limit row is inactive in EVERY lane of the wavefront (the usual case for most joints: 16 .. 64 envs per wave)
    // takes the scalar friction update only -- a wave-uniform branch, the same arithmetic for the lanes (case 1 below).
    bool lim[6];
#pragma unroll
    for (int i = 0; i < 6; i++) {
        tq[i] = ff[i] + sg[i]*fl[i];
        lim[i] = wave_any(sg[i] != T(0));
        const T a = sym6(Minv, i, i), Af = a + Rf[i];
        rAf[i] = trcp(Af);
        if (lim[i]) { const T Al = a + Rl[i], cx = sg[i]*a; rAl[i] = trcp(Al); rdet[i] = trcp(Af*Al - cx*cx); }
        else { rAl[i] = T(0); rdet[i] = T(0); }
    }
    T change = T(0);                                      // largest |d acc_i| caused by the LAST sweep: the residual a caller can watch
    for (int it = 0; it < iters; it++) {
        change = T(0);
#pragma unroll
        for (int i = 0; i < 6; i++) {
            T w = T(0);
#pragma unroll
            for (int j = 0; j < 6; j++) w += sym6(Minv, i, j)*tq[j];
            const T a = sym6(Minv, i, i);
            const T wo = w - a*tq[i];                     // contribution of the other joints
            const T cf = bf[i] + wo, F = fmax_[i];
            if (!lim[i]) {                                // no lane has this joint at a limit: case (1) alone
                const T fn = tclamp(-cf*rAf[i], -F, F);
                ff[i] = fn; fl[i] = T(0);
                change = tmax(change, tabs(fn - tq[i])*a);
                tq[i] = fn;
                continue;
            }
            const T cl = bl[i] + sg[i]*wo;
            const T Af = a + Rf[i], Al = a + Rl[i], cx = sg[i]*a;
            // (1) limit row inactive
            const T f1 = tclamp(-cf*rAf[i], -F, F);
            const bool ok1 = (sg[i] == T(0)) || (cl + cx*f1 >= T(0));
            // (2) both interior
            const T f2 = (cx*cl - cf*Al)*rdet[i], l2 = (cx*cf - Af*cl)*rdet[i];
            const bool ok2 = tabs(f2) <= F && l2 >= T(0);
            // (3) friction saturated, limit active
            const T lp = tmax(-(cl + cx*F)*rAl[i], T(0)), lm = tmax(-(cl - cx*F)*rAl[i], T(0));
            const bool okp = cf + Af*F + cx*lp <= T(0);
            const T f3 = okp ? F : -F, l3 = okp ? lp : lm;
            const T fn = ok1 ? f1 : (ok2 ? f2 : f3);
            const T ln = ok1 ? T(0) : (ok2 ? l2 : l3);
            ff[i] = fn; fl[i] = ln;
            const T tn = fn + sg[i]*ln;
            change = tmax(change, tabs(tn - tq[i])*a);
            tq[i] = tn;
        }
    }
    residual = change;
#pragma unroll
    for (int i = 0; i < 6; i++) {
        T t = r.a0[i];
#pragma unroll
        for (int j = 0; j < 6; j++) t += sym6(Minv, i, j)*tq[j];
        acc[i] = t;
    }
}

template <typename T>
SO100_HD void arm_integrate(T q[6], T v[6], T qc[6], const T acc[6], T dq[6]) {
    const T h = T(so100g::TIMESTEP);
#pragma unroll
    for (int i = 0; i < 6; i++) {
        v[i] += h*acc[i];
        const T y = h*v[i] - qc[i], t = q[i] + y;
        qc[i] = (t - q[i]) - y;
        q[i] = t;
        dq[i] = y;                              // the (compensated) increment: what the next substep's trig update rotates by
    }
}

template <typename T>
SO100_HD void arm_solve_integrate(T q[6], T v[6], T qc[6], const T ctrl[6], T ff[6], T fl[6], unsigned flags, int iters, Arm<T>& A, T dq[6],
                                  T* residual = nullptr, T* acc_out = nullptr) {
    // (A.M / A.Dinv / A.Minv hold this substep's factorised mass matrix, A.bias its bias force)
    T tau[6], acc[6];
    arm_tau(q, v, ctrl, A, tau);
    const bool constrained = (flags & (F_FRICTIONLOSS | F_LIMITS)) != 0u;
    if (!constrained) {
#pragma unroll
        for (int i = 0; i < 6; i++) acc[i] = tau[i];
        ldl6_solve(A.M, A.Dinv, acc);
    } else {
        ArmRows<T> r;
        arm_rows(q, v, tau, ff, fl, flags, A, r);
        T res;
        arm_pgs(ff, fl, iters, A, r, acc, res);
        if (residual) *residual = tmax(*residual, res);
    }
    if (acc_out) {
#pragma unroll
        for (int i = 0; i < 6; i++) acc_out[i] = acc[i];
    }
    arm_integrate(q, v, qc, acc, dq);
}

template <typename T>
SO100_HD void arm_finish(T q[6], T v[6], T qc[6], const T ctrl[6], T ff[6], T fl[6], unsigned flags, int iters, Arm<T>& A, T dq[6], T* residual = nullptr) {
    arm_factor(flags, A);
    arm_solve_integrate(q, v, qc, ctrl, ff, fl, flags, iters, A, dq, residual);
}

template <typename T>
SO100_HD void arm_substep(T q[6], T v[6], T qc[6], const T ctrl[6], T ff[6], T fl[6], unsigned flags, int iters, Arm<T>& A,
                          bool first, T dq[6], T* residual = nullptr) {
    // first substep of an env step: exact sin/cos; later ones: incremental update by the previous substep's increment
    if (first) arm_trig(q, A); else arm_trig_update(q, dq, A);
    arm_bias(v, A);
    arm_mass(A);
    arm_finish(q, v, qc, ctrl, ff, fl, flags, iters, A, dq, residual);
}

// ---------------------------------------------------------------------------------------------
// world-frame poses the task layer reads (computed from the sin/cos of the substep they belong to)
// ---------------------------------------------------------------------------------------------
template <typename T> struct TaskPoses {
    T wrist[3];            // xpos of Wrist_Pitch_Roll (link 3)
    T jaw_pos[3];          // xpos of Fixed_Jaw (link 4)
    T jaw_mat[9];          // xmat of Fixed_Jaw, row-major
    T cam_pos[3], cam_mat[9];
};
template <int K, typename T> SO100_HD void fk_link(const T s[6], const T c[6], T pos[3], T R[9]) {
    // pos += R p_K ; R <- R C_K Rot_K
    const T p[3] = { T(so100g::LINK_P[K][0]), T(so100g::LINK_P[K][1]), T(so100g::LINK_P[K][2]) };
#pragma unroll
    for (int i = 0; i < 3; i++) pos[i] += R[3*i]*p[0] + R[3*i+1]*p[1] + R[3*i+2]*p[2];
#pragma unroll
    for (int i = 0; i < 3; i++) {          // each row r of R: r <- (r C) Rot  == to_child applied to the row
        T row[3] = { R[3*i], R[3*i+1], R[3*i+2] }, out[3];
        to_child<K>(row, s[K], c[K], out);
        R[3*i] = out[0]; R[3*i+1] = out[1]; R[3*i+2] = out[2];
    }
}
template <typename T> SO100_HD void task_poses(const T s[6], const T c[6], bool want_cam, TaskPoses<T>& P) {
    T pos[3] = { T(0), T(0), T(0) };
    T R[9] = { T(1), T(0), T(0), T(0), T(1), T(0), T(0), T(0), T(1) };
    fk_link<0>(s, c, pos, R); fk_link<1>(s, c, pos, R); fk_link<2>(s, c, pos, R);
    fk_link<3>(s, c, pos, R);
    P.wrist[0] = pos[0]; P.wrist[1] = pos[1]; P.wrist[2] = pos[2];
    fk_link<4>(s, c, pos, R);
#pragma unroll
    for (int i = 0; i < 3; i++) P.jaw_pos[i] = pos[i];
#pragma unroll
    for (int i = 0; i < 9; i++) P.jaw_mat[i] = R[i];
    if (want_cam) {
#pragma unroll
        for (int i = 0; i < 3; i++) {
            P.cam_pos[i] = pos[i] + R[3*i]*T(so100g::CAM_P[0]) + R[3*i+1]*T(so100g::CAM_P[1]) + R[3*i+2]*T(so100g::CAM_P[2]);
#pragma unroll
            for (int j = 0; j < 3; j++)
                P.cam_mat[3*i+j] = R[3*i]*T(so100g::CAM_R[j]) + R[3*i+1]*T(so100g::CAM_R[3+j]) + R[3*i+2]*T(so100g::CAM_R[6+j]);
        }
    }
}

}  // namespace so100
