// so100_contact.hpp -- finger-pad contacts of the so100 gripper: pad/floor and pad/cube.
//
// Reference geometry: the 8 box geoms of class "finger_collision" on Fixed_Jaw / Moving_Jaw (arm:108-111, 120-123) with the
// contact parameters of arm:61 (solimp "2 1 0.01", solref "0.01 1", friction 1), the floor plane (scene:39), the cube
// (scene:32).  In the reference scene the pads collide with the floor (F_PADS_FLOOR: part of the reference physics) but
// not with the cube (scene:47-48 exclude block_a against both jaws); BASELINE.json configs[4] lifts exactly that exclusion
// (F_PADS_CUBE).  MuJoCo stages restated: mj_collision narrowphase
// (mjc_PlaneBox; a box-box routine in the manner of mjc_BoxBox: 15-axis separating-axis test, then face clipping or an
// edge-edge point), mj_contactParam's 1:1 mixing of solref / solimp with mj_assignImp's clamp, pyramidal rows (condim 3),
// and the constraint solve.  Same algorithm, slot order and constants as oracle/so100_oracle.c (the checker), in fp32.
//
// Solver.  Contact rows are general Jacobian rows over the arm's 6 dofs (and the cube's 6 when a pad touches the cube), so
// the joint-space block PGS of so100_physics.hpp (rows +-e_i only) does not apply.  A lane that has pad contacts solves its
// whole substep in the PRIMAL instead, like MuJoCo's default Newton solver and like the cube/floor block of so100_cube.hpp:
//     minimise over x = qacc   1/2 x'Mx - x'tau  (+ the cube's 1/2 m |x_l - a0|^2 + 1/2 I |x_a|^2)  +  sum_r s_r(J_r x - aref_r)
// with s_r the row penalties (friction loss: Huber; limits and pyramid edges: one-sided quadratic), Newton steps on the
// exact Hessian M + J'DJ of the current active set (6 x 6, or 12 x 12 when arm and cube are coupled), full step when the
// quadratic model holds, exact line search otherwise.  Lanes without pad contacts keep the block PGS + separate cube solve.
// Contact Jacobians are never stored: a row's J x is the edge direction dotted with the point acceleration of the link's
// spatial acceleration, and J'f is the contact wrench projected on the joint axes (world FK of the 6 joint frames).
//
// Contact records live in a small per-env store: LDS [record][lane] in the multi-wave kernels (detection on one wave, solve
// on another), a private array in the one-wave kernel and on the host (tests/_hostcheck).  Budget: MAXC contacts per env;
// further contacts are dropped in detection order (cube/floor, pad/floor by pad, pad/cube by pad) and counted.
#pragma once
#include "so100_cube.hpp"
#ifndef SO100_LS_PASSES
#define SO100_LS_PASSES 2      // line-search trials per Newton iteration in fp32 (6: -25 % row passes measured with 2, same residuals)
#endif
#if !defined(__HIPCC__)
#include <cstdio>
#endif
#ifndef SO100_LS_TIGHT_AFTER
#define SO100_LS_TIGHT_AFTER 4
#endif
#ifndef SO100_LEAN_DX
#define SO100_LEAN_DX 0.02     // size of a Newton step, SO100_LEAN_ABS + SO100_LEAN_DX |x| (rad/s^2), up to which its round-off is removed by
#endif                         // refining the LINEAR system (see primal_newton: lean_refine)
#ifndef SO100_LEAN_ABS
#define SO100_LEAN_ABS 8.0
#endif

namespace so100 {

enum : unsigned { F_PADS_FLOOR = 16u, F_PADS_CUBE = 32u, F_LINKS_FLOOR = 64u, F_LINKS_CUBE = 128u };
constexpr unsigned F_ANY_CONTACT = F_PADS_FLOOR | F_PADS_CUBE | F_LINKS_FLOOR | F_LINKS_CUBE;
constexpr unsigned F_ARM_CUBE = F_PADS_CUBE | F_LINKS_CUBE;                      // pairs that couple the arm's solve with the cube's
constexpr unsigned F_ANY_LINKS = F_LINKS_FLOOR | F_LINKS_CUBE;                   // records on links other than the jaws: the solver's general form      // any of these: the contact store / contact wave / primal solve are in play

constexpr int MAXPADC = 16;                  // budget of PAD contacts per env (oracle: model.max_contacts); detection order = pad/floor
                                             // by pad, then pad/cube by pad; further ones are dropped and counted
constexpr int MAXC = MAXPADC + 4;            // + the cube's <= 4 floor contacts, which join the list when a pad touches the cube
constexpr int CF = 12;                       // floats per contact record
enum { C_PX = 0, C_PY, C_PZ, C_NX, C_NY, C_NZ, C_KD, C_RINV, C_KIND, C_VX, C_VY, C_VZ };
// C_P point (world, midway between the surfaces), C_N normal geom1 -> geom2 (the tangents follow from it: mju_makeFrame),
// C_KD = K imp dist, C_RINV = 1/R of the 4 edge rows, C_V = B * relative point velocity (the velocity part of -aref),
// C_KIND: kind | id << 3 | mask0 << 11 (an integer held in a float) -- kind: 0 cube/floor, 1 / 2 pad/floor on link 4 / 5, 3 / 4 pad/cube
// with the pad on link 4 / 5, 5 link proxy/floor (F_LINKS_FLOOR; id = 144 + 2 proxy + capsule end, on link proxy + 1), 6 link proxy/cube (F_LINKS_CUBE; id = 160 + link, link 0 / 1); id: which geometric feature made the contact (pad corner, manifold slot: stable from substep to substep);
// mask0: the pyramid edges that carried force at the end of the previous substep's solve (the Newton's first guess of the active set).
// The solver leaves (hash(id) << 4 | final mask), one byte, of every record in the store's "previous" list for the next substep's
// detection (4-bit hash: distinct for the same corner of different pads; a collision only costs a worse first guess).
SO100_HD int contact_id_hash(int id) { return (id ^ (id >> 4)) & 15; }
// the arm link (0..5) that carries the record's arm-side geom (kind != 0)
static_assert(so100g::NPROX == 5 && so100g::PROX_LINK[0] == 1 && so100g::PROX_LINK[4] == 5, "proxy k sits on link k + 1");
SO100_HD int contact_link(int kind, int id) { return kind == 5 ? ((id - 144) >> 1) + 1 : kind == 6 ? id - 160 : ((kind == 2 || kind == 4) ? 5 : 4); }
SO100_HD bool contact_on_cube(int kind) { return kind == 3 || kind == 4 || kind == 6; }      // arm geom (geom1) against the cube (geom2)
// 32-bit mix of a feature id; a contact SET's signature is the wrapping sum of it over the set's PAD contacts (order-free: the
// cooperating lanes add their shares).  Parity tests compare it with the same sum over the oracle's contact list (so100o_contact.feat).
SO100_HD int contact_id_mix(int id) {
    unsigned h = (unsigned)(id + 1)*0x9E3779B1u; h ^= h >> 15; h *= 0x85EBCA77u; h ^= h >> 13;
    return (int)h;
}

// Cooperative lanes.  When a wavefront holds fewer envs than lanes (the multi-wave kernels spread small batches over all CUs: 16 or
// 32 envs per 64-lane wave), the contact wave gives each env a group of `nparts` = 4 or 2 ADJACENT lanes (a quad, or half of one):
// every lane of the group holds the same env (same inputs, same x, g, H: all replicated arithmetic is bit-identical), the loops over
// pads / contact records are dealt out over the group (record s belongs to lane s mod nparts), and the partial sums meet in
// quad-permute DPP additions.  Stores with COOP = false (private arrays: the one-wave kernel, the host) are a group of one.
#if defined(__HIP_DEVICE_COMPILE__)
SO100_HD float quad_xor1(float v) { return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xF, 0xF, true)); }   // lanes 0<->1, 2<->3
SO100_HD float quad_xor2(float v) { return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x4E, 0xF, 0xF, true)); }   // lanes 0<->2, 1<->3
SO100_HD int quad_xor1(int v) { return __builtin_amdgcn_mov_dpp(v, 0xB1, 0xF, 0xF, true); }
SO100_HD int quad_xor2(int v) { return __builtin_amdgcn_mov_dpp(v, 0x4E, 0xF, 0xF, true); }
#else
template <typename T> SO100_HD T quad_xor1(T v) { return v; }
template <typename T> SO100_HD T quad_xor2(T v) { return v; }
#endif
template <class Store, typename T> SO100_HD T coop_sum(const Store& cs, T v) {
    if constexpr (Store::COOP) { if (cs.nparts >= 2) v += quad_xor1(v); if (cs.nparts == 4) v += quad_xor2(v); }
    return v;
}
// the value held by the group's first lane
template <class Store> SO100_HD int coop_first(const Store& cs, int v) {
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (Store::COOP) {
        if (cs.nparts == 4) v = __builtin_amdgcn_mov_dpp(v, 0x00, 0xF, 0xF, true);        // quad_perm [0,0,0,0]
        else if (cs.nparts == 2) v = __builtin_amdgcn_mov_dpp(v, 0xA0, 0xF, 0xF, true);   // quad_perm [0,0,2,2]
    }
#endif
    return v;
}
template <class Store> SO100_HD int coop_sum_int(const Store& cs, int v) {          // wrapping sum (unsigned arithmetic)
    if constexpr (Store::COOP) {
        if (cs.nparts >= 2) v = (int)((unsigned)v + (unsigned)quad_xor1(v));
        if (cs.nparts == 4) v = (int)((unsigned)v + (unsigned)quad_xor2(v));
    }
    return v;
}
template <class Store> SO100_HD int coop_or(const Store& cs, int v) {
    if constexpr (Store::COOP) { if (cs.nparts >= 2) v |= quad_xor1(v); if (cs.nparts == 4) v |= quad_xor2(v); }
    return v;
}

template <typename T> struct ContactsPriv {                 // one env's records in a private array
    static constexpr bool COOP = false; static constexpr int part = 0, nparts = 1;
    T a[MAXC*CF]; int n = 0, dropped = 0, sig = 0;
    unsigned char pcode[MAXC]; int prev_n = 0;
    SO100_HD T get(int s, int f) const { return a[s*CF + f]; }
    SO100_HD void set(int s, int f, T v) { a[s*CF + f] = v; }
    SO100_HD int getp(int k) const { return pcode[k]; }
    SO100_HD void setp(int k, int v) { pcode[k] = (unsigned char)v; }
};
template <typename T> struct ContactsLds {                  // [record][field][lane] image shared by the waves of a workgroup
    static constexpr bool COOP = true;
    T* base; int lane; unsigned char* pbase; int part = 0, nparts = 1;                 // lane: the ENV's column; part of nparts: this lane's share of it
    int n = 0, dropped = 0, prev_n = 0, sig = 0;                                      // pbase: [MAXC][64] bytes OUTSIDE any aliased region
    SO100_HD T get(int s, int f) const { return base[(s*CF + f)*64 + lane]; }
    SO100_HD void set(int s, int f, T v) { base[(s*CF + f)*64 + lane] = v; }
    SO100_HD int getp(int k) const { return pbase[k*64 + lane]; }
    SO100_HD void setp(int k, int v) { pbase[k*64 + lane] = (unsigned char)v; }
};

// ---- world-frame kinematics of the six joint frames (axis z_k, origin o_k) and of the two jaw links ------------------------
template <typename T> struct WorldFK { T z[6][3], o[6][3], oz[6][3], R4[9], R5[9]; };      // oz_k = o_k x z_k: the linear part of joint k's screw about the world origin
template <int K, typename T> SO100_HD void wfk_step(const T s[6], const T c[6], T pos[3], T R[9], WorldFK<T>& W) {
    fk_link<K>(s, c, pos, R);
    constexpr int AX = so100g::LINK_AXIS[K];
#pragma unroll
    for (int i = 0; i < 3; i++) { W.z[K][i] = R[3*i + AX]; W.o[K][i] = pos[i]; }
    cross(W.o[K], W.z[K], W.oz[K]);
}
template <typename T> SO100_HD void world_fk(const T s[6], const T c[6], WorldFK<T>& W) {
    T pos[3] = { T(0), T(0), T(0) };
    T R[9] = { T(1), T(0), T(0), T(0), T(1), T(0), T(0), T(0), T(1) };
    wfk_step<0>(s, c, pos, R, W); wfk_step<1>(s, c, pos, R, W); wfk_step<2>(s, c, pos, R, W); wfk_step<3>(s, c, pos, R, W);
    wfk_step<4>(s, c, pos, R, W);
#pragma unroll
    for (int i = 0; i < 9; i++) W.R4[i] = R[i];
    wfk_step<5>(s, c, pos, R, W);
#pragma unroll
    for (int i = 0; i < 9; i++) W.R5[i] = R[i];
}
// spatial motion (about the world origin) of links 4 and 5 produced by the joint-space vector x (velocities or accelerations)
template <typename T> struct Spatial { T a[3], b[3]; };      // angular part, linear part at the origin
template <typename T> SO100_HD void link_spatial(const WorldFK<T>& W, const T x[6], Spatial<T>& S4, Spatial<T>& S5) {
    T a[3] = { T(0), T(0), T(0) }, b[3] = { T(0), T(0), T(0) };
#pragma unroll
    for (int i = 0; i < 6; i++) {
#pragma unroll
        for (int k = 0; k < 3; k++) { a[k] += x[i]*W.z[i][k]; b[k] += x[i]*W.oz[i][k]; }
        if (i == 4) {
#pragma unroll
            for (int k = 0; k < 3; k++) { S4.a[k] = a[k]; S4.b[k] = b[k]; }
        }
    }
#pragma unroll
    for (int k = 0; k < 3; k++) { S5.a[k] = a[k]; S5.b[k] = b[k]; }
}
template <typename T> SO100_HD void point_motion(const Spatial<T>& S, const T p[3], T out[3]) {
    T t[3]; cross(S.a, p, t);
    out[0] = t[0] + S.b[0]; out[1] = t[1] + S.b[1]; out[2] = t[2] + S.b[2];
}
// link 5's or link 4's, selected per lane BY VALUE (a reference select between two register structs would send both to memory)
template <typename T> SO100_HD Spatial<T> pick_spatial(bool on5, const Spatial<T>& S4, const Spatial<T>& S5) {
    Spatial<T> S;
#pragma unroll
    for (int k = 0; k < 3; k++) { S.a[k] = on5 ? S5.a[k] : S4.a[k]; S.b[k] = on5 ? S5.b[k] : S4.b[k]; }
    return S;
}

// ---- contact parameters ------------------------------------------------------------------------------------------------
// impedance d(r) of the pad-involved pairs: solimp (0.9999, 0.975, 0.0055, 0.5, 2) after mixing + clamp (so100_model_gen.h)
template <typename T> SO100_HD T impedance_pad(T r) {
    const T x = r * T(1.0/so100g::PADC_WIDTH);
    const T d0 = T(so100g::PADC_D0), dm = T(so100g::PADC_DMAX);
    if (x >= T(1)) return dm;
    if (x <= T(0)) return d0;
    const T y = x <= T(0.5) ? T(2)*x*x : T(1) - T(2)*(T(1) - x)*(T(1) - x);
    return d0 + y*(dm - d0);
}

// mju_makeFrame: t1 = e_y unless |n_y| >= 0.5 (then e_z), orthogonalised against n and normalised; t2 = n x t1
template <typename T> SO100_HD void contact_frame(const T n[3], T t1[3], T t2[3]) {
    const bool usey = n[1] < T(0.5) && n[1] > T(-0.5);
    const T dp = usey ? n[1] : n[2];
    t1[0] = -dp*n[0]; t1[1] = (usey ? T(1) : T(0)) - dp*n[1]; t1[2] = (usey ? T(0) : T(1)) - dp*n[2];
    const T rn = trcp(tsqrt(dot(t1, t1)));
    t1[0] *= rn; t1[1] *= rn; t1[2] *= rn;
    cross(n, t1, t2);
}

// append one contact (returns false when the pad budget is exhausted).  vrel = relative point velocity geom2 - geom1.
template <typename T, class Store>
SO100_HD void contact_put(Store& cs, int s, int kind, int id, const T p[3], const T n[3], T dist, const T vrel[3]);
template <typename T, class Store>
SO100_HD bool contact_add(Store& cs, int kind, int id, const T p[3], const T n[3], T dist, const T vrel[3]) {
    if (kind != 0 && cs.n >= MAXPADC) { cs.dropped++; return false; }
    if (cs.n >= MAXC) { cs.dropped++; return false; }
    contact_put(cs, cs.n++, kind, id, p, n, dist, vrel);
    return true;
}
// write record s (no budget logic: the caller owns the slot)
template <typename T, class Store>
SO100_HD void contact_put(Store& cs, int s, int kind, int id, const T p[3], const T n[3], T dist, const T vrel[3]) {
    if (kind != 0) cs.sig = (int)((unsigned)cs.sig + (unsigned)contact_id_mix(id));      // signature of the pad-contact set (this lane's share)
    int mask0 = 15;                                            // a new contact: expect all four edges to push (an impact sticks first)
#pragma unroll 1
    for (int k = 0; k < cs.prev_n; k++) { const int c = cs.getp(k); if ((c >> 4) == contact_id_hash(id)) mask0 = c & 15; }
    // impedance, reference and regulariser: R = 2 mu^2 (1 - imp)/imp * (1 + mu^2) * (translational invweight0 of both bodies), mu = 1
    T imp, K, B, tran;
    if (kind == 0) { imp = impedance(tabs(dist)); K = T(so100g::SOLREF_K); B = T(so100g::SOLREF_B); tran = T(1.0/so100g::CUBE_MASS); }
    else if (kind == 5) {                                      // link proxy / floor: both geoms carry MuJoCo's default parameters
        imp = impedance(tabs(dist)); K = T(so100g::SOLREF_K); B = T(so100g::SOLREF_B);
        const int l = contact_link(kind, id);
        tran = l == 1 ? T(so100g::LINK_INVWEIGHT_TRAN[1]) : l == 2 ? T(so100g::LINK_INVWEIGHT_TRAN[2]) : l == 3 ? T(so100g::LINK_INVWEIGHT_TRAN[3])
             : l == 4 ? T(so100g::LINK_INVWEIGHT_TRAN[4]) : T(so100g::LINK_INVWEIGHT_TRAN[5]);
    } else if (kind == 6) {                                    // link proxy / cube: default parameters on both sides
        imp = impedance(tabs(dist)); K = T(so100g::SOLREF_K); B = T(so100g::SOLREF_B);
        tran = (contact_link(kind, id) == 0 ? T(so100g::LINK_INVWEIGHT_TRAN[0]) : T(so100g::LINK_INVWEIGHT_TRAN[1])) + T(1.0/so100g::CUBE_MASS);
    } else {
        imp = impedance_pad(tabs(dist)); K = T(so100g::PADC_K); B = T(so100g::PADC_B);
        tran = (kind == 1 || kind == 3) ? T(so100g::LINK_INVWEIGHT_TRAN[4]) : T(so100g::LINK_INVWEIGHT_TRAN[5]);
        if (kind == 3 || kind == 4) tran += T(1.0/so100g::CUBE_MASS);
    }
    const T R = T(4)*tran*(T(1) - imp)*trcp(imp);
    cs.set(s, C_PX, p[0]); cs.set(s, C_PY, p[1]); cs.set(s, C_PZ, p[2]);
    cs.set(s, C_NX, n[0]); cs.set(s, C_NY, n[1]); cs.set(s, C_NZ, n[2]);
    cs.set(s, C_KD, K*imp*dist); cs.set(s, C_RINV, trcp(R)); cs.set(s, C_KIND, T(kind | (id << 3) | (mask0 << 11)));
    cs.set(s, C_VX, B*vrel[0]); cs.set(s, C_VY, B*vrel[1]); cs.set(s, C_VZ, B*vrel[2]);
}

// the cube's point velocity / acceleration: x = (linear, world frame; angular, BODY frame) like MuJoCo's free-joint dofs
template <typename T> SO100_HD void cube_point_motion(const T Rc[9], const T cpos[3], const T x[6], const T p[3], T out[3]) {
    const T ww[3] = { Rc[0]*x[3] + Rc[1]*x[4] + Rc[2]*x[5], Rc[3]*x[3] + Rc[4]*x[4] + Rc[5]*x[5], Rc[6]*x[3] + Rc[7]*x[4] + Rc[8]*x[5] };
    const T r[3] = { p[0] - cpos[0], p[1] - cpos[1], p[2] - cpos[2] };
    T t[3]; cross(ww, r, t);
    out[0] = x[0] + t[0]; out[1] = x[1] + t[1]; out[2] = x[2] + t[2];
}

// ---- narrowphase -----------------------------------------------------------------------------------------------------------
// mjc_PlaneBox against the floor z = 0: corners at / below the plane and below the box centre, in corner order, at most 4.
// R row-major, world <- box.  `emit(p, dist, corner)` is called per contact.
template <typename T, class Emit>
SO100_HD void plane_box(const T c[3], const T R[9], const T h[3], Emit emit) {
    int cnt = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const T v0 = (k & 1) ? h[0] : -h[0], v1 = (k & 2) ? h[1] : -h[1], v2 = (k & 4) ? h[2] : -h[2];
        const T lz = R[6]*v0 + R[7]*v1 + R[8]*v2;
        if (!(c[2] + lz > T(0) || lz > T(0)) && cnt < 4) {
            const T dist = c[2] + lz;
            const T p[3] = { R[0]*v0 + R[1]*v1 + R[2]*v2 + c[0], R[3]*v0 + R[4]*v1 + R[5]*v2 + c[1], lz + c[2] - T(0.5)*dist };
            emit(p, dist, k);
            cnt++;
        }
    }
}

// the same test, returning only WHICH corners make a contact (bit k = corner k; at most 4, in corner order)
template <typename T>
SO100_HD unsigned plane_box_mask(const T c[3], const T R[9], const T h[3]) {
    unsigned m = 0u; int cnt = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const T v0 = (k & 1) ? h[0] : -h[0], v1 = (k & 2) ? h[1] : -h[1], v2 = (k & 4) ? h[2] : -h[2];
        const T lz = R[6]*v0 + R[7]*v1 + R[8]*v2;
        if (!(c[2] + lz > T(0) || lz > T(0)) && cnt < 4) { m |= 1u << k; cnt++; }
    }
    return m;
}

// Box-box (see oracle/so100_oracle.c: so100o_box_box for the algorithm; this is the same sequence of operations).
// Boxes: centre, rotation (row-major, world <- box: COLUMNS are the box axes), half sizes.  Normal from A to B.
// `emit(p, dist)` per contact, at most 8; returns the normal in nrm.
template <typename T, class Emit>
SO100_HD int box_box(const T cA[3], const T RA[9], const T hA[3], const T cB[3], const T RB[9], const T hB[3], T nrm[3], Emit emit) {
    T a[3][3], b[3][3];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int k = 0; k < 3; k++) { a[i][k] = RA[3*k + i]; b[i][k] = RB[3*k + i]; }
    const T dd[3] = { cB[0] - cA[0], cB[1] - cA[1], cB[2] - cA[2] };
    T Rm[3][3], Ra[3][3];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) { Rm[i][j] = dot(a[i], b[j]); Ra[i][j] = tabs(Rm[i][j]); }
    T best = T(-1e30), bn[3] = { T(0), T(0), T(0) }; int code = -1;
    bool sep_found = false;
#pragma unroll
    for (int i = 0; i < 3; i++) {
        const T proj = dot(dd, a[i]);
        const T sep = tabs(proj) - (hA[i] + hB[0]*Ra[i][0] + hB[1]*Ra[i][1] + hB[2]*Ra[i][2]);
        sep_found = sep_found || sep > T(0);
        if (sep > best) { best = sep; code = i; const T sg = proj < T(0) ? T(-1) : T(1); bn[0] = sg*a[i][0]; bn[1] = sg*a[i][1]; bn[2] = sg*a[i][2]; }
    }
#pragma unroll
    for (int j = 0; j < 3; j++) {
        const T proj = dot(dd, b[j]);
        const T sep = tabs(proj) - (hB[j] + hA[0]*Ra[0][j] + hA[1]*Ra[1][j] + hA[2]*Ra[2][j]);
        sep_found = sep_found || sep > T(0);
        if (sep > best) { best = sep; code = 3 + j; const T sg = proj < T(0) ? T(-1) : T(1); bn[0] = sg*b[j][0]; bn[1] = sg*b[j][1]; bn[2] = sg*b[j][2]; }
    }
    T ebest = T(-1e30), en[3] = { T(0), T(0), T(0) }; int ecode = -1;
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) {
            T L[3]; cross(a[i], b[j], L);
            const T len = tsqrt(dot(L, L));
            if (len >= T(1e-6)) {
                const T rl = trcp(len);
                L[0] *= rl; L[1] *= rl; L[2] *= rl;
                const T proj = dot(dd, L);
                T ra = T(0), rb = T(0);
#pragma unroll
                for (int k = 0; k < 3; k++) { ra += hA[k]*tabs(dot(a[k], L)); rb += hB[k]*tabs(dot(b[k], L)); }
                const T sep = tabs(proj) - (ra + rb);
                sep_found = sep_found || sep > T(0);
                if (sep > ebest) { ebest = sep; ecode = 6 + 3*i + j; const T sg = proj < T(0) ? T(-1) : T(1); en[0] = sg*L[0]; en[1] = sg*L[1]; en[2] = sg*L[2]; }
            }
        }
    if (sep_found) return 0;
    if (ecode >= 0 && ebest*T(1.05) > best + T(1e-9)) { best = ebest; code = ecode; bn[0] = en[0]; bn[1] = en[1]; bn[2] = en[2]; }
    nrm[0] = bn[0]; nrm[1] = bn[1]; nrm[2] = bn[2];

    if (code >= 6) {
        // edge-edge: supporting edge of A towards +n, of B towards -n; closest points of the two lines, clamped to the edges
        const int i = (code - 6) / 3, j = (code - 6) % 3;
        T pa[3] = { cA[0], cA[1], cA[2] }, pb[3] = { cB[0], cB[1], cB[2] };
#pragma unroll
        for (int q = 0; q < 3; q++) {
            if (q != i) { const T sg = dot(bn, a[q]) > T(0) ? T(1) : T(-1); pa[0] += sg*hA[q]*a[q][0]; pa[1] += sg*hA[q]*a[q][1]; pa[2] += sg*hA[q]*a[q][2]; }
            if (q != j) { const T sg = dot(bn, b[q]) > T(0) ? T(-1) : T(1); pb[0] += sg*hB[q]*b[q][0]; pb[1] += sg*hB[q]*b[q][1]; pb[2] += sg*hB[q]*b[q][2]; }
        }
        T ai[3], bj[3], hAi = T(0), hBj = T(0), uu = T(0);
#pragma unroll
        for (int q = 0; q < 3; q++) {
            if (q == i) { ai[0] = a[q][0]; ai[1] = a[q][1]; ai[2] = a[q][2]; hAi = hA[q]; }
            if (q == j) { bj[0] = b[q][0]; bj[1] = b[q][1]; bj[2] = b[q][2]; hBj = hB[q]; }
        }
        uu = dot(ai, bj);
        const T w0[3] = { pb[0] - pa[0], pb[1] - pa[1], pb[2] - pa[2] };
        const T d1 = dot(ai, w0), d2 = dot(bj, w0), den = T(1) - uu*uu;
        T sa = T(0), tb = T(0);
        if (den > T(1e-12)) { const T rd = trcp(den); sa = (d1 - uu*d2)*rd; tb = (uu*d1 - d2)*rd; }
        sa = tclamp(sa, -hAi, hAi); tb = tclamp(tb, -hBj, hBj);
        const T p[3] = { T(0.5)*((pa[0] + sa*ai[0]) + (pb[0] + tb*bj[0])), T(0.5)*((pa[1] + sa*ai[1]) + (pb[1] + tb*bj[1])),
                         T(0.5)*((pa[2] + sa*ai[2]) + (pb[2] + tb*bj[2])) };
        emit(p, best);
        return 1;
    }

    // face contact: X = reference box (owner of the axis), Y = incident box
    const bool refA = code < 3; const int r = refA ? code : code - 3;
    T x[3][3], y[3][3], cX[3], cY[3], hX[3], hY[3], nref[3];
#pragma unroll
    for (int i = 0; i < 3; i++) {
#pragma unroll
        for (int k = 0; k < 3; k++) { x[i][k] = refA ? a[i][k] : b[i][k]; y[i][k] = refA ? b[i][k] : a[i][k]; }
        cX[i] = refA ? cA[i] : cB[i]; cY[i] = refA ? cB[i] : cA[i]; hX[i] = refA ? hA[i] : hB[i]; hY[i] = refA ? hB[i] : hA[i];
        nref[i] = refA ? bn[i] : -bn[i];
    }
    int mi = 0; T mv = T(-1);
#pragma unroll
    for (int k = 0; k < 3; k++) { const T v = tabs(dot(nref, y[k])); if (v > mv) { mv = v; mi = k; } }
    // select the axes by index without dynamic register indexing
    T ym[3], yp1[3], yp2[3], xr[3], xu1[3], xu2[3], hYm = T(0), hYp1 = T(0), hYp2 = T(0), hXr = T(0), hu = T(0), hv = T(0);
#pragma unroll
    for (int k = 0; k < 3; k++) {
        if (k == mi) { ym[0] = y[k][0]; ym[1] = y[k][1]; ym[2] = y[k][2]; hYm = hY[k]; }
        if (k == (mi + 1) % 3) { yp1[0] = y[k][0]; yp1[1] = y[k][1]; yp1[2] = y[k][2]; hYp1 = hY[k]; }
        if (k == (mi + 2) % 3) { yp2[0] = y[k][0]; yp2[1] = y[k][1]; yp2[2] = y[k][2]; hYp2 = hY[k]; }
        if (k == r) { xr[0] = x[k][0]; xr[1] = x[k][1]; xr[2] = x[k][2]; hXr = hX[k]; }
        if (k == (r + 1) % 3) { xu1[0] = x[k][0]; xu1[1] = x[k][1]; xu1[2] = x[k][2]; hu = hX[k]; }
        if (k == (r + 2) % 3) { xu2[0] = x[k][0]; xu2[1] = x[k][1]; xu2[2] = x[k][2]; hv = hX[k]; }
    }
    (void)xr;
    const T fs = dot(nref, ym) > T(0) ? T(-1) : T(1);
    T fc[3], rc[3];
#pragma unroll
    for (int k = 0; k < 3; k++) { rc[k] = cX[k] + hXr*nref[k]; fc[k] = cY[k] + fs*hYm*ym[k] - rc[k]; }
    // incident face in reference-face coordinates (u, v, height w): centre c0 + alpha e1 + beta e2
    const T c0[3] = { dot(fc, xu1), dot(fc, xu2), dot(fc, nref) };
    const T e1[3] = { hYp1*dot(yp1, xu1), hYp1*dot(yp1, xu2), hYp1*dot(yp1, nref) };
    const T e2[3] = { hYp2*dot(yp2, xu1), hYp2*dot(yp2, xu2), hYp2*dot(yp2, nref) };
    int cnt = 0;
    auto out = [&](T u, T v, T w) {
        if (w <= T(0) && cnt < 8) {
            const T p[3] = { rc[0] + u*xu1[0] + v*xu2[0] + T(0.5)*w*nref[0], rc[1] + u*xu1[1] + v*xu2[1] + T(0.5)*w*nref[1],
                             rc[2] + u*xu1[2] + v*xu2[2] + T(0.5)*w*nref[2] };
            emit(p, w);
            cnt++;
        }
    };
    // slots 0-7: per incident edge its Liang-Barsky entry point (or start vertex) and, when it leaves the rectangle early, its exit point
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const T sa0 = (k == 0 || k == 3) ? T(-1) : T(1), sb0 = (k < 2) ? T(-1) : T(1);
        const int k2 = (k + 1) & 3;
        const T sa1 = (k2 == 0 || k2 == 3) ? T(-1) : T(1), sb1 = (k2 < 2) ? T(-1) : T(1);
        const T P0[3] = { c0[0] + sa0*e1[0] + sb0*e2[0], c0[1] + sa0*e1[1] + sb0*e2[1], c0[2] + sa0*e1[2] + sb0*e2[2] };
        const T P1[3] = { c0[0] + sa1*e1[0] + sb1*e2[0], c0[1] + sa1*e1[1] + sb1*e2[1], c0[2] + sa1*e1[2] + sb1*e2[2] };
        T t0 = T(0), t1 = T(1); bool ok = true;
#pragma unroll
        for (int ax = 0; ax < 2; ax++) {
            const T dq = P1[ax] - P0[ax], lim = ax == 0 ? hu : hv;
#pragma unroll
            for (int side = -1; side <= 1; side += 2) {
                const T pden = -T(side)*dq, pnum = T(side)*P0[ax] - lim;
                if (pden == T(0)) { if (pnum > T(0)) ok = false; }
                else {
                    const T t = pnum/pden;
                    if (pden > T(0)) { if (t > t0) t0 = t; } else { if (t < t1) t1 = t; }
                }
            }
        }
        if (t0 > t1) ok = false;
        if (ok) out(P0[0] + t0*(P1[0] - P0[0]), P0[1] + t0*(P1[1] - P0[1]), P0[2] + t0*(P1[2] - P0[2]));
        if (ok && t1 < T(1)) out(P0[0] + t1*(P1[0] - P0[0]), P0[1] + t1*(P1[1] - P0[1]), P0[2] + t1*(P1[2] - P0[2]));
    }
    {   // slots 8-11: rectangle corners inside the incident parallelogram
        const T det = e1[0]*e2[1] - e1[1]*e2[0];
        if (tabs(det) > T(1e-18)) {
            const T rd = T(1)/det;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const T su = (k == 0 || k == 3) ? T(-1) : T(1), sv = (k < 2) ? T(-1) : T(1);
                const T uu = su*hu - c0[0], vv = sv*hv - c0[1];
                const T al = (uu*e2[1] - vv*e2[0])*rd, be = (e1[0]*vv - e1[1]*uu)*rd;
                if (tabs(al) < T(1) && tabs(be) < T(1)) out(su*hu, sv*hv, c0[2] + al*e1[2] + be*e2[2]);
            }
        }
    }
    return cnt;
}

// Capsule (segment a..b, radius r: geom1) against a box (geom2): the stand-in narrowphase of oracle/so100_oracle.c: so100o_capsule_box, same
// operations in the same order.  The segment's point nearest to the box minimises a convex piecewise-quadratic function of the segment parameter t
// whose derivative is linear between the <= 6 parameters at which a coordinate crosses a face plane: evaluate it at those and at the ends,
// interpolate in the piece that holds the root (exact, no iteration), then a sphere-box test there.  One contact, normal capsule -> box.  Returns 0 / 1.
template <typename T>
SO100_HD int capsule_box(const T a[3], const T b[3], T r, const T c[3], const T R[9], const T h[3], T pos[3], T nrm[3], T& dist) {
    T la[3], d[3], s[3], q[3], e[3], tj[8], gj[8];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        la[k] = R[k]*(a[0] - c[0]) + R[3 + k]*(a[1] - c[1]) + R[6 + k]*(a[2] - c[2]);
        d[k] = R[k]*(b[0] - a[0]) + R[3 + k]*(b[1] - a[1]) + R[6 + k]*(b[2] - a[2]);
    }
    tj[0] = T(0); tj[1] = T(1);
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const T inv = tabs(d[k]) > T(1e-12) ? T(1)/d[k] : T(0);
        tj[2 + 2*k] = tclamp((-h[k] - la[k])*inv, T(0), T(1)); tj[3 + 2*k] = tclamp((h[k] - la[k])*inv, T(0), T(1));
    }
#pragma unroll
    for (int j = 0; j < 8; j++) {
        T g = T(0);
#pragma unroll
        for (int k = 0; k < 3; k++) { const T sk = la[k] + tj[j]*d[k]; g += (sk - tclamp(sk, -h[k], h[k]))*d[k]; }
        gj[j] = g;
    }
    T thi = T(2), ghi = T(0), tlo = T(-1), glo = T(0);
#pragma unroll
    for (int j = 0; j < 8; j++) if (gj[j] >= T(0) && tj[j] < thi) { thi = tj[j]; ghi = gj[j]; }
    if (thi > T(1.5)) { thi = T(1); ghi = T(0); }
#pragma unroll
    for (int j = 0; j < 8; j++) if (gj[j] < T(0) && tj[j] <= thi && tj[j] > tlo) { tlo = tj[j]; glo = gj[j]; }
    T t = thi;
    if (tlo >= T(0) && ghi > T(0)) t = tlo - glo*(thi - tlo)/(ghi - glo);
    T tin = T(0), tout = T(1);                                 // the axis itself passes through the box (slab test): the middle of that interval
#pragma unroll
    for (int k = 0; k < 3; k++) {
        if (tabs(d[k]) > T(1e-12)) {
            const T t1 = (-h[k] - la[k])/d[k], t2 = (h[k] - la[k])/d[k];
            tin = tmax(tin, tmin(t1, t2)); tout = tmin(tout, tmax(t1, t2));
        } else if (tabs(la[k]) > h[k]) tin = T(2);
    }
    if (tin <= tout) t = T(0.5)*(tin + tout);
#pragma unroll
    for (int k = 0; k < 3; k++) { s[k] = la[k] + t*d[k]; q[k] = tclamp(s[k], -h[k], h[k]); e[k] = s[k] - q[k]; }
    const T len = tsqrt(dot(e, e));
    T n[3] = { T(0), T(0), T(0) }, dst;
    if (len > T(1e-9)) { const T rl = trcp(len); n[0] = e[0]*rl; n[1] = e[1]*rl; n[2] = e[2]*rl; dst = len - r; }
    else {
        int ax = 0; T best = h[0] - tabs(s[0]);
#pragma unroll
        for (int k = 1; k < 3; k++) if (h[k] - tabs(s[k]) < best) { best = h[k] - tabs(s[k]); ax = k; }
#pragma unroll
        for (int k = 0; k < 3; k++) if (k == ax) { n[k] = s[k] < T(0) ? T(-1) : T(1); q[k] = n[k]*h[k]; }
        dst = -best - r;
    }
    if (dst > T(0)) return 0;
    const T pl[3] = { q[0] + T(0.5)*dst*n[0], q[1] + T(0.5)*dst*n[1], q[2] + T(0.5)*dst*n[2] };
#pragma unroll
    for (int k = 0; k < 3; k++) {
        pos[k] = c[k] + R[3*k]*pl[0] + R[3*k + 1]*pl[1] + R[3*k + 2]*pl[2];
        nrm[k] = -(R[3*k]*n[0] + R[3*k + 1]*n[1] + R[3*k + 2]*n[2]);
    }
    dist = dst;
    return 1;
}

// The box (in jaw coordinates) that encloses all pads of one jaw link: centre and half sizes, compile-time from the pad table.
SO100_HD constexpr double pad_hull_lo(int link, int ax) {
    double lo = 1e30;
    for (int g = 0; g < so100g::NPAD; g++) if (so100g::PAD_LINK[g] == link) { const double v = so100g::PAD_POS[g][ax] - so100g::PAD_SIZE[g][ax]; lo = v < lo ? v : lo; }
    return lo;
}
SO100_HD constexpr double pad_hull_hi(int link, int ax) {
    double hi = -1e30;
    for (int g = 0; g < so100g::NPAD; g++) if (so100g::PAD_LINK[g] == link) { const double v = so100g::PAD_POS[g][ax] + so100g::PAD_SIZE[g][ax]; hi = v > hi ? v : hi; }
    return hi;
}
// lowest point (world z) of that box for a jaw at origin o with rotation R (row-major, world <- jaw): no pad of the jaw reaches lower
template <int LINK, typename T> SO100_HD T pad_hull_lowest(const T o[3], const T R[9]) {
    constexpr double cx = 0.5*(pad_hull_lo(LINK, 0) + pad_hull_hi(LINK, 0)), cy = 0.5*(pad_hull_lo(LINK, 1) + pad_hull_hi(LINK, 1)), cz = 0.5*(pad_hull_lo(LINK, 2) + pad_hull_hi(LINK, 2));
    constexpr double hx = 0.5*(pad_hull_hi(LINK, 0) - pad_hull_lo(LINK, 0)), hy = 0.5*(pad_hull_hi(LINK, 1) - pad_hull_lo(LINK, 1)), hz = 0.5*(pad_hull_hi(LINK, 2) - pad_hull_lo(LINK, 2));
    return o[2] + (R[6]*T(cx) + R[7]*T(cy) + R[8]*T(cz)) - (tabs(R[6])*T(hx) + tabs(R[7])*T(hy) + tabs(R[8])*T(hz)) - T(1e-6);     // (1e-6 m: fp32 round-off of this bound vs the per-pad test)
}

// ---- detection: fills the contact store for one env --------------------------------------------------------------------
// v = q-dot of the arm, cube pose / velocity; W = world FK of this substep.  Store order: pad/floor by pad, pad/cube by pad
// (these two share the budget MAXPADC), then -- only when a pad touches the cube, i.e. arm and cube must be solved together --
// the cube's own floor contacts (otherwise the cube keeps its separate Newton solve in so100_cube.hpp).
// Returns true when at least one pad/cube contact exists.
template <typename T, class Store>
SO100_HD bool detect_pad_contacts(const WorldFK<T>& W, const T v[6], const Cube<T>& cube, const T Rc[9], unsigned flags, bool cube_live, Store& cs) {
    cs.n = 0; cs.dropped = 0; cs.sig = 0;                     // (cs.prev_n / the previous list stay: contact_add looks the new contacts up in it)
    Spatial<T> V4{}, V5{};
    const T nz[3] = { T(0), T(0), T(1) };
    const T hc[3] = { T(so100g::CUBE_HALF), T(so100g::CUBE_HALF), T(so100g::CUBE_HALF) };
    bool coupled = false;
    // every pad lies within PAD_REACH of its jaw's origin: a jaw higher than that above the floor / farther than that (+ the cube's
    // circumradius) from the cube cannot touch; the per-pad tests below run only for lanes that fail this bound
    constexpr double PAD_REACH = 0.115;
    const T zmin = tmin(W.o[4][2], W.o[5][2]);
    T dc2 = T(1e30);
    if ((flags & F_PADS_CUBE) != 0u && cube_live) {
#pragma unroll
        for (int L = 4; L <= 5; L++) {
            const T dx = W.o[L][0] - cube.pos[0], dy = W.o[L][1] - cube.pos[1], dz = W.o[L][2] - cube.pos[2];
            dc2 = tmin(dc2, dx*dx + dy*dy + dz*dz);
        }
    }
    // pad/floor: the box around each jaw's pads against the plane -- exact for the box, so a jaw hovering a millimetre over the table does not
    // send its wave into the per-pad tests (the old bound, zmin < PAD_REACH, let every arm within 11 cm of the table through)
    const bool near_floor = (flags & F_PADS_FLOOR) != 0u && zmin < T(PAD_REACH)
                            && tmin(pad_hull_lowest<4, T>(W.o[4], W.R4), pad_hull_lowest<5, T>(W.o[5], W.R5)) <= T(0);
    const bool near_cube = dc2 < T((PAD_REACH + so100g::CUBE_HALF*1.7320508075688772)*(PAD_REACH + so100g::CUBE_HALF*1.7320508075688772));
    if (near_floor || near_cube) link_spatial(W, v, V4, V5);   // the jaws' spatial velocities (for the contacts' velocity terms): only where a pad may touch
    auto pad_frame = [&](int g, T R[9], T o[3], T h[3], T c[3]) {
        const bool l5 = so100g::PAD_LINK[g] == 5;
#pragma unroll
        for (int k = 0; k < 9; k++) R[k] = l5 ? W.R5[k] : W.R4[k];
#pragma unroll
        for (int k = 0; k < 3; k++) o[k] = l5 ? W.o[5][k] : W.o[4][k];
        const T pp[3] = { T(so100g::PAD_POS[g][0]), T(so100g::PAD_POS[g][1]), T(so100g::PAD_POS[g][2]) };
        h[0] = T(so100g::PAD_SIZE[g][0]); h[1] = T(so100g::PAD_SIZE[g][1]); h[2] = T(so100g::PAD_SIZE[g][2]);
        c[0] = o[0] + R[0]*pp[0] + R[1]*pp[1] + R[2]*pp[2]; c[1] = o[1] + R[3]*pp[0] + R[4]*pp[1] + R[5]*pp[2]; c[2] = o[2] + R[6]*pp[0] + R[7]*pp[1] + R[8]*pp[2];
        return l5;
    };
    bool coop_floor_done = false;
    if constexpr (Store::COOP) {
        if (cs.nparts > 1) {
            // Pad/floor pass dealt out over the env's lanes (pad g belongs to lane g mod nparts): first every lane COUNTS the contacts
            // of its pads, the counts meet (3 bits per pad), then every lane writes its records at the slots the serial order gives
            // them -- pad by pad, corner by corner, the first MAXPADC kept.  Same list as the serial pass, 4 (2) times faster.
            coop_floor_done = true;
            if (near_floor) {
                // pass 1: which corners of MY pads touch (a 4-bit-per-pad count for the group, the corner masks stay with the lane)
                int counts = 0; unsigned cmask = 0u;
#pragma unroll 1
                for (int g = cs.part, j = 0; g < so100g::NPAD; g += cs.nparts, j++) {
                    T R[9], o[3], h[3], c[3];
                    pad_frame(g, R, o, h, c);
                    if (c[2] - (tabs(R[6])*h[0] + tabs(R[7])*h[1] + tabs(R[8])*h[2]) <= T(0)) {
                        const unsigned m = plane_box_mask<T>(c, R, h);
                        cmask |= m << (8*j); counts |= __builtin_popcount(m) << (3*g);
                    }
                }
                counts = coop_or(cs, counts);
                int total = 0;
#pragma unroll
                for (int g = 0; g < so100g::NPAD; g++) total += (counts >> (3*g)) & 7;
                // pass 2: every lane writes its records at the slots the serial order gives them (pad by pad, corner by corner)
#pragma unroll 1
                for (int g = cs.part, j = 0; g < so100g::NPAD; g += cs.nparts, j++) {
                    unsigned m = (cmask >> (8*j)) & 255u;
                    if (m == 0u) continue;
                    int slot = 0;
                    for (int hh = 0; hh < g; hh++) slot += (counts >> (3*hh)) & 7;
                    T R[9], o[3], h[3], c[3];
                    const bool l5 = pad_frame(g, R, o, h, c);
                    const Spatial<T> V = pick_spatial(l5, V4, V5);
#pragma unroll 1
                    while (m != 0u && slot < MAXPADC) {
                        const int k = __builtin_ctz(m); m &= m - 1u;
                        const T v0 = (k & 1) ? h[0] : -h[0], v1 = (k & 2) ? h[1] : -h[1], v2 = (k & 4) ? h[2] : -h[2];
                        const T lz = R[6]*v0 + R[7]*v1 + R[8]*v2, dist = c[2] + lz;
                        const T p[3] = { R[0]*v0 + R[1]*v1 + R[2]*v2 + c[0], R[3]*v0 + R[4]*v1 + R[5]*v2 + c[1], lz + c[2] - T(0.5)*dist };
                        T vr[3]; point_motion(V, p, vr);
                        contact_put(cs, slot++, l5 ? 2 : 1, 8*g + k, p, nz, dist, vr);
                    }
                }
                cs.n = total < MAXPADC ? total : MAXPADC; cs.dropped = total - cs.n;
            }
        }
    }
    bool coop_cube_done = false;
    if constexpr (Store::COOP) {
        if (cs.nparts > 1 && (flags & F_PADS_CUBE) != 0u && cube_live) {
            // Pad/cube pass dealt out over the env's lanes in ROUNDS of nparts pads (round r: pad r nparts + part, so the lanes of a round hold
            // consecutive pads): every lane runs the box-box routine for its pad into registers, the round's counts meet, and the records go
            // to the slots the serial pad order gives them.  A closing grasp touches 2-4 pads: their 1400-instruction separating-axis +
            // clipping runs side by side instead of one after the other (the coupled env is what a CONTACT5 launch waits for).
            coop_cube_done = true;
            if (near_cube) {                                   // (cs.n / cs.dropped are the same in all lanes of the group after the floor pass)
                int base = cs.n, dropped = cs.dropped;
#pragma unroll 1
                for (int r = 0; r < so100g::NPAD/cs.nparts; r++) {
                    const int g = r*cs.nparts + cs.part;
                    T R[9], o[3], h[3], c[3];
                    const bool l5 = pad_frame(g, R, o, h, c);
                    T bp[8][3], bd[8], nrm[3] = { T(0), T(0), T(1) };
#pragma unroll
                    for (int q = 0; q < 8; q++) { bp[q][0] = bp[q][1] = bp[q][2] = T(0); bd[q] = T(0); }
                    int k = 0;
                    const T dx = c[0] - cube.pos[0], dy = c[1] - cube.pos[1], dz = c[2] - cube.pos[2];
                    const T rr = tsqrt(h[0]*h[0] + h[1]*h[1] + h[2]*h[2]) + T(so100g::CUBE_HALF*1.7320508075688772);
                    if (dx*dx + dy*dy + dz*dz < rr*rr) {
                        int cnt = 0;
                        k = box_box<T>(c, R, h, cube.pos, Rc, hc, nrm, [&](const T* p, T dist) {
#pragma unroll
                            for (int q = 0; q < 8; q++) if (q == cnt) { bp[q][0] = p[0]; bp[q][1] = p[1]; bp[q][2] = p[2]; bd[q] = dist; }     // (no dynamic register indexing)
                            cnt++;
                        });
                    }
                    const int counts = coop_or(cs, k << (4*cs.part));
                    int before = 0, total = 0;
#pragma unroll
                    for (int j = 0; j < 4; j++) { const int cj = (counts >> (4*j)) & 15; total += cj; before += j < cs.part ? cj : 0; }
                    const Spatial<T> V = pick_spatial(l5, V4, V5);
#pragma unroll
                    for (int i = 0; i < 8; i++) {
                        if (i < k && base + before + i < MAXPADC) {
                            T va[3], vc[3]; point_motion(V, bp[i], va); cube_point_motion(Rc, cube.pos, cube.vel, bp[i], vc);
                            const T vr[3] = { vc[0] - va[0], vc[1] - va[1], vc[2] - va[2] };
                            contact_put(cs, base + before + i, l5 ? 4 : 3, 64 + 8*g + i, bp[i], nrm, bd[i], vr);
                        }
                    }
                    coupled = coupled || total > 0;
                    const int nn = base + total < MAXPADC ? base + total : MAXPADC;
                    dropped += base + total - nn; base = nn;
                }
                cs.n = base; cs.dropped = dropped;
            }
        }
    }
    const bool serial_lane = !Store::COOP || cs.part == 0;   // the serial passes and the cube's own floor contacts run on the group's first lane
#pragma unroll 1
    for (int pass = 0; pass < 2; pass++) {                   // pass 0: pad/floor, pass 1: pad/cube
        if (pass == 0 && coop_floor_done) continue;
        if (pass == 1 && (coop_cube_done || !serial_lane)) continue;
        if (pass == 0 && !near_floor) continue;
        if (pass == 1 && !near_cube) continue;
        if (pass == 0 && (flags & F_PADS_FLOOR) == 0u) continue;
        if (pass == 1 && ((flags & F_PADS_CUBE) == 0u || !cube_live)) continue;
#pragma unroll 1
        for (int g = 0; g < so100g::NPAD; g++) {
            T R[9], o[3], h[3], c[3];
            const bool l5 = pad_frame(g, R, o, h, c);
            if (pass == 0) {
                if (c[2] - (tabs(R[6])*h[0] + tabs(R[7])*h[1] + tabs(R[8])*h[2]) <= T(0)) {
                    plane_box<T>(c, R, h, [&](const T* p, T dist, int corner) {
                        T vr[3]; point_motion(pick_spatial(l5, V4, V5), p, vr);
                        contact_add(cs, l5 ? 2 : 1, 8*g + corner, p, nz, dist, vr);
                    });
                }
            } else {
                const T dx = c[0] - cube.pos[0], dy = c[1] - cube.pos[1], dz = c[2] - cube.pos[2];
                const T rr = tsqrt(h[0]*h[0] + h[1]*h[1] + h[2]*h[2]) + T(so100g::CUBE_HALF*1.7320508075688772);
                if (dx*dx + dy*dy + dz*dz < rr*rr) {          // bounding spheres overlap: run the separating-axis test
                    // the points are emitted before the normal is returned: records are appended with a placeholder, then patched
                    const int s0 = cs.n;
                    T nrm[3];
                    int slot = 0;
                    const int k = box_box<T>(c, R, h, cube.pos, Rc, hc, nrm, [&](const T* p, T dist) {
                        T va[3], vc[3]; point_motion(pick_spatial(l5, V4, V5), p, va); cube_point_motion(Rc, cube.pos, cube.vel, p, vc);
                        const T vr[3] = { vc[0] - va[0], vc[1] - va[1], vc[2] - va[2] };
                        contact_add(cs, l5 ? 4 : 3, 64 + 8*g + (slot++ & 7), p, nz, dist, vr);
                    });
                    coupled = coupled || k > 0;
                    for (int s = s0; s < cs.n; s++) { cs.set(s, C_NX, nrm[0]); cs.set(s, C_NY, nrm[1]); cs.set(s, C_NZ, nrm[2]); }
                }
            }
        }
    }
    if ((flags & F_LINKS_FLOOR) != 0u && serial_lane) {
        // Link proxies (stand-in capsules for the arm's collision meshes, so100_model_def.h) against the floor: mjc_PlaneCapsule = a
        // plane-sphere test at either end of the segment.  Ends: the link's joint origin and its child's (links 1-3), the far end of
        // the jaw's pads (links 4, 5).  Velocity of the contact point from the joint screws of the links below it.
#pragma unroll 1
        for (int k = 0; k < so100g::NPROX; k++) {
            const int l = k + 1;                               // PROX_LINK[k] (static_assert below)
            const T r = T(so100g::PROX_RADIUS[k]);
#pragma unroll 1
            for (int e = 0; e < 2; e++) {
                T c[3];
                if (e == 0 || l <= 3) {
                    const int j = e == 0 ? l : l + 1;
#pragma unroll
                    for (int a = 0; a < 3; a++) c[a] = j == 1 ? W.o[1][a] : j == 2 ? W.o[2][a] : j == 3 ? W.o[3][a] : j == 4 ? W.o[4][a] : W.o[5][a];
                } else {
                    const T* R = l == 4 ? W.R4 : W.R5; const T* o = l == 4 ? W.o[4] : W.o[5];
                    const T f0 = T(so100g::PROX_FAR[k][0]), f1 = T(so100g::PROX_FAR[k][1]), f2 = T(so100g::PROX_FAR[k][2]);
                    c[0] = o[0] + R[0]*f0 + R[1]*f1 + R[2]*f2; c[1] = o[1] + R[3]*f0 + R[4]*f1 + R[5]*f2; c[2] = o[2] + R[6]*f0 + R[7]*f1 + R[8]*f2;
                }
                const T dist = c[2] - r;
                if (dist > T(0)) continue;
                const T p[3] = { c[0], c[1], c[2] - r - T(0.5)*dist };
                T vr[3] = { T(0), T(0), T(0) };
#pragma unroll
                for (int i = 0; i < 6; i++) {
                    T col[3]; cross(W.z[i], p, col);
                    const T vi = i <= l ? v[i] : T(0);
                    vr[0] += vi*(col[0] + W.oz[i][0]); vr[1] += vi*(col[1] + W.oz[i][1]); vr[2] += vi*(col[2] + W.oz[i][2]);
                }
                contact_add(cs, 5, 144 + 2*k + e, p, nz, dist, vr);
            }
        }
    }
    if ((flags & F_LINKS_CUBE) != 0u && cube_live && serial_lane) {
        // Q7: the capsules of Rotation_Pitch (joint origin 0 -> 1) and Upper_Arm (1 -> 2) against the cube; stand-in narrowphase capsule_box
#pragma unroll 1
        for (int k = 0; k < so100g::NCPROX; k++) {
            T a[3], b[3];
#pragma unroll
            for (int i = 0; i < 3; i++) { a[i] = k == 0 ? W.o[0][i] : W.o[1][i]; b[i] = k == 0 ? W.o[1][i] : W.o[2][i]; }
            const T r = k == 0 ? T(so100g::CPROX_RADIUS[0]) : T(so100g::CPROX_RADIUS[1]);
            // bounding spheres: segment midpoint / half length + radius against the cube's circumsphere
            const T mx = T(0.5)*(a[0] + b[0]) - cube.pos[0], my = T(0.5)*(a[1] + b[1]) - cube.pos[1], mz = T(0.5)*(a[2] + b[2]) - cube.pos[2];
            const T hl2 = T(0.25)*((b[0] - a[0])*(b[0] - a[0]) + (b[1] - a[1])*(b[1] - a[1]) + (b[2] - a[2])*(b[2] - a[2]));
            const T reach = tsqrt(hl2) + r + T(so100g::CUBE_HALF*1.7320508075688772);
            if (mx*mx + my*my + mz*mz >= reach*reach) continue;
            T p[3], nrm[3], dist;
            if (capsule_box<T>(a, b, r, cube.pos, Rc, hc, p, nrm, dist) == 0) continue;
            T va[3] = { T(0), T(0), T(0) }, vc[3];
#pragma unroll
            for (int i = 0; i < 2; i++) {
                T col[3]; cross(W.z[i], p, col);
                const T vi = i <= k ? v[i] : T(0);
                va[0] += vi*(col[0] + W.oz[i][0]); va[1] += vi*(col[1] + W.oz[i][1]); va[2] += vi*(col[2] + W.oz[i][2]);
            }
            cube_point_motion(Rc, cube.pos, cube.vel, p, vc);
            const T vr[3] = { vc[0] - va[0], vc[1] - va[1], vc[2] - va[2] };
            if (contact_add(cs, 6, 160 + k, p, nrm, dist, vr)) coupled = true;
        }
    }
    if (coupled && serial_lane && (flags & F_FLOOR) != 0u) {
        plane_box<T>(cube.pos, Rc, hc, [&](const T* p, T dist, int corner) {
            T vr[3]; cube_point_motion(Rc, cube.pos, cube.vel, p, vr);
            contact_add(cs, 0, 128 + corner, p, nz, dist, vr);
        });
    }
    if constexpr (Store::COOP) {
        if (cs.nparts > 1 && (flags & (F_PADS_CUBE | F_ANY_LINKS)) != 0u) {   // what the first lane appended is the group's
            cs.n = coop_first(cs, cs.n); cs.dropped = coop_first(cs, cs.dropped); coupled = coop_first(cs, coupled ? 1 : 0) != 0;
        }
    }
    if (cs.n > 0) cs.prev_n = cs.n;                            // (the solve fills the list: every row pass leaves id | final mask per record)
    if constexpr (Store::COOP) { if (cs.nparts > 1) cs.sig = coop_sum_int(cs, cs.sig); }     // the lanes' shares of the set's signature
    return coupled;
}
// signature of the set of PAD contacts in the store (cube/floor records, kind 0, are the cube block's business): the wrapping sum of
// contact_id_mix over the records, accumulated by contact_put while the store is filled
template <class Store> SO100_HD int contact_signature(const Store& cs) { return cs.sig; }

// ---- N x N SPD systems, N = 6 or 12: LDL^T in place on the packed lower triangle -----------------------------------------
template <int N, typename T> SO100_HD void ldln(T M[N*(N+1)/2], T Dinv[N]) {
#pragma unroll
    for (int j = 0; j < N; j++) {
        T d = M[SO100_TRI(j, j)];
#pragma unroll
        for (int k = 0; k < j; k++) d -= M[SO100_TRI(j, k)]*M[SO100_TRI(j, k)]*M[SO100_TRI(k, k)];
        M[SO100_TRI(j, j)] = d;
        Dinv[j] = trcp(d);
#pragma unroll
        for (int i = j + 1; i < N; i++) {
            T t = M[SO100_TRI(i, j)];
#pragma unroll
            for (int k = 0; k < j; k++) t -= M[SO100_TRI(i, k)]*M[SO100_TRI(j, k)]*M[SO100_TRI(k, k)];
            M[SO100_TRI(i, j)] = t*Dinv[j];
        }
    }
}
template <int N, typename T> SO100_HD void ldln_solve(const T L[N*(N+1)/2], const T Dinv[N], T x[N]) {
#pragma unroll
    for (int i = 1; i < N; i++)
#pragma unroll
        for (int k = 0; k < i; k++) x[i] -= L[SO100_TRI(i, k)]*x[k];
#pragma unroll
    for (int i = 0; i < N; i++) x[i] *= Dinv[i];
#pragma unroll
    for (int i = N - 2; i >= 0; i--)
#pragma unroll
        for (int k = i + 1; k < N; k++) x[i] -= L[SO100_TRI(k, i)]*x[k];
}
// M (packed lower) from its LDL^T factor (arm_factor leaves only the factor in A.M)
template <typename T> SO100_HD void ldl6_reconstruct(const T L[21], T M[21]) {
#pragma unroll
    for (int i = 0; i < 6; i++)
#pragma unroll
        for (int j = 0; j <= i; j++) {
            T t = T(0);
#pragma unroll
            for (int k = 0; k <= j; k++) {
                const T lik = k == i ? T(1) : L[SO100_TRI(i, k)], ljk = k == j ? T(1) : L[SO100_TRI(j, k)];
                t += lik*ljk*L[SO100_TRI(k, k)];
            }
            M[SO100_TRI(i, j)] = t;
        }
}

// ---- the primal problem of one env's substep ----------------------------------------------------------------------------
// ND = 6: arm only (pad/floor contacts);  ND = 12: arm + cube (a pad touches the cube): x = [arm qacc (6) ; cube qacc (6)]
// LINKS = false: every arm-side geom sits on link 4 or 5 (the finger pads: the default physics) -- point accelerations from the two jaws'
// spatial accelerations, wrench totals per jaw.  LINKS = true (F_LINKS_FLOOR): a record may sit on any link 1..5; per record the six
// Jacobian columns z_i x (p - o_i) are formed once and serve the point acceleration, the gradient and the Hessian alike.
template <int ND, typename T, class Store, bool LINKS = false> struct PrimalProblem {
    const WorldFK<T>& W; Store& cs;
    const T* Marm;             // packed lower 6x6
    const T* tau;              // arm smooth force
    const ArmRows<T>& rows;    // friction / limit row constants
    const T* Rc; const T* cpos; const T* a0c;      // cube rotation (world <- body), centre, smooth linear acceleration (ND = 12)
    int* zones;                // active-set memory of the arm's own rows: per joint 3 bits (friction 0 quadratic / 1 low / 2 high, limit active)
    static constexpr int NH = ND*(ND + 1)/2;

    // MODE 0: only the active set x selects (recorded, compared with the remembered one: `same`); 1: gradient at x, 2: + Hessian.
    // (The cost VALUE is never formed: no step of the solver compares costs, see primal_newton.)
    // FORCED: gradient and Hessian of the QUADRATIC whose active set is the remembered one (*zones, the records' mask0) instead of
    // the one x selects: its minimiser is the solution whenever the active set did not change since the previous substep.
    // A plain MODE 2 (or MODE 0) pass records the active set it saw (*zones, the records' mask, the store's previous list) and reports in
    // `same` whether that is the set of the quadratic that produced x (the remembered one): if it is, and x is that quadratic's
    // minimiser (a full Newton step), x minimises the true cost -- the piecewise-quadratic problem's exact convergence test.
    template <int MODE, bool FORCED = false> SO100_HD void eval(const T x[ND], T g[ND], T H[NH], bool* same = nullptr) const {
        const int zin = *zones;
        int zout = 0, differ = 0;
        // ---- contacts first, into zero-initialised accumulators (this lane's share of the records; the shares meet below)
        if (MODE == 2) {
#pragma unroll
            for (int i = 0; i < NH; i++) H[i] = T(0);
        }
        T gc[6] = { T(0), T(0), T(0), T(0), T(0), T(0) };           // the contacts' gradient on the cube's dofs (ND = 12)
        // contacts
        Spatial<T> S4{}, S5{};
        if (!LINKS) link_spatial(W, x, S4, S5);
        T F45[3] = { T(0), T(0), T(0) }, T45[3] = { T(0), T(0), T(0) }, F5[3] = { T(0), T(0), T(0) }, T5[3] = { T(0), T(0), T(0) };   // wrenches on links 4 + 5, on link 5
        T ga[6] = { T(0), T(0), T(0), T(0), T(0), T(0) };           // LINKS: the contacts' gradient on the arm's dofs, accumulated per record
#pragma unroll 1
        for (int s = cs.part; s < cs.n; s += cs.nparts) {
            const T p[3] = { cs.get(s, C_PX), cs.get(s, C_PY), cs.get(s, C_PZ) };
            const T kd = cs.get(s, C_KD), D = cs.get(s, C_RINV);
            const int code = (int)cs.get(s, C_KIND), kind = code & 7;
            // ND == 6 (arm alone): every record is a pad/floor contact, whose frame is made of world axes (mju_makeFrame of +z:
            // n = e_z, t1 = e_y, t2 = -e_x): projections on the frame are component picks
            T n[3] = { T(0), T(0), T(1) }, t1[3] = { T(0), T(1), T(0) }, t2[3] = { T(-1), T(0), T(0) };
            if (ND == 12) { n[0] = cs.get(s, C_NX); n[1] = cs.get(s, C_NY); n[2] = cs.get(s, C_NZ); contact_frame(n, t1, t2); }
            const bool on5 = kind == 2 || kind == 4, arm_side = ND == 6 || kind != 0, cube_side = ND == 12 && (kind == 0 || contact_on_cube(kind));
            const T sgn = (ND == 12 && contact_on_cube(kind)) ? T(-1) : T(1);   // the arm link is geom1 in pad/cube and link/cube pairs, geom2 in floor pairs
            const int link = LINKS ? contact_link(kind, (code >> 3) & 255) : (on5 ? 5 : 4);
            T w[3] = { cs.get(s, C_VX), cs.get(s, C_VY), cs.get(s, C_VZ) };
            T col[6][3];                                        // LINKS: column i of the point Jacobian, zero where joint i does not move the point
            if (LINKS) {
#pragma unroll
                for (int i = 0; i < 6; i++) {
                    cross(W.z[i], p, col[i]);
                    const bool moves = arm_side && i <= link;
#pragma unroll
                    for (int k = 0; k < 3; k++) { col[i][k] = moves ? col[i][k] + W.oz[i][k] : T(0); w[k] += sgn*x[i]*col[i][k]; }
                }
            } else if (arm_side) { T ap[3]; point_motion(pick_spatial(on5, S4, S5), p, ap); w[0] += sgn*ap[0]; w[1] += sgn*ap[1]; w[2] += sgn*ap[2]; }
            if (ND == 12 && cube_side) { T ac[3]; cube_point_motion(Rc, cpos, x + 6, p, ac); w[0] += ac[0]; w[1] += ac[1]; w[2] += ac[2]; }
            const T jn = (ND == 6 ? w[2] : dot(n, w)) + kd, j1 = ND == 6 ? w[1] : dot(t1, w), j2 = ND == 6 ? -w[0] : dot(t2, w);
            const T jar[4] = { jn + j1, jn - j1, jn + j2, jn - j2 };        // edges n +- mu t1, n +- mu t2 (mu = 1)
            T m_[4]; bool act[4]; int mask = 0;
#pragma unroll
            for (int e = 0; e < 4; e++) {
                act[e] = FORCED ? ((code >> (11 + e)) & 1) != 0 : jar[e] < T(0);
                m_[e] = act[e] ? jar[e] : T(0);
                mask |= act[e] ? 1 << e : 0;
            }
            if ((MODE == 2 || MODE == 0) && !FORCED) {
                differ |= mask ^ ((code >> 11) & 15);
                cs.set(s, C_KIND, T((code & 2047) | (mask << 11)));
                cs.setp(s, (contact_id_hash((code >> 3) & 255) << 4) | mask);
            }
            if (MODE >= 1) {
                const T cn = D*(m_[0] + m_[1] + m_[2] + m_[3]), c1 = D*(m_[0] - m_[1]), c2 = D*(m_[2] - m_[3]);
                T Fv[3] = { -c2, c1, cn };
                if (ND == 12) { Fv[0] = cn*n[0] + c1*t1[0] + c2*t2[0]; Fv[1] = cn*n[1] + c1*t1[1] + c2*t2[1]; Fv[2] = cn*n[2] + c1*t1[2] + c2*t2[2]; }
                if (LINKS) {
#pragma unroll
                    for (int i = 0; i < 6; i++) ga[i] += sgn*dot(col[i], Fv);
                } else if (arm_side) {
                    T tq[3]; cross(p, Fv, tq);
                    const T w5 = on5 ? sgn : T(0);
#pragma unroll
                    for (int k = 0; k < 3; k++) { F45[k] += sgn*Fv[k]; T45[k] += sgn*tq[k]; F5[k] += w5*Fv[k]; T5[k] += w5*tq[k]; }
                }
                if (ND == 12 && cube_side) {
                    const T r[3] = { p[0] - cpos[0], p[1] - cpos[1], p[2] - cpos[2] };
                    T tw[3]; cross(r, Fv, tw);
                    gc[0] += Fv[0]; gc[1] += Fv[1]; gc[2] += Fv[2];
                    gc[3] += Rc[0]*tw[0] + Rc[3]*tw[1] + Rc[6]*tw[2];
                    gc[4] += Rc[1]*tw[0] + Rc[4]*tw[1] + Rc[7]*tw[2];
                    gc[5] += Rc[2]*tw[0] + Rc[5]*tw[1] + Rc[8]*tw[2];
                }
            }
            if (MODE == 2) {
                // explicit rows of the active edges: J_e = [sgn d_e . (z_i x (p - o_i)), i <= link ; d_e ; Rc'((p - c) x d_e)]
                T cn_[ND], c1_[ND], c2_[ND];                    // the point Jacobian's columns projected on n, t1, t2
#pragma unroll
                for (int i = 0; i < 6; i++) {
                    if (LINKS) {                                           // (columns formed above, zero where the joint does not move the point)
                        if (ND == 6) { cn_[i] = col[i][2]; c1_[i] = col[i][1]; c2_[i] = -col[i][0]; }
                        else { cn_[i] = sgn*dot(n, col[i]); c1_[i] = sgn*dot(t1, col[i]); c2_[i] = sgn*dot(t2, col[i]); }
                        continue;
                    }
                    const bool moves = arm_side && (i < 5 || on5);
                    T col1[3]; cross(W.z[i], p, col1);                     // z x (p - o) = z x p + o x z
                    col1[0] += W.oz[i][0]; col1[1] += W.oz[i][1]; col1[2] += W.oz[i][2];
                    if (ND == 6) { cn_[i] = moves ? col1[2] : T(0); c1_[i] = moves ? col1[1] : T(0); c2_[i] = moves ? -col1[0] : T(0); }
                    else { cn_[i] = moves ? sgn*dot(n, col1) : T(0); c1_[i] = moves ? sgn*dot(t1, col1) : T(0); c2_[i] = moves ? sgn*dot(t2, col1) : T(0); }
                }
                if (ND == 12) {
                    const T r[3] = { p[0] - cpos[0], p[1] - cpos[1], p[2] - cpos[2] };
                    const T on = cube_side ? T(1) : T(0);
                    auto cube_cols = [&](const T d[3], T o[ND]) {
                        T tw[3]; cross(r, d, tw);
                        o[6] = on*d[0]; o[7] = on*d[1]; o[8] = on*d[2];
                        o[9]  = on*(Rc[0]*tw[0] + Rc[3]*tw[1] + Rc[6]*tw[2]);
                        o[10] = on*(Rc[1]*tw[0] + Rc[4]*tw[1] + Rc[7]*tw[2]);
                        o[11] = on*(Rc[2]*tw[0] + Rc[5]*tw[1] + Rc[8]*tw[2]);
                    };
                    cube_cols(n, cn_); cube_cols(t1, c1_); cube_cols(t2, c2_);
                }
                // sum over the active edges e of D (cn +- c_k)(cn +- c_k)' = the 3 x 3 weight matrix of the edge set in the (n, t1, t2)
                // basis: w_nn = sum, w_11 = a0 + a1, w_22 = a2 + a3, w_n1 = a0 - a1, w_n2 = a2 - a3 (w_12 = 0)
                const T a0 = act[0] ? D : T(0), a1 = act[1] ? D : T(0), a2 = act[2] ? D : T(0), a3 = act[3] ? D : T(0);
                const T w11 = a0 + a1, w22 = a2 + a3, wnn = w11 + w22, wn1 = a0 - a1, wn2 = a2 - a3;
#pragma unroll
                for (int i = 0; i < ND; i++) {
                    const T un = wnn*cn_[i] + wn1*c1_[i] + wn2*c2_[i], u1 = wn1*cn_[i] + w11*c1_[i], u2 = wn2*cn_[i] + w22*c2_[i];
#pragma unroll
                    for (int j = 0; j <= i; j++) H[SO100_TRI(i, j)] += un*cn_[j] + u1*c1_[j] + u2*c2_[j];
                }
            }
        }
        // ---- the shares of the group's lanes meet (a group of one: nothing happens)
        if constexpr (Store::COOP) {
            if (cs.nparts > 1) {
                if (MODE >= 1 && !LINKS) {
#pragma unroll
                    for (int k = 0; k < 3; k++) { F45[k] = coop_sum(cs, F45[k]); T45[k] = coop_sum(cs, T45[k]); F5[k] = coop_sum(cs, F5[k]); T5[k] = coop_sum(cs, T5[k]); }
                }
                if (MODE >= 1 && LINKS) {
#pragma unroll
                    for (int i = 0; i < 6; i++) ga[i] = coop_sum(cs, ga[i]);
                }
                if (ND == 12 && MODE >= 1) {
#pragma unroll
                    for (int k = 0; k < 6; k++) gc[k] = coop_sum(cs, gc[k]);
                }
                if (MODE == 2) {
#pragma unroll
                    for (int i = 0; i < NH; i++) H[i] = coop_sum(cs, H[i]);
                }
                differ = coop_or(cs, differ);
            }
        }
        // ---- smooth part, arm: 1/2 x'Mx - x'tau (+ the cube's), then the contact wrenches -> joint space: g_i += z_i . (T - o_i x F), links >= i
        if (MODE >= 1) {
#pragma unroll
            for (int i = 0; i < 6; i++) {
                T t = T(0);
#pragma unroll
                for (int j = 0; j < 6; j++) t += sym6(Marm, i, j)*x[j];
                const T* Fa = i == 5 ? F5 : F45; const T* Ta = i == 5 ? T5 : T45;    // z . (T - o x F) = z . T + (o x z) . F
                g[i] = (t - tau[i]) + (LINKS ? ga[i] : dot(W.z[i], Ta) + dot(W.oz[i], Fa));
            }
        }
        if (MODE == 2) {
#pragma unroll
            for (int i = 0; i < 21; i++) H[i] += Marm[i];
        }
        if (ND == 12 && MODE >= 1) {
            const T m = T(so100g::CUBE_MASS), I = T(so100g::CUBE_INERTIA);
#pragma unroll
            for (int i = 0; i < 3; i++) {
                const T dl = x[6 + i] - a0c[i], da = x[9 + i];
                g[6 + i] = m*dl + gc[i]; g[9 + i] = I*da + gc[3 + i];
                if (MODE == 2) { H[SO100_TRI(6 + i, 6 + i)] += m; H[SO100_TRI(9 + i, 9 + i)] += I; }
            }
        }
        // friction-loss rows (Huber) and limit rows (one-sided) of the arm: J = +-e_i.  Branch-free: the Huber gradient is the
        // clamped quadratic's, absent rows have D = 0 (rows.Df / rows.Dl) -- 20 instructions per joint instead of 80 behind branches.
#pragma unroll
        for (int i = 0; i < 6; i++) {
            const T F = rows.fmax_[i], D = rows.Df[i], fq = D*(x[i] + rows.cfv[i]);      // the quadratic zone's force
            const int zone = FORCED ? ((zin >> (3*i)) & 3) : (F > T(0) ? (fq <= -F ? 1 : fq >= F ? 2 : 0) : 0);
            const T fr = zone == 1 ? -F : zone == 2 ? F : fq;
            if (MODE >= 1) g[i] += fr;
            if (MODE == 2) H[SO100_TRI(i, i)] += zone == 0 ? D : T(0);
            zout |= zone << (3*i);
            const T sg = rows.sg[i], Dl = rows.Dl[i], jl = sg*x[i] + rows.clv[i];
            const bool act = FORCED ? ((zin >> (3*i + 2)) & 1) != 0 : (Dl > T(0) && jl < T(0));
            if (MODE >= 1) g[i] += act ? sg*Dl*jl : T(0);
            if (MODE == 2) H[SO100_TRI(i, i)] += act ? Dl : T(0);
            zout |= act ? 4 << (3*i) : 0;
        }
        if ((MODE == 2 || MODE == 0) && !FORCED) { differ |= zout ^ zin; *zones = zout; }
        if ((MODE == 2 || MODE == 0) && !FORCED && same) *same = differ == 0;
    }

    // derivatives of phi(alpha) = cost(x + alpha dx) at alpha: d1 = phi', d2 = phi'' (of the current active set).  phi' is
    // continuous, increasing and piecewise linear, so Newton on it is exact within one piece.
    SO100_HD void line_deriv(const T x[ND], const T dx[ND], T alpha, T& d1, T& d2) const {
        T xa[ND];
#pragma unroll
        for (int i = 0; i < ND; i++) xa[i] = x[i] + alpha*dx[i];
        d1 = T(0); d2 = T(0);
#pragma unroll
        for (int i = 0; i < 6; i++) {
            T t = T(0), u = T(0);
#pragma unroll
            for (int j = 0; j < 6; j++) { t += sym6(Marm, i, j)*xa[j]; u += sym6(Marm, i, j)*dx[j]; }
            d1 += dx[i]*(t - tau[i]); d2 += dx[i]*u;
        }
        if (ND == 12) {
            const T m = T(so100g::CUBE_MASS), I = T(so100g::CUBE_INERTIA);
#pragma unroll
            for (int i = 0; i < 3; i++) {
                d1 += m*(xa[6 + i] - a0c[i])*dx[6 + i] + I*xa[9 + i]*dx[9 + i];
                d2 += m*dx[6 + i]*dx[6 + i] + I*dx[9 + i]*dx[9 + i];
            }
        }
#pragma unroll
        for (int i = 0; i < 6; i++) {
            const T F = rows.fmax_[i], D = rows.Df[i], fq = D*(xa[i] + rows.cfv[i]);
            const bool quad = fq > -F && fq < F;
            d1 += tclamp(fq, -F, F)*dx[i]; d2 += quad ? D*dx[i]*dx[i] : T(0);
            const T sg = rows.sg[i], Dl = rows.Dl[i], jl = sg*xa[i] + rows.clv[i];
            const bool act = Dl > T(0) && jl < T(0);
            d1 += act ? Dl*jl*sg*dx[i] : T(0); d2 += act ? Dl*dx[i]*dx[i] : T(0);
        }
        Spatial<T> S4{}, S5{}, D4{}, D5{};
        if (!LINKS) { link_spatial(W, xa, S4, S5); link_spatial(W, dx, D4, D5); }
        T c1_ = T(0), c2_ = T(0);                                 // the contacts' share of this lane
#pragma unroll 1
        for (int s = cs.part; s < cs.n; s += cs.nparts) {
            const T p[3] = { cs.get(s, C_PX), cs.get(s, C_PY), cs.get(s, C_PZ) };
            const T kd = cs.get(s, C_KD), D = cs.get(s, C_RINV);
            const int code = (int)cs.get(s, C_KIND), kind = code & 7;
            T n[3] = { T(0), T(0), T(1) }, t1[3] = { T(0), T(1), T(0) }, t2[3] = { T(-1), T(0), T(0) };
            if (ND == 12) { n[0] = cs.get(s, C_NX); n[1] = cs.get(s, C_NY); n[2] = cs.get(s, C_NZ); contact_frame(n, t1, t2); }
            const bool on5 = kind == 2 || kind == 4, arm_side = ND == 6 || kind != 0, cube_side = ND == 12 && (kind == 0 || contact_on_cube(kind));
            const T sgn = (ND == 12 && contact_on_cube(kind)) ? T(-1) : T(1);
            T w[3] = { cs.get(s, C_VX), cs.get(s, C_VY), cs.get(s, C_VZ) }, wd[3] = { T(0), T(0), T(0) };
            if (LINKS) {
                const int link = contact_link(kind, (code >> 3) & 255);
#pragma unroll
                for (int i = 0; i < 6; i++) {
                    T c3[3]; cross(W.z[i], p, c3);
                    const bool moves = arm_side && i <= link;
#pragma unroll
                    for (int k = 0; k < 3; k++) { const T ck = moves ? c3[k] + W.oz[i][k] : T(0); w[k] += sgn*xa[i]*ck; wd[k] += sgn*dx[i]*ck; }
                }
            } else if (arm_side) {
                T ap[3], ad[3]; point_motion(pick_spatial(on5, S4, S5), p, ap); point_motion(pick_spatial(on5, D4, D5), p, ad);
#pragma unroll
                for (int k = 0; k < 3; k++) { w[k] += sgn*ap[k]; wd[k] += sgn*ad[k]; }
            }
            if (ND == 12 && cube_side) {
                T ac[3], ad[3]; cube_point_motion(Rc, cpos, xa + 6, p, ac); cube_point_motion(Rc, cpos, dx + 6, p, ad);
#pragma unroll
                for (int k = 0; k < 3; k++) { w[k] += ac[k]; wd[k] += ad[k]; }
            }
            const T jn = (ND == 6 ? w[2] : dot(n, w)) + kd, j1 = ND == 6 ? w[1] : dot(t1, w), j2 = ND == 6 ? -w[0] : dot(t2, w);
            const T dn = ND == 6 ? wd[2] : dot(n, wd), e1 = ND == 6 ? wd[1] : dot(t1, wd), e2 = ND == 6 ? -wd[0] : dot(t2, wd);
            const T jar[4] = { jn + j1, jn - j1, jn + j2, jn - j2 }, jd[4] = { dn + e1, dn - e1, dn + e2, dn - e2 };
#pragma unroll
            for (int e = 0; e < 4; e++) if (jar[e] < T(0)) { c1_ += D*jar[e]*jd[e]; c2_ += D*jd[e]*jd[e]; }
        }
        d1 += coop_sum(cs, c1_); d2 += coop_sum(cs, c2_);
    }
};

#if !defined(__HIPCC__)
static long g_dbg_cnewton_iters = 0, g_dbg_cnewton_ls = 0, g_dbg_cnewton_calls = 0, g_dbg_cnewton_passes = 0, g_dbg_cnewton_signpasses = 0, g_dbg_cnewton_gradpasses = 0, g_dbg_cnewton_lastiter = 0;     // host-only instrumentation
static int g_dbg_cnewton_trace = 0;
static long g_dbg_cnewton_hist[4][16] = {};               // per solve: histogram of [0] gradient + Hessian passes, [1] sign passes, [2] gradient passes, [3] line-search passes
#endif
#if defined(SO100_CONTACT_STATS) && defined(__HIPCC__)
__device__ unsigned long long so100_cstats[8];          // calls, iterations, line-search passes, evals, capped calls (tools/micro/contact_bench.hip)
#endif
#if defined(SO100_CONTACT_STATS) && defined(__HIP_DEVICE_COMPILE__)
#define SO100_CSTAT(i) atomicAdd(&so100_cstats[i], 1ull)
#elif !defined(__HIPCC__)
#define SO100_CSTAT(i) ((i) == 3 || (i) == 2 ? (void)g_dbg_cnewton_passes++ : (i) == 5 ? (void)g_dbg_cnewton_signpasses++ : (i) == 6 ? (void)g_dbg_cnewton_gradpasses++ : (i) == 4 ? (void)g_dbg_cnewton_lastiter++ : (void)0)
#else
#define SO100_CSTAT(i) ((void)0)
#endif

// (`work`, profiling builds only: += 1 per gradient + Hessian pass, 1 << 8 per sign pass, 1 << 16 per gradient pass, 1 << 24 per line-search pass.)
// Newton on the primal problem.  x: warm start in, solution out.  Returns the size of the last Newton step when the
// iteration budget ran out before the solve ended (0 otherwise): the solver residual a caller can watch.
//
// 1. (warm) One full step on the quadratic of the REMEMBERED active set (eval<2, FORCED>): see the comment in the body.
// 2. A plain pass at the new point tells which set x selects.  Same set as the quadratic that produced x by a full step =>
//    x is the minimiser of the true cost (the exact convergence test of a piecewise-quadratic problem); one more step on the
//    same set's Hessian removes the first step's round-off and the solve ends: two row passes.
// 3. Otherwise safeguarded Newton: eval<2> at the trial point x + dx gives gradient and Hessian there at once -- the step
//    stands if the set did not change over it (-> 2.), if the merit E = g'diag(M)^-1 g fell to a quarter, or if the slope
//    phi'(1) = g(x + dx).dx is at most half of |phi'(0)| (phi convex, phi' continuous and piecewise linear).  Only when a step
//    overshoots (many rows switching on: an impact) the exact line search runs (safeguarded Newton on phi', one derivative
//    pass per trial).  No cost values are compared anywhere: 1/2 x'Mx reaches 1e5 during an impact, so in fp32 a cost
//    DIFFERENCE of the size of a Newton decrement is below the round-off of the two costs.
// Fallback stop in fp32.  The pad rows are stiff (1/R = 3e3 against M ~ 0.1): a row's force is D * jar with jar = J.x + b a
// small difference of O(10) terms, so forces carry a relative round-off of ~3e-4 and rows at a kink may flicker between
// passes.  TWO consecutive steps of at most 1e-3 relative size without progress in E end the solve (one is not enough: x may
// sit at a kink whose other side wants to go elsewhere).
template <int ND, typename T> SO100_HD T grad_merit(const T g[ND]) {
    T E = T(0);
#pragma unroll
    for (int i = 0; i < 6; i++) E += g[i]*g[i];
    E *= T(1.0/so100g::ARMATURE);
    if (ND == 12) E += (g[6]*g[6] + g[7]*g[7] + g[8]*g[8])*T(1.0/so100g::CUBE_MASS) + (g[9]*g[9] + g[10]*g[10] + g[11]*g[11])*T(1.0/so100g::CUBE_INERTIA);
    return E;
}

// y = H x for a packed-lower symmetric N x N matrix
template <int N, typename T> SO100_HD void symn_mul(const T H[N*(N+1)/2], const T x[N], T y[N]) {
#pragma unroll
    for (int i = 0; i < N; i++) {
        T t = T(0);
#pragma unroll
        for (int j = 0; j < N; j++) t += (i >= j ? H[SO100_TRI(i, j)] : H[SO100_TRI(j, i)])*x[j];
        y[i] = t;
    }
}

// One step of iterative refinement of H dx = -g on the factor (L, Dinv) with the unfactored Hc: dx += H^-1 (-(g + Hc dx)).
// Returns true when the step is LARGE relative to x (an impact: |g| is large, and the fp32 residual of the linear system is then
// dominated by cancellation between g and Hc dx -- the caller refines with the gradient evaluated at the new point instead).
template <int ND, typename T>
SO100_HD bool lean_refine(const T* Hc, const T* L, const T* Dinv, const T g[ND], const T x[ND], T dx[ND]) {
    T dmax = T(0), xmax = T(0);
#pragma unroll
    for (int i = 0; i < ND; i++) { dmax = tmax(dmax, tabs(dx[i])); xmax = tmax(xmax, tabs(x[i])); }
    if (dmax > T(SO100_LEAN_ABS) + T(SO100_LEAN_DX)*xmax) return true;
    T r[ND];
    symn_mul<ND>(Hc, dx, r);
#pragma unroll
    for (int i = 0; i < ND; i++) r[i] = -(g[i] + r[i]);
    ldln_solve<ND>(L, Dinv, r);
#pragma unroll
    for (int i = 0; i < ND; i++) dx[i] += r[i];
    return false;
}

// LEAN.  The problem is piecewise quadratic, so after a FULL Newton step
// x -> x + dx on the quadratic Q_S of an active set S the only open question is whether x + dx still selects S:
//   * it does  => x + dx minimises the true cost.  No gradient or Hessian at the new point is needed to know that -- a pass
//     that only evaluates the rows' signs (eval<0>: ~1/5 of a gradient + Hessian pass) answers it; the round-off of the step is
//     removed by one iterative-refinement step of the LINEAR system on the Hessian that made the step (r = -(g + H dx) is the
//     gradient of Q_S at x + dx; the unfactored H is kept for it: 21 registers -- 6-unknown problem only; the 12-unknown problem
//     and large steps refine with the gradient evaluated at the new point instead, still without a Hessian or a factorisation);
//   * it does not => gradient + Hessian of the set x + dx selects (eval<2> there), as before.
// Round 2 ran a full gradient + Hessian pass + factorisation at every trial point just to learn `same`: 2 such passes per solve
// in resting contact (1 + 1 sign pass now), 3.06 on average under the bench's random policy.
template <int ND, typename T, class Store, bool LINKS>
SO100_HD T primal_newton(const PrimalProblem<ND, T, Store, LINKS>& P, int iters, T x[ND], bool warm, int* work = nullptr) {
    constexpr int NH = ND*(ND + 1)/2;
    constexpr bool LEAN = true, LINREF = ND == 6;       // LINREF: keep the unfactored Hessian for the linear refinement (21 registers; 78 would not fit)
    const bool f32 = sizeof(T) == 4;
    T last = T(0);
#if !defined(__HIPCC__)
    g_dbg_cnewton_calls++;
    struct HistOnExit {                                       // host-only: per-solve pass histograms (tools/host_newton_stats.py)
        long p0 = g_dbg_cnewton_passes, l0 = g_dbg_cnewton_ls, s0 = g_dbg_cnewton_signpasses, g0 = g_dbg_cnewton_gradpasses;
        ~HistOnExit() {
            const long l = g_dbg_cnewton_ls - l0, f = g_dbg_cnewton_passes - p0 - l, sg = g_dbg_cnewton_signpasses - s0, gr = g_dbg_cnewton_gradpasses - g0;
            g_dbg_cnewton_hist[0][f > 15 ? 15 : f]++; g_dbg_cnewton_hist[1][sg > 15 ? 15 : sg]++; g_dbg_cnewton_hist[2][gr > 15 ? 15 : gr]++; g_dbg_cnewton_hist[3][l > 15 ? 15 : l]++;
        }
    } hist_on_exit_;
#endif
    T g[ND], H[NH], Hc[LINREF ? NH : 1];
    bool same = false;
    if (warm) {
        // Active-set warm start.  The rows are stiff (a force-carrying row sits at jar = -R f, a hair below zero), so the sign of jar at
        // the previous acceleration says nothing about which rows will carry force now -- the previous substep's final active set does.
        // One step on the quadratic of THAT set lands on the solution whenever the set did not change (the usual case in sustained
        // contact); the plain iteration below starts from there, verifies, and repairs the set where it did change.
        T Dinv[ND], dx[ND];
        P.template eval<2, true>(x, g, H);
        if (work) *work += 1;
        SO100_CSTAT(3);
        if (LINREF) {
#pragma unroll
            for (int i = 0; i < NH; i++) Hc[i] = H[i];
        }
#pragma unroll
        for (int i = 0; i < ND; i++) dx[i] = -g[i];
        ldln<ND>(H, Dinv);
        ldln_solve<ND>(H, Dinv, dx);
        bool big = false;
        if (LEAN) big = LINREF ? lean_refine<ND>(Hc, H, Dinv, g, x, dx) : true;
#pragma unroll
        for (int i = 0; i < ND; i++) x[i] += dx[i];
        if (LEAN) {
            P.template eval<0>(x, g, H, &same);               // signs only: does x select the remembered set? (records the set it selects)
            if (work) *work += 1 << 8;
            SO100_CSTAT(5);
            if (same && !big) return T(0);
            if (same) {                                       // a large step: its round-off is removed with the gradient evaluated AT the new point
                P.template eval<1>(x, g, Hc);                 // (Hc is not written by a gradient pass; H holds the factor)
                if (work) *work += 1 << 16;
                SO100_CSTAT(6);
#pragma unroll
                for (int i = 0; i < ND; i++) dx[i] = -g[i];
                ldln_solve<ND>(H, Dinv, dx);
#pragma unroll
                for (int i = 0; i < ND; i++) x[i] += dx[i];
                return T(0);
            }
        }
    }
    P.template eval<2>(x, g, H, &same);
    if (work) *work += 1;
    SO100_CSTAT(0); SO100_CSTAT(3);
    // onq: x came from a full Newton step on the quadratic of the very active set it selects, i.e. it IS the minimiser up to the
    // round-off of that step; one more step on the same quadratic (iterative refinement) and the solve ends without a verifying pass.
    // (LEAN: that case has returned above; here the remembered set is the one x selects since the sign pass, `same` says nothing.)
    bool onq = !LEAN && warm && same;
    T E0 = grad_merit<ND>(g);
    bool small_prev = false;                                  // the previous step was small and brought no progress in the merit
    for (int it = 0; it < iters; it++) {
        T Dinv[ND], dx[ND];
        SO100_CSTAT(1);
        if (it == iters - 1) SO100_CSTAT(4);
#if !defined(__HIPCC__)
        g_dbg_cnewton_iters++;
#endif
        if (LINREF) {
#pragma unroll
            for (int i = 0; i < NH; i++) Hc[i] = H[i];
        }
#pragma unroll
        for (int i = 0; i < ND; i++) dx[i] = -g[i];
        ldln<ND>(H, Dinv);
        ldln_solve<ND>(H, Dinv, dx);
        T dmax = T(0), xmax = T(0), gdx = T(0);
#pragma unroll
        for (int i = 0; i < ND; i++) {
            const T sc = (ND == 12 && i >= 9) ? T(so100g::CUBE_HALF) : T(1);      // cube angular acceleration measured at the cube's corner
            dmax = tmax(dmax, tabs(dx[i])*sc); xmax = tmax(xmax, tabs(x[i])*sc);
            gdx += g[i]*dx[i];
        }
#if !defined(__HIPCC__)
        if (g_dbg_cnewton_trace) printf("  it %d E %.3e g.dx %.3e dmax %.3e xmax %.3e\n", it, (double)E0, (double)gdx, (double)dmax, (double)xmax);
#endif
        const T tol = (f32 ? T(1e-4) : T(1e-11))*(T(1) + T(0.01)*xmax);
        const bool small = dmax < (f32 ? T(1e-3) : T(1e-10))*(T(1) + xmax);
        if (dmax < tol || onq || (small_prev && small)) {     // converged: the step no longer changes the acceleration
#pragma unroll
            for (int i = 0; i < ND; i++) x[i] += dx[i];
            last = T(0);
            break;
        }
        last = dmax;
        T xn[ND], gn[ND];
#pragma unroll
        for (int i = 0; i < ND; i++) xn[i] = x[i] + dx[i];
        if (LEAN) {
            // the remembered set is the one x selects (the plain pass at x recorded it): did the full step stay on its quadratic?
            P.template eval<0>(xn, gn, H, &same);
            if (work) *work += 1 << 8;
            SO100_CSTAT(5);
            if (same) {                                       // (H still holds the factor of the step's Hessian: eval<0> does not touch it)
                T d2[ND];
#pragma unroll
                for (int i = 0; i < ND; i++) d2[i] = dx[i];
                const bool big = LINREF ? lean_refine<ND>(Hc, H, Dinv, g, x, d2) : true;
#pragma unroll
                for (int i = 0; i < ND; i++) x[i] += d2[i];
                if (big) {
                    P.template eval<1>(x, gn, Hc);
                    if (work) *work += 1 << 16;
                    SO100_CSTAT(6);
#pragma unroll
                    for (int i = 0; i < ND; i++) d2[i] = -gn[i];
                    ldln_solve<ND>(H, Dinv, d2);
#pragma unroll
                    for (int i = 0; i < ND; i++) x[i] += d2[i];
                }
                last = T(0);
                break;
            }
        }
        P.template eval<2>(xn, gn, H, &same);                 // gradient + Hessian at the trial point: next iteration's, if accepted
        if (work) *work += 1;
        SO100_CSTAT(3);
        if (!LEAN && same) {                                  // the full step stayed on its quadratic: the minimiser, up to round-off
#pragma unroll
            for (int i = 0; i < ND; i++) { x[i] = xn[i]; g[i] = gn[i]; }
            onq = true; last = T(0);
            continue;
        }
        const T E1 = grad_merit<ND>(gn);
        T d1 = T(0);
#pragma unroll
        for (int i = 0; i < ND; i++) d1 += gn[i]*dx[i];
        const bool progress = E1 <= T(0.25)*E0;
        if (!progress && small) {
            // Either working precision is reached (see above) or x sits at a kink whose other side wants to go elsewhere: the
            // next step tells -- a second small one ends the solve, a large one carries on from the trial point.
#pragma unroll
            for (int i = 0; i < ND; i++) { x[i] = xn[i]; g[i] = gn[i]; }
            E0 = E1; small_prev = true;
            continue;
        }
        small_prev = false;
#ifndef SO100_ACCEPT_RULE
#define SO100_ACCEPT_RULE (progress || d1 <= T(0.5)*tabs(gdx))
#endif
        if (SO100_ACCEPT_RULE) {
#pragma unroll
            for (int i = 0; i < ND; i++) { x[i] = xn[i]; g[i] = gn[i]; }
            E0 = E1;
            continue;
        }
        // the full step overshot: exact line search in (0, 1) by safeguarded Newton on phi', first trial from the secant of
        // phi'(0) = g.dx < 0 and phi'(1) = d1 > 0
        T lo = T(0), hi = T(1), alpha = gdx/(gdx - d1), d2;
        // A crude minimiser (2 trials, |phi'| <= 1/4 |phi'(0)|) is the cheapest way through an impact, but near the solution it can trap the
        // iteration in a 2-cycle over a kink of a friction-loss row (measured: 1.5e-4 of the solves under the bench's random policy ran into
        // their iteration cap that way): from the SO100_LS_TIGHT_AFTER-th iteration on the search is run to |phi'| <= 1/50 |phi'(0)| with up
        // to 8 trials -- an exact line search cannot cycle (the cost decreases strictly)
        const bool tight = f32 && it >= SO100_LS_TIGHT_AFTER;
        const T ls_tol = f32 ? (tight ? T(0.02) : T(0.25)) : T(1e-10);
#pragma unroll 1
        for (int ls = 0; ls < (f32 ? (tight ? 8 : (ND == 12 ? 6 : SO100_LS_PASSES)) : 40); ls++) {      // (the coupled problem with its 8 g cube needs the better minimiser)
            P.line_deriv(x, dx, alpha, d1, d2);
            if (work) *work += 1 << 24;
            SO100_CSTAT(2);
#if !defined(__HIPCC__)
            g_dbg_cnewton_ls++;
#endif
            if (tabs(d1) <= ls_tol*tabs(gdx)) break;
            if (d1 < T(0)) lo = alpha; else hi = alpha;
            T an = d2 > T(0) ? alpha - d1*trcp(d2) : alpha;
            if (!(an > lo && an < hi)) an = T(0.5)*(lo + hi);
            if (tabs(an - alpha) <= (f32 ? T(1e-3) : T(1e-10))*tabs(alpha)) { alpha = an; break; }
            alpha = an;
        }
#if !defined(__HIPCC__)
        if (g_dbg_cnewton_trace) printf("     line search alpha %.6g\n", (double)alpha);
#endif
#pragma unroll
        for (int i = 0; i < ND; i++) x[i] += alpha*dx[i];
        P.template eval<2>(x, g, H);
        if (work) *work += 1;
        SO100_CSTAT(3);
        E0 = grad_merit<ND>(g);
    }
#if !defined(__HIPCC__)
    if (g_dbg_cnewton_trace) { printf("     x ="); for (int i = 0; i < ND; i++) printf(" %.5g", (double)x[i]); printf("  zones %x onq %d\n", *P.zones, (int)onq); }
#endif
    return last;
}

// ---- the constraint solve of a lane with pad contacts ----------------------------------------------------------------------
// tau: the arm's smooth force (actuation - bias), Marm: its mass matrix (packed lower, unfactored), r: row constants
// (arm_row_consts / arm_rows).  acc: warm start in (previous substep's arm acceleration), solution out.  coupled: solve the
// cube together with the arm (cs then holds its floor contacts too); cwarm = the cube's warm start in so100_cube.hpp's
// convention (qacc - qacc_smooth), xcube = its acceleration out.  Returns the solver residual (0 when converged).
template <bool LINKS = false, typename T, class Store>
SO100_HD T contact_solve(const T tau[6], const ArmRows<T>& r, const T Marm[21], const WorldFK<T>& W, Store& cs, bool coupled,
                         const T cpos[3], const T cwarm[6], const T Rc[9], const T applied[3], int iters, T acc[6], T xcube[6], int* zones, int* work = nullptr) {
    const T a0c[3] = { applied[0]*T(1.0/so100g::CUBE_MASS), applied[1]*T(1.0/so100g::CUBE_MASS), applied[2]*T(1.0/so100g::CUBE_MASS) - T(so100g::GRAVITY) };
    T res;
    if (*zones < 0) {                                          // no memory yet: the arm rows' zones at the warm-start acceleration, contacts all edges
        int z = 0;
#pragma unroll
        for (int i = 0; i < 6; i++) {
            const T F = r.fmax_[i];
            if (F > T(0)) { const T jar = acc[i] + r.cfv[i]; z |= (jar <= -r.Rf[i]*F ? 1 : jar >= r.Rf[i]*F ? 2 : 0) << (3*i); }
            if (r.sg[i] != T(0) && r.sg[i]*acc[i] + r.clv[i] < T(0)) z |= 4 << (3*i);
        }
        *zones = z;
    }
    const bool warm = true;
    if (!coupled) {
        PrimalProblem<6, T, Store, LINKS> P{ W, cs, Marm, tau, r, Rc, cpos, a0c, zones };
        res = primal_newton<6, T, Store, LINKS>(P, iters, acc, warm, work);
    } else {
        PrimalProblem<12, T, Store, LINKS> P{ W, cs, Marm, tau, r, Rc, cpos, a0c, zones };
        T x[12];
#pragma unroll
        for (int i = 0; i < 6; i++) x[i] = acc[i];
#pragma unroll
        for (int i = 0; i < 3; i++) { x[6 + i] = cwarm[i] + a0c[i]; x[9 + i] = cwarm[3 + i]; }
        res = primal_newton<12, T, Store, LINKS>(P, iters, x, warm, work);
#pragma unroll
        for (int i = 0; i < 6; i++) { acc[i] = x[i]; xcube[i] = x[6 + i]; }
    }
    return res;
}

// ---- one arm (+ cube) substep with pad contacts, single lane ----------------------------------------------------------
// Called INSTEAD of arm_solve_integrate (+ cube_finish) by lanes that have pad contacts.  A holds the factorised mass matrix
// (arm_factor) and the bias force.  aw: the arm's qacc warm start (previous substep's acceleration), updated.
// coupled: the cube is solved with the arm and integrated here.
template <typename T, class Store>
SO100_HD void contact_solve_integrate(T q[6], T v[6], T qc[6], const T ctrl[6], T ff[6], T fl[6], T aw[6], unsigned flags, int iters,
                                      Arm<T>& A, const WorldFK<T>& W, Store& cs, bool coupled, Cube<T>& cube, const T Rc[9],
                                      const T applied[3], T dq[6], T* residual, int* zones) {
    T tau[6], acc[6], xcube[6];
    arm_tau(q, v, ctrl, A, tau);
    ArmRows<T> r;
    arm_row_consts(q, v, flags, r);
    T Marm[21];
    ldl6_reconstruct(A.M, Marm);
#pragma unroll
    for (int i = 0; i < 6; i++) acc[i] = aw[i];
    const T res = (flags & F_ANY_LINKS) != 0u ? contact_solve<true>(tau, r, Marm, W, cs, coupled, cube.pos, cube.warm, Rc, applied, iters, acc, xcube, zones)
                                                : contact_solve<false>(tau, r, Marm, W, cs, coupled, cube.pos, cube.warm, Rc, applied, iters, acc, xcube, zones);
    if (coupled) {
        // the cube: warm start memory (x - qacc_smooth, the convention of so100_cube.hpp) and semi-implicit Euler
        const T a0c[3] = { applied[0]*T(1.0/so100g::CUBE_MASS), applied[1]*T(1.0/so100g::CUBE_MASS), applied[2]*T(1.0/so100g::CUBE_MASS) - T(so100g::GRAVITY) };
#pragma unroll
        for (int i = 0; i < 3; i++) { cube.warm[i] = xcube[i] - a0c[i]; cube.warm[3 + i] = xcube[3 + i]; }
        const T al[3] = { xcube[0], xcube[1], xcube[2] }, aa[3] = { xcube[3], xcube[4], xcube[5] };
        cube_integrate(cube, al, aa);
    }
    if (residual) *residual = tmax(*residual, res);
    arm_row_forces(r, acc, ff, fl);          // the block PGS's warm-start memory (a lane may be back on that path next substep)
#pragma unroll
    for (int i = 0; i < 6; i++) aw[i] = acc[i];
    arm_integrate(q, v, qc, acc, dq);
}

// One whole substep of one env with the pad-contact flags on, single lane (the one-wave step kernel and the host tests; the
// multi-wave kernels run the same stages spread over their waves).  stat (optional): [0] contacts, [1] coupled, [2] dropped,
// [3] signature of the pad-contact set (contact_signature).
// cs / zones: the caller's contact store and active-set memory, kept from substep to substep (zones = -1 to start with).
template <typename T>
SO100_HD void substep_with_pads(T q[6], T v[6], T qc[6], const T ctrl[6], T ff[6], T fl[6], T aw[6], Cube<T>& cube, const T applied[3],
                                unsigned flags, int solver_iters, int contact_iters, Arm<T>& A, bool first, T dq[6], T* residual,
                                ContactsPriv<T>& cs, int& zones, int* stat = nullptr) {
    if (first) arm_trig(q, A); else arm_trig_update(q, dq, A);
    arm_bias(v, A);
    arm_mass(A);
    arm_factor(flags, A);
    const bool cube_live = (flags & F_CUBE_PINNED) == 0u;
    WorldFK<T> W;
    world_fk(A.s, A.c, W);
    T qn[4] = { cube.quat[0], cube.quat[1], cube.quat[2], cube.quat[3] };
    quat_normalize(qn);
    T Rc[9]; quat_to_mat(qn, Rc);
    const bool coupled = detect_pad_contacts(W, v, cube, Rc, flags, cube_live, cs);
    if (stat) { stat[0] = cs.n; stat[1] = coupled ? 1 : 0; stat[2] = cs.dropped; stat[3] = cs.n > 0 ? contact_signature(cs) : 0; }
    if (cs.n > 0) {
        contact_solve_integrate(q, v, qc, ctrl, ff, fl, aw, flags, contact_iters, A, W, cs, coupled, cube, Rc, applied, dq, residual, &zones);
        if (!coupled) cube_substep(cube, applied, flags, contact_iters);
    } else {
        arm_solve_integrate(q, v, qc, ctrl, ff, fl, flags, solver_iters, A, dq, residual, aw);
        cube_substep(cube, applied, flags, contact_iters);
    }
}

}  // namespace so100
