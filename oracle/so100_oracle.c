/* so100_oracle.c -- TEST INFRASTRUCTURE, NOT A PRODUCT PATH (see so100_oracle.h).
 *
 * Part 1: physics.  MuJoCo 3.3.1's mj_step, restated for the so100 scene in fp64.  The structure
 * follows MuJoCo's stage order: kinematics -> comPos -> camlight -> crb ->
 * factorM -> collision -> makeConstraint -> comVel -> rne -> actuation -> qacc_smooth -> constraint
 * solve (PGS on the dual) -> Euler.  World-aligned spatial vectors [angular; linear] are referred
 * to the WORLD ORIGIN (MuJoCo refers them to the kinematic tree's subtree COM, which only changes
 * round-off, not results).
 *
 * Part 2: task layer.  Line-by-line restatement of the reference's Python, including its NumPy-2
 * scalar promotion (float32 where the reference computes in float32): see each function's cite.
 */
#include "so100_oracle.h"
#include "../so100_mujoco_rl_amd/csrc/so100_model_def.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define NB SO100O_NBODY
#define NV SO100O_NV
#define NQ SO100O_NQ
#define CUBE 8
#define PI 3.14159265358979323846

/* ================================================================================================
 * small math (mju_* equivalents)
 * ============================================================================================== */
static double dot3(const double* a, const double* b) { return a[0]*b[0] + a[1]*b[1] + a[2]*b[2]; }
static void cross3(double* r, const double* a, const double* b) {
    double x = a[1]*b[2] - a[2]*b[1], y = a[2]*b[0] - a[0]*b[2], z = a[0]*b[1] - a[1]*b[0];
    r[0] = x; r[1] = y; r[2] = z;
}
static void quat_mul(double* r, const double* a, const double* b) {
    double w = a[0]*b[0] - a[1]*b[1] - a[2]*b[2] - a[3]*b[3];
    double x = a[0]*b[1] + a[1]*b[0] + a[2]*b[3] - a[3]*b[2];
    double y = a[0]*b[2] - a[1]*b[3] + a[2]*b[0] + a[3]*b[1];
    double z = a[0]*b[3] + a[1]*b[2] - a[2]*b[1] + a[3]*b[0];
    r[0] = w; r[1] = x; r[2] = y; r[3] = z;
}
static void quat_normalize(double* q) {
    double n = sqrt(q[0]*q[0] + q[1]*q[1] + q[2]*q[2] + q[3]*q[3]);
    if (n < SO100_MJMINVAL) { q[0] = 1; q[1] = q[2] = q[3] = 0; return; }
    q[0] /= n; q[1] /= n; q[2] /= n; q[3] /= n;
}
static void quat2mat(double* m, const double* q) {   /* row-major 3x3, mju_quat2Mat */
    double q00 = q[0]*q[0], q11 = q[1]*q[1], q22 = q[2]*q[2], q33 = q[3]*q[3];
    double q01 = q[0]*q[1], q02 = q[0]*q[2], q03 = q[0]*q[3];
    double q12 = q[1]*q[2], q13 = q[1]*q[3], q23 = q[2]*q[3];
    m[0] = q00 + q11 - q22 - q33; m[4] = q00 - q11 + q22 - q33; m[8] = q00 - q11 - q22 + q33;
    m[1] = 2*(q12 - q03); m[2] = 2*(q13 + q02);
    m[3] = 2*(q12 + q03); m[5] = 2*(q23 - q01);
    m[6] = 2*(q13 - q02); m[7] = 2*(q23 + q01);
}
static void mat_vec3(double* r, const double* m, const double* v) {
    double x = m[0]*v[0] + m[1]*v[1] + m[2]*v[2];
    double y = m[3]*v[0] + m[4]*v[1] + m[5]*v[2];
    double z = m[6]*v[0] + m[7]*v[1] + m[8]*v[2];
    r[0] = x; r[1] = y; r[2] = z;
}
static void rot_vec_quat(double* r, const double* v, const double* q) {
    double m[9]; quat2mat(m, q); mat_vec3(r, m, v);
}
static void axis_angle2quat(double* q, const double* axis, double angle) {
    double s = sin(angle * 0.5);
    q[0] = cos(angle * 0.5); q[1] = axis[0]*s; q[2] = axis[1]*s; q[3] = axis[2]*s;
}
/* intrinsic xyz euler (MuJoCo default eulerseq "xyz"): q = qx * qy * qz */
static void euler2quat(double* q, const double* e) {
    static const double ax[3][3] = { {1,0,0}, {0,1,0}, {0,0,1} };
    double t[4] = {1,0,0,0}, r[4];
    for (int i = 0; i < 3; i++) { axis_angle2quat(r, ax[i], e[i]); quat_mul(t, t, r); }
    memcpy(q, t, sizeof t);
    quat_normalize(q);
}
/* 6-vector helpers, [angular; linear] */
static void cross_motion(double* r, const double* v, const double* s) {
    double a[3], b[3], c[3];
    cross3(a, v, s); cross3(b, v, s + 3); cross3(c, v + 3, s);
    r[0] = a[0]; r[1] = a[1]; r[2] = a[2];
    r[3] = b[0] + c[0]; r[4] = b[1] + c[1]; r[5] = b[2] + c[2];
}
static void cross_force(double* r, const double* v, const double* f) {
    double a[3], b[3], c[3];
    cross3(a, v, f); cross3(b, v + 3, f + 3); cross3(c, v, f + 3);
    r[0] = a[0] + b[0]; r[1] = a[1] + b[1]; r[2] = a[2] + b[2];
    r[3] = c[0]; r[4] = c[1]; r[5] = c[2];
}
static void mat6_vec(double* r, const double* m, const double* v) {
    double t[6];
    for (int i = 0; i < 6; i++) { double s = 0; for (int j = 0; j < 6; j++) s += m[6*i + j]*v[j]; t[i] = s; }
    memcpy(r, t, sizeof t);
}
static double dot6(const double* a, const double* b) {
    double s = 0; for (int i = 0; i < 6; i++) s += a[i]*b[i]; return s;
}

/* ================================================================================================
 * model compiler: what MuJoCo's XML compiler + mj_setConst derive for this scene
 * ============================================================================================== */
static void cholesky(const double* A, double* L, int n);
static void chol_solve(const double* L, int n, const double* b, double* x);
void so100o__crb(const so100o_model* m, so100o_data* d);

void so100o_model_init(so100o_model* m) { so100o_model_init_with_inertials(m, NULL); }

/* inert: NULL = the MJCF's <inertial> elements (arm:73-74 ... 113-114), else 6 rows "mass ipos(3) iquat(4) diaginertia(3)" for Rotation_Pitch ...
 * Moving_Jaw -- the escape hatch for SURVEY.md A.1's assumption that the scene's inertiafromgeom="true" does not reach the attached arm model
 * (csrc/gen_model.cpp takes the same table: make gen INERTIALS=file).  Everything derived from them (M0, invweights, kv, proxy radii) follows. */
void so100o_model_init_with_inertials(so100o_model* m, const double* inert) {
    double LMASS[SO100_NLINK], LIPOS[SO100_NLINK][3], LIQUAT[SO100_NLINK][4], LDIAG[SO100_NLINK][3];
    for (int k = 0; k < SO100_NLINK; k++) {
        const double* r = inert ? inert + 11*k : NULL;
        LMASS[k] = r ? r[0] : SO100_LINK_MASS[k];
        for (int a = 0; a < 3; a++) { LIPOS[k][a] = r ? r[1 + a] : SO100_LINK_IPOS[k][a]; LDIAG[k][a] = r ? r[8 + a] : SO100_LINK_DIAGINERTIA[k][a]; }
        for (int a = 0; a < 4; a++) LIQUAT[k][a] = r ? r[4 + a] : SO100_LINK_IQUAT[k][a];
    }
    memset(m, 0, sizeof *m);
    /* bodies 0 world, 1 Base: identity frames, no joints */
    for (int b = 0; b < NB; b++) {
        m->body_quat[b][0] = 1; m->body_iquat[b][0] = 1;
        m->body_dofadr[b] = -1; m->body_qposadr[b] = -1;
    }
    m->body_parent[0] = 0; m->body_parent[1] = 0;
    for (int k = 0; k < SO100_NLINK; k++) {
        int b = k + 2;
        m->body_parent[b] = b - 1;
        m->body_jnt[b] = 1; m->body_dofadr[b] = k; m->body_qposadr[b] = k;
        memcpy(m->body_pos[b], SO100_LINK_POS[k], sizeof(double)*3);
        if (SO100_LINK_ORI_KIND[k] == 0) {
            memcpy(m->body_quat[b], SO100_LINK_ORI[k], sizeof(double)*4);
            quat_normalize(m->body_quat[b]);
        } else {
            euler2quat(m->body_quat[b], SO100_LINK_ORI[k]);
        }
        memcpy(m->body_ipos[b], LIPOS[k], sizeof(double)*3);
        memcpy(m->body_iquat[b], LIQUAT[k], sizeof(double)*4);
        quat_normalize(m->body_iquat[b]);
        m->body_mass[b] = LMASS[k];
        memcpy(m->body_inertia[b], LDIAG[k], sizeof(double)*3);
        memcpy(m->jnt_axis[b], SO100_JNT_AXIS[k], sizeof(double)*3);
        m->jnt_range[k][0] = SO100_JNT_RANGE[k][0]; m->jnt_range[k][1] = SO100_JNT_RANGE[k][1];
        m->armature[k] = SO100_JNT_ARMATURE; m->frictionloss[k] = SO100_JNT_FRICTIONLOSS;
    }
    /* cube: mass and inertia from the box geom (inertiafromgeom, scene:2) */
    m->body_parent[CUBE] = 0; m->body_jnt[CUBE] = 2; m->body_dofadr[CUBE] = 6; m->body_qposadr[CUBE] = 6;
    {
        double a = 2.0 * SO100_CUBE_HALF;
        double mass = SO100_GEOM_DENSITY * a * a * a;
        double I = mass * (a*a + a*a) / 12.0;
        m->body_mass[CUBE] = mass;
        m->body_inertia[CUBE][0] = m->body_inertia[CUBE][1] = m->body_inertia[CUBE][2] = I;
    }
    memcpy(m->cam_pos, SO100_CAM_POS, sizeof(double)*3);
    euler2quat(m->cam_quat, SO100_CAM_EULER);
    m->kp = SO100_ACT_KP;
    m->timestep = SO100_TIMESTEP;
    m->gravity[2] = SO100_GRAVITY_Z;
    m->qpos0[9] = 1.0;                               /* cube quat identity */
    for (int g = 0; g < SO100_NPAD; g++) {
        m->pad_body[g] = SO100_PAD_LINK[g] + 2;
        memcpy(m->pad_pos[g], SO100_PAD_POS[g], sizeof(double)*3);
        memcpy(m->pad_size[g], SO100_PAD_SIZE[g], sizeof(double)*3);
    }
    m->pad_solref[0] = SO100_PAD_SOLREF_TIMECONST; m->pad_solref[1] = SO100_PAD_SOLREF_DAMPRATIO;
    m->pad_solimp[0] = SO100_PAD_SOLIMP_D0; m->pad_solimp[1] = SO100_PAD_SOLIMP_DMAX; m->pad_solimp[2] = SO100_PAD_SOLIMP_WIDTH;
    m->pad_solimp[3] = SO100_SOLIMP_MID; m->pad_solimp[4] = SO100_SOLIMP_POWER;
    m->pad_friction = SO100_PAD_FRICTION;
    m->def_solref[0] = SO100_SOLREF_TIMECONST; m->def_solref[1] = SO100_SOLREF_DAMPRATIO;
    m->def_solimp[0] = SO100_SOLIMP_D0; m->def_solimp[1] = SO100_SOLIMP_DMAX; m->def_solimp[2] = SO100_SOLIMP_WIDTH;
    m->def_solimp[3] = SO100_SOLIMP_MID; m->def_solimp[4] = SO100_SOLIMP_POWER;
    m->def_friction = SO100_GEOM_FRICTION;
    m->max_contacts = 16;                            /* the product's contact budget per env (csrc/so100_contact.hpp) */
    /* link proxies (stand-in capsules for the absent collision meshes; rule: so100_model_def.h) */
    for (int k = 0; k < SO100_NPROX; k++) {
        const int l = SO100_PROX_LINK[k], b = l + 2;
        m->prox_body[k] = b;
        memset(m->prox_p[k], 0, sizeof m->prox_p[k]);
        if (l <= 3) memcpy(m->prox_p[k][1], SO100_LINK_POS[l + 1], sizeof(double)*3);
        else {
            double xlo = 1e30, xhi = -1e30, ymax = 0;
            for (int g = 0; g < SO100_NPAD; g++) if (SO100_PAD_LINK[g] == l) {
                xlo = fmin(xlo, SO100_PAD_POS[g][0] - SO100_PAD_SIZE[g][0]); xhi = fmax(xhi, SO100_PAD_POS[g][0] + SO100_PAD_SIZE[g][0]);
                if (fabs(SO100_PAD_POS[g][1]) + SO100_PAD_SIZE[g][1] > fabs(ymax)) ymax = SO100_PAD_POS[g][1] < 0 ? SO100_PAD_POS[g][1] - SO100_PAD_SIZE[g][1] : SO100_PAD_POS[g][1] + SO100_PAD_SIZE[g][1];
            }
            m->prox_p[k][1][0] = 0.5*(xlo + xhi); m->prox_p[k][1][1] = ymax; m->prox_p[k][1][2] = 0.0;
        }
        const double* I = LDIAG[l]; const double mass = LMASS[l];
        double h[3] = { 0.5*sqrt(6.0*(I[1] + I[2] - I[0])/mass), 0.5*sqrt(6.0*(I[0] + I[2] - I[1])/mass), 0.5*sqrt(6.0*(I[0] + I[1] - I[2])/mass) };
        const double hmax = fmax(h[0], fmax(h[1], h[2]));
        double r = 0.5*(h[0] + h[1] + h[2] - hmax);
        if (l >= 4 && r > SO100_PROX_JAW_RADIUS_MAX) r = SO100_PROX_JAW_RADIUS_MAX;
        m->prox_radius[k] = r;
    }
    /* the same rule on links 0 (Rotation_Pitch) and 1 (Upper_Arm), against the cube (SO100O_F_LINKS_CUBE; scene:44-48 excludes every other arm body) */
    for (int k = 0; k < SO100O_NCPROX; k++) {
        const int l = k;
        m->cprox_body[k] = l + 2;
        memset(m->cprox_p[k], 0, sizeof m->cprox_p[k]);
        memcpy(m->cprox_p[k][1], SO100_LINK_POS[l + 1], sizeof(double)*3);
        const double* I = LDIAG[l]; const double mass = LMASS[l];
        double h[3] = { 0.5*sqrt(6.0*(I[1] + I[2] - I[0])/mass), 0.5*sqrt(6.0*(I[0] + I[2] - I[1])/mass), 0.5*sqrt(6.0*(I[0] + I[1] - I[2])/mass) };
        const double hmax = fmax(h[0], fmax(h[1], h[2]));
        m->cprox_radius[k] = 0.5*(h[0] + h[1] + h[2] - hmax);
    }

    /* mj_setConst: dof_M0, dof_invweight0, body_invweight0 at qpos0, then kv from dampratio */
    so100o_data* d = (so100o_data*)calloc(1, sizeof *d);
    so100o_reset_data(m, d);
    so100o_kinematics(m, d);
    /* crb is computed inside so100o_forward; do it here by hand */
    so100o__crb(m, d);
    double Minv_col[NV], e[NV];
    for (int i = 0; i < NV; i++) m->dof_M0[i] = d->M[i*NV + i];
    cholesky(d->M, d->L, NV);
    double diag[NV];
    for (int i = 0; i < NV; i++) {
        memset(e, 0, sizeof e); e[i] = 1;
        chol_solve(d->L, NV, e, Minv_col);
        diag[i] = Minv_col[i];
    }
    for (int i = 0; i < 6; i++) m->dof_invweight0[i] = diag[i];
    {   /* free joint: translational / rotational averages */
        double tr = (diag[6] + diag[7] + diag[8]) / 3.0, ro = (diag[9] + diag[10] + diag[11]) / 3.0;
        for (int i = 6; i < 9; i++)  m->dof_invweight0[i] = tr;
        for (int i = 9; i < 12; i++) m->dof_invweight0[i] = ro;
        m->body_invweight0[CUBE][0] = tr; m->body_invweight0[CUBE][1] = ro;
    }
    /* body_invweight0 of the arm links (mj_setConst): A = J M^-1 J^T with J = mj_jacBodyCom at qpos0 (6 x nv: linear rows
     * then angular rows); translational / rotational weight = mean of the respective diagonal block */
    for (int b = 2; b < CUBE; b++) {
        double Jb[6][NV], MiJ[6][NV];
        memset(Jb, 0, sizeof Jb);
        for (int j = 0; j <= b - 2; j++) {
            double w[3]; cross3(w, d->cdof[j], d->xipos[b]);           /* omega x p + v_origin */
            for (int a = 0; a < 3; a++) { Jb[a][j] = w[a] + d->cdof[j][3 + a]; Jb[3 + a][j] = d->cdof[j][a]; }
        }
        for (int r = 0; r < 6; r++) chol_solve(d->L, NV, Jb[r], MiJ[r]);
        double tr = 0, ro = 0;
        for (int r = 0; r < 3; r++) for (int i = 0; i < NV; i++) { tr += Jb[r][i]*MiJ[r][i]; ro += Jb[3 + r][i]*MiJ[3 + r][i]; }
        m->body_invweight0[b][0] = tr / 3.0; m->body_invweight0[b][1] = ro / 3.0;
    }
    /* position actuator damping from dampratio (mj_setConst): kv = dampratio*2*sqrt(kp*M0) */
    for (int i = 0; i < 6; i++) m->kv[i] = SO100_ACT_DAMPRATIO * 2.0 * sqrt(m->kp * m->dof_M0[i]);
    free(d);
}

void so100o_reset_data(const so100o_model* m, so100o_data* d) {
    /* mj_resetData: the whole buffer is zeroed (xpos, xmat, cam_* included), qpos = qpos0 */
    memset(d, 0, sizeof *d);
    memcpy(d->qpos, m->qpos0, sizeof d->qpos);
}

/* ================================================================================================
 * position stage: mj_kinematics, mj_comPos (cinert, cdof), mj_camlight
 * ============================================================================================== */
void so100o_kinematics(const so100o_model* m, so100o_data* d) {
    memset(d->xpos[0], 0, sizeof d->xpos[0]);
    d->xquat[0][0] = 1; d->xquat[0][1] = d->xquat[0][2] = d->xquat[0][3] = 0;
    quat2mat(d->xmat[0], d->xquat[0]);
    memset(d->xipos[0], 0, sizeof d->xipos[0]);
    quat2mat(d->ximat[0], d->xquat[0]);

    for (int b = 1; b < NB; b++) {
        double xpos[3], xquat[4];
        if (m->body_jnt[b] == 2) {                   /* free joint */
            int qa = m->body_qposadr[b], da = m->body_dofadr[b];
            memcpy(xpos, d->qpos + qa, sizeof xpos);
            memcpy(xquat, d->qpos + qa + 3, sizeof xquat);
            quat_normalize(xquat);
            for (int k = 0; k < 6; k++) memcpy(d->xanchor[da + k], xpos, sizeof xpos);
        } else {
            int p = m->body_parent[b];
            if (p) {
                double v[3]; mat_vec3(v, d->xmat[p], m->body_pos[b]);
                for (int i = 0; i < 3; i++) xpos[i] = d->xpos[p][i] + v[i];
                quat_mul(xquat, d->xquat[p], m->body_quat[b]);
            } else {
                memcpy(xpos, m->body_pos[b], sizeof xpos);
                memcpy(xquat, m->body_quat[b], sizeof xquat);
            }
            if (m->body_jnt[b] == 1) {               /* hinge, jnt_pos = 0 */
                int da = m->body_dofadr[b], qa = m->body_qposadr[b];
                double qloc[4];
                rot_vec_quat(d->xaxis[da], m->jnt_axis[b], xquat);
                memcpy(d->xanchor[da], xpos, sizeof xpos);
                axis_angle2quat(qloc, m->jnt_axis[b], d->qpos[qa] - m->qpos0[qa]);
                quat_mul(xquat, xquat, qloc);
                /* off-centre correction vanishes: jnt_pos = 0 => xpos = xanchor */
            }
        }
        quat_normalize(xquat);
        memcpy(d->xquat[b], xquat, sizeof xquat);
        memcpy(d->xpos[b], xpos, sizeof xpos);
        quat2mat(d->xmat[b], xquat);
    }
    /* inertial frames (mj_local2Global) */
    for (int b = 1; b < NB; b++) {
        double v[3], q[4];
        mat_vec3(v, d->xmat[b], m->body_ipos[b]);
        for (int i = 0; i < 3; i++) d->xipos[b][i] = d->xpos[b][i] + v[i];
        quat_mul(q, d->xquat[b], m->body_iquat[b]);
        quat2mat(d->ximat[b], q);
    }
    /* camera (mj_camlight, fixed mode) */
    {
        int b = SO100_CAM_LINK + 2; double v[3], q[4];
        mat_vec3(v, d->xmat[b], m->cam_pos);
        for (int i = 0; i < 3; i++) d->cam_xpos[i] = d->xpos[b][i] + v[i];
        quat_mul(q, d->xquat[b], m->cam_quat);
        quat2mat(d->cam_xmat, q);
    }
    /* mj_comPos: spatial inertia about the world origin, world axes */
    for (int b = 1; b < NB; b++) {
        double* I6 = d->cinert[b];
        memset(I6, 0, sizeof(double)*36);
        double mass = m->body_mass[b];
        if (mass <= 0) continue;
        const double* R = d->ximat[b]; const double* c = d->xipos[b];
        double Ic[9];
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) {
            double s = 0;
            for (int k = 0; k < 3; k++) s += R[3*i + k] * m->body_inertia[b][k] * R[3*j + k];
            Ic[3*i + j] = s;
        }
        double cc = dot3(c, c);
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++)
            I6[6*i + j] = Ic[3*i + j] + mass * ((i == j ? cc : 0.0) - c[i]*c[j]);
        double cx[9] = { 0, -c[2], c[1],  c[2], 0, -c[0],  -c[1], c[0], 0 };
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) {
            I6[6*i + 3 + j] = mass * cx[3*i + j];
            I6[6*(3 + i) + j] = mass * cx[3*j + i];
        }
        for (int i = 0; i < 3; i++) I6[6*(3 + i) + 3 + i] = mass;
    }
    /* cdof */
    for (int b = 1; b < NB; b++) {
        if (m->body_jnt[b] == 1) {
            int da = m->body_dofadr[b];
            memcpy(d->cdof[da], d->xaxis[da], sizeof(double)*3);
            cross3(d->cdof[da] + 3, d->xanchor[da], d->xaxis[da]);   /* axis x (O - anchor) */
        } else if (m->body_jnt[b] == 2) {
            int da = m->body_dofadr[b];
            for (int k = 0; k < 3; k++) {
                memset(d->cdof[da + k], 0, sizeof(double)*6);
                d->cdof[da + k][3 + k] = 1.0;
                double ax[3] = { d->xmat[b][k], d->xmat[b][3 + k], d->xmat[b][6 + k] };
                memcpy(d->xaxis[da + 3 + k], ax, sizeof ax);
                memcpy(d->cdof[da + 3 + k], ax, sizeof ax);
                cross3(d->cdof[da + 3 + k] + 3, d->xpos[b], ax);
            }
        }
    }
}

/* dof chain parent: serial arm 0..5, free joint 6..11 */
static int dof_parent(int i) { return (i == 0 || i == 6) ? -1 : i - 1; }
static int dof_body(int i) { return i < 6 ? i + 2 : CUBE; }

/* mj_crb: composite rigid body inertia and joint-space inertia matrix (dense) */
void so100o__crb(const so100o_model* m, so100o_data* d) {
    memcpy(d->crb, d->cinert, sizeof d->crb);
    for (int b = NB - 1; b > 0; b--) {
        int p = m->body_parent[b];
        if (p > 0) for (int i = 0; i < 36; i++) d->crb[p][i] += d->crb[b][i];
    }
    memset(d->M, 0, sizeof d->M);
    for (int i = 0; i < NV; i++) {
        double buf[6];
        mat6_vec(buf, d->crb[dof_body(i)], d->cdof[i]);
        for (int j = i; j >= 0; j = dof_parent(j)) {
            double v = dot6(d->cdof[j], buf);
            d->M[i*NV + j] = v; d->M[j*NV + i] = v;
        }
        d->M[i*NV + i] += m->armature[i];
    }
}

/* mj_comVel */
static void com_vel(const so100o_model* m, so100o_data* d) {
    memset(d->cvel, 0, sizeof d->cvel);
    for (int b = 1; b < NB; b++) {
        double cvel[6];
        memcpy(cvel, d->cvel[m->body_parent[b]], sizeof cvel);
        if (m->body_jnt[b] == 1) {
            int da = m->body_dofadr[b];
            cross_motion(d->cdof_dot[da], cvel, d->cdof[da]);
            for (int k = 0; k < 6; k++) cvel[k] += d->cdof[da][k] * d->qvel[da];
        } else if (m->body_jnt[b] == 2) {
            int da = m->body_dofadr[b];
            for (int j = 0; j < 3; j++) {
                memset(d->cdof_dot[da + j], 0, sizeof(double)*6);
                for (int k = 0; k < 6; k++) cvel[k] += d->cdof[da + j][k] * d->qvel[da + j];
            }
            for (int j = 3; j < 6; j++) cross_motion(d->cdof_dot[da + j], cvel, d->cdof[da + j]);
            for (int j = 3; j < 6; j++)
                for (int k = 0; k < 6; k++) cvel[k] += d->cdof[da + j][k] * d->qvel[da + j];
        }
        memcpy(d->cvel[b], cvel, sizeof cvel);
    }
}

/* mj_rne(flg_acc): out = M qacc + bias.  Requires kinematics + com_vel. */
static void rne(const so100o_model* m, so100o_data* d, const double* qacc, double* out) {
    double cacc[NB][6], cfrc[NB][6];
    memset(cacc, 0, sizeof cacc); memset(cfrc, 0, sizeof cfrc);
    for (int k = 0; k < 3; k++) cacc[0][3 + k] = -m->gravity[k];
    for (int b = 1; b < NB; b++) {
        memcpy(cacc[b], cacc[m->body_parent[b]], sizeof cacc[b]);
        int da = m->body_dofadr[b], nd = m->body_jnt[b] == 1 ? 1 : (m->body_jnt[b] == 2 ? 6 : 0);
        for (int j = 0; j < nd; j++) {
            for (int k = 0; k < 6; k++) cacc[b][k] += d->cdof_dot[da + j][k] * d->qvel[da + j];
            if (qacc) for (int k = 0; k < 6; k++) cacc[b][k] += d->cdof[da + j][k] * qacc[da + j];
        }
        double t[6], t1[6];
        mat6_vec(cfrc[b], d->cinert[b], cacc[b]);
        mat6_vec(t, d->cinert[b], d->cvel[b]);
        cross_force(t1, d->cvel[b], t);
        for (int k = 0; k < 6; k++) cfrc[b][k] += t1[k];
    }
    for (int b = NB - 1; b > 0; b--) {
        int p = m->body_parent[b];
        if (p > 0) for (int k = 0; k < 6; k++) cfrc[p][k] += cfrc[b][k];
    }
    for (int i = 0; i < NV; i++) out[i] = dot6(d->cdof[i], cfrc[dof_body(i)]);
}

void so100o_rne(const so100o_model* m, so100o_data* d, const double* qacc_or_null, double* out) {
    so100o_kinematics(m, d);
    com_vel(m, d);
    rne(m, d, qacc_or_null, out);
    if (qacc_or_null) for (int i = 0; i < NV; i++) out[i] += m->armature[i] * qacc_or_null[i];
}

/* dense Cholesky A = L L^T (mj_factorM's role) */
static void cholesky(const double* A, double* L, int n) {
    memset(L, 0, sizeof(double)*n*n);
    for (int j = 0; j < n; j++) {
        double s = A[j*n + j];
        for (int k = 0; k < j; k++) s -= L[j*n + k]*L[j*n + k];
        L[j*n + j] = sqrt(s);
        for (int i = j + 1; i < n; i++) {
            double t = A[i*n + j];
            for (int k = 0; k < j; k++) t -= L[i*n + k]*L[j*n + k];
            L[i*n + j] = t / L[j*n + j];
        }
    }
}
static void chol_solve(const double* L, int n, const double* b, double* x) {
    double y[SO100O_MAXEFC];
    for (int i = 0; i < n; i++) {
        double s = b[i];
        for (int k = 0; k < i; k++) s -= L[i*n + k]*y[k];
        y[i] = s / L[i*n + i];
    }
    for (int i = n - 1; i >= 0; i--) {
        double s = y[i];
        for (int k = i + 1; k < n; k++) s -= L[k*n + i]*x[k];
        x[i] = s / L[i*n + i];
    }
}

/* ================================================================================================
 * constraint rows (mj_makeConstraint + mj_makeImpedance) and the PGS solve (mj_solPGS)
 * ============================================================================================== */
static double impedance_of(const double solimp[5], double pos_minus_margin) {      /* getimpedance() */
    const double d0 = solimp[0], dm = solimp[1], w = solimp[2], mid = solimp[3], p = solimp[4];
    if (d0 == dm || w <= SO100_MJMINVAL) return 0.5*(d0 + dm);
    double x = fabs(pos_minus_margin / w), y;
    if (x >= 1) return dm;
    if (x <= 0) return d0;
    if (p == 1) y = x;
    else if (x <= mid) y = pow(x, p) / pow(mid, p - 1);
    else               y = 1 - pow(1 - x, p) / pow(1 - mid, p - 1);
    return d0 + y * (dm - d0);
}
static void solref_KB_of(const double solref[2], double dmax, double h, double* K, double* B) {
    double tc = solref[0], dr = solref[1];
    if (tc < 2*h) tc = 2*h;                                  /* refsafe */
    *K = 1.0 / fmax(SO100_MJMINVAL, dmax*dmax*tc*tc*dr*dr);
    *B = 2.0 / fmax(SO100_MJMINVAL, dmax*tc);
}
static const double DEF_SOLIMP[5] = { SO100_SOLIMP_D0, SO100_SOLIMP_DMAX, SO100_SOLIMP_WIDTH, SO100_SOLIMP_MID, SO100_SOLIMP_POWER };
static const double DEF_SOLREF[2] = { SO100_SOLREF_TIMECONST, SO100_SOLREF_DAMPRATIO };
static double impedance(double pos_minus_margin) { return impedance_of(DEF_SOLIMP, pos_minus_margin); }   /* default solimp */
static void solref_KB(double h, double* K, double* B) { solref_KB_of(DEF_SOLREF, SO100_SOLIMP_DMAX, h, K, B); }
/* mj_contactParam (equal priority, solmix 1 : 1) followed by mj_assignRef / mj_assignImp (the clamp comes AFTER the mix) */
static void mix_contact_params(const double ref1[2], const double imp1[5], double fr1, const double ref2[2], const double imp2[5], double fr2,
                               double solref[2], double solimp[5], double* mu) {
    const double mix = 0.5;
    if (ref1[0] > 0 && ref2[0] > 0) for (int i = 0; i < 2; i++) solref[i] = mix*ref1[i] + (1 - mix)*ref2[i];
    else for (int i = 0; i < 2; i++) solref[i] = fmin(ref1[i], ref2[i]);
    for (int i = 0; i < 5; i++) solimp[i] = mix*imp1[i] + (1 - mix)*imp2[i];
    solimp[0] = fmin(SO100_MJMAXIMP, fmax(SO100_MJMINIMP, solimp[0]));
    solimp[1] = fmin(SO100_MJMAXIMP, fmax(SO100_MJMINIMP, solimp[1]));
    solimp[2] = fmax(0.0, solimp[2]);
    solimp[3] = fmin(SO100_MJMAXIMP, fmax(SO100_MJMINIMP, solimp[3]));
    solimp[4] = fmax(1.0, solimp[4]);
    *mu = fmax(fr1, fr2);
}
static int add_row(so100o_data* d, int type, int id, const double* J, double pos, double floss,
                   double diagApprox, double K, double B) {
    int r = d->nefc++;
    d->efc_type[r] = type; d->efc_id[r] = id; d->efc_pos[r] = pos; d->efc_floss[r] = floss;
    memcpy(d->efc_J[r], J, sizeof(double)*NV);
    double imp = impedance(pos);
    double vel = 0; for (int i = 0; i < NV; i++) vel += J[i]*d->qvel[i];
    double Kr = (type == 0) ? 0.0 : K;                       /* friction rows: K = 0 */
    d->efc_aref[r] = -B*vel - Kr*imp*pos;
    d->efc_R[r] = fmax(SO100_MJMINVAL, (1 - imp)*diagApprox/imp);
    return r;
}

/* ------------------------------------------------------------------------------------------------
 * narrowphase.  Boxes are (centre c, rotation R row-major with the box axes as COLUMNS, half sizes h).
 * ---------------------------------------------------------------------------------------------- */
/* mjc_PlaneBox against the floor z = 0 (normal +z): every corner that is at / below the plane AND below the box centre, in
 * corner order, at most 4; contact point midway between the corner and the plane */
static int plane_box_ids(const double c[3], const double R[9], const double h[3], double pos[4][3], double dist[4], int corner_id[4]) {
    int cnt = 0;
    const double cdist = c[2];
    for (int k = 0; k < 8 && cnt < 4; k++) {
        const double v[3] = { (k & 1) ? h[0] : -h[0], (k & 2) ? h[1] : -h[1], (k & 4) ? h[2] : -h[2] };
        double corner[3]; mat_vec3(corner, R, v);
        const double ldist = corner[2];
        if (cdist + ldist > 0 || ldist > 0) continue;
        dist[cnt] = cdist + ldist;
        pos[cnt][0] = corner[0] + c[0]; pos[cnt][1] = corner[1] + c[1]; pos[cnt][2] = corner[2] + c[2] - dist[cnt]*0.5;
        if (corner_id) corner_id[cnt] = k;
        cnt++;
    }
    return cnt;
}
int so100o_plane_box(const double c[3], const double R[9], const double h[3], double pos[4][3], double dist[4]) {
    return plane_box_ids(c, R, h, pos, dist, NULL);
}

/* Box-box, in the manner of mjc_BoxBox: separating-axis test over the 15 candidate axes, then
 *   face axis  -> the incident face of the other box is clipped against the side planes of the reference face; contacts are
 *                 the clipped polygon's vertices that lie at / below the reference face (at most 8), each midway between
 *                 the two surfaces along the normal;
 *   edge axis  -> one contact at the midpoint of the closest points of the two edges.
 * The clipped polygon is enumerated in a fixed slot order (no dynamic vertex lists, so that the fp32 device code can follow
 * the same order): per incident edge k = 0..3 its Liang-Barsky entry point (or start vertex) and, if it leaves the rectangle
 * before its end vertex, its exit point; then the reference rectangle's corners that lie inside the incident face.
 * An edge axis must beat the best face axis by 5 % (+ 1e-9) to be chosen: face contacts are the stabler manifold.
 * The normal points from A to B.  MuJoCo's own function is not available to compare against: "parity unpinned". */
int so100o_box_box(const double cA[3], const double RA[9], const double hA[3], const double cB[3], const double RB[9],
                   const double hB[3], double pos[8][3], double normal[3], double dist[8]) {
    double a[3][3], b[3][3], dd[3];                         /* box axes in the world */
    for (int i = 0; i < 3; i++) for (int k = 0; k < 3; k++) { a[i][k] = RA[3*k + i]; b[i][k] = RB[3*k + i]; }
    for (int k = 0; k < 3; k++) dd[k] = cB[k] - cA[k];
    double Rm[3][3], Ra[3][3];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { Rm[i][j] = dot3(a[i], b[j]); Ra[i][j] = fabs(Rm[i][j]); }
    double best = -1e300, bn[3] = {0, 0, 0}; int code = -1;
    /* face axes of A (code 0-2) and of B (3-5) */
    for (int i = 0; i < 3; i++) {
        const double proj = dot3(dd, a[i]);
        const double sep = fabs(proj) - (hA[i] + hB[0]*Ra[i][0] + hB[1]*Ra[i][1] + hB[2]*Ra[i][2]);
        if (sep > 0) return 0;
        if (sep > best) { best = sep; code = i; const double sg = proj < 0 ? -1.0 : 1.0; for (int k = 0; k < 3; k++) bn[k] = sg*a[i][k]; }
    }
    for (int j = 0; j < 3; j++) {
        const double proj = dot3(dd, b[j]);
        const double sep = fabs(proj) - (hB[j] + hA[0]*Ra[0][j] + hA[1]*Ra[1][j] + hA[2]*Ra[2][j]);
        if (sep > 0) return 0;
        if (sep > best) { best = sep; code = 3 + j; const double sg = proj < 0 ? -1.0 : 1.0; for (int k = 0; k < 3; k++) bn[k] = sg*b[j][k]; }
    }
    /* edge axes a_i x b_j (code 6 + 3 i + j) */
    double ebest = -1e300, en[3] = {0, 0, 0}; int ecode = -1;
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) {
        double L[3]; cross3(L, a[i], b[j]);
        const double len = sqrt(dot3(L, L));
        if (len < 1e-6) continue;                           /* parallel edges: the face axes cover it */
        for (int k = 0; k < 3; k++) L[k] /= len;
        const double proj = dot3(dd, L);
        double ra = 0, rb = 0;
        for (int k = 0; k < 3; k++) { ra += hA[k]*fabs(dot3(a[k], L)); rb += hB[k]*fabs(dot3(b[k], L)); }
        const double sep = fabs(proj) - (ra + rb);
        if (sep > 0) return 0;
        if (sep > ebest) { ebest = sep; ecode = 6 + 3*i + j; const double sg = proj < 0 ? -1.0 : 1.0; for (int k = 0; k < 3; k++) en[k] = sg*L[k]; }
    }
    if (ecode >= 0 && ebest*1.05 > best + 1e-9) { best = ebest; code = ecode; memcpy(bn, en, sizeof bn); }
    memcpy(normal, bn, sizeof bn);

    if (code >= 6) {
        /* edge-edge: the supporting edge of A in direction +n and of B in direction -n; closest points of the two lines */
        const int i = (code - 6) / 3, j = (code - 6) % 3;
        double pa[3], pb[3];
        for (int k = 0; k < 3; k++) { pa[k] = cA[k]; pb[k] = cB[k]; }
        for (int q = 0; q < 3; q++) {
            if (q != i) { const double sg = dot3(bn, a[q]) > 0 ? 1.0 : -1.0; for (int k = 0; k < 3; k++) pa[k] += sg*hA[q]*a[q][k]; }
            if (q != j) { const double sg = dot3(bn, b[q]) > 0 ? -1.0 : 1.0; for (int k = 0; k < 3; k++) pb[k] += sg*hB[q]*b[q][k]; }
        }
        /* lines pa + s a_i, pb + t b_j */
        double w0[3]; for (int k = 0; k < 3; k++) w0[k] = pb[k] - pa[k];
        const double uu = Rm[i][j], d1 = dot3(a[i], w0), d2 = dot3(b[j], w0), den = 1.0 - uu*uu;
        double sa = 0, tb = 0;
        if (den > 1e-12) { sa = (d1 - uu*d2)/den; tb = (uu*d1 - d2)/den; }
        sa = sa < -hA[i] ? -hA[i] : (sa > hA[i] ? hA[i] : sa);
        tb = tb < -hB[j] ? -hB[j] : (tb > hB[j] ? hB[j] : tb);
        for (int k = 0; k < 3; k++) pos[0][k] = 0.5*((pa[k] + sa*a[i][k]) + (pb[k] + tb*b[j][k]));
        dist[0] = best;
        return 1;
    }

    /* face contact: X = reference box (owner of the axis), Y = incident box */
    const int refA = code < 3, r = refA ? code : code - 3;
    const double (*x)[3] = refA ? a : b; const double (*y)[3] = refA ? b : a;
    const double* cX = refA ? cA : cB; const double* cY = refA ? cB : cA;
    const double* hX = refA ? hA : hB; const double* hY = refA ? hB : hA;
    double nref[3]; for (int k = 0; k < 3; k++) nref[k] = refA ? bn[k] : -bn[k];          /* outward normal of the reference face */
    int mi = 0; double mv = -1;
    for (int k = 0; k < 3; k++) { const double v = fabs(dot3(nref, y[k])); if (v > mv) { mv = v; mi = k; } }
    const double fs = dot3(nref, y[mi]) > 0 ? -1.0 : 1.0;   /* incident face: the one whose outward normal opposes nref most */
    const int p1 = (mi + 1) % 3, p2 = (mi + 2) % 3, u1 = (r + 1) % 3, u2 = (r + 2) % 3;
    double fc[3], rc[3];
    for (int k = 0; k < 3; k++) { fc[k] = cY[k] + fs*hY[mi]*y[mi][k] - (cX[k] + hX[r]*nref[k]); rc[k] = cX[k] + hX[r]*nref[k]; }
    /* incident face in reference-face coordinates (u, v in the rectangle's plane, w = signed height above it): centre + alpha e1 + beta e2 */
    const double c0[3] = { dot3(fc, x[u1]), dot3(fc, x[u2]), dot3(fc, nref) };
    const double e1[3] = { hY[p1]*dot3(y[p1], x[u1]), hY[p1]*dot3(y[p1], x[u2]), hY[p1]*dot3(y[p1], nref) };
    const double e2[3] = { hY[p2]*dot3(y[p2], x[u1]), hY[p2]*dot3(y[p2], x[u2]), hY[p2]*dot3(y[p2], nref) };
    const double hu = hX[u1], hv = hX[u2];
    static const double sa[4] = { -1, 1, 1, -1 }, sb[4] = { -1, -1, 1, 1 };             /* quad vertices in order */
    double cand[12][3]; int valid[12];
    for (int k = 0; k < 4; k++) {
        const int k2 = (k + 1) & 3;
        double P0[3], P1[3];
        for (int q = 0; q < 3; q++) { P0[q] = c0[q] + sa[k]*e1[q] + sb[k]*e2[q]; P1[q] = c0[q] + sa[k2]*e1[q] + sb[k2]*e2[q]; }
        /* Liang-Barsky on |u| <= hu, |v| <= hv */
        double t0 = 0, t1 = 1; int ok = 1;
        const double dq[2] = { P1[0] - P0[0], P1[1] - P0[1] }, lim[2] = { hu, hv };
        for (int ax = 0; ax < 2 && ok; ax++) {
            for (int side = -1; side <= 1; side += 2) {
                const double pden = -side*dq[ax], pnum = side*P0[ax] - lim[ax];          /* side*(P0 + t dq) <= lim  <=>  t*(-pden) <= -pnum */
                if (pden == 0) { if (pnum > 0) ok = 0; }
                else {
                    const double t = pnum/pden;
                    if (pden > 0) { if (t > t0) t0 = t; }     /* entering */
                    else          { if (t < t1) t1 = t; }     /* leaving  */
                }
            }
        }
        if (t0 > t1) ok = 0;
        valid[2*k] = ok; valid[2*k + 1] = ok && t1 < 1.0;
        for (int q = 0; q < 3; q++) { cand[2*k][q] = P0[q] + t0*(P1[q] - P0[q]); cand[2*k + 1][q] = P0[q] + t1*(P1[q] - P0[q]); }
    }
    {   /* rectangle corners inside the incident parallelogram: (u,v) = c0 + alpha e1 + beta e2, |alpha|, |beta| < 1 */
        const double det = e1[0]*e2[1] - e1[1]*e2[0];
        for (int k = 0; k < 4; k++) {
            const double uu = sa[k]*hu - c0[0], vv = sb[k]*hv - c0[1];
            int ok = fabs(det) > 1e-18;
            double al = 0, be = 0;
            if (ok) { al = (uu*e2[1] - vv*e2[0])/det; be = (e1[0]*vv - e1[1]*uu)/det; ok = fabs(al) < 1.0 && fabs(be) < 1.0; }
            valid[8 + k] = ok;
            cand[8 + k][0] = sa[k]*hu; cand[8 + k][1] = sb[k]*hv; cand[8 + k][2] = c0[2] + al*e1[2] + be*e2[2];
        }
    }
    int cnt = 0;
    for (int k = 0; k < 12 && cnt < 8; k++) {
        if (!valid[k] || cand[k][2] > 0) continue;
        const double w = cand[k][2];
        for (int q = 0; q < 3; q++) pos[cnt][q] = rc[q] + cand[k][0]*x[u1][q] + cand[k][1]*x[u2][q] + 0.5*w*nref[q];
        dist[cnt] = w;
        cnt++;
    }
    return cnt;
}

/* Capsule (segment a..b, radius r; geom1) against a box (centre c, rotation R row-major world <- box, half sizes h; geom2): a STAND-IN for
 * mjc_CapsuleBox, like the capsules themselves.  The point of the segment nearest to the box minimises f(t) = dist^2(a + t (b - a), box) over
 * [0, 1], a convex piecewise-quadratic function: f' is non-decreasing and LINEAR between the (at most six) parameters at which a coordinate
 * of the point crosses a face plane.  f' is evaluated at those and at the two ends; the root lies between the last candidate with f' < 0
 * and the first with f' >= 0, by linear interpolation -- exact, no iteration.  Then a sphere-box test at that point.  One contact: position
 * midway between the two surfaces, normal from the capsule to the box.  Returns 0 / 1. */
static double capsule_box_slope(const double la[3], const double d[3], const double h[3], double t) {
    double g = 0;
    for (int k = 0; k < 3; k++) { const double sk = la[k] + t*d[k]; g += (sk - fmin(fmax(sk, -h[k]), h[k]))*d[k]; }
    return g;
}
int so100o_capsule_box(const double a[3], const double b[3], double r, const double c[3], const double R[9], const double h[3],
                       double pos[3], double normal[3], double* dist) {
    double la[3], d[3], s[3], q[3], e[3], tj[8], gj[8];
    for (int k = 0; k < 3; k++) {                          /* box frame: local = R' (world - c) */
        la[k] = R[k]*(a[0] - c[0]) + R[3 + k]*(a[1] - c[1]) + R[6 + k]*(a[2] - c[2]);
        d[k] = R[k]*(b[0] - a[0]) + R[3 + k]*(b[1] - a[1]) + R[6 + k]*(b[2] - a[2]);
    }
    tj[0] = 0.0; tj[1] = 1.0;
    for (int k = 0; k < 3; k++) {
        const double inv = fabs(d[k]) > 1e-12 ? 1.0/d[k] : 0.0;      /* (a coordinate that does not move crosses no plane: candidate 0 again) */
        tj[2 + 2*k] = fmin(fmax((-h[k] - la[k])*inv, 0.0), 1.0); tj[3 + 2*k] = fmin(fmax((h[k] - la[k])*inv, 0.0), 1.0);
    }
    for (int j = 0; j < 8; j++) gj[j] = capsule_box_slope(la, d, h, tj[j]);
    double thi = 2.0, ghi = 0.0, tlo = -1.0, glo = 0.0;
    for (int j = 0; j < 8; j++) if (gj[j] >= 0 && tj[j] < thi) { thi = tj[j]; ghi = gj[j]; }
    if (thi > 1.5) { thi = 1.0; ghi = 0.0; }               /* f' < 0 on the whole segment: the far end */
    for (int j = 0; j < 8; j++) if (gj[j] < 0 && tj[j] <= thi && tj[j] > tlo) { tlo = tj[j]; glo = gj[j]; }
    double t = thi;
    if (tlo >= 0.0 && ghi > 0) t = tlo - glo*(thi - tlo)/(ghi - glo);
    /* the axis itself passes through the box (f = 0 on an interval, found by the slab test): the middle of that interval */
    double tin = 0.0, tout = 1.0;
    for (int k = 0; k < 3; k++) {
        if (fabs(d[k]) > 1e-12) {
            const double t1 = (-h[k] - la[k])/d[k], t2 = (h[k] - la[k])/d[k];
            tin = fmax(tin, fmin(t1, t2)); tout = fmin(tout, fmax(t1, t2));
        } else if (fabs(la[k]) > h[k]) tin = 2.0;
    }
    if (tin <= tout) t = 0.5*(tin + tout);
    for (int k = 0; k < 3; k++) { s[k] = la[k] + t*d[k]; q[k] = fmin(fmax(s[k], -h[k]), h[k]); e[k] = s[k] - q[k]; }
    const double len = sqrt(dot3(e, e));
    double n[3] = { 0, 0, 0 }, dst;
    if (len > 1e-9) { for (int k = 0; k < 3; k++) n[k] = e[k]/len; dst = len - r; }
    else {                                                  /* the axis point is inside the box: leave through the nearest face */
        int ax = 0; double best = h[0] - fabs(s[0]);
        for (int k = 1; k < 3; k++) if (h[k] - fabs(s[k]) < best) { best = h[k] - fabs(s[k]); ax = k; }
        n[ax] = s[ax] < 0 ? -1.0 : 1.0; q[ax] = n[ax]*h[ax]; dst = -best - r;
    }
    if (dst > 0) return 0;
    double pl[3];
    for (int k = 0; k < 3; k++) pl[k] = q[k] + 0.5*dst*n[k];
    for (int k = 0; k < 3; k++) {
        pos[k] = c[k] + R[3*k]*pl[0] + R[3*k + 1]*pl[1] + R[3*k + 2]*pl[2];
        normal[k] = -(R[3*k]*n[0] + R[3*k + 1]*n[1] + R[3*k + 2]*n[2]);
    }
    *dist = dst;
    return 1;
}

/* point Jacobian row of body b at world point p along direction dir, accumulated into J with sign sg */
static void jac_point_dir(const so100o_data* d, int b, const double p[3], const double dir[3], double sg, double* J) {
    if (b <= 1) return;                                     /* world / welded base */
    const int j0 = b == CUBE ? 6 : 0, j1 = b == CUBE ? NV : b - 1;     /* arm body b moves with dofs 0 .. b-2 */
    for (int j = j0; j < j1; j++) {
        double w[3]; cross3(w, d->cdof[j], p);              /* omega x p + v_origin */
        for (int a = 0; a < 3; a++) w[a] += d->cdof[j][3 + a];
        J[j] += sg*dot3(dir, w);
    }
}

static int add_contact(const so100o_model* m, so100o_data* d, int kind, int geom, int feat, int b1, int b2, const double pos[3],
                       const double normal[3], double dist, double mu, const double solref[2], const double solimp[5]) {
    /* the product's contact budget (csrc/so100_contact.hpp: MAXPADC) counts PAD contacts, in detection order pad/floor by
     * pad then pad/cube by pad; the cube's own <= 4 floor contacts are outside it */
    if (kind != 0) {
        int npad = 0; for (int i = 0; i < d->ncon; i++) npad += d->con[i].kind != 0;
        if ((m->max_contacts > 0 && npad >= m->max_contacts) || d->ncon >= SO100O_MAXCON) { d->ncon_dropped++; return -1; }
    }
    so100o_contact* c = &d->con[d->ncon];
    c->b1 = b1; c->b2 = b2; c->kind = kind; c->geom = geom; c->feat = feat; c->dist = dist; c->mu = mu;
    memcpy(c->pos, pos, sizeof c->pos); memcpy(c->solref, solref, sizeof c->solref); memcpy(c->solimp, solimp, sizeof c->solimp);
    /* mju_makeFrame: t1 = (0,1,0) if |n_y| < 0.5 else (0,0,1), orthogonalised against n; t2 = n x t1 */
    double* f = c->frame;
    memcpy(f, normal, sizeof(double)*3);
    f[3] = 0; f[4] = 0; f[5] = 0;
    if (f[1] < 0.5 && f[1] > -0.5) f[4] = 1; else f[5] = 1;
    { const double dp = dot3(f, f + 3); for (int k = 0; k < 3; k++) f[3 + k] -= dp*f[k];
      const double nn = sqrt(dot3(f + 3, f + 3)); for (int k = 0; k < 3; k++) f[3 + k] /= nn; }
    cross3(f + 6, f, f + 3);
    /* pyramidal rows (condim 3): J = (jac_b2 - jac_b1)(p) . (n +- mu t_k); aref = -B Jv - K imp dist;
     * diagApprox = tran (1 + mu^2), tran = body_invweight0 of both bodies; R = 2 mu^2 R(first edge), impratio 1 */
    double K, B, J[NV];
    solref_KB_of(solref, solimp[1], m->timestep, &K, &B);
    const double imp = impedance_of(solimp, dist);
    const double tran = m->body_invweight0[b1][0] + m->body_invweight0[b2][0];
    const double R0 = fmax(SO100_MJMINVAL, (1 - imp)*(tran + mu*mu*tran)/imp), Rpy = 2*mu*mu*R0;
    c->efc0 = d->nefc;
    for (int k = 0; k < 2; k++)
        for (int sgn = 1; sgn >= -1; sgn -= 2) {
            double dir[3];
            for (int a = 0; a < 3; a++) dir[a] = f[a] + sgn*mu*f[3 + 3*k + a];
            memset(J, 0, sizeof J);
            jac_point_dir(d, b2, pos, dir, 1.0, J);
            jac_point_dir(d, b1, pos, dir, -1.0, J);
            const int r = d->nefc++;
            d->efc_type[r] = 2; d->efc_id[r] = kind == 0 ? 4*d->ncon + 2*k + (sgn < 0) : -1;
            d->efc_pos[r] = dist; d->efc_floss[r] = 0;
            memcpy(d->efc_J[r], J, sizeof J);
            double vel = 0; for (int i = 0; i < NV; i++) vel += J[i]*d->qvel[i];
            d->efc_aref[r] = -B*vel - K*imp*dist;
            d->efc_R[r] = Rpy;
        }
    return d->ncon++;
}

static void make_constraints(const so100o_model* m, so100o_data* d, unsigned flags) {
    double K, B, J[NV];
    solref_KB(m->timestep, &K, &B);
    d->nefc = 0; d->ncon = 0; d->ncon_dropped = 0;
    if (flags & SO100O_F_FRICTIONLOSS)
        for (int i = 0; i < 6; i++) if (m->frictionloss[i] > 0) {
            memset(J, 0, sizeof J); J[i] = 1;
            add_row(d, 0, i, J, 0.0, m->frictionloss[i], m->dof_invweight0[i], K, B);
        }
    if (flags & SO100O_F_LIMITS)
        for (int i = 0; i < 6; i++)
            for (int side = -1; side <= 1; side += 2) {
                double dist = side * (m->jnt_range[i][(side + 1)/2] - d->qpos[i]);
                if (dist < 0) {
                    memset(J, 0, sizeof J); J[i] = -side;
                    add_row(d, 1, 2*i + (side + 1)/2, J, dist, 0.0, m->dof_invweight0[i], K, B);
                }
            }
    const double nz[3] = {0, 0, 1};
    /* contacts in geom-pair order: the arm's geoms come first (attached first, scene:24-26), so pad/floor and pad/cube pairs
     * precede cube/floor.  Cube/floor keeps its place at the front here for the benefit of the PGS warm start identities;
     * the converged solution does not depend on the order. */
    if ((flags & SO100O_F_FLOOR) && !(flags & SO100O_F_CUBE_PINNED)) {
        double hs[3] = { SO100_CUBE_HALF, SO100_CUBE_HALF, SO100_CUBE_HALF }, pos[4][3], dist[4], ref[2], imp[5], mu;
        mix_contact_params(m->def_solref, m->def_solimp, m->def_friction, m->def_solref, m->def_solimp, m->def_friction, ref, imp, &mu);
        int cid[4];
        const int n = plane_box_ids(d->xpos[CUBE], d->xmat[CUBE], hs, pos, dist, cid);
        for (int k = 0; k < n; k++) add_contact(m, d, 0, 0, 128 + cid[k], 0, CUBE, pos[k], nz, dist[k], mu, ref, imp);
    }
    if (flags & SO100O_F_PADS_FLOOR) {
        double ref[2], imp[5], mu;
        mix_contact_params(m->def_solref, m->def_solimp, m->def_friction, m->pad_solref, m->pad_solimp, m->pad_friction, ref, imp, &mu);
        for (int g = 0; g < SO100O_NPAD; g++) {
            const int b = m->pad_body[g];
            double c[3], v[3], pos[4][3], dist[4]; int cid[4];
            mat_vec3(v, d->xmat[b], m->pad_pos[g]);
            for (int k = 0; k < 3; k++) c[k] = d->xpos[b][k] + v[k];
            const int n = plane_box_ids(c, d->xmat[b], m->pad_size[g], pos, dist, cid);
            for (int k = 0; k < n; k++) add_contact(m, d, 1, g, 8*g + cid[k], 0, b, pos[k], nz, dist[k], mu, ref, imp);
        }
    }
    if ((flags & SO100O_F_PADS_CUBE) && !(flags & SO100O_F_CUBE_PINNED)) {
        double ref[2], imp[5], mu, hs[3] = { SO100_CUBE_HALF, SO100_CUBE_HALF, SO100_CUBE_HALF };
        mix_contact_params(m->pad_solref, m->pad_solimp, m->pad_friction, m->def_solref, m->def_solimp, m->def_friction, ref, imp, &mu);
        for (int g = 0; g < SO100O_NPAD; g++) {
            const int b = m->pad_body[g];
            double c[3], v[3], pos[8][3], dist[8], nrm[3];
            mat_vec3(v, d->xmat[b], m->pad_pos[g]);
            for (int k = 0; k < 3; k++) c[k] = d->xpos[b][k] + v[k];
            const int n = so100o_box_box(c, d->xmat[b], m->pad_size[g], d->xpos[CUBE], d->xmat[CUBE], hs, pos, nrm, dist);
            for (int k = 0; k < n; k++) add_contact(m, d, 2, g, 64 + 8*g + k, b, CUBE, pos[k], nrm, dist[k], mu, ref, imp);
        }
    }
    if (flags & SO100O_F_LINKS_FLOOR) {                     /* (after the pad pairs: the order the product's contact store is filled in) */
        /* mjc_PlaneCapsule: a plane-sphere test at either end of the segment; contact point midway between the sphere's lowest point and the plane */
        double ref[2], imp[5], mu;
        mix_contact_params(m->def_solref, m->def_solimp, m->def_friction, m->def_solref, m->def_solimp, m->def_friction, ref, imp, &mu);
        for (int k = 0; k < SO100O_NPROX; k++) {
            const int b = m->prox_body[k];
            for (int e = 0; e < 2; e++) {
                double c[3], v[3];
                mat_vec3(v, d->xmat[b], m->prox_p[k][e]);
                for (int a = 0; a < 3; a++) c[a] = d->xpos[b][a] + v[a];
                const double dist = c[2] - m->prox_radius[k];
                if (dist > 0) continue;
                const double pos[3] = { c[0], c[1], c[2] - m->prox_radius[k] - 0.5*dist };
                add_contact(m, d, 3, k, 144 + 2*k + e, 0, b, pos, nz, dist, mu, ref, imp);
            }
        }
    }
    if ((flags & SO100O_F_LINKS_CUBE) && !(flags & SO100O_F_CUBE_PINNED)) {          /* Q7: Rotation_Pitch / Upper_Arm vs block_a (scene:44-48 excludes the rest) */
        double ref[2], imp[5], mu, hs[3] = { SO100_CUBE_HALF, SO100_CUBE_HALF, SO100_CUBE_HALF };
        mix_contact_params(m->def_solref, m->def_solimp, m->def_friction, m->def_solref, m->def_solimp, m->def_friction, ref, imp, &mu);
        for (int k = 0; k < SO100O_NCPROX; k++) {
            const int b = m->cprox_body[k];
            double e[2][3], v[3], pos[3], nrm[3], dist;
            for (int j = 0; j < 2; j++) {
                mat_vec3(v, d->xmat[b], m->cprox_p[k][j]);
                for (int a = 0; a < 3; a++) e[j][a] = d->xpos[b][a] + v[a];
            }
            if (so100o_capsule_box(e[0], e[1], m->cprox_radius[k], d->xpos[CUBE], d->xmat[CUBE], hs, pos, nrm, &dist))
                add_contact(m, d, 4, k, 160 + k, b, CUBE, pos, nrm, dist, mu, ref, imp);
        }
    }
}

/* projected Gauss-Seidel on the dual: min 1/2 f'(A+R)f + f'b, f in the row's admissible set */
static void solve_pgs(const so100o_model* m, so100o_data* d, int iters) {
    int n = d->nefc;
    memset(d->qfrc_constraint, 0, sizeof d->qfrc_constraint);
    d->solver_iter_used = 0; d->solver_last_change = 0;
    if (n == 0) { memcpy(d->qacc, d->qacc_smooth, sizeof d->qacc); return; }
    static _Thread_local double AR[SO100O_MAXEFC*SO100O_MAXEFC], MinvJT[SO100O_MAXEFC][NV];
    for (int r = 0; r < n; r++) chol_solve(d->L, NV, d->efc_J[r], MinvJT[r]);
    for (int r = 0; r < n; r++) {
        for (int c = 0; c < n; c++) {
            double s = 0; for (int i = 0; i < NV; i++) s += d->efc_J[r][i]*MinvJT[c][i];
            AR[r*n + c] = s;
        }
        AR[r*n + r] += d->efc_R[r];
        double s = 0; for (int i = 0; i < NV; i++) s += d->efc_J[r][i]*d->qacc_smooth[i];
        d->efc_b[r] = s - d->efc_aref[r];
    }
    /* warm start from the previous forces of the same row identity */
    for (int r = 0; r < n; r++) {
        int id = d->efc_id[r];
        d->efc_force[r] = d->efc_type[r] == 0 ? d->warm_fric[id]
                        : d->efc_type[r] == 1 ? d->warm_limit[id/2] : (id >= 0 && id < 16 ? d->warm_contact[id] : 0.0);
        if (d->efc_type[r] != 0 && d->efc_force[r] < 0) d->efc_force[r] = 0;
    }
    int maxit = iters > 0 ? iters : 2000;
    for (int it = 0; it < maxit; it++) {
        double change = 0;
        for (int r = 0; r < n; r++) {
            double res = d->efc_b[r];
            for (int c = 0; c < n; c++) res += AR[r*n + c]*d->efc_force[c];
            double f = d->efc_force[r] - res / AR[r*n + r];
            if (d->efc_type[r] == 0) { double fl = d->efc_floss[r]; f = f < -fl ? -fl : (f > fl ? fl : f); }
            else if (f < 0) f = 0;
            double df = fabs(f - d->efc_force[r]);
            if (df > change) change = df;
            d->efc_force[r] = f;
        }
        d->solver_iter_used = it + 1; d->solver_last_change = change;
        if (iters <= 0 && change < 1e-15) break;
    }
    memset(d->warm_fric, 0, sizeof d->warm_fric);
    memset(d->warm_limit, 0, sizeof d->warm_limit);
    memset(d->warm_contact, 0, sizeof d->warm_contact);
    for (int r = 0; r < n; r++) {
        int id = d->efc_id[r];
        if (d->efc_type[r] == 0) d->warm_fric[id] = d->efc_force[r];
        else if (d->efc_type[r] == 1) d->warm_limit[id/2] = d->efc_force[r];
        else if (id >= 0 && id < 16) d->warm_contact[id] = d->efc_force[r];
        for (int i = 0; i < NV; i++) d->qfrc_constraint[i] += d->efc_J[r][i]*d->efc_force[r];
    }
    double dq[NV];
    chol_solve(d->L, NV, d->qfrc_constraint, dq);
    for (int i = 0; i < NV; i++) d->qacc[i] = d->qacc_smooth[i] + dq[i];
}

/* ------------------------------------------------------------------------------------------------
 * Newton on the primal (mj_solNewton's problem): minimise over x = qacc
 *     1/2 (x - a0)' M (x - a0) + sum_r s_r(J_r x - aref_r),
 * s_r = the row's penalty: friction-loss rows Huber (quadratic |jar| <= R floss, linear outside, force saturating at
 * +-floss), limit / contact rows 1/(2R) min(0, jar)^2.  Newton direction from the exact Hessian M + J' D J of the current
 * active set, exact line search on the convex piecewise-quadratic restriction, iterated to machine precision.  Same optimum
 * as the PGS on the dual above (the dual of this problem is that one); tests/test_oracle_physics.py checks it.
 * nvs = number of leading dofs solved for (6 when the cube is pinned, else 12).
 * ---------------------------------------------------------------------------------------------- */
static double row_penalty(const so100o_data* d, int r, double jar, double* force, double* hess) {
    const double R = d->efc_R[r], D = 1.0/R;
    if (d->efc_type[r] == 0) {
        const double fl = d->efc_floss[r];
        if (jar <= -R*fl) { *force = fl;  *hess = 0; return -0.5*R*fl*fl - fl*jar; }
        if (jar >=  R*fl) { *force = -fl; *hess = 0; return -0.5*R*fl*fl + fl*jar; }
        *force = -D*jar; *hess = D; return 0.5*D*jar*jar;
    }
    if (jar < 0) { *force = -D*jar; *hess = D; return 0.5*D*jar*jar; }
    *force = 0; *hess = 0; return 0;
}
static double primal_cost(const so100o_data* d, int nvs, const double* x, double* jar, double* grad, double* H) {
    const int n = d->nefc;
    double dx[NV], Mdx[NV], cost = 0;
    for (int i = 0; i < nvs; i++) dx[i] = x[i] - d->qacc_smooth[i];
    for (int i = 0; i < nvs; i++) { double t = 0; for (int j = 0; j < nvs; j++) t += d->M[i*NV + j]*dx[j]; Mdx[i] = t; cost += 0.5*dx[i]*t; }
    if (grad) memcpy(grad, Mdx, sizeof(double)*nvs);
    if (H) for (int i = 0; i < nvs; i++) for (int j = 0; j < nvs; j++) H[i*nvs + j] = d->M[i*NV + j];
    for (int r = 0; r < n; r++) {
        double jr = -d->efc_aref[r], f, h;
        for (int i = 0; i < nvs; i++) jr += d->efc_J[r][i]*x[i];
        jar[r] = jr;
        cost += row_penalty(d, r, jr, &f, &h);
        if (grad) for (int i = 0; i < nvs; i++) grad[i] -= d->efc_J[r][i]*f;
        if (H && h != 0) for (int i = 0; i < nvs; i++) for (int j = 0; j < nvs; j++) H[i*nvs + j] += h*d->efc_J[r][i]*d->efc_J[r][j];
    }
    return cost;
}
static void solve_newton(const so100o_model* m, so100o_data* d, int nvs) {
    const int n = d->nefc;
    memset(d->qfrc_constraint, 0, sizeof d->qfrc_constraint);
    memcpy(d->qacc, d->qacc_smooth, sizeof d->qacc);
    d->solver_iter_used = 0; d->solver_last_change = 0;
    if (n == 0) return;
    static _Thread_local double jar[SO100O_MAXEFC], jp[SO100O_MAXEFC];
    double x[NV], g[NV], H[NV*NV], Lh[NV*NV], p[NV];
    /* warm start: the previous acceleration if it is cheaper than the unconstrained one (mj_fwdConstraint does the same) */
    memcpy(x, d->qacc_smooth, sizeof x);
    {
        const double c0 = primal_cost(d, nvs, d->qacc_smooth, jar, NULL, NULL), c1 = primal_cost(d, nvs, d->qacc_warmstart, jar, NULL, NULL);
        if (c1 < c0) memcpy(x, d->qacc_warmstart, sizeof(double)*nvs);
    }
    for (int it = 0; it < 200; it++) {
        const double cost = primal_cost(d, nvs, x, jar, g, H);
        double gn = 0; for (int i = 0; i < nvs; i++) gn = fmax(gn, fabs(g[i]));
        d->solver_iter_used = it; d->solver_last_change = gn;
        if (gn < 1e-13) break;
        cholesky(H, Lh, nvs);
        for (int i = 0; i < nvs; i++) g[i] = -g[i];
        chol_solve(Lh, nvs, g, p);
        for (int i = 0; i < nvs; i++) g[i] = -g[i];
        /* exact line search: phi'(alpha) = p'M(x + alpha p - a0) - sum_r f_r(jar_r + alpha jp_r) jp_r, increasing in alpha */
        double q1 = 0, q2 = 0;
        for (int i = 0; i < nvs; i++) {
            double t1 = 0, t2 = 0;
            for (int j = 0; j < nvs; j++) { t1 += d->M[i*NV + j]*(x[j] - d->qacc_smooth[j]); t2 += d->M[i*NV + j]*p[j]; }
            q1 += p[i]*t1; q2 += p[i]*t2;
        }
        for (int r = 0; r < n; r++) { double t = 0; for (int i = 0; i < nvs; i++) t += d->efc_J[r][i]*p[i]; jp[r] = t; }
        double lo = 0, hi = -1, alpha = 1;
        for (int ls = 0; ls < 200; ls++) {
            double d1 = q1 + alpha*q2, d2 = q2;
            for (int r = 0; r < n; r++) {
                double f, h; row_penalty(d, r, jar[r] + alpha*jp[r], &f, &h);
                d1 -= f*jp[r]; d2 += h*jp[r]*jp[r];
            }
            if (d1 < 0) lo = alpha; else hi = alpha;
            double an = d2 > 0 ? alpha - d1/d2 : alpha;
            if (!(an > lo && (hi < 0 || an < hi))) an = hi < 0 ? 2*alpha : 0.5*(lo + hi);
            if (fabs(an - alpha) <= 1e-15*fabs(alpha) || d1 == 0) { alpha = an; break; }
            alpha = an;
        }
        double step = 0, xm = 0;
        for (int i = 0; i < nvs; i++) { x[i] += alpha*p[i]; step = fmax(step, fabs(alpha*p[i])); xm = fmax(xm, fabs(x[i])); }
        (void)cost;
        if (step < 1e-15*(1.0 + xm)) break;                  /* the iterate no longer moves: machine precision */
    }
    primal_cost(d, nvs, x, jar, g, NULL);
    for (int r = 0; r < n; r++) {
        double f, h; row_penalty(d, r, jar[r], &f, &h);
        d->efc_force[r] = f;
        for (int i = 0; i < NV; i++) d->qfrc_constraint[i] += d->efc_J[r][i]*f;
    }
    memcpy(d->qacc, x, sizeof(double)*nvs);
    /* keep the PGS force memory coherent for callers that switch solvers */
    memset(d->warm_fric, 0, sizeof d->warm_fric); memset(d->warm_limit, 0, sizeof d->warm_limit); memset(d->warm_contact, 0, sizeof d->warm_contact);
    for (int r = 0; r < n; r++) {
        const int id = d->efc_id[r];
        if (d->efc_type[r] == 0) d->warm_fric[id] = d->efc_force[r];
        else if (d->efc_type[r] == 1) d->warm_limit[id/2] = d->efc_force[r];
        else if (id >= 0 && id < 16) d->warm_contact[id] = d->efc_force[r];
    }
}

/* ================================================================================================
 * mj_forward / mj_step
 * ============================================================================================== */
void so100o_forward(const so100o_model* m, so100o_data* d, unsigned flags, int iters) {
    so100o_kinematics(m, d);
    so100o__crb(m, d);
    cholesky(d->M, d->L, NV);
    make_constraints(m, d, flags);
    com_vel(m, d);
    rne(m, d, NULL, d->qfrc_bias);
    /* mj_fwdActuation: position servo, ctrl clamped to ctrlrange, force to forcerange */
    memset(d->qfrc_actuator, 0, sizeof d->qfrc_actuator);
    for (int i = 0; i < 6; i++) {
        double u = d->ctrl[i];
        u = u < SO100_ACT_CTRL_LO ? SO100_ACT_CTRL_LO : (u > SO100_ACT_CTRL_HI ? SO100_ACT_CTRL_HI : u);
        double f = m->kp*u - m->kp*d->qpos[i] - m->kv[i]*d->qvel[i];
        f = f < SO100_ACT_FORCE_LO ? SO100_ACT_FORCE_LO : (f > SO100_ACT_FORCE_HI ? SO100_ACT_FORCE_HI : f);
        d->qfrc_actuator[i] = f;
    }
    for (int i = 0; i < NV; i++)
        d->qfrc_smooth[i] = -d->qfrc_bias[i] + d->qfrc_applied[i] + d->qfrc_actuator[i];
    if (flags & SO100O_F_CUBE_PINNED) for (int i = 6; i < NV; i++) d->qfrc_smooth[i] = 0;
    chol_solve(d->L, NV, d->qfrc_smooth, d->qacc_smooth);
    if (iters < 0) solve_newton(m, d, (flags & SO100O_F_CUBE_PINNED) ? 6 : NV);
    else solve_pgs(m, d, iters);
    if (flags & SO100O_F_CUBE_PINNED) for (int i = 6; i < NV; i++) d->qacc[i] = 0;
}

static void euler(const so100o_model* m, so100o_data* d, unsigned flags) {
    double h = m->timestep;
    int nvi = (flags & SO100O_F_CUBE_PINNED) ? 6 : NV;
    for (int i = 0; i < nvi; i++) d->qvel[i] += h*d->qacc[i];
    for (int i = 0; i < 6; i++) d->qpos[i] += h*d->qvel[i];
    if (!(flags & SO100O_F_CUBE_PINNED)) {
        for (int i = 0; i < 3; i++) d->qpos[6 + i] += h*d->qvel[6 + i];
        /* mju_quatIntegrate */
        double w[3] = { d->qvel[9], d->qvel[10], d->qvel[11] };
        double nrm = sqrt(dot3(w, w)), qrot[4];
        if (nrm < SO100_MJMINVAL) { w[0] = 1; w[1] = w[2] = 0; }
        else { w[0] /= nrm; w[1] /= nrm; w[2] /= nrm; }
        axis_angle2quat(qrot, w, h*nrm);
        quat_normalize(d->qpos + 9);
        quat_mul(d->qpos + 9, d->qpos + 9, qrot);
    }
    d->time += h;
    memcpy(d->qacc_warmstart, d->qacc, sizeof d->qacc);
}

void so100o_step(const so100o_model* m, so100o_data* d, unsigned flags, int iters, int nstep) {
    for (int s = 0; s < nstep; s++) {
        so100o_forward(m, d, flags, iters);
        euler(m, d, flags);
    }
}

/* ================================================================================================
 * Philox4x32-10 (Salmon et al., SC'11) -- the device uses the same generator
 * ============================================================================================== */
void so100o_philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                       uint32_t k0, uint32_t k1, uint32_t out[4]) {
    for (int r = 0; r < 10; r++) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
void so100o_uniform4(uint64_t seed, uint32_t env, uint32_t counter, uint32_t stream, float out[4]) {
    uint32_t r[4];
    so100o_philox4x32(env, counter, stream, 0u, (uint32_t)seed, (uint32_t)(seed >> 32), r);
    for (int i = 0; i < 4; i++) out[i] = (float)(r[i] >> 8) * (1.0f / 16777216.0f);
}

/* ================================================================================================
 * task layer
 * ============================================================================================== */
static const double JOINT_STEP_SCALE_F32 = 0.07500000298023224;   /* float32(0.075), utils.py:9 */
static const double REST_POSITION[6] = { 0.0, -3.141, 3.117, 1.0, 0.0, 0.0 };         /* utils.py:11 */
static const double START_POSITION[6] = { 0.0, -2.04, 1.19, 1.5, -1.58, 0.5 };        /* env03_v1.py:10 */
#include "../so100_mujoco_rl_amd/csrc/so100_start_positions.inc"                                  /* utils.py:13-50 */

static const double SPACE_START[2][3] = { {-0.05, -0.4, 0.01}, {0.05, -0.3, 0.01} };  /* env03_v1.py:13-16 */
static const double SPACE_END_03[2][3] = { {-0.35, -0.45, 0.01}, {0.35, -0.25, 0.01} }; /* env03_v1.py:17-20 */
static const double SPACE_END_05[2][3] = { {-0.45, -0.45, 0.01}, {0.45, -0.25, 0.5} };  /* env05_v1.py:17-20 */

static int reach_kind(int kind) { return kind <= 2 || kind == 6; }   /* obs 15, env_base_01.py / env_base_06.py */
int so100o_obs_dim(int kind) { return reach_kind(kind) ? 15 : 8; }

double so100o_joint_penalty(double a, double lo, double hi) {          /* env_base_01.py:153-163 */
    double penalty = 0.0;
    double lt = lo + 0.05*(hi - lo), ut = hi - 0.05*(hi - lo);
    if (a < lt) penalty -= (lt - a)*10.0;
    else if (a > ut) penalty -= (a - ut)*10.0;
    return penalty;
}
static double joint_reward(const so100o_model* m, const double* q) {   /* env_base_01.py:144-151 */
    double r = 0.0;
    for (int i = 0; i < 6; i++) r += so100o_joint_penalty(q[i], m->jnt_range[i][0], m->jnt_range[i][1]);
    return r;
}
void so100o_end_effector(const double xpos[3], const double xmat[9], double out[3]) {
    /* env_base_01.py:118-127; local point is a float32 array: float32(-0.1) */
    const double ly = -0.10000000149011612;
    for (int i = 0; i < 3; i++) out[i] = xpos[i] + xmat[3*i + 1]*ly;
}
double so100o_reward_base(const so100o_model* m, const double q[6], const double block[3],
                          const double ee[3], const double wrist[3], int has_prev) {
    /* env_base_01.py:180-239 */
    double reward = 0.0;
    double dx = block[0] - ee[0], dy = block[1] - ee[1], dz = block[2] - ee[2];
    double distance = sqrt(dx*dx + dy*dy + dz*dz);
    if (block[1] < -0.1) {
        double pitch = q[1];
        if (has_prev && pitch < -0.7*PI) reward += (pitch + 0.7*PI)*0.7;
    }
    if (has_prev && ee[2] < 0.02) reward += (ee[2] - 0.02)*20.0;
    if (has_prev && wrist[2] < 0.08) {
        double w = (wrist[2] - 0.08)*10.0;
        w = w < -0.8 ? -0.8 : (w > 0.8 ? 0.8 : w);
        reward += w;
    }
    double ddr = -distance + 0.02;
    if (ddr > 0.0) ddr = 0.0;
    reward += ddr*0.5;
    reward += joint_reward(m, q);
    return reward;
}
int so100o_project(const double cam_xpos[3], const double cam_xmat[9], const double p[3], int uv[2]) {
    /* env_base_02.py:88-122 */
    double rel[3] = { p[0] - cam_xpos[0], p[1] - cam_xpos[1], p[2] - cam_xpos[2] }, c[3];
    for (int i = 0; i < 3; i++) c[i] = cam_xmat[i]*rel[0] + cam_xmat[3 + i]*rel[1] + cam_xmat[6 + i]*rel[2];
    double fovy = SO100_CAM_FOVY_DEG * (PI/180.0);
    double fy = 0.5*1920.0/tan(fovy/2), fx = fy;
    double u = fx*c[0]/c[2] + 1080.0/2, v = fy*c[1]/c[2] + 1920.0/2;
    if (isnan(u) || isnan(v)) return 0;
    if (isinf(u) || isinf(v)) return 0;          /* Python's int(inf) raises; treated as "None" */
    if (fabs(u) > 2e9 || fabs(v) > 2e9) return 0;
    int iu = (int)u, iv = (int)v;                 /* truncation toward zero, like int() */
    if (iu < 0 || iu >= 1080 || iv < 0 || iv >= 1920) return 0;
    uv[0] = 1080 - iu; uv[1] = 1920 - iv;
    return 1;
}

static void draw(const so100o_env* e, const float* inject, int phase, float* u8) {
    if (inject) { memcpy(u8, inject + 8*phase, sizeof(float)*8); return; }
    so100o_uniform4(e->seed, e->env_id, e->rng_counter, (uint32_t)(2*phase), u8);
    so100o_uniform4(e->seed, e->env_id, e->rng_counter, (uint32_t)(2*phase + 1), u8 + 4);
}

static void set_initial_values_03(so100o_env* e) {            /* env03_v1.py:35-57, env04_v1.py:25-46 */
    for (int i = 0; i < 6; i++) e->cmd[i] = (double)(float)START_POSITION[i];
    e->have_center = (e->kind == 4); e->last_center[0] = e->last_center[1] = -1.0;
    e->lost_count = 0;
    for (int i = 0; i < 3; i++) {
        e->space_min[i] = SPACE_START[0][i]; e->space_max[i] = SPACE_START[1][i];
        e->block_target[i] = (SPACE_START[0][i] + SPACE_START[1][i]) / 2;
    }
    e->block_speed = 0.0; e->target_dt = 0.01; e->target_time = 0.0;
    e->block_position_updated = 0;
}

void so100o_env_init(const so100o_model* m, so100o_env* e, int kind, unsigned flags, int iters,
                     uint64_t seed, uint32_t env_id) {
    memset(e, 0, sizeof *e);
    e->kind = kind; e->flags = flags; e->iters = iters; e->frame_skip = 16;
    e->max_episode_steps = kind == 1 ? 4000 : 6000;           /* so100_mujoco_rl/__init__.py:5-38 */
    e->seed = seed; e->env_id = env_id;
    so100o_reset_data(m, &e->d);
    if (!reach_kind(kind)) {                                  /* env_base_02.py:32,51 */
        set_initial_values_03(e);
        for (int i = 0; i < 3; i++) e->d.qpos[6 + i] = e->block_target[i];
    }
}

static void obs_base(const so100o_env* e, float* obs) {       /* env_base_01.py:241-270 */
    const so100o_data* d = &e->d;
    double ee[3]; so100o_end_effector(d->xpos[6], d->xmat[6], ee);
    const double* b = d->xpos[CUBE];
    for (int i = 0; i < 6; i++) obs[i] = (float)d->qpos[i];
    for (int i = 0; i < 3; i++) { obs[6 + i] = (float)(b[i] - ee[i]); obs[9 + i] = (float)b[i]; obs[12 + i] = (float)ee[i]; }
}
/* env_base_02.py:129-176 get_projected_cube_bounding_box (+ _get_cube_corners): the 8 axis-aligned corners of the
 * 2 cm cube around p are projected one by one, Nones dropped, fewer than 2 survivors => None (returns 0);
 * box = (min_x, min_y, max_x, max_y).  Unused upstream; the product's analytic stand-in for render + YOLO in Env03/04. */
int so100o_project_bbox(const double cam_xpos[3], const double cam_xmat[9], const double p[3], int box[4]) {
    const double d = 0.02 / 2;
    int n = 0;
    for (int k = 0; k < 8; k++) {                             /* corner order of the reference: x outer, y, z inner */
        const double c[3] = { (k & 4) ? p[0] + d : p[0] - d, (k & 2) ? p[1] + d : p[1] - d, (k & 1) ? p[2] + d : p[2] - d };
        int uv[2];
        if (!so100o_project(cam_xpos, cam_xmat, c, uv)) continue;
        if (n == 0) { box[0] = box[2] = uv[0]; box[1] = box[3] = uv[1]; }
        else {
            if (uv[0] < box[0]) box[0] = uv[0];
            if (uv[0] > box[2]) box[2] = uv[0];
            if (uv[1] < box[1]) box[1] = uv[1];
            if (uv[1] > box[3]) box[3] = uv[1];
        }
        n++;
    }
    return n >= 2;
}

static void obs_cam(const so100o_env* e, const float* u8, int noise, float* obs) {  /* env05_v1.py:32-75 */
    const so100o_data* d = &e->d;
    double cx = -1.0, cy = -1.0; int uv[2];
    if (e->kind == 5) {
        if (so100o_project(d->cam_xpos, d->cam_xmat, d->qpos + 6, uv)) {
            cx = uv[0] / 1080.0; cy = uv[1] / 1920.0;
            if (noise) { cx += -0.05 + 0.1*(double)u8[4]; cy += -0.05 + 0.1*(double)u8[5]; }
        }
    } else {
        /* Env03 / Env04 (render + YOLO upstream, env_base_02.py:178-222): the detector is replaced by the bounding box of
         * the projected cube corners with YOLO's centre arithmetic, center = (x1 + x2) // 2 / width (env_base_02.py:206-209) */
        int box[4];
        if (so100o_project_bbox(d->cam_xpos, d->cam_xmat, d->qpos + 6, box)) {
            cx = ((box[0] + box[2]) / 2) / 1080.0; cy = ((box[1] + box[3]) / 2) / 1920.0;
        }
    }
    for (int i = 0; i < 6; i++) obs[i] = (float)e->cmd[i];
    obs[6] = (float)cx; obs[7] = (float)cy;
}

static void set_random_block_position(so100o_env* e, double dlo, const float* u) {
    /* env01_v1.py:45-52 (dlo=0.18), env02_v1.py:52-68 and env06_v1.py:52-69 (dlo=0.22); u[1] is the discarded draw */
    double dist = dlo + (0.42 - dlo)*(double)u[0];
    double theta = -0.5*PI + (-0.25*PI + (0.5*PI)*(double)u[2]);
    double p[3] = { dist*cos(theta), dist*sin(theta), 0.0 };
    memcpy(e->d.qpos + 6, p, sizeof p);
    if (e->kind == 2 || e->kind == 6) {
        if (!e->have_last_block_pos) { memcpy(e->last_block_pos, p, sizeof p); e->have_last_block_pos = 1; }
        else memcpy(e->last_block_pos, e->block_pos, sizeof p);
        memcpy(e->block_pos, p, sizeof p); e->have_block_pos = 1;
    }
}

void so100o_env_reset(const so100o_model* m, so100o_env* e, const float* inject, float* obs) {
    float u[8];
    draw(e, inject, 1, u);
    e->rng_counter++;
    so100o_reset_data(m, &e->d);                              /* MujocoEnv.reset -> mj_resetData */
    e->elapsed_steps = 0; e->episode_return = 0; e->episode_length = 0;
    switch (e->kind) {
    case 1: {                                                 /* env01_v1.py:39-63 */
        set_random_block_position(e, 0.18, u);
        int idx = (int)((double)u[3] * 36.0); if (idx > 35) idx = 35;
        for (int i = 0; i < 5; i++) e->d.qpos[i] = SO100_VALID_START_POSITIONS[idx][i];  /* Jaw skipped */
        obs_base(e, obs);
    } break;
    case 2: case 6:                                           /* env02_v1.py:70-81, env06_v1.py:71-82 */
        set_random_block_position(e, 0.22, u);
        for (int i = 0; i < 6; i++) e->d.qpos[i] = REST_POSITION[i];
        obs_base(e, obs);
        break;
    default:                                                  /* env03_v1.py:203-215 */
        set_initial_values_03(e);
        for (int i = 0; i < 3; i++) e->d.qpos[6 + i] = e->block_target[i];
        for (int i = 0; i < 6; i++) e->d.qpos[i] = START_POSITION[i];
        obs_cam(e, u, e->kind == 5, obs);                     /* cam pose is zero => (-1,-1) */
        break;
    }
}

static double norm3d(const double* a, const double* b) {
    double x = a[0]-b[0], y = a[1]-b[1], z = a[2]-b[2]; return sqrt(x*x + y*y + z*z);
}

void so100o_env_step(const so100o_model* m, so100o_env* e, const float* a, const float* inject,
                     int autoreset, float* obs, double* reward_out, int* terminated, int* truncated,
                     float* terminal_obs) {
    so100o_data* d = &e->d;
    float u[8];
    draw(e, inject, 0, u);
    e->rng_counter++;
    double reward = 0.0; int term = 0;
    const int od = so100o_obs_dim(e->kind);
    /* product behaviour (so100_task.hpp::env_step_pre): a non-finite action runs as zero and ends the episode below */
    float a_ok[6]; int bad_action = 0;
    for (int i = 0; i < 6; i++) bad_action |= !isfinite(a[i]);
    for (int i = 0; i < 6; i++) a_ok[i] = bad_action ? 0.0f : a[i];
    a = a_ok;

    if (reach_kind(e->kind)) {
        /* env01_v1.py:15-37 / env02_v1.py:18-50 / env06_v1.py:18-50 */
        double ee[3]; so100o_end_effector(d->xpos[6], d->xmat[6], ee);
        reward = so100o_reward_base(m, d->qpos, d->xpos[CUBE], ee, d->xpos[5], e->has_prev);
        e->has_prev = 1;
        for (int i = 0; i < 6; i++)      /* np.float64 + (np.float32 * weak python float) */
            d->ctrl[i] = d->qpos[i] + (double)(float)((float)a[i] * (float)JOINT_STEP_SCALE_F32);
        if ((e->kind == 2 || e->kind == 6) && norm3d(d->xpos[CUBE], ee) < 0.03) {
            if (e->kind == 6) {                                   /* gripper term: env_base_06.py:149-162, 253-256 */
                double jn = (d->qpos[5] + 0.2) / 2.2;
                jn = jn < 0.0 ? 0.0 : jn > 1.0 ? 1.0 : jn;
                reward += 100.0 * (1.0 / (1.0 + exp(-10 * (jn - 0.3))));
            }
            reward += norm3d(e->block_pos, e->last_block_pos) * 20;
            if (e->kind == 2) set_random_block_position(e, 0.22, u);   /* commented out in env06_v1.py:36 */
        }
        so100o_step(m, d, e->flags, e->iters, e->frame_skip);
        obs_base(e, obs);
    } else {
        /* env03_v1.py:124-201 (kind 3, 5) / env04_v1.py:62-160 (kind 4) */
        double frac = d->time / 12.0; if (frac > 1.0) frac = 1.0;
        if (e->kind != 4) {
            const double (*end)[3] = e->kind == 5 ? SPACE_END_05 : SPACE_END_03;
            for (int i = 0; i < 3; i++) {                     /* _update_block_space :59-68 */
                e->space_min[i] = SPACE_START[0][i] + frac*(end[0][i] - SPACE_START[0][i]);
                e->space_max[i] = SPACE_START[1][i] + frac*(end[1][i] - SPACE_START[1][i]);
            }
            e->block_speed = frac <= 0.05 ? 0.0 : 0.0 + (frac - 0.05)*(2.0 - 0.0)/(1.0 - 0.05);  /* :70-75 */
            {                                                 /* _update_block_target :77-93 */
                double dist_t = norm3d(e->block_target, d->qpos + 6);
                if (!(d->time - e->target_time < e->target_dt && dist_t > 0.02)) {
                    for (int i = 0; i < 3; i++)
                        e->block_target[i] = e->space_min[i] + (e->space_max[i] - e->space_min[i])*(double)u[i];
                    e->target_dt = 1.2 + (5.1 - 1.2)*(double)u[3];
                    e->target_time = d->time;
                }
            }
            {                                                 /* _update_block_position :95-122 */
                double dir[3], dist = norm3d(e->block_target, d->qpos + 6);
                if (dist > 0) {
                    for (int i = 0; i < 3; i++) dir[i] = (e->block_target[i] - d->qpos[6 + i]) / dist;
                    double sd = e->block_speed * m->timestep; if (sd > dist) sd = dist;
                    for (int i = 0; i < 3; i++) d->qpos[6 + i] = d->qpos[6 + i] + dir[i]*sd;
                    for (int i = 0; i < 3; i++) d->qvel[6 + i] = 0.0;
                    for (int i = 0; i < 3; i++) d->qfrc_applied[6 + i] = -m->body_mass[CUBE]*m->gravity[i];
                }
            }
        }
        float newcmd[6];
        double old[6]; memcpy(old, e->cmd, sizeof old);
        for (int i = 0; i < 6; i++) {                         /* float32 accumulation under NumPy 2 */
            newcmd[i] = (float)e->cmd[i] + (float)a[i] * (float)JOINT_STEP_SCALE_F32;
            d->ctrl[i] = (double)newcmd[i];
        }
        so100o_step(m, d, e->flags, e->iters, e->frame_skip);
        obs_cam(e, u, e->kind == 5, obs);
        float cxf = obs[6], cyf = obs[7];
        if (cxf == -1.0f && cyf == -1.0f) {
            if (e->lost_count > 30) term = 1;
            e->lost_count++;
            if (e->kind == 4) { obs[6] = (float)e->last_center[0]; obs[7] = (float)e->last_center[1]; }
        } else {
            e->last_center[0] = cxf; e->last_center[1] = cyf; e->have_center = 1; e->lost_count = 0;
        }
        reward = 0.5;
        if (e->have_center) {
            float fx = 0.5f - (float)e->last_center[0], fy = 0.5f - (float)e->last_center[1];
            float dd = sqrtf(fx*fx + fy*fy);
            if (e->kind == 4) {                               /* env04_v1.py:108-131 */
                float bonus = (float)exp((double)(-10.0f * dd));   /* np.exp on float32 */
                float r32 = (float)reward + bonus;
                r32 = r32 + (-1.0f * dd);
                reward = (double)r32;
                if (dd < 0.1f && !e->block_position_updated) {
                    e->block_position_updated = 1;
                    for (int i = 0; i < 3; i++) {
                        e->block_target[i] = e->space_min[i] + (e->space_max[i] - e->space_min[i])*(double)u[i];
                        d->qpos[6 + i] = e->block_target[i];
                    }
                    reward = (double)((float)reward + 10.0f);
                }
            } else {
                reward = (double)((float)reward + (-1.0f * dd));   /* env03_v1.py:168-176 */
            }
        }
        reward += joint_reward(m, old);
        if (e->kind == 4) {                                   /* env04_v1.py:137-148 */
            double wr = so100o_joint_penalty(old[4], START_POSITION[4] - 0.2, START_POSITION[4] + 0.2);
            wr = wr < -0.2 ? -0.2 : (wr > 0.0 ? 0.0 : wr);
            reward += wr*0.5;
        } else {                                              /* env03_v1.py:182-189, env_base_01.py:165-178 */
            float pen = 0.0f, av[6];
            for (int i = 0; i < 6; i++) av[i] = (newcmd[i] - (float)old[i]) / (float)m->timestep;
            if (e->have_last_angvel)
                for (int i = 0; i < 6; i++) pen += fabsf(av[i] - (float)e->last_angvel[i]) * 0.0025f;
            for (int i = 0; i < 6; i++) e->last_angvel[i] = av[i];
            e->have_last_angvel = 1;
            reward += (double)((-pen) * (float)frac);
        }
        obs[6] = 5.0f*obs[6]; obs[7] = 5.0f*obs[7];
        for (int i = 0; i < 6; i++) e->cmd[i] = (double)newcmd[i];
    }

    /* Non-finite state guard (NOT reference behaviour; mirrors so100_task.hpp::env_step_finish): MuJoCo answers a
     * NaN / > 1e10 state with a warning + mj_resetData and the reference adds nothing, so a NaN action would poison
     * the rest of the episode.  The product ends that env's episode instead (terminated, reward 0, terminal obs 0). */
    {
        int bad = bad_action || !isfinite(reward);
        for (int i = 0; i < 13; i++) bad |= !isfinite(d->qpos[i]) || (i < 9 && fabs(d->qpos[i]) >= 1e10);
        for (int i = 0; i < 12; i++) bad |= !isfinite(d->qvel[i]) || fabs(d->qvel[i]) >= 1e10;
        if (bad) {
            term = 1; reward = 0.0; e->bad_state = 1;
            for (int i = 0; i < od; i++) obs[i] = 0.0f;
            if (!reach_kind(e->kind)) { for (int i = 0; i < 6; i++) e->last_angvel[i] = 0.0; e->have_last_angvel = 0; }
        }
    }

    /* TimeLimit (gymnasium) + DummyVecEnv auto-reset (stable_baselines3) */
    e->elapsed_steps++;
    int trunc = (e->max_episode_steps > 0 && e->elapsed_steps >= e->max_episode_steps);
    e->episode_return += reward; e->episode_length++;
    *reward_out = reward; *terminated = term; *truncated = trunc;
    if ((term || trunc) && autoreset) {
        if (terminal_obs) memcpy(terminal_obs, obs, sizeof(float)*od);
        so100o_env_reset(m, e, inject, obs);
    }
}

/* ---- batch helpers -------------------------------------------------------------------------- */
so100o_env* so100o_envs_alloc(int n) { return (so100o_env*)calloc((size_t)n, sizeof(so100o_env)); }
void so100o_envs_free(so100o_env* e) { free(e); }
so100o_env* so100o_envs_at(so100o_env* e, int i) { return e + i; }
void so100o_envs_step_range(const so100o_model* m, so100o_env* envs, int begin, int end,
                            const float* actions, int autoreset, float* obs, float* rew,
                            uint8_t* term, uint8_t* trunc, float* terminal_obs) {
    for (int i = begin; i < end; i++) {
        int od = so100o_obs_dim(envs[i].kind), t, tr; double r;
        so100o_env_step(m, envs + i, actions + 6*i, NULL, autoreset, obs + od*i, &r, &t, &tr,
                        terminal_obs ? terminal_obs + od*i : NULL);
        rew[i] = (float)r; term[i] = (uint8_t)t; trunc[i] = (uint8_t)tr;
    }
}

/* the same over `nthreads` host threads, one contiguous env slice each (envs are independent): bench.py's cpu_baseline on all the host's
 * cores -- a Python thread pool spends more time dispatching 256 slices than the slices take */
#include <pthread.h>
typedef struct { const so100o_model* m; so100o_env* envs; int begin, end; const float* actions; int autoreset; float* obs; float* rew;
                 uint8_t* term; uint8_t* trunc; float* tobs; } so100o_slice;
static void* slice_main(void* a) {
    so100o_slice* s = (so100o_slice*)a;
    so100o_envs_step_range(s->m, s->envs, s->begin, s->end, s->actions, s->autoreset, s->obs, s->rew, s->term, s->trunc, s->tobs);
    return NULL;
}
int so100o_envs_step_threads(const so100o_model* m, so100o_env* envs, int n, int nthreads, const float* actions, int autoreset, float* obs,
                             float* rew, uint8_t* term, uint8_t* trunc, float* terminal_obs) {
    if (nthreads < 1) nthreads = 1;
    if (nthreads > n) nthreads = n;
    pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t)*(size_t)nthreads);
    so100o_slice* sl = (so100o_slice*)malloc(sizeof(so100o_slice)*(size_t)nthreads);
    if (!th || !sl) { free(th); free(sl); return -1; }
    int started = 0;
    for (int k = 0; k < nthreads; k++) {
        sl[k] = (so100o_slice){ m, envs, (int)((long long)n*k/nthreads), (int)((long long)n*(k + 1)/nthreads), actions, autoreset, obs, rew, term, trunc, terminal_obs };
        if (pthread_create(&th[k], NULL, slice_main, &sl[k]) != 0) { slice_main(&sl[k]); th[k] = (pthread_t)0; } else started++;
    }
    for (int k = 0; k < nthreads; k++) if (th[k] != (pthread_t)0) pthread_join(th[k], NULL);
    free(th); free(sl);
    return started;
}

/* struct sizes, so the ctypes mirror in so100_oracle.py can be verified at load time */
int so100o_sizeof(int which) {
    return which == 0 ? (int)sizeof(so100o_model) : which == 1 ? (int)sizeof(so100o_data) : (int)sizeof(so100o_env);
}
