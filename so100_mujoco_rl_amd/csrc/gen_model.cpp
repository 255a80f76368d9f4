// gen_model.cpp -- offline "model compiler" for the so100 scene (host tool, fp64).
//
// Reads the RAW numbers in so100_model_def.h (transcribed from the reference MJCF) and derives what
// MuJoCo's XML compiler + mj_setConst derive, in the LINK-FRAME form the HIP
// kernels use:
//   * per link: fixed child->parent rotation C_k (from quat / intrinsic-xyz euler), offset p_k,
//     joint axis index, mass, COM c_k, first moment h_k = m c_k, inertia about the COM and about the
//     link origin expressed in link axes (from iquat + diaginertia),
//   * camera pose in the Fixed_Jaw frame,
//   * dof_M0 = diag M(qpos0) incl. armature, dof_invweight0 = diag M(qpos0)^-1,
//     actuator kv = dampratio * 2 sqrt(kp * dof_M0), default-solref K and B, friction-row R,
//   * cube mass / inertia from the box geom.
// Output: so100_model_gen.h (constexpr tables).  Build + run:  make -C so100_mujoco_rl_amd/csrc gen
// Escape hatch for SURVEY.md A.1's inertia assumption (the arm keeps its explicit <inertial> elements under the scene's inertiafromgeom="true";
// if MuJoCo's compiler instead derives them from the mesh geoms, the numbers are only known to who can run MuJoCo):
//   make gen INERTIALS=file      file = 6 lines "mass  ipos(3)  iquat(4, wxyz)  diaginertia(3)" = MjModel.body_mass / body_ipos / body_iquat /
//   body_inertia of the bodies Rotation_Pitch ... Moving_Jaw; '#' starts a comment.  The oracle takes the same table (so100o_model_init_with_inertials).
// The result is cross-checked against the oracle's independent derivation in tests/test_model_def.py.
#include <cmath>
#include <cstdio>
#include <cstring>
#include "so100_model_def.h"

static void qnorm(double* q) { double n = std::sqrt(q[0]*q[0]+q[1]*q[1]+q[2]*q[2]+q[3]*q[3]); for (int i = 0; i < 4; i++) q[i] /= n; }
static void qmul(double* r, const double* a, const double* b) {
    double t[4] = { a[0]*b[0]-a[1]*b[1]-a[2]*b[2]-a[3]*b[3], a[0]*b[1]+a[1]*b[0]+a[2]*b[3]-a[3]*b[2],
                    a[0]*b[2]-a[1]*b[3]+a[2]*b[0]+a[3]*b[1], a[0]*b[3]+a[1]*b[2]-a[2]*b[1]+a[3]*b[0] };
    std::memcpy(r, t, sizeof t);
}
static void q2m(double* m, const double* q) {
    double w=q[0],x=q[1],y=q[2],z=q[3];
    m[0]=w*w+x*x-y*y-z*z; m[1]=2*(x*y-w*z); m[2]=2*(x*z+w*y);
    m[3]=2*(x*y+w*z); m[4]=w*w-x*x+y*y-z*z; m[5]=2*(y*z-w*x);
    m[6]=2*(x*z-w*y); m[7]=2*(y*z+w*x); m[8]=w*w-x*x-y*y+z*z;
}
static void euler2q(double* q, const double* e) {      // intrinsic xyz: q = qx qy qz
    double t[4] = {1,0,0,0};
    for (int i = 0; i < 3; i++) { double r[4] = { std::cos(e[i]/2), 0, 0, 0 }; r[1+i] = std::sin(e[i]/2); qmul(t, t, r); }
    std::memcpy(q, t, sizeof t); qnorm(q);
}
static void mm(double* r, const double* a, const double* b) {
    double t[9]; for (int i=0;i<3;i++) for (int j=0;j<3;j++) { double s=0; for (int k=0;k<3;k++) s+=a[3*i+k]*b[3*k+j]; t[3*i+j]=s; }
    std::memcpy(r, t, sizeof t);
}
static void mv(double* r, const double* a, const double* v) { double t[3]; for (int i=0;i<3;i++) t[i]=a[3*i]*v[0]+a[3*i+1]*v[1]+a[3*i+2]*v[2]; std::memcpy(r,t,sizeof t); }

static void emit(FILE* f, const char* name, const double* v, int rows, int cols) {
    std::fprintf(f, "static constexpr double %s[%d][%d] = {\n", name, rows, cols);
    for (int r = 0; r < rows; r++) {
        std::fprintf(f, "    {");
        for (int c = 0; c < cols; c++) std::fprintf(f, " %.17g%s", v[r*cols+c], c+1<cols ? "," : "");
        std::fprintf(f, " },\n");
    }
    std::fprintf(f, "};\n");
}
static void emit1(FILE* f, const char* name, const double* v, int n) {
    std::fprintf(f, "static constexpr double %s[%d] = {", name, n);
    for (int c = 0; c < n; c++) std::fprintf(f, " %.17g%s", v[c], c+1<n ? "," : "");
    std::fprintf(f, " };\n");
}

int main(int argc, char** argv) {
    const int N = SO100_NLINK;
    // the arm links' inertials: the MJCF's <inertial> elements, or the override table (see the header comment)
    double LMASS[SO100_NLINK], LIPOS[SO100_NLINK][3], LIQUAT[SO100_NLINK][4], LDIAG[SO100_NLINK][3];
    for (int k = 0; k < N; k++) {
        LMASS[k] = SO100_LINK_MASS[k];
        std::memcpy(LIPOS[k], SO100_LINK_IPOS[k], sizeof LIPOS[k]); std::memcpy(LIQUAT[k], SO100_LINK_IQUAT[k], sizeof LIQUAT[k]); std::memcpy(LDIAG[k], SO100_LINK_DIAGINERTIA[k], sizeof LDIAG[k]);
    }
    const char* inert_file = argc > 2 && argv[2][0] ? argv[2] : nullptr;
    if (inert_file) {
        FILE* fi = std::fopen(inert_file, "r");
        if (!fi) { std::perror(inert_file); return 1; }
        char line[512]; int k = 0;
        while (k < N && std::fgets(line, sizeof line, fi)) {
            if (char* h = std::strchr(line, '#')) *h = 0;
            double v[11];
            if (std::sscanf(line, "%lf %lf %lf %lf %lf %lf %lf %lf %lf %lf %lf", v, v+1, v+2, v+3, v+4, v+5, v+6, v+7, v+8, v+9, v+10) != 11) continue;
            LMASS[k] = v[0]; for (int a = 0; a < 3; a++) { LIPOS[k][a] = v[1+a]; LDIAG[k][a] = v[8+a]; } for (int a = 0; a < 4; a++) LIQUAT[k][a] = v[4+a];
            k++;
        }
        std::fclose(fi);
        if (k != N) { std::fprintf(stderr, "%s: %d of %d inertial lines (11 numbers each)\n", inert_file, k, N); return 1; }
    }
    double C[N][9], P[N][3], COM[N][3], H[N][3], ICOM[N][6], IORG[N][6], MASS[N];
    int AX[N];
    for (int k = 0; k < N; k++) {
        double q[4];
        if (SO100_LINK_ORI_KIND[k] == 0) { std::memcpy(q, SO100_LINK_ORI[k], sizeof q); qnorm(q); }
        else euler2q(q, SO100_LINK_ORI[k]);
        q2m(C[k], q);
        std::memcpy(P[k], SO100_LINK_POS[k], sizeof P[k]);
        AX[k] = SO100_JNT_AXIS[k][0] == 1 ? 0 : (SO100_JNT_AXIS[k][1] == 1 ? 1 : 2);
        MASS[k] = LMASS[k];
        std::memcpy(COM[k], LIPOS[k], sizeof COM[k]);
        double iq[4]; std::memcpy(iq, LIQUAT[k], sizeof iq); qnorm(iq);
        double R[9], I[9];
        q2m(R, iq);
        for (int i=0;i<3;i++) for (int j=0;j<3;j++) { double s=0; for (int a=0;a<3;a++) s+=R[3*i+a]*LDIAG[k][a]*R[3*j+a]; I[3*i+j]=s; }
        double c[3] = {COM[k][0],COM[k][1],COM[k][2]}, cc = c[0]*c[0]+c[1]*c[1]+c[2]*c[2], IO[9];
        for (int i=0;i<3;i++) for (int j=0;j<3;j++) IO[3*i+j] = I[3*i+j] + MASS[k]*((i==j?cc:0.0) - c[i]*c[j]);
        const int ix[6][2] = {{0,0},{1,1},{2,2},{0,1},{0,2},{1,2}};
        for (int e = 0; e < 6; e++) { ICOM[k][e] = I[3*ix[e][0]+ix[e][1]]; IORG[k][e] = IO[3*ix[e][0]+ix[e][1]]; }
        for (int i=0;i<3;i++) H[k][i] = MASS[k]*c[i];
    }
    // camera in the Fixed_Jaw frame
    double camq[4], CAMR[9]; euler2q(camq, SO100_CAM_EULER); q2m(CAMR, camq);

    // M(qpos0): world-frame composite inertia of the serial chain at q = 0
    double Rw[N][9], xw[N][3], axw[N][3];
    { double Rp[9] = {1,0,0,0,1,0,0,0,1}, xp[3] = {0,0,0};
      for (int k = 0; k < N; k++) {
          double v[3]; mv(v, Rp, P[k]); for (int i=0;i<3;i++) xw[k][i] = xp[i]+v[i];
          mm(Rw[k], Rp, C[k]);
          for (int i=0;i<3;i++) axw[k][i] = Rw[k][3*i+AX[k]];
          std::memcpy(Rp, Rw[k], sizeof Rp); std::memcpy(xp, xw[k], sizeof xp);
      } }
    double M[N][N] = {{0}};
    for (int i = 0; i < N; i++) for (int j = 0; j <= i; j++) {
        // M_ij = sum over bodies b >= i of [ m (a_i x r_ib).(a_j x r_jb) + a_i' I_b a_j ]
        double s = 0;
        for (int b = i; b < N; b++) {
            double cw[3], t[3]; mv(t, Rw[b], COM[b]); for (int a=0;a<3;a++) cw[a] = xw[b][a]+t[a];
            double ri[3], rj[3]; for (int a=0;a<3;a++) { ri[a]=cw[a]-xw[i][a]; rj[a]=cw[a]-xw[j][a]; }
            double vi[3] = { axw[i][1]*ri[2]-axw[i][2]*ri[1], axw[i][2]*ri[0]-axw[i][0]*ri[2], axw[i][0]*ri[1]-axw[i][1]*ri[0] };
            double vj[3] = { axw[j][1]*rj[2]-axw[j][2]*rj[1], axw[j][2]*rj[0]-axw[j][0]*rj[2], axw[j][0]*rj[1]-axw[j][1]*rj[0] };
            s += MASS[b]*(vi[0]*vj[0]+vi[1]*vj[1]+vi[2]*vj[2]);
            double Ib[9] = { ICOM[b][0],ICOM[b][3],ICOM[b][4], ICOM[b][3],ICOM[b][1],ICOM[b][5], ICOM[b][4],ICOM[b][5],ICOM[b][2] };
            double Iw[9], Rt[9]; for (int a=0;a<3;a++) for (int c=0;c<3;c++) Rt[3*a+c]=Rw[b][3*c+a];
            mm(Iw, Rw[b], Ib); mm(Iw, Iw, Rt);
            double u[3]; mv(u, Iw, axw[j]);
            s += axw[i][0]*u[0]+axw[i][1]*u[1]+axw[i][2]*u[2];
        }
        M[i][j] = M[j][i] = s;
    }
    for (int i = 0; i < N; i++) M[i][i] += SO100_JNT_ARMATURE;
    double M0[N], INVW0[N], KV[N];
    {   // inverse by Gauss-Jordan (6x6 SPD)
        double A[N][2*N];
        for (int i=0;i<N;i++) for (int j=0;j<N;j++) { A[i][j]=M[i][j]; A[i][N+j]=(i==j); }
        for (int c=0;c<N;c++) { double p=A[c][c]; for (int j=0;j<2*N;j++) A[c][j]/=p;
            for (int r=0;r<N;r++) if (r!=c) { double f=A[r][c]; for (int j=0;j<2*N;j++) A[r][j]-=f*A[c][j]; } }
        for (int i=0;i<N;i++) { M0[i]=M[i][i]; INVW0[i]=A[i][N+i]; KV[i]=SO100_ACT_DAMPRATIO*2.0*std::sqrt(SO100_ACT_KP*M0[i]); }
    }
    // body_invweight0 of the two jaw links (mj_setConst): mean diagonal of J M^-1 J^T, J = translational Jacobian of the
    // link's COM at qpos0 (only the translational weight enters the diagApprox of a condim-3 pyramidal contact)
    double INVW_TRAN[N];
    {
        double Ainv[N][N];
        {   double A[N][2*N];
            for (int i=0;i<N;i++) for (int j=0;j<N;j++) { A[i][j]=M[i][j]; A[i][N+j]=(i==j); }
            for (int c=0;c<N;c++) { double p=A[c][c]; for (int j=0;j<2*N;j++) A[c][j]/=p;
                for (int r=0;r<N;r++) if (r!=c) { double f=A[r][c]; for (int j=0;j<2*N;j++) A[r][j]-=f*A[c][j]; } }
            for (int i=0;i<N;i++) for (int j=0;j<N;j++) Ainv[i][j]=A[i][N+j]; }
        for (int b = 0; b < N; b++) {
            double cw[3], t[3]; mv(t, Rw[b], COM[b]); for (int a=0;a<3;a++) cw[a] = xw[b][a]+t[a];
            double J[3][N] = {{0}};
            for (int j = 0; j <= b; j++) {
                double r[3] = { cw[0]-xw[j][0], cw[1]-xw[j][1], cw[2]-xw[j][2] };
                J[0][j] = axw[j][1]*r[2]-axw[j][2]*r[1]; J[1][j] = axw[j][2]*r[0]-axw[j][0]*r[2]; J[2][j] = axw[j][0]*r[1]-axw[j][1]*r[0];
            }
            double tr = 0;
            for (int a=0;a<3;a++) for (int i=0;i<N;i++) for (int j=0;j<N;j++) tr += J[a][i]*Ainv[i][j]*J[a][j];
            INVW_TRAN[b] = tr/3.0;
        }
    }
    // contact parameters of every pair that involves a finger pad (pad/floor, pad/cube): mj_contactParam mixes the two geoms'
    // solref / solimp 1:1 (equal priority, solmix 1), friction = max; mj_assignImp then clamps d0, dmax into [1e-4, 0.9999]
    auto clampimp = [](double v) { return v < SO100_MJMINIMP ? SO100_MJMINIMP : (v > SO100_MJMAXIMP ? SO100_MJMAXIMP : v); };
    double ptc = 0.5*(SO100_PAD_SOLREF_TIMECONST + SO100_SOLREF_TIMECONST), pdr = 0.5*(SO100_PAD_SOLREF_DAMPRATIO + SO100_SOLREF_DAMPRATIO);
    double pd0 = clampimp(0.5*(SO100_PAD_SOLIMP_D0 + SO100_SOLIMP_D0)), pdm = clampimp(0.5*(SO100_PAD_SOLIMP_DMAX + SO100_SOLIMP_DMAX));
    double pw = 0.5*(SO100_PAD_SOLIMP_WIDTH + SO100_SOLIMP_WIDTH);
    if (ptc < 2*SO100_TIMESTEP) ptc = 2*SO100_TIMESTEP;
    double PK = 1.0/(pdm*pdm*ptc*ptc*pdr*pdr), PB = 2.0/(pdm*ptc);
    double pmu = SO100_PAD_FRICTION > SO100_GEOM_FRICTION ? SO100_PAD_FRICTION : SO100_GEOM_FRICTION;
    double tc = SO100_SOLREF_TIMECONST; if (tc < 2*SO100_TIMESTEP) tc = 2*SO100_TIMESTEP;
    double K = 1.0/(SO100_SOLIMP_DMAX*SO100_SOLIMP_DMAX*tc*tc*SO100_SOLREF_DAMPRATIO*SO100_SOLREF_DAMPRATIO);
    double B = 2.0/(SO100_SOLIMP_DMAX*tc);
    double a = 2*SO100_CUBE_HALF, cm = SO100_GEOM_DENSITY*a*a*a, ci = cm*(a*a+a*a)/12.0;

    const char* path = argc > 1 ? argv[1] : "so100_model_gen.h";
    FILE* f = std::fopen(path, "w");
    if (!f) { std::perror(path); return 1; }
    std::fprintf(f, "// GENERATED by so100_mujoco_rl_amd/csrc/gen_model.cpp from so100_model_def.h -- do not edit.\n"
                    "// Link-frame model constants of the so100 arm scene (see gen_model.cpp for definitions).\n"
                    "#pragma once\nnamespace so100g {\n");
    std::fprintf(f, "static constexpr bool INERTIALS_OVERRIDDEN = %s;      // true: built with `make gen INERTIALS=...` (MuJoCo's compiled body inertials instead of the MJCF's <inertial> elements)\n", inert_file ? "true" : "false");
    emit(f, "LINK_C", &C[0][0], N, 9);
    emit(f, "LINK_P", &P[0][0], N, 3);
    std::fprintf(f, "static constexpr int LINK_AXIS[%d] = { %d, %d, %d, %d, %d, %d };\n", N, AX[0],AX[1],AX[2],AX[3],AX[4],AX[5]);
    emit1(f, "LINK_MASS", MASS, N);
    emit(f, "LINK_COM", &COM[0][0], N, 3);
    emit(f, "LINK_H", &H[0][0], N, 3);
    std::fprintf(f, "// symmetric 3x3 as (xx, yy, zz, xy, xz, yz), link axes\n");
    emit(f, "LINK_ICOM", &ICOM[0][0], N, 6);
    emit(f, "LINK_IORG", &IORG[0][0], N, 6);
    emit(f, "JNT_RANGE", &SO100_JNT_RANGE[0][0], N, 2);
    emit1(f, "CAM_P", SO100_CAM_POS, 3);
    emit1(f, "CAM_R", CAMR, 9);
    emit1(f, "DOF_M0", M0, N);
    emit1(f, "DOF_INVWEIGHT0", INVW0, N);
    emit1(f, "ACT_KV", KV, N);
    std::fprintf(f, "static constexpr double ACT_KP = %.17g, ACT_FORCE = %.17g, ACT_CTRL = %.17g;\n", (double)SO100_ACT_KP, (double)SO100_ACT_FORCE_HI, (double)SO100_ACT_CTRL_HI);
    std::fprintf(f, "static constexpr double ARMATURE = %.17g, FRICTIONLOSS = %.17g;\n", (double)SO100_JNT_ARMATURE, (double)SO100_JNT_FRICTIONLOSS);
    std::fprintf(f, "static constexpr double TIMESTEP = %.17g, GRAVITY = %.17g;\n", (double)SO100_TIMESTEP, -(double)SO100_GRAVITY_Z);
    std::fprintf(f, "static constexpr double SOLREF_K = %.17g, SOLREF_B = %.17g;\n", K, B);
    std::fprintf(f, "static constexpr double SOLIMP_D0 = %.17g, SOLIMP_DMAX = %.17g, SOLIMP_WIDTH = %.17g;\n", (double)SO100_SOLIMP_D0, (double)SO100_SOLIMP_DMAX, (double)SO100_SOLIMP_WIDTH);
    std::fprintf(f, "static constexpr double CUBE_MASS = %.17g, CUBE_INERTIA = %.17g, CUBE_HALF = %.17g, GEOM_FRICTION = %.17g;\n", cm, ci, (double)SO100_CUBE_HALF, (double)SO100_GEOM_FRICTION);
    std::fprintf(f, "static constexpr double CAM_FOVY_DEG = %.17g;\n", (double)SO100_CAM_FOVY_DEG);
    std::fprintf(f, "static constexpr int CAM_LINK = %d;\n", SO100_CAM_LINK);
    std::fprintf(f, "// finger pads (box geoms on links 4 and 5) and the contact parameters of pairs that involve one\n");
    std::fprintf(f, "static constexpr int NPAD = %d;\n", SO100_NPAD);
    std::fprintf(f, "static constexpr int PAD_LINK[%d] = { %d, %d, %d, %d, %d, %d, %d, %d };\n", SO100_NPAD, SO100_PAD_LINK[0], SO100_PAD_LINK[1], SO100_PAD_LINK[2],
                 SO100_PAD_LINK[3], SO100_PAD_LINK[4], SO100_PAD_LINK[5], SO100_PAD_LINK[6], SO100_PAD_LINK[7]);
    emit(f, "PAD_POS", &SO100_PAD_POS[0][0], SO100_NPAD, 3);
    emit(f, "PAD_SIZE", &SO100_PAD_SIZE[0][0], SO100_NPAD, 3);
    emit1(f, "LINK_INVWEIGHT_TRAN", INVW_TRAN, N);
    std::fprintf(f, "static constexpr double PADC_K = %.17g, PADC_B = %.17g, PADC_D0 = %.17g, PADC_DMAX = %.17g, PADC_WIDTH = %.17g, PADC_MU = %.17g;\n", PK, PB, pd0, pdm, pw, pmu);
    {   // link proxies (stand-in capsules for the absent collision meshes): the rule of so100_model_def.h, as in oracle/so100_oracle.c
        double far_[SO100_NPROX][3] = {{0}}, rad[SO100_NPROX];
        for (int k = 0; k < SO100_NPROX; k++) {
            const int l = SO100_PROX_LINK[k];
            if (l <= 3) for (int a = 0; a < 3; a++) far_[k][a] = SO100_LINK_POS[l + 1][a];
            else {
                double xlo = 1e30, xhi = -1e30, ymax = 0;
                for (int g = 0; g < SO100_NPAD; g++) if (SO100_PAD_LINK[g] == l) {
                    xlo = std::fmin(xlo, SO100_PAD_POS[g][0] - SO100_PAD_SIZE[g][0]); xhi = std::fmax(xhi, SO100_PAD_POS[g][0] + SO100_PAD_SIZE[g][0]);
                    if (std::fabs(SO100_PAD_POS[g][1]) + SO100_PAD_SIZE[g][1] > std::fabs(ymax))
                        ymax = SO100_PAD_POS[g][1] < 0 ? SO100_PAD_POS[g][1] - SO100_PAD_SIZE[g][1] : SO100_PAD_POS[g][1] + SO100_PAD_SIZE[g][1];
                }
                far_[k][0] = 0.5*(xlo + xhi); far_[k][1] = ymax; far_[k][2] = 0.0;
            }
            const double* I = LDIAG[l]; const double mass = LMASS[l];
            const double h[3] = { 0.5*std::sqrt(6.0*(I[1] + I[2] - I[0])/mass), 0.5*std::sqrt(6.0*(I[0] + I[2] - I[1])/mass), 0.5*std::sqrt(6.0*(I[0] + I[1] - I[2])/mass) };
            const double hmax = std::fmax(h[0], std::fmax(h[1], h[2]));
            rad[k] = 0.5*(h[0] + h[1] + h[2] - hmax);
            if (l >= 4 && rad[k] > SO100_PROX_JAW_RADIUS_MAX) rad[k] = SO100_PROX_JAW_RADIUS_MAX;
        }
        std::fprintf(f, "// link proxies (F_LINKS_FLOOR): capsule k on link k + 1, from that link's joint origin to PROX_FAR[k] -- for links 1-3 the child's joint origin -- in link coordinates\n");
        std::fprintf(f, "static constexpr int NPROX = %d;\n", SO100_NPROX);
        std::fprintf(f, "static constexpr int PROX_LINK[%d] = { %d, %d, %d, %d, %d };\n", SO100_NPROX, SO100_PROX_LINK[0], SO100_PROX_LINK[1], SO100_PROX_LINK[2], SO100_PROX_LINK[3], SO100_PROX_LINK[4]);
        emit(f, "PROX_FAR", &far_[0][0], SO100_NPROX, 3);
        emit1(f, "PROX_RADIUS", rad, SO100_NPROX);
        // link / cube proxies (F_LINKS_CUBE): capsule k on link k (0 Rotation_Pitch, 1 Upper_Arm), same rule
        double cfar[SO100_NCPROX][3], crad[SO100_NCPROX];
        for (int k = 0; k < SO100_NCPROX; k++) {
            for (int a = 0; a < 3; a++) cfar[k][a] = SO100_LINK_POS[k + 1][a];
            const double* I = LDIAG[k]; const double mass = LMASS[k];
            const double h[3] = { 0.5*std::sqrt(6.0*(I[1] + I[2] - I[0])/mass), 0.5*std::sqrt(6.0*(I[0] + I[2] - I[1])/mass), 0.5*std::sqrt(6.0*(I[0] + I[1] - I[2])/mass) };
            crad[k] = 0.5*(h[0] + h[1] + h[2] - std::fmax(h[0], std::fmax(h[1], h[2])));
        }
        std::fprintf(f, "// link / cube proxies (F_LINKS_CUBE): capsule k on link k (Rotation_Pitch, Upper_Arm), from that link's joint origin to its child's (CPROX_FAR[k], link coordinates)\n");
        std::fprintf(f, "static constexpr int NCPROX = %d;\n", SO100_NCPROX);
        emit(f, "CPROX_FAR", &cfar[0][0], SO100_NCPROX, 3);
        emit1(f, "CPROX_RADIUS", crad, SO100_NCPROX);
    }
    std::fprintf(f, "}  // namespace so100g\n");
    std::fclose(f);
    return 0;
}
