// debug: substep_with_pads<float> on the device AND on the host for states read from a text file
// line: q6 v6 ctrl6 cubepos3 cubequat4 cubevel6 flags  -> prints per state: stat, records (device | host), qvel after (device | host)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../../so100_mujoco_rl_amd/csrc/so100_physics.hpp"
#include "../../so100_mujoco_rl_amd/csrc/so100_cube.hpp"
#include "../../so100_mujoco_rl_amd/csrc/so100_contact.hpp"
using namespace so100;
constexpr int NIN = 32, NOUT = 12 + 4 + 2*CF;
SO100_HD void one(const float* p, float* o) {
    float q[6], v[6], ctrl[6], qc[6] = {0,0,0,0,0,0}, ff[6] = {0,0,0,0,0,0}, fl[6] = {0,0,0,0,0,0}, aw[6] = {0,0,0,0,0,0}, dq[6] = {0,0,0,0,0,0}, ap[3] = {0,0,0}, res = 0;
    Cube<float> cb;
    for (int i = 0; i < 6; i++) { q[i] = p[i]; v[i] = p[6+i]; ctrl[i] = p[12+i]; cb.vel[i] = p[25+i]; cb.warm[i] = 0; }
    for (int i = 0; i < 3; i++) cb.pos[i] = p[18+i];
    for (int i = 0; i < 4; i++) cb.quat[i] = p[21+i];
    const unsigned flags = (unsigned)p[31];
    Arm<float> A; ContactsPriv<float> cs; int zones = -1, st[4];
    substep_with_pads<float>(q, v, qc, ctrl, ff, fl, aw, cb, ap, flags, 4, 30, A, true, dq, &res, cs, zones, st);
    for (int i = 0; i < 6; i++) { o[i] = v[i]; o[6+i] = cb.vel[i]; }
    for (int i = 0; i < 4; i++) o[12+i] = (float)st[i];
    for (int i = 0; i < 2*CF; i++) o[16+i] = cs.a[i];
}
__global__ void k(const float* in, float* out, int n) { const int i = threadIdx.x; if (i < n) one(in + NIN*i, out + NOUT*i); }
int main(int argc, char** argv) {
    FILE* f = fopen(argv[1], "r"); std::vector<float> in; double v; while (fscanf(f, "%lf", &v) == 1) in.push_back((float)v);
    const int n = (int)in.size()/NIN; float *di, *dout; std::vector<float> out(NOUT*n), href(NOUT*n);
    hipMalloc(&di, in.size()*4); hipMalloc(&dout, out.size()*4); hipMemcpy(di, in.data(), in.size()*4, hipMemcpyHostToDevice);
    k<<<1, 64>>>(di, dout, n); hipMemcpy(out.data(), dout, out.size()*4, hipMemcpyDeviceToHost);
    for (int i = 0; i < n; i++) one(in.data() + NIN*i, href.data() + NOUT*i);
    for (int i = 0; i < n; i++) {
        printf("state %d  n %g coupled %g\n", i, out[NOUT*i+12], out[NOUT*i+13]);
        printf("  dev  v:"); for (int j = 0; j < 12; j++) printf(" %.6g", out[NOUT*i+j]); printf("\n  host v:"); for (int j = 0; j < 12; j++) printf(" %.6g", href[NOUT*i+j]);
        printf("\n  dev  rec0:"); for (int j = 0; j < CF; j++) printf(" %.8g", out[NOUT*i+16+j]); printf("\n  host rec0:"); for (int j = 0; j < CF; j++) printf(" %.8g", href[NOUT*i+16+j]); printf("\n");
    }
    return 0;
}
