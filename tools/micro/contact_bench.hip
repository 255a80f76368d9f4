// Dynamic cost of the pad-contact path on the GPU: one wave (64 envs) steps NSUB substeps with substep_with_pads<float> from a
// pose whose pads rest on the floor; prints cycles per substep and the solver's dynamic counters.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=fast -fno-slp-vectorize -DSO100_CONTACT_STATS -I so100_mujoco_rl_amd/csrc -o /tmp/contact_bench tools/micro/contact_bench.hip && /tmp/contact_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include "so100_contact.hpp"
using namespace so100;
__global__ void __launch_bounds__(64) k_run(const float* q0, float* out, long long* cyc, unsigned flags, int nsub, int iters, int mode) {
    const int t = threadIdx.x;
    float q[6], v[6] = {0,0,0,0,0,0}, qc[6] = {0,0,0,0,0,0}, ctrl[6], ff[6] = {0,0,0,0,0,0}, fl[6] = {0,0,0,0,0,0}, aw[6] = {0,0,0,0,0,0}, dq[6] = {0,0,0,0,0,0};
    for (int i = 0; i < 6; i++) { q[i] = q0[i] + 1e-3f*t*(i == 0); ctrl[i] = q[i]; }
    if (mode == 1) ctrl[1] += 0.5f;                       // servo the shoulder down: the gripper lands on the floor and rests there
    Cube<float> cb{}; cb.pos[0] = 0.2f; cb.pos[1] = -0.2f; cb.pos[2] = 0.0099f; cb.quat[0] = 1.0f;
    const float ap[3] = {0, 0, 0};
    Arm<float> A; float res = 0; int st[3] = {0, 0, 0}; int nmax = 0;
    ContactsPriv<float> cs; int zones = -1;                // the solver's active-set memory, carried from substep to substep
    // settle first (untimed), then time
    const int settle = mode == 2 ? 4800 : 960;
    for (int s = 0; s < settle; s++) {
        if (mode == 2 && (s & 15) == 0) for (int i = 0; i < 6; i++) ctrl[i] = q[i];       // Env01's ctrl = measured angle + 0 action: the arm sags onto the floor
        substep_with_pads<float>(q, v, qc, ctrl, ff, fl, aw, cb, ap, flags, 2, iters, A, (s & 15) == 0, dq, &res, cs, zones, st);
    }
    res = 0;
    const long long c0 = __builtin_readcyclecounter();
    for (int s = 0; s < nsub; s++) {
        if (mode == 2 && (s & 15) == 0) for (int i = 0; i < 6; i++) ctrl[i] = q[i];
        substep_with_pads<float>(q, v, qc, ctrl, ff, fl, aw, cb, ap, flags, 2, iters, A, (s & 15) == 0, dq, &res, cs, zones, st); nmax = st[0] > nmax ? st[0] : nmax;
    }
    const long long c1 = __builtin_readcyclecounter();
    if (t == 0) cyc[0] = c1 - c0;
    out[t] = q[1]; out[64 + t] = res; out[128 + t] = (float)nmax;
}
int main() {
    const float poses[3][6] = { {0.0f, -1.6f, 1.9f, 1.5f, 0.0f, 0.3f}, {0.0f, -1.6f, 1.9f, 1.5f, 0.0f, 0.3f}, {0.0f, -1.7f, 1.2f, 0.3f, 0.0f, 0.3f} };
    float *dq, *dout; long long* dc;
    hipMalloc(&dq, 24); hipMalloc(&dout, 192*4); hipMalloc(&dc, 8);
    for (int pose = 0; pose < 3; pose++)
        for (unsigned flags : {7u, 23u}) {
            hipMemcpy(dq, poses[pose], 24, hipMemcpyHostToDevice);
            unsigned long long z[8] = {0}; hipMemcpyToSymbol(HIP_SYMBOL(so100::so100_cstats), z, sizeof z);
            const int nsub = 1600;
            hipLaunchKernelGGL(k_run, dim3(1), dim3(64), 0, 0, dq, dout, dc, flags, nsub, 6, pose);
            hipDeviceSynchronize();
            long long c; float out[192]; unsigned long long st[8];
            hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost); hipMemcpy(out, dout, sizeof out, hipMemcpyDeviceToHost);
            hipMemcpyFromSymbol(st, HIP_SYMBOL(so100::so100_cstats), sizeof st);
            float rmax = 0, nm = 0; for (int i = 0; i < 64; i++) { rmax = out[64 + i] > rmax ? out[64 + i] : rmax; nm = out[128 + i] > nm ? out[128 + i] : nm; }
            printf("mode %d (0 = holding in the air, 1 = pressed on the floor, 2 = sagged onto the floor under ctrl = q) flags %2u: %8.0f cycles/substep  | per lane-substep: newton calls %.3f iterations %.3f evals %.3f ls passes %.3f capped %.4f | max contacts %.0f  max residual %.2e  q1[0] %.5f\n",
                   pose, flags, (double)c/nsub, st[0]/(64.0*(nsub + (pose == 2 ? 4800 : 960))), st[1]/(64.0*(nsub + (pose == 2 ? 4800 : 960))), st[3]/(64.0*(nsub + (pose == 2 ? 4800 : 960))), st[2]/(64.0*(nsub + (pose == 2 ? 4800 : 960))), st[4]/(64.0*(nsub + (pose == 2 ? 4800 : 960))), nm, rmax, out[0]);
        }
    return 0;
}
