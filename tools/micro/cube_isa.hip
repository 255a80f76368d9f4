// Compile-only helper: instruction counts of the cube half-substeps in isolation.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=fast -fno-slp-vectorize -S --cuda-device-only -I so100_mujoco_rl_amd/csrc -o /tmp/cube.s tools/micro/cube_isa.hip
#include <hip/hip_runtime.h>
#include "so100_cube.hpp"
using namespace so100;
__global__ void __launch_bounds__(64) k_prepare(const float* in, float* out, unsigned flags) {
    Cube<float> c; const int t = threadIdx.x;
    for (int i = 0; i < 3; i++) c.pos[i] = in[t + 64*i];
    for (int i = 0; i < 4; i++) c.quat[i] = in[t + 64*(3 + i)];
    for (int i = 0; i < 6; i++) { c.vel[i] = in[t + 64*(7 + i)]; c.warm[i] = in[t + 64*(13 + i)]; }
    float applied[3] = { 0, 0, in[t + 64*19] };
    CubePrep<float> P; cube_prepare<float>(c, applied, flags, P);
    float s = 0;
    for (int a = 0; a < 4; a++) { s += P.r.act[a] ? 1.0f : 0.0f; s += P.r.R[a]; for (int b = 0; b < 3; b++) s += P.r.rl[a][b] + P.r.dl[a][b]; for (int b = 0; b < 4; b++) s += P.r.b[a][b] + P.r.arinv[a][b]; }
    out[t] = s + P.a0[2];
}
__global__ void __launch_bounds__(64) k_step(const float* in, float* out, unsigned flags, int iters) {
    Cube<float> c; const int t = threadIdx.x;
    for (int i = 0; i < 3; i++) c.pos[i] = in[t + 64*i];
    for (int i = 0; i < 4; i++) c.quat[i] = in[t + 64*(3 + i)];
    for (int i = 0; i < 6; i++) { c.vel[i] = in[t + 64*(7 + i)]; c.warm[i] = in[t + 64*(13 + i)]; }
    float applied[3] = { 0, 0, in[t + 64*19] };
    cube_substep<float>(c, applied, flags, iters);
    for (int i = 0; i < 3; i++) out[t + 64*i] = c.pos[i];
    for (int i = 0; i < 4; i++) out[t + 64*(3 + i)] = c.quat[i];
    for (int i = 0; i < 6; i++) { out[t + 64*(7 + i)] = c.vel[i]; out[t + 64*(13 + i)] = c.warm[i]; }
}
