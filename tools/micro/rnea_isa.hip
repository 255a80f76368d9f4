// Compile-only helper: static instruction counts of the RNEA bias force and of the CRBA mass matrix (tools/rnea_count.sh).
#include <hip/hip_runtime.h>
#include "so100_physics.hpp"
using namespace so100;
__global__ void __launch_bounds__(64) k_rnea(const float* in, float* out) {
    const int t = threadIdx.x;
    Arm<float> A; float v[6];
    for (int i = 0; i < 6; i++) { A.s[i] = in[t + 64*i]; A.c[i] = in[t + 64*(6 + i)]; v[i] = in[t + 64*(12 + i)]; }
    arm_bias<float>(v, A);
    for (int i = 0; i < 6; i++) out[t + 64*i] = A.bias[i];
}
__global__ void __launch_bounds__(64) k_crba(const float* in, float* out) {
    const int t = threadIdx.x;
    Arm<float> A;
    for (int i = 0; i < 6; i++) { A.s[i] = in[t + 64*i]; A.c[i] = in[t + 64*(6 + i)]; }
    arm_mass<float>(A);
    for (int i = 0; i < 21; i++) out[t + 64*i] = A.M[i];
}
