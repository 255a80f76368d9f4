// Micro-benchmark: VALU issue rate of W = 1 .. 4 wavefronts resident on ONE SIMD of gfx950.
//   hipcc -O3 --offload-arch=gfx950 -o simd_share tools/micro/simd_share.hip && ./simd_share
// One workgroup of 4 W waves on an otherwise idle chip: the dispatcher deals a workgroup's waves round-robin over the CU's 4
// SIMDs, so each SIMD holds W of them.  Every wave runs the same straight-line stream of independent v_fmac_f32 chains (inline
// asm: neither packed nor reordered) and times itself with the shader clock.  Reported: cycles per instruction as seen by one
// wave, and per SIMD (= the former / W if the waves really share one SIMD): the MI355X guide says a wave64 FMA occupies the
// SIMD-32 for 2 cycles but ONE wave alone issues only every 4 -- i.e. two waves per SIMD should each still see ~4.5.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <algorithm>

template <int CH, int UNROLL>
__global__ void __launch_bounds__(1024) body(float* out, int iters, long long* cyc) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float x[CH], y[CH], z[CH];
    for (int c = 0; c < CH; c++) { x[c] = 1.0f + lane*1e-3f + c + wave; y[c] = 1.0f + 1e-6f*(lane + c); z[c] = 1e-7f*(c + 1 + lane); }
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < UNROLL; u++)
#pragma unroll
            for (int c = 0; c < CH; c++) asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(x[c]) : "v"(y[c]), "v"(z[c]));
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0; for (int c = 0; c < CH; c++) s += x[c];
    out[threadIdx.x] = s;
    if (lane == 0) cyc[wave] = t1 - t0;
}

template <int CH> void run(int W, float* out, long long* cyc) {
    constexpr int UNROLL = 64; const int iters = 4096; long long c[16] = {0};
    for (int rep = 0; rep < 3; rep++) {
        hipLaunchKernelGGL((body<CH, UNROLL>), dim3(1), dim3(256*W), 0, 0, out, iters, cyc);
        (void)hipMemcpy(c, cyc, 8*4*W, hipMemcpyDeviceToHost);
    }
    const double n = (double)iters*UNROLL*CH;
    long long mx = *std::max_element(c, c + 4*W), mn = *std::min_element(c, c + 4*W);
    printf("W = %d waves per SIMD, %d independent chains per wave: %5.2f .. %5.2f cycles / instruction per wave, %5.2f cycles / wave-instruction per SIMD"
           "  => %4.1f %% of the 2-cycle SIMD-32 rate\n", W, CH, mn/n, mx/n, mx/n/W, 100.0*2.0/(mx/n/W));
}

int main() {
    float* out; long long* cyc; (void)hipMalloc(&out, 4096); (void)hipMalloc(&cyc, 128);
    for (int W = 1; W <= 4; W++) run<8>(W, out, cyc);
    for (int W = 1; W <= 4; W++) run<2>(W, out, cyc);
    for (int W = 1; W <= 4; W++) run<1>(W, out, cyc);
    return 0;
}
