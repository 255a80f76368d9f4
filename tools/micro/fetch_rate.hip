// Micro-benchmark: what bounds ONE lone wavefront running straight-line VALU code on gfx950 (the 4096-env regime:
// one physics wave per SIMD, nothing to interleave with)?
//   hipcc -O3 --offload-arch=gfx950 -o fetch_rate tools/micro/fetch_rate.hip && ./fetch_rate
// CH independent dependency chains (1 = every instruction depends on the previous one), body of UNROLL*CH instructions
// (inline asm, so the compiler neither packs nor reorders), executed `iters` times.  Variants: 3-VGPR v_fma_f32 (VOP3,
// 8 bytes), v_fmac_f32 (VOP2, 4 bytes), v_fmamk_f32 with a 32-bit literal (8 bytes), v_pk_fma_f32 (2 FMAs / lane).
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f2 __attribute__((ext_vector_type(2)));

template <int UNROLL, int KIND, int CH>
__global__ void body(float* out, int iters, long long* cyc) {
    const int lane = threadIdx.x;
    float x[CH], y[CH], z[CH]; f2 px[CH], py[CH], pz[CH];
    for (int c = 0; c < CH; c++) {
        x[c] = 1.0f + lane * 1e-3f + c; y[c] = 1.0f + 1e-6f * (lane + c); z[c] = 1e-7f * (c + 1 + lane);
        px[c] = f2{x[c], x[c] + 1.0f}; py[c] = f2{y[c], y[c]}; pz[c] = f2{z[c], z[c]};
    }
    const long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < UNROLL; u++)
#pragma unroll
            for (int c = 0; c < CH; c++) {
                if (KIND == 0) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x[c]) : "v"(y[c]), "v"(z[c]));
                if (KIND == 1) asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(x[c]) : "v"(y[c]), "v"(z[c]));
                if (KIND == 2) asm volatile("v_fmamk_f32 %0, %1, 0x3f800001, %0" : "+v"(x[c]) : "v"(y[c]));
                if (KIND == 3) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(px[c]) : "v"(py[c]), "v"(pz[c]));
            }
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0; for (int c = 0; c < CH; c++) s += x[c] + px[c].x + px[c].y;
    out[lane] = s;
    if (lane == 0) *cyc = t1 - t0;
}

template <int UNROLL, int KIND, int CH> void run(float* out, long long* cyc, const char* name) {
    const int iters = 65536 / UNROLL; long long c = 0;
    for (int rep = 0; rep < 3; rep++) { hipLaunchKernelGGL((body<UNROLL, KIND, CH>), dim3(1), dim3(64), 0, 0, out, iters, cyc); (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost); }
    const double n = (double)iters * UNROLL * CH;
    printf("%-34s chains %d  body %6d instr: %5.2f cycles / instruction\n", name, CH, UNROLL * CH, c / n);
}

int main() {
    float* out; long long* cyc; (void)hipMalloc(&out, 256); (void)hipMalloc(&cyc, 8);
    run<64, 0, 1>(out, cyc, "v_fma_f32 3 VGPR (VOP3, 8 B)"); run<64, 0, 2>(out, cyc, "v_fma_f32 3 VGPR (VOP3, 8 B)");
    run<64, 0, 4>(out, cyc, "v_fma_f32 3 VGPR (VOP3, 8 B)"); run<64, 0, 8>(out, cyc, "v_fma_f32 3 VGPR (VOP3, 8 B)"); run<1024, 0, 8>(out, cyc, "v_fma_f32 3 VGPR (VOP3, 8 B)");
    run<64, 1, 1>(out, cyc, "v_fmac_f32 (VOP2, 4 B)"); run<64, 1, 2>(out, cyc, "v_fmac_f32 (VOP2, 4 B)");
    run<64, 1, 4>(out, cyc, "v_fmac_f32 (VOP2, 4 B)"); run<64, 1, 8>(out, cyc, "v_fmac_f32 (VOP2, 4 B)"); run<1024, 1, 8>(out, cyc, "v_fmac_f32 (VOP2, 4 B)");
    run<64, 2, 1>(out, cyc, "v_fmamk_f32 literal (8 B)"); run<64, 2, 8>(out, cyc, "v_fmamk_f32 literal (8 B)"); run<1024, 2, 8>(out, cyc, "v_fmamk_f32 literal (8 B)");
    run<64, 3, 1>(out, cyc, "v_pk_fma_f32 (2 FMA / lane)"); run<64, 3, 4>(out, cyc, "v_pk_fma_f32 (2 FMA / lane)"); run<64, 3, 8>(out, cyc, "v_pk_fma_f32 (2 FMA / lane)");
    return 0;
}
