// Micro-benchmark: do MFMA and VALU work overlap on one SIMD (gfx950)?  (a) inside one wave's in-order stream,
// (b) between the two waves of a SIMD. Each loop iteration = 1 v_mfma_f32_32x32x16_f16 (two accumulators alternate)
// followed by NEXP v_exp_f32 and NFMA v_fma_f32 on independent registers. Cycles by s_memtime (wave 0 of block 0)
// and wall time by HIP events.
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %s\n", hipGetErrorString(e)); return 1; } } while (0)
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int NMFMA, int NEXP, int NFMA>
__device__ __forceinline__ void body(f32x16 &c0, f32x16 &c1, u32x4 a, u32x4 b, float (&e)[8], float k, float d) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        if (NMFMA) {
            if (u == 0) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c0) : "v"(a), "v"(b));
            else        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c1) : "v"(a), "v"(b));
        }
#pragma unroll
        for (int i = 0; i < NEXP; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(e[i & 7]));
#pragma unroll
        for (int i = 0; i < NFMA; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(e[(i + NEXP) & 7]) : "v"(k), "v"(d));
    }
}

// ROLE = 0: every wave runs the mixed body. ROLE = 1: waves 0-3 of a 512-thread block run MFMA only, waves 4-7 VALU only.
template <int NMFMA, int NEXP, int NFMA, int ROLE>
__global__ void mix(float *out, unsigned long long *cyc, int iters) {
    f32x16 c0, c1;
    for (int r = 0; r < 16; ++r) { c0[r] = 0.f; c1[r] = 0.f; }
    u32x4 a = {0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u}, b = {0u, 0u, 0u, threadIdx.x};
    float e[8];
    for (int i = 0; i < 8; ++i) e[i] = -1.f - i - threadIdx.x * 1e-3f;
    const float k = 0.999f, d = 1e-3f;
    const bool mfma_role = (threadIdx.x >> 6) < 4;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (ROLE == 0) {
        for (int it = 0; it < iters; ++it) body<NMFMA, NEXP, NFMA>(c0, c1, a, b, e, k, d);
    } else if (mfma_role) {
        for (int it = 0; it < iters; ++it) body<1, 0, 0>(c0, c1, a, b, e, k, d);
    } else {
        for (int it = 0; it < iters; ++it) body<0, NEXP, NFMA>(c0, c1, a, b, e, k, d);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += c0[r] + c1[r];
    for (int i = 0; i < 8; ++i) s += e[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) cyc[threadIdx.x >> 6] = t1 - t0;
}

template <int NMFMA, int NEXP, int NFMA, int ROLE>
int run(float *d_out, unsigned long long *d_cyc, int wps) {
    const int iters = 4000;
    dim3 grid(ROLE ? 256 : 256 * wps), block(ROLE ? 512 : 256);
    hipLaunchKernelGGL((mix<NMFMA, NEXP, NFMA, ROLE>), grid, block, 0, 0, d_out, d_cyc, 10);
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    CHK(hipEventRecord(e0));
    hipLaunchKernelGGL((mix<NMFMA, NEXP, NFMA, ROLE>), grid, block, 0, 0, d_out, d_cyc, iters);
    CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
    float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long h[8]; CHK(hipMemcpy(h, d_cyc, sizeof(h), hipMemcpyDeviceToHost));
    const double per_pair = (double)h[0] / iters / 2.0, per_pair_hi = (double)h[ROLE ? 4 : 0] / iters / 2.0;
    printf("%s mfma=%d exp=%d fma=%d waves/SIMD=%d | %.3f ms | %.1f ticks per (MFMA+VALU group) wave0", ROLE ? "ROLE-SPLIT" : "MIXED     ",
           NMFMA, NEXP, NFMA, ROLE ? 2 : wps, ms, per_pair);
    if (ROLE) printf(" (MFMA wave), %.1f (VALU wave 4)", per_pair_hi);
    printf(" | wall %.1f ns per group per SIMD\n", ms * 1e6 / (iters * 2.0 * (ROLE ? 1 : wps)));
    return 0;
}

int main() {
    float *d_out; unsigned long long *d_cyc;
    CHK(hipMalloc(&d_out, 512 * 512 * 4)); CHK(hipMalloc(&d_cyc, 64));
    for (int wps = 1; wps <= 2; ++wps) {
        run<1, 0, 0, 0>(d_out, d_cyc, wps);
        run<0, 4, 0, 0>(d_out, d_cyc, wps);
        run<1, 2, 0, 0>(d_out, d_cyc, wps);
        run<1, 4, 0, 0>(d_out, d_cyc, wps);
        run<1, 6, 0, 0>(d_out, d_cyc, wps);
        run<0, 0, 6, 0>(d_out, d_cyc, wps);
        run<1, 0, 3, 0>(d_out, d_cyc, wps);
        run<1, 0, 6, 0>(d_out, d_cyc, wps);
        run<1, 0, 10, 0>(d_out, d_cyc, wps);
        run<1, 2, 3, 0>(d_out, d_cyc, wps);
    }
    run<1, 4, 0, 1>(d_out, d_cyc, 2);
    run<1, 0, 6, 1>(d_out, d_cyc, 2);
    run<1, 2, 3, 1>(d_out, d_cyc, 2);
    return 0;
}
