// Micro-benchmark: issue cost (cycles per wave64 instruction per SIMD) of the VALU ops in the softmax inner loop.
// One wave per SIMD (1024 waves) and two waves per SIMD; independent chains so latency does not matter.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %s\n", hipGetErrorString(e)); return 1; } } while (0)
#define REP8(X) X X X X X X X X
template <int OP> __global__ void k(float *out, int iters, float seed) {
    float a0 = seed + threadIdx.x, a1 = a0 * 1.1f, a2 = a0 * 1.2f, a3 = a0 * 1.3f, a4 = a0 * 1.4f, a5 = a0 * 1.5f, a6 = a0 * 1.6f, a7 = a0 * 1.7f;
    const float c = 0.999f, d = 1e-3f;
    for (int i = 0; i < iters; ++i) {
        if (OP == 0) { REP8(asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));) }
        if (OP == 1) { REP8(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(d));) }
        if (OP == 2) { REP8(asm volatile("v_exp_f16 %0, %0\n v_exp_f16 %1, %1\n v_exp_f16 %2, %2\n v_exp_f16 %3, %3\n v_exp_f16 %4, %4\n v_exp_f16 %5, %5\n v_exp_f16 %6, %6\n v_exp_f16 %7, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));) }
        if (OP == 3) { REP8(asm volatile("v_max3_f32 %0, %0, %1, %2\n v_max3_f32 %1, %1, %2, %3\n v_max3_f32 %2, %2, %3, %4\n v_max3_f32 %3, %3, %4, %5\n v_max3_f32 %4, %4, %5, %6\n v_max3_f32 %5, %5, %6, %7\n v_max3_f32 %6, %6, %7, %0\n v_max3_f32 %7, %7, %0, %1" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));) }
        if (OP == 4) { REP8(asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1\n v_cvt_pk_bf16_f32 %1, %1, %2\n v_cvt_pk_bf16_f32 %2, %2, %3\n v_cvt_pk_bf16_f32 %3, %3, %4\n v_cvt_pk_bf16_f32 %4, %4, %5\n v_cvt_pk_bf16_f32 %5, %5, %6\n v_cvt_pk_bf16_f32 %6, %6, %7\n v_cvt_pk_bf16_f32 %7, %7, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));) }
        if (OP == 5) {   // packed fma on register pairs
            typedef float f2 __attribute__((ext_vector_type(2)));
            f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, cc = {c, c}, dd = {d, d};
            REP8(asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(cc), "v"(dd));)
            a0 = p0[0]; a1 = p0[1]; a2 = p1[0]; a3 = p1[1]; a4 = p2[0]; a5 = p2[1]; a6 = p3[0]; a7 = p3[1];
        }
        if (OP == 6) { REP8(asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));) }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
template <int OP> int run(const char *name, float *d_out) {
    const int iters = 2000;
    for (int wps = 1; wps <= 2; ++wps) {
        dim3 grid(256 * wps), block(256);    // 4 waves per block -> one per SIMD (x wps blocks per CU)
        hipLaunchKernelGGL(k<OP>, grid, block, 0, 0, d_out, 10, 1.0f);
        hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
        CHK(hipEventRecord(e0)); hipLaunchKernelGGL(k<OP>, grid, block, 0, 0, d_out, iters, 1.0f); CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
        float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
        const double insts_per_simd = (double)iters * 64 * wps;
        printf("%-20s waves/SIMD=%d  %.3f ms  -> %.2f ns per wave-instruction per SIMD (= %.1f cycles @2.4GHz)\n", name, wps, ms, ms * 1e6 / insts_per_simd, ms * 1e6 / insts_per_simd * 2.4);
    }
    return 0;
}
int main() {
    float *d_out; CHK(hipMalloc(&d_out, 512 * 256 * 4));
    run<1>("v_fma_f32", d_out); run<6>("v_mul_f32", d_out); run<5>("v_pk_fma_f32", d_out); run<3>("v_max3_f32", d_out); run<4>("v_cvt_pk_bf16_f32", d_out);
    run<0>("v_exp_f32", d_out); run<2>("v_exp_f16", d_out);
    return 0;
}
