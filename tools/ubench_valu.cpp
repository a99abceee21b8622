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
        if (OP == 7) { REP8(asm volatile("v_cvt_pk_f16_f32 %0, %0, %1\n v_cvt_pk_f16_f32 %1, %1, %2\n v_cvt_pk_f16_f32 %2, %2, %3\n v_cvt_pk_f16_f32 %3, %3, %4\n v_cvt_pk_f16_f32 %4, %4, %5\n v_cvt_pk_f16_f32 %5, %5, %6\n v_cvt_pk_f16_f32 %6, %6, %7\n v_cvt_pk_f16_f32 %7, %7, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));) }
        if (OP == 8) { REP8(asm volatile("v_cvt_pkrtz_f16_f32 %0, %0, %1\n v_cvt_pkrtz_f16_f32 %1, %1, %2\n v_cvt_pkrtz_f16_f32 %2, %2, %3\n v_cvt_pkrtz_f16_f32 %3, %3, %4\n v_cvt_pkrtz_f16_f32 %4, %4, %5\n v_cvt_pkrtz_f16_f32 %5, %5, %6\n v_cvt_pkrtz_f16_f32 %6, %6, %7\n v_cvt_pkrtz_f16_f32 %7, %7, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));) }
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
// (round 6) MFMA issue rate, f16 against bf16 operands: 4 independent accumulators per wave, one or two waves per SIMD. The d = 40 self-attention
// kernel runs 13 % slower on f16 than on bf16 with the same instruction mix (481 against 424 us at 16 rows): is it the matrix pipe?
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <int BF> __global__ void km(float *out, int iters, float seed) {
    f32x16 c0, c1, c2, c3;
    for (int r = 0; r < 16; ++r) { c0[r] = seed; c1[r] = seed + 1; c2[r] = seed + 2; c3[r] = seed + 3; }
    f16x8 ah, bh; bf16x8 ab, bb;
    for (int j = 0; j < 8; ++j) { ah[j] = (_Float16)(0.001f * (threadIdx.x + j)); bh[j] = (_Float16)(0.002f * j + seed); ab[j] = (__bf16)(0.001f * (threadIdx.x + j)); bb[j] = (__bf16)(0.002f * j + seed); }
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (BF) { c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, c1, 0, 0, 0);
                      c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, c3, 0, 0, 0); }
            else { c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, c1, 0, 0, 0);
                   c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, c3, 0, 0, 0); }
        }
    }
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += c0[r] + c1[r] + c2[r] + c3[r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int BF> int run_mfma(const char *name, float *d_out) {
    const int iters = 4000;
    for (int wps = 1; wps <= 2; ++wps) {
        dim3 grid(256 * wps), block(256);
        hipLaunchKernelGGL(km<BF>, grid, block, 0, 0, d_out, 10, 1.0f);
        hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
        CHK(hipEventRecord(e0)); hipLaunchKernelGGL(km<BF>, grid, block, 0, 0, d_out, iters, 1.0f); CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
        float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
        const double n = (double)iters * 16 * wps;       // MFMAs per SIMD
        const double tf = n * 1024 * 2.0 * 32 * 32 * 16 / (ms * 1e-3) / 1e12;
        printf("%-28s waves/SIMD=%d  %.3f ms  -> %.1f ns per MFMA per SIMD (= %.1f cycles @2.4GHz), %.0f TFLOP/s\n", name, wps, ms, ms * 1e6 / n, ms * 1e6 / n * 2.4, tf);
    }
    return 0;
}
int main() {
    float *d_out; CHK(hipMalloc(&d_out, 512 * 256 * 4));
    run<7>("v_cvt_pk_f16_f32", d_out); run<8>("v_cvt_pkrtz_f16_f32", d_out);
    run_mfma<0>("v_mfma_f32_32x32x16_f16", d_out); run_mfma<1>("v_mfma_f32_32x32x16_bf16", d_out);
    run<1>("v_fma_f32", d_out); run<6>("v_mul_f32", d_out); run<5>("v_pk_fma_f32", d_out); run<3>("v_max3_f32", d_out); run<4>("v_cvt_pk_bf16_f32", d_out);
    run<0>("v_exp_f32", d_out); run<2>("v_exp_f16", d_out);
    return 0;
}
