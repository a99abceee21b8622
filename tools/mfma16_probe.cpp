// Hardware probe (test infrastructure): semantics of v_permlane16_swap and the operand / result layout of
// v_mfma_f32_16x16x32_f16 on gfx950, as assumed by the 16x16 PV path of pww_attn.hip.
//   hipcc --offload-arch=gfx950 -O2 tools/mfma16_probe.cpp -o tools/mfma16_probe && tools/mfma16_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void probe(unsigned *swap_out, const float *A, const float *B, float *D) {
    const int l = threadIdx.x;
    // (1) permlane16_swap(x, y): which lanes of x / y come back in r[0] / r[1]?
    const auto r = __builtin_amdgcn_permlane16_swap((unsigned)l, 100u + l, false, false);
    swap_out[l] = r[0];
    swap_out[64 + l] = r[1];
    // (2) 16x16x32: assumed A[m = l & 15][k = 8 * (l >> 4) + j], B[k = 8 * (l >> 4) + j][n = l & 15], D[m = 4 * (l >> 4) + r][n = l & 15]
    f16x8 a, b;
    for (int j = 0; j < 8; ++j) {
        a[j] = (_Float16)A[(l & 15) * 32 + 8 * (l >> 4) + j];
        b[j] = (_Float16)B[(8 * (l >> 4) + j) * 16 + (l & 15)];
    }
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    for (int rr = 0; rr < 4; ++rr) D[(4 * (l >> 4) + rr) * 16 + (l & 15)] = c[rr];
}

int main() {
    unsigned *ds; float *dA, *dB, *dD;
    float A[16 * 32], B[32 * 16], D[256];
    for (int i = 0; i < 512; ++i) { A[i] = (float)((i * 7) % 11 - 5); B[i] = (float)((i * 5) % 13 - 6); }
    hipMalloc(&ds, 128 * 4); hipMalloc(&dA, sizeof(A)); hipMalloc(&dB, sizeof(B)); hipMalloc(&dD, sizeof(D));
    hipMemcpy(dA, A, sizeof(A), hipMemcpyHostToDevice); hipMemcpy(dB, B, sizeof(B), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, ds, dA, dB, dD);
    unsigned s[128]; hipMemcpy(s, ds, sizeof(s), hipMemcpyDeviceToHost); hipMemcpy(D, dD, sizeof(D), hipMemcpyDeviceToHost);
    printf("permlane16_swap(x = lane, y = 100 + lane):\n r[0]:"); for (int i = 0; i < 64; ++i) printf(" %u", s[i]);
    printf("\n r[1]:"); for (int i = 0; i < 64; ++i) printf(" %u", s[64 + i]);
    double err = 0;
    for (int m = 0; m < 16; ++m) for (int n = 0; n < 16; ++n) { double ref = 0; for (int k = 0; k < 32; ++k) ref += A[m * 32 + k] * B[k * 16 + n]; err = fmax(err, fabs(ref - D[m * 16 + n])); }
    printf("\nmfma_f32_16x16x32_f16 with the assumed A/B/D layout: max |D - A B| = %g\n", err);
    return 0;
}
