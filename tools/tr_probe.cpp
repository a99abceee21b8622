// Probe of ds_read_b64_tr_b16 (gfx950): LDS holds halfs with value = element index; lane l supplies byte address
// addr(l) and we print the four elements it receives, for a few address patterns.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short s4 __attribute__((ext_vector_type(4)));
__global__ void k(float *out, int mode) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    for (int i = threadIdx.x; i < 2048; i += 64) ((_Float16 *)smem)[i] = (_Float16)i;
    __syncthreads();
    const int l = threadIdx.x;
    int addr;
    if (mode == 0) addr = l * 8;                                   // lane-linear pieces
    else if (mode == 1) addr = ((l & 15) >> 2) * 128 + (l & 3) * 8 + (l >> 4) * 512;   // 16-lane group: 4 rows of 64 elements, 4 pieces per row
    else addr = ((l & 15) >> 2) * 80 + (l & 3) * 8 + (l >> 4) * 32;                      // rows of 40 elements (80 B), groups 16 columns apart
    __attribute__((address_space(3))) s4 *p = (__attribute__((address_space(3))) s4 *)(smem + addr);
    s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(p);
    for (int j = 0; j < 4; ++j) { short h = v[j]; out[l * 4 + j] = (float)__builtin_bit_cast(_Float16, h); }
}
int main() {
    float *d; hipMalloc(&d, 256 * 4); float h[256];
    for (int mode = 0; mode < 3; ++mode) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 8192, 0, d, mode);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("mode %d\n", mode);
        for (int l = 0; l < 64; ++l) printf("  lane %2d: %5.0f %5.0f %5.0f %5.0f%s", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3], (l & 3) == 3 ? "\n" : "");
    }
    return 0;
}
