// Micro-benchmark (round 6): what does ONE L2 read request (TCC_EA0_RDREQ, the counter FETCH_SIZE is made of) move -- the sectors a wave asked
// for, or the whole 128-byte line? The attention kernels read 80-byte head slices of 640-byte rows with 16-byte-per-lane buffer loads, and the
// counters of rounds 2 - 5 were read as "one request = 64 bytes" (calibrated on a kernel whose rows had the same slicing: the calibration
// could not see the difference). Here a buffer far larger than the 256 MiB Infinity Cache is read with 16-byte-per-lane loads that touch
//   full  : every byte of every 128-byte line                      (8 lanes per line)
//   half  : the first 64 bytes of every line                       (4 lanes per line)
//   b80   : the first 80 bytes of every 640-byte row               (5 lanes per row: one head's slice)
//   b160  : the first 160 bytes of every 640-byte row              (10 lanes per row: two adjacent heads)
//   sect  : the first 32 bytes of every line                       (2 lanes per line)
// and reports time, touched lines per second and "useful" bytes per second. If a request moves whole lines, `half` and `sect` take as long
// as `full` per LINE; if it moves sectors, they take half / a quarter. Run under `rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum`
// (and FETCH_SIZE in its own pass) to tie the request count to the lines touched.
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench_linefill tools/ubench_linefill.cpp
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// unit = `span` bytes at the start of every `pitch` bytes; lanes_per_unit = span / 16
template <int PITCH, int SPAN>
__global__ void __launch_bounds__(256) reader(const char *buf, long units, unsigned *out) {
    constexpr int LPU = SPAN / 16;
    const long nchunk = units * LPU;
    unsigned acc = 0;
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nchunk; i += stride) {
        const long u = i / LPU;
        const int c = (int)(i - u * LPU);
        const u32x4 v = *reinterpret_cast<const u32x4 *>(buf + u * PITCH + c * 16);
        acc ^= v[0] ^ v[1] ^ v[2] ^ v[3];
    }
    if (acc == 0x12345u) out[0] = acc;      // (never: keeps the loads)
}

template <int PITCH, int SPAN>
int run(const char *name, const char *buf, size_t bytes, unsigned *out) {
    const long units = (long)(bytes / PITCH);
    const dim3 grid(256 * 16), block(256);
    hipLaunchKernelGGL((reader<PITCH, SPAN>), grid, block, 0, 0, buf, units / 16, out);
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        CHK(hipEventRecord(e0));
        hipLaunchKernelGGL((reader<PITCH, SPAN>), grid, block, 0, 0, buf, units, out);
        CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
        float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    // 128-byte lines a unit touches (units start on a line boundary when PITCH % 128 == 0)
    const double lines = (double)units * ((SPAN + 127) / 128);
    const double useful = (double)units * SPAN;
    printf("%-6s pitch %4d span %4d : %8.3f ms  %7.1f G lines/s  useful %7.1f GB/s  whole lines %7.1f GB/s  (%.0f lines, %.0f 64-byte blocks, %.0f sectors touched)\n", name, PITCH, SPAN, best,
           lines / best * 1e-6, useful / best * 1e-6, lines * 128 / best * 1e-6, lines, (double)units * ((SPAN + 63) / 64), (double)units * ((SPAN + 31) / 32));
    return 0;
}

int main(int argc, char **argv) {
    const size_t bytes = (size_t)(argc > 1 ? atol(argv[1]) : 2048) << 20;      // MiB; default 2 GiB (8 x the Infinity Cache)
    char *buf; unsigned *out;
    CHK(hipMalloc(&buf, bytes)); CHK(hipMalloc(&out, 64));
    CHK(hipMemset(buf, 1, bytes)); CHK(hipDeviceSynchronize());
    if (run<128, 128>("full", buf, bytes, out)) return 1;
    if (run<128, 64>("half", buf, bytes, out)) return 1;
    if (run<128, 32>("sect", buf, bytes, out)) return 1;
    if (run<640, 80>("b80", buf, bytes, out)) return 1;
    if (run<640, 160>("b160", buf, bytes, out)) return 1;
    if (run<640, 640>("row", buf, bytes, out)) return 1;
    CHK(hipFree(buf)); CHK(hipFree(out));
    return 0;
}
