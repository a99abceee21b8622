// Shared device/host helpers for libpww_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/pww_hip.h"

// PWW_EXPERIMENTS = 1 builds libpww_hip_experiments.so: the product library plus the forms that were built, measured and NOT made a default
// (round 3's in-launch score statistic with its workgroup hand-off: pww_cross_attn_fwd_fused[_ex]; round 5's single-buffered key-split
// self-attention kernel and the to_out epilogue kernel; the A/B instantiations behind PWW_DEBUG). Tests and tools load it; the product never does.
#ifndef PWW_EXPERIMENTS
#define PWW_EXPERIMENTS 0
#endif

namespace pww {

typedef _Float16 f16;
typedef __bf16 bf16;
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
// 16-byte staging register. NOT HIP's uint4 (a struct): `cond ? a : b` on struct lvalues selects between
// ADDRESSES and forces both operands into scratch memory; on a native vector it is a register select.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

template <typename T> struct Vec;
template <> struct Vec<f16> { typedef f16x8 v8; typedef f16x4 v4; };
template <> struct Vec<bf16> { typedef bf16x8 v8; typedef bf16x4 v4; };

// D(32x32 f32) += A(32x16) * B(16x32). Lane l supplies A[l&31][8*(l>>5)+j] and B[8*(l>>5)+j][l&31],
// j=0..7; receives D[(r&3)+8*(r>>2)+4*(l>>5)][l&31] in register r (guide: cdna_hip_programming §3).
__device__ __forceinline__ f32x16 mfma32(f16x8 a, f16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mfma32(bf16x8 a, bf16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

template <typename V8> __device__ __forceinline__ V8 zero8() {
    const u32x4 z = {0u, 0u, 0u, 0u};
    return __builtin_bit_cast(V8, z);
}

// Swap bits 2 and 3 of a 5-bit row index. Feeding K rows to the MFMA in this order makes each
// lane's 8 consecutive accumulator registers correspond to 8 consecutive keys (see pww_attn.hip).
__device__ __forceinline__ int swap23(int i) { return (i & ~12) | ((i & 4) << 1) | ((i & 8) >> 1); }

// ---- host side -------------------------------------------------------------------------------
// A/B knobs of experiments DESIGN.md declares closed (each measured, the default is the winner): ONE environment variable,
// PWW_DEBUG="name=value,name=value", parsed once per process. Nothing in here changes results beyond rounding; the product never sets it.
struct DebugKnobs {
    int attn_rf = 1;               // 0: running-maximum softmax instead of the range-free one
    int attn_ksplit = 1;           // 0: no key-split workgroups for under-filled launches
    int attn_fold = 1;             // folded-reference d = 40 kernel: 0 never / 1 bf16 and fp16 / 2 bf16 only
    int attn_nw8 = 1;              // 0: no 8-wave workgroups
    int attn_pair_major = 16;      // pair-major workgroup order from this many (image, head) pairs on (0: never)
    int attn_head_pairs = 2;       // head slices that are not whole 128-byte lines: groups of 2 (4) adjacent heads on one XCD; 0: round 5's order
    int cross_head_major = 1;      // multi-block cross kernel: the heads of one (image, query chunk) adjacent on one XCD; 0: round 5's order
    int attn_wide_store = 1;       // 0: 8-byte instead of 16-byte epilogue stores
    int cross_wg_per_cu = 4;       // upper bound on the resident workgroups per CU the hand-off launch counts on
    int cross_assume_resident = 0; // TEST HOOK: count on n workgroups per CU whatever the occupancy query says (drives the time-out path)
    int cross_gate_weight = 2;     // cost of a gated-in image's query block in units of a gated-out one's (1: ignore the hint)
    int cross_tile_nbuf = 2;       // 1: never double-buffer the bias tile
    int cross_bias_lds = 2;        // bias rows: 0 per lane from global memory / 1 LDS tile in single-block launches only / 2 LDS tile everywhere
    int cross_lean = 1;            // pass-2-only launches on the small kernel (pww_cross_lean.hip): 0 never / 1 where it fits (default) / 2 also for large batches
    int cross_lean_multi = 1;      // batched launches (> 1024 blocks of 128 rows, head dim <= 64): the small kernel walking several blocks per workgroup; 0: the general kernel (round 5)
    int cross_lean_nw = 0;         // waves per workgroup of that kernel: 0 by problem size / 2 / 4
    int attn_ksplit1 = 0;          // 1: the small self-attention launches (N >= 512 at d = 80 / 96, N >= 128 at d = 128 / 160) on the single-buffered key-split kernel of round 5 (measured slower: default off)
    int attn_hot_sum = 8;          // f16 d = 40 self-attention: a workgroup whose first key stage leaves a row sum below this follows the running maximum lazily (0: never -- round 5; -1: always)
    int attn_fold_limit_f16 = 0;   // > 0: overrides the f16 magnitude-guard limit of the folded-reference kernel (exp2 units; built-in 36)
    int attn_ksplit_half = 0;      // (experiments library) d = 80 / 96 key-split self-attention: 1 = 2 row groups x 4 key groups of a 32-key block each (8 waves, round 6: a tie) / 0 = 2 x 2 of a 64-key sub-tile each
    int attn_ksplit_nw = 2;        // row groups of the key-split d = 80 / 96 self-attention workgroup: 2 (x 2 key groups) or 4 (x 2)
};
const DebugKnobs &debug_knobs();

void set_error(const char *fmt, ...);
int check_hip(hipError_t e, const char *what);
bool arch_ok();

// Kernel-only timing (pww_profile_*): if this thread armed a timing slot, hand out its event pair once.
bool profile_take(hipEvent_t *start, hipEvent_t *stop, hipStream_t stream);

// Phase time stamps (pww_debug_timeline): the process-wide debug buffer, or null.
unsigned long long *debug_timeline();
size_t debug_timeline_bytes();

// Launch of an attention-class kernel. An armed launch goes through hipExtLaunchKernelGGL, which stamps the dispatch's own
// start / end device timestamps into the two events: the duration rocprofv3 reports for the kernel, free of host latency and of
// the gap between dispatches that an event pair AROUND a launch includes.
template <typename K, typename P>
static inline void launch_attn_kernel(K kern, dim3 grid, dim3 block, size_t lds, hipStream_t stream, const P &params) {
    hipEvent_t e0, e1;
    if (profile_take(&e0, &e1, stream)) hipExtLaunchKernelGGL(kern, grid, block, (unsigned)lds, stream, e0, e1, 0, params);
    else hipLaunchKernelGGL(kern, grid, block, lds, stream, params);
}

// The same for kernels that take their arguments one by one (mask build, CFG combine: the streaming helpers of the path).
template <typename K, typename... A>
static inline void launch_timed(K kern, dim3 grid, dim3 block, size_t lds, hipStream_t stream, A... args) {
    hipEvent_t e0, e1;
    if (profile_take(&e0, &e1, stream)) hipExtLaunchKernelGGL(kern, grid, block, (unsigned)lds, stream, e0, e1, 0, args...);
    else hipLaunchKernelGGL(kern, grid, block, lds, stream, args...);
}

}  // namespace pww
