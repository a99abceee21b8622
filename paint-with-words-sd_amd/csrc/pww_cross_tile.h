// Helpers shared by the two cross-attention translation units (pww_cross.hip: the general kernel with the statistic formed in the launch;
// pww_cross_lean.hip: the small pass-2-only kernel of the product path).
#pragma once
#include "pww_attn_core.h"

namespace pww {

// Q fragments through a buffer descriptor: UNCONDITIONAL loads (validity goes into the offset: an out-of-range offset returns zeros).
// A load under an `if` -- rows past N, fragment halves past D -- sits in its own basic block and hipcc then waits for everything in
// flight at the join: a prefetch issued that way is no prefetch (the query-block loops of this kernel ran one exposed global-load
// latency per block until round 3). row_off = byte offset of the lane's row within the (image, head) slice, or OOB_OFF.
template <typename T, int KS, typename SRD>
__device__ __forceinline__ void load_q_frags_buf(typename Vec<T>::v8 (&qf)[KS], SRD srd, unsigned row_off, int hi, int D) {
    typedef typename Vec<T>::v8 V8;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        const int d0 = ks * 16 + hi * 8;
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(srd, d0 < D ? row_off + (unsigned)d0 * 2u : OOB_OFF, 0, 0);
        qf[ks] = __builtin_bit_cast(V8, v);
    }
}

// ---- the score statistic's partials at kernel entry (pww_cross_out.hip; pww_cross_lean.hip carries the same steps inline) -----------------
// [B][nparts][4] fp64 { max, min, sum, sum of squares }: lane i of EVERY wave requests partials i, i + 64, ... (PARTS_UNROLL per lane from
// the prologue's load batch) and folds them with shuffles -- the same order in every wave of every workgroup, no LDS, no barrier.
constexpr int PARTS_UNROLL = 4;
struct PartsRegs { u32x4 lo[PARTS_UNROLL], hi[PARTS_UNROLL]; };
struct PartsWant {
    bool f_max, f_min, f_sum, f_sq;
    __device__ __forceinline__ PartsWant(int stat_kind, bool all) {
        f_max = all || stat_kind == PWW_STAT_MAX || stat_kind == PWW_STAT_ABSMAX;
        f_min = all || stat_kind == PWW_STAT_MIN || stat_kind == PWW_STAT_ABSMAX;
        f_sum = all || stat_kind == PWW_STAT_MEAN || stat_kind == PWW_STAT_STD;
        f_sq = all || stat_kind == PWW_STAT_STD;
    }
};

__device__ __forceinline__ void parts_request(PartsRegs &r, const double *parts, int nparts, int b, bool enabled, const PartsWant &w, int lane) {
    // (disabled: the descriptor covers zero bytes -- the loads return zeros without touching memory)
    const unsigned bytes = (enabled && parts) ? (unsigned)nparts * 32u : 0u;
    const auto srd = __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(parts + (parts ? (long)b * nparts * 4 : 0)), 0, bytes, 0x00020000);
    const unsigned lo0 = (w.f_max || w.f_min) ? (unsigned)lane * 32u : OOB_OFF, hi0 = (w.f_sum || w.f_sq) ? (unsigned)lane * 32u + 16u : OOB_OFF;
#pragma unroll
    for (int j = 0; j < PARTS_UNROLL; ++j) {
        r.lo[j] = __builtin_amdgcn_raw_buffer_load_b128(srd, lo0 + (unsigned)(j * 2048), 0, 0);
        r.hi[j] = __builtin_amdgcn_raw_buffer_load_b128(srd, hi0 + (unsigned)(j * 2048), 0, 0);
    }
}

// st = folded { max, min, sum, sum of squares } of image b (fields that were not asked for hold the neutral element)
__device__ __forceinline__ void parts_fold(double (&st)[4], const PartsRegs &r, const double *parts, int nparts, int b, const PartsWant &w, int lane) {
    auto as_double = [](unsigned lo, unsigned hi32) { return __longlong_as_double((long long)(((unsigned long long)hi32 << 32) | lo)); };
    float vmax = -INFINITY, vmin = INFINITY;      // (a partial's extreme is a float stored as a double)
    double dsum = 0.0, dsq = 0.0;
#pragma unroll
    for (int j = 0; j < PARTS_UNROLL; ++j) {
        const bool mine = lane + j * 64 < nparts;
        if (w.f_max) vmax = fmaxf(vmax, mine ? (float)as_double(r.lo[j][0], r.lo[j][1]) : -INFINITY);
        if (w.f_min) vmin = fminf(vmin, mine ? (float)as_double(r.lo[j][2], r.lo[j][3]) : INFINITY);
        if (w.f_sum) dsum += mine ? as_double(r.hi[j][0], r.hi[j][1]) : 0.0;
        if (w.f_sq) dsq += mine ? as_double(r.hi[j][2], r.hi[j][3]) : 0.0;
    }
    for (int i = lane + PARTS_UNROLL * 64; i < nparts; i += 64) {      // (more than 256 partials per image: rare, not prefetched)
        const double *pp = parts + ((long)b * nparts + i) * 4;
        if (w.f_max) vmax = fmaxf(vmax, (float)pp[0]);
        if (w.f_min) vmin = fminf(vmin, (float)pp[1]);
        if (w.f_sum) dsum += pp[2];
        if (w.f_sq) dsq += pp[3];
    }
    if (w.f_max) {
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, off));
    }
    if (w.f_min) {
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) vmin = fminf(vmin, __shfl_xor(vmin, off));
    }
    if (w.f_sum) {
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) dsum += __shfl_xor(dsum, off);
    }
    if (w.f_sq) {
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) dsq += __shfl_xor(dsq, off);
    }
    st[0] = (double)vmax; st[1] = (double)vmin; st[2] = dsum; st[3] = dsq;
}

}  // namespace pww
