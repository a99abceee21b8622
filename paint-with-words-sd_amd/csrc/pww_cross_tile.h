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
}  // namespace pww
