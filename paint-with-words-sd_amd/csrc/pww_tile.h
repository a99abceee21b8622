// Tile machinery shared by the fused attention kernel (pww_attn.hip) and the score-reduction
// pre-pass (pww_reduce.hip): Q fragments in registers, K tiles staged through LDS, and the
// "swapped" score tile S^T = K Q^T on the 32x32x16 MFMA so that every lane owns ONE query row.
//
// Geometry (gfx950, wave64):
//   - a wave owns 32 query rows; lane l works for row (l & 31); hi = l >> 5 selects which half of
//     each 16-wide contraction slice / which 8 of every 16 keys the lane holds.
//   - a KV tile is KVBLK = 64 keys = two 32-key blocks. For block kb the MFMA A operand is
//     K[key = kb*32 + swap23(l & 31)][d = ks*16 + hi*8 .. +7], the B operand is the lane's own
//     Q[row][same d range]. Accumulator register r of block kb then holds the raw score of
//     key = kb*32 + 16*(r >> 3) + 8*hi + (r & 7): 8 consecutive keys per 8 consecutive registers,
//     which is exactly the B-operand packing the PV MFMA wants (no cross-lane traffic for P).
#pragma once
#include "pww_common.h"

namespace pww {

constexpr int KVBLK = 64;

template <int KS> struct KTile {
    static constexpr int DPAD = KS * 16;           // head dim padded to the MFMA k granularity
    static constexpr int CHK = DPAD / 8;           // 16-byte chunks per key row
    static constexpr int STRIDE = DPAD * 2 + 16;   // bytes; +16 makes the row->slot map odd => conflict-free b128 reads
    static constexpr int BYTES = KVBLK * STRIDE;
    static constexpr int NCHUNK = KVBLK * CHK;
};

// Q fragments of one wave: lane (hi, row) keeps Q[row][ks*16 + hi*8 .. +7] for every k-step.
template <typename T, int KS>
__device__ __forceinline__ void load_q_frags(typename Vec<T>::v8 (&qf)[KS], const T *qrow_ptr,
                                              bool qvalid, int hi, int D) {
    typedef typename Vec<T>::v8 V8;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        const int d0 = ks * 16 + hi * 8;
        V8 v = zero8<V8>();
        if (qvalid && d0 < D) v = *reinterpret_cast<const V8 *>(qrow_ptr + d0);
        qf[ks] = v;
    }
}

// Global -> registers for one K tile (keys key0 .. key0+63). Out-of-range keys and the head-dim
// padding are zero-filled so the MFMA never sees uninitialised LDS.
template <typename T, int KS, int NT, int KPT>
__device__ __forceinline__ void ktile_load(u32x4 (&kreg)[KPT], const T *Kp, long k_sm, int key0,
                                           int M, int D, int tid) {
    typedef KTile<KS> KT;
#pragma unroll
    for (int i = 0; i < KPT; ++i) {
        const int c = tid + i * NT;
        // unconditional load from an always-valid address (clamped row, chunk 0 for padding): a load inside
        // an `if` would get an s_waitcnt vmcnt(0) right behind it; out-of-range data is zeroed at store time
        const int key = min(c / KT::CHK, KVBLK - 1), ch = c % KT::CHK;
        const int gk = min(key0 + key, M - 1), d0 = ch * 8 < D ? ch * 8 : 0;
        kreg[i] = *reinterpret_cast<const u32x4 *>(Kp + (long)gk * k_sm + d0);
    }
}

template <int KS, int NT, int KPT>
__device__ __forceinline__ void ktile_store(const u32x4 (&kreg)[KPT], char *Ks, int tid, int key0, int M, int D) {
    typedef KTile<KS> KT;
    const u32x4 z = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int i = 0; i < KPT; ++i) {
        const int c = tid + i * NT;
        if (c < KT::NCHUNK) {
            const int key = c / KT::CHK, ch = c % KT::CHK;
            const bool ok = key0 + key < M && ch * 8 < D;
            *reinterpret_cast<u32x4 *>(Ks + key * KT::STRIDE + ch * 16) = ok ? kreg[i] : z;
        }
    }
}

// s[kb] = K_block(kb) Q^T for the two 32-key blocks of the tile; a block entirely past M is
// skipped (wave-uniform) and left at zero.
// EVERY K fragment of the tile is requested from LDS before the first MFMA (round 5). The straightforward form -- one ds_read_b128 in
// front of each MFMA -- compiles to read -> s_waitcnt lgkmcnt(0) -> MFMA per k-step (hipcc re-uses one register quad for all the
// fragments): 2 KS exposed LDS latencies per tile, which is what a tile cost at one wave per SIMD (the N <= 1024 self-attention and
// every cross-attention launch at 2 folded rows: ~3000 cycles per tile for ~700 cycles of MFMA). Same MFMAs in the same order: bit-identical scores.
// ONLY = 0 / 1: just that 32-key block of the tile (key-split workgroups whose key groups own HALF a tile each); -1: both.
template <typename T, int KS, int ONLY = -1>
__device__ __forceinline__ void score_tile(f32x16 (&s)[2], const typename Vec<T>::v8 (&qf)[KS],
                                           const char *Ks, int key0, int M, int l31, int hi) {
    typedef typename Vec<T>::v8 V8;
    typedef KTile<KS> KT;
    const char *base = Ks + swap23(l31) * KT::STRIDE + hi * 16;
    const bool live0 = ONLY != 1 && key0 < M, live1 = ONLY != 0 && key0 + 32 < M;      // wave-uniform
    V8 kf[2][KS];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
        if (kb == 0 ? live0 : live1) {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) kf[kb][ks] = *reinterpret_cast<const V8 *>(base + kb * 32 * KT::STRIDE + ks * 32);
        }
    }
    __builtin_amdgcn_sched_barrier(0);          // keep the requests ahead of the MFMAs (hipcc sinks them otherwise)
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        if (kb == 0 ? live0 : live1) {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) acc = mfma32(kf[kb][ks], qf[ks], acc);
        }
        s[kb] = acc;
    }
}

// key index (within the tile) of accumulator register r of block kb for this lane.
__device__ __forceinline__ int key_of(int kb, int r, int hi) {
    return kb * 32 + 16 * (r >> 3) + 8 * hi + (r & 7);
}

}  // namespace pww
