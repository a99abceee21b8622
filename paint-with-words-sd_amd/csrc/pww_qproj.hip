// Query projection of a cross-attention layer with the score statistic formed in the GEMM's epilogue.
//
//     Q = X W_q^T                               (paint_with_words/paint_with_words.py:76   query = self.to_q(hidden_states))
//     partial(image, tile) = (max, min, sum, sum of squares) of Q_tile K_h^T over the tile's heads, rows and the M prompt keys
//                                               (:87 attention_scores = Q K^T, reduced by weight_function: :402-405 qk.max(),
//                                                runner.py:104, README.md:152 qk.std())
//
// Why here: the statistic has to exist before ANY softmax of the image can start (:106 -> :112). Rounds 2-3 formed it inside the
// attention launch -- pass 1 over Q, a device-scope hand-off between all workgroups of an image (5-7 us, the launch had to be
// resident as a whole, a 1 s spin limit for the impossible case), pass 2 over Q again. The producer of Q already holds every
// Q tile in registers: multiplying the finished [32 tokens x head] tile with the request-cached K_h (77 keys) costs 9-30 extra MFMAs
// per head and tile, and the KERNEL BOUNDARY between this launch and the attention launch is the synchronisation. The attention
// kernel folds <= a few hundred fp64 partials per image at entry (pww_cross.hip, CrossParams::ext_part) and runs its second pass only.
//
// Shape of the work (gfx950, wave64):
//   * workgroup = 4 waves, tile = (32 TW) tokens of ONE image x TN = 32 NB channels (whole heads), TW x KW = 4: TW waves own 32 tokens
//     each, KW waves split the contraction (small layers: more workgroups; the partial accumulators meet in LDS).
//   * Q^T tile = W_tile X_tile^T on v_mfma_f32_32x32x16: A = W rows fed in swap23 order, B = the lane's own token row -- so register r of
//     accumulator block nb is channel 32 nb + 16 (r >> 3) + 8 hi + (r & 7) of token (lane & 31): 8 consecutive registers = 8 consecutive
//     channels = (after rounding to T) one 16-byte piece of the Q row AND one B-operand fragment of the score MFMA (pww_tile.h).
//   * both operands stream global -> registers (buffer loads, a ring of P k-steps in flight; rows past N read the image's last row); K of the tile's heads sits in LDS (requested first, parked while the ring fills).
//   * epilogue: accumulators -> T -> LDS tile; the workgroup stores the tile with coalesced 16-byte row pieces, and the (token wave,
//     head) units are dealt to the 4 waves: score blocks S^T = K_h Q^T on the MFMA, masked (rows >= N, keys >= M), reduced per lane,
//     per wave -> ONE fp64 partial per WAVE, written before the tile's stores go out (no barrier, nothing in the kernel waits for a
//     store). Statistics are those of the ROUNDED Q the attention kernel reads.
#include <string.h>
#include <type_traits>
#include "pww_attn_core.h"

namespace pww {

struct QprojParams {
    const void *x, *w, *k;
    void *q;
    const float *gate;        // [B] or null: images with gate 0 (unconditional rows of a CFG-folded batch) get Q but no statistic
    double *partials;         // [B][nparts][4] = { max, min, sum, sum of squares }; rows of gated-out images are not written
    int B, N, Cin, C, D, M;
    long x_sb, x_sn, q_sb, q_sn, k_sb, k_sm;     // elements
    int ntile, ncg, nparts;   // token tiles per image, channel groups, partials per image (= 4 ntile ncg: one per wave)
    int fields;               // which of the four fields anybody will read: bit 0 max, 1 min, 2 sum, 3 sum of squares
    unsigned long long *timeline;     // debug: per-workgroup phase time stamps (pww_debug_timeline), normally null
    unsigned timeline_wgs;
};

__device__ __forceinline__ void qp_stamp(const QprojParams &p, int slot) {
    if (p.timeline && threadIdx.x == 0 && blockIdx.x < p.timeline_wgs) p.timeline[(long)blockIdx.x * TL_SLOTS + slot] = wall_clock64();
}

constexpr int QP_MAX_KEYS = 128;
constexpr int QP_KC = 64;             // contraction elements per staged chunk: 128 bytes = one full cache line per operand row
constexpr int QP_SROW = QP_KC * 2 + 16;     // LDS row of a staged chunk: +16 bytes => 16 rows read at one column hit 16 different 4-bank groups

template <int N, typename F> __device__ __forceinline__ void static_for(F &&f) {
    if constexpr (N > 0) { static_for<N - 1>(f); f(std::integral_constant<int, N - 1>{}); }
}

// One (32 tokens x one head) unit of the statistic: score blocks S^T[32 keys][32 tokens] = K_h Q^T for NKB key blocks on the MFMA (Q
// fragments from the LDS tile, K rows from the K tile: the operands of pww_tile.h's score_tile), masked and reduced into the lane's
// running extremes and fp64 sums. Fragments of slice ks + 1 are requested before the MFMAs of slice ks. Rows of the K tile past M hold
// whatever LDS holds: their scores are masked.
// KSC > 0: the head spans exactly KSC 16-channel slices (a compile-time constant: d = 40 -> 3, 64 -> 4, 80 -> 5, 160 -> 10): every
// fragment read of the unit is issued before its first MFMA (one LDS latency per unit instead of one per slice); KSC == 0: run-time loop.
template <typename T, int NKB, int ROWB, int KSC>
__device__ __forceinline__ void qp_stat_unit(const char *qp, const char *kp, int hh, int D, const float (&neg)[16], int hi, bool rvalid, int fields,
                                             float &vmax, float &vmin, double &dsum, double &dsq) {
    typedef typename Vec<T>::v8 V8;
    const int ks_lo = (hh * D) >> 4, ks_hi = ((hh + 1) * D - 1) >> 4;
    const bool ragged = (D & 15) != 0;         // head boundaries inside a 16-channel slice (d = 40): the foreign half of a boundary slice is zeroed
    const int c_lo = hh * D, c_hi = c_lo + D;  // the head's channels within the tile
    auto q_frag = [&](int ks) {
        V8 f = *reinterpret_cast<const V8 *>(qp + ks * 32);
        const int c0 = ks * 16 + hi * 8;
        if (ragged && (c0 < c_lo || c0 >= c_hi)) f = zero8<V8>();
        return f;
    };
    f32x16 s[NKB];
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) s[kb][r] = 0.f;
    if constexpr (KSC > 0) {
        // groups of at most 5 slices (80 fragment registers: a head of 160 channels takes two groups) -- the statistic must not be what
        // sets the kernel's register count (two waves per SIMD for the 128 x 160 tile)
        constexpr int GRP = KSC <= 5 ? KSC : 5;
        static_assert(KSC % GRP == 0, "slice groups");
#pragma unroll
        for (int g0 = 0; g0 < KSC; g0 += GRP) {
            V8 qa[GRP], ka[GRP][NKB];
#pragma unroll
            for (int i = 0; i < GRP; ++i) {
                qa[i] = q_frag(ks_lo + g0 + i);
#pragma unroll
                for (int kb = 0; kb < NKB; ++kb) ka[i][kb] = *reinterpret_cast<const V8 *>(kp + kb * 32 * ROWB + (ks_lo + g0 + i) * 32);
            }
#pragma unroll
            for (int i = 0; i < GRP; ++i)
#pragma unroll
                for (int kb = 0; kb < NKB; ++kb) s[kb] = mfma32(ka[i][kb], qa[i], s[kb]);
        }
    } else {
    V8 qf = q_frag(ks_lo), kf[NKB];
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) kf[kb] = *reinterpret_cast<const V8 *>(kp + kb * 32 * ROWB + ks_lo * 32);
    for (int ks = ks_lo; ks <= ks_hi; ++ks) {
        const int kn = ks < ks_hi ? ks + 1 : ks;
        const V8 qn = q_frag(kn);
        V8 kfn[NKB];
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) kfn[kb] = *reinterpret_cast<const V8 *>(kp + kb * 32 * ROWB + kn * 32);
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) s[kb] = mfma32(kf[kb], qf, s[kb]);
        qf = qn;
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) kf[kb] = kfn[kb];
    }
    }
    // one pass per field that anybody will read, each behind a wave-uniform branch (the shipped weight functions read ONE: max).
    // Rows of the K tile past M are ZERO (the kernel fills them): their scores are exactly 0 and leave the sums alone; for the
    // extremes the lane adds neg[r] (0 for a live key of the last block, -inf for a padding key) -- no compares, no lane masks here
    float usum = 0.f, usq = 0.f, umax = -INFINITY, umin = INFINITY;
    if (fields & 1) {
#pragma unroll
        for (int kb = 0; kb < NKB - 1; ++kb)
#pragma unroll
            for (int r = 0; r < 16; r += 4) umax = fmaxf(umax, fmaxf(fmaxf(s[kb][r], s[kb][r + 1]), fmaxf(s[kb][r + 2], s[kb][r + 3])));
#pragma unroll
        for (int r = 0; r < 16; ++r) umax = fmaxf(umax, s[NKB - 1][r] + neg[r]);
    }
    if (fields & 2) {
#pragma unroll
        for (int kb = 0; kb < NKB - 1; ++kb)
#pragma unroll
            for (int r = 0; r < 16; r += 4) umin = fminf(umin, fminf(fminf(s[kb][r], s[kb][r + 1]), fminf(s[kb][r + 2], s[kb][r + 3])));
#pragma unroll
        for (int r = 0; r < 16; ++r) umin = fminf(umin, s[NKB - 1][r] - neg[r]);
    }
    if (fields & 4) {
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) usum += s[kb][r];
    }
    if (fields & 8) {
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) usq = fmaf(s[kb][r], s[kb][r], usq);
    }
    if (rvalid) {       // (a token row past N contributes nothing)
        vmax = fmaxf(vmax, umax); vmin = fminf(vmin, umin);
        dsum += (double)usum; dsq += (double)usq;
    }
}

// Tile shape <NB, TW, CW, KW>: the workgroup's 4 waves are TW token waves x CW channel waves x KW contraction waves.
//   tile = (32 TW) tokens x (32 NB) channels; a wave accumulates 32 tokens x (32 NB / CW) channels over 1 / KW of the contraction.
// Operands are STAGED THROUGH LDS with coalesced loads: 8 consecutive threads move the 128 contiguous bytes a chunk holds of one
// operand row (round 4's first version streamed both operands global -> registers in MFMA layout: every wave-load touched 32 cache
// lines for 32 bytes each and the same lines were requested by 4 consecutive k-steps -- 24 us for the B = 2 layers against 8.5 us of
// the stock GEMM, profiles/r04_qproj_v1_register_direct.md). Chunk c + 1 is requested into registers before chunk c is computed and
// parked after it (one staging buffer, two barriers per chunk): the K tile, the staging buffer and the Q tile share 160 KB.
template <typename T, int NB, int TW, int CW, int KW, int NSETS>
__global__ void __launch_bounds__(256, 1) qproj_stat_kernel(const QprojParams p) {
    typedef typename Vec<T>::v8 V8;
    static_assert(TW * CW * KW == 4 && NB % CW == 0, "four waves per workgroup");
    constexpr int TN = NB * 32, TM = TW * 32, NBW = NB / CW;
    constexpr int ROWB = TN * 2 + 16;                 // LDS row of TN channels (K tile, Q tile): same bank argument as QP_SROW
    constexpr int CPR = TN / 8;                       // 16-byte chunks per such row
    constexpr int RED_BYTES = NBW * 16 * 64 * 4;      // one wave's fp32 accumulators
    constexpr int NRED = KW == 1 ? 0 : (KW == 2 ? TW * CW : 2);
    constexpr int KCH = (QP_MAX_KEYS * CPR + 255) / 256;
    constexpr int PLANE = (TN + TM) * QP_SROW;        // one contraction wave's staged chunk: TN rows of W, then TM rows of X
    constexpr int WSL = TN / 32, XSL = TW;            // 32-row slabs per plane (one slab = one 16-byte piece per thread)
    constexpr int WPT = KW * WSL, XPT = KW * XSL;     // pieces per thread and chunk
    constexpr int STAGE_BYTES = KW * PLANE;

    extern __shared__ __attribute__((aligned(16))) char smem[];      // [K tile: whole 32-key blocks][work: staged chunk / reduction buffers / Q tile]
    const int kt_rows = ((p.M + 31) >> 5) << 5;       // whole 32-key blocks: the rows past M are zero-filled (scores of exactly 0)
    const int kt_bytes = kt_rows * ROWB;
    char *Kt = smem;
    char *work = smem + kt_bytes;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);       // (a scalar: everything derived from it -- tile roles, loop bounds of the statistic -- stays in SGPRs)
    const int hi = lane >> 5, l31 = lane & 31;
    const int tw = wave % TW, cw = (wave / TW) % CW, kw = wave / (TW * CW);
    qp_stamp(p, 0);
    const unsigned long long tl_c0 = p.timeline ? clock64() : 0ull;

    // workgroup -> (image, token tile, channel group): XCD x (= blockIdx & 7) always works on channel group x % ncg, so an XCD's L2
    // holds ONE slice of W_q (<= 820 KB) for the whole launch
    int cg, t;
    if ((8 % p.ncg) == 0) {
        const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3, per = 8 / p.ncg;
        cg = xcd % p.ncg;
        t = j * per + xcd / p.ncg;
    } else {
        cg = blockIdx.x % p.ncg;
        t = blockIdx.x / p.ncg;
    }
    if (t >= p.B * p.ntile) return;          // (whole workgroup: no barrier is skipped by a part of it)
    const int b = t / p.ntile, tile = t - b * p.ntile;
    const int row0 = tile * TM;              // first token of the tile

    const T *Xb = reinterpret_cast<const T *>(p.x) + b * p.x_sb;
    const T *Wg = reinterpret_cast<const T *>(p.w) + (long)cg * TN * p.Cin;
    const T *Kb = reinterpret_cast<const T *>(p.k) + b * p.k_sb + cg * TN;
    T *Qb = reinterpret_cast<T *>(p.q) + b * p.q_sb + cg * TN;
    const float gate = p.gate ? p.gate[b] : 1.f;          // (requested early, looked at in the epilogue)

    // exact descriptors, every variable part of an address in the VECTOR offset: anything past the end (token rows >= N, the chunk
    // after the last one) is out of range by the vector offset alone and returns zeros without touching memory
    const auto srd_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<T *>(Xb), 0, (unsigned)((((long)p.N - 1) * p.x_sn + p.Cin) * 2), 0x00020000);
    const auto srd_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<T *>(Wg), 0, (unsigned)((long)TN * p.Cin * 2), 0x00020000);
    const auto srd_k = __builtin_amdgcn_make_buffer_rsrc(const_cast<T *>(Kb), 0, (unsigned)((((long)p.M - 1) * p.k_sm + TN) * 2), 0x00020000);

    // ---- K of the tile's heads: requested FIRST (its data returns first: vmcnt retires in order), parked in LDS while chunk 0 arrives
    u32x4 kreg[KCH];
    const int nkchunk = p.M * CPR;
#pragma unroll
    for (int i = 0; i < KCH; ++i) {
        const int c = tid + i * 256, row = c / CPR, ch = c - row * CPR;
        kreg[i] = __builtin_amdgcn_raw_buffer_load_b128(srd_k, c < nkchunk ? (unsigned)((row * p.k_sm + ch * 8) * 2) : OOB_OFF, 0, 0);
    }

    // ---- staging plan of this thread: piece = 16 bytes; slab s of a plane = rows 32 s .. 32 s + 31, thread -> (row tid >> 3, piece tid & 7)
    const int nch = p.Cin / (QP_KC * KW);                       // chunks per contraction wave (the host checks divisibility)
    const unsigned kspan = (unsigned)(p.Cin / KW) * 2u;         // bytes of a row one contraction wave covers
    const int prow = tid >> 3, pcol = (tid & 7) * 16;
    const unsigned wv0 = (unsigned)(prow * p.Cin * 2 + pcol), wslab = (unsigned)(32 * p.Cin * 2);
    unsigned xv[XSL];
#pragma unroll
    for (int j = 0; j < XSL; ++j) {
        const int n = row0 + j * 32 + prow;
        xv[j] = n < p.N ? (unsigned)((long)n * p.x_sn * 2) + (unsigned)pcol : OOB_OFF;
    }
    char *park_base = work + prow * QP_SROW + pcol;
    // A RING of NSETS register sets: chunks c + 1 ... c + NSETS are in flight while chunk c is computed (one chunk of look-ahead left
    // every chunk's load latency exposed: 20 MFMAs per wave and chunk are 0.3 us, the X rows come from HBM: 1.5 - 2 us under load)
    u32x4 wreg[NSETS][WPT], xreg[NSETS][XPT];
    auto request = [&](auto set, int c) {      // chunk c of every contraction wave -> register set (c >= nch: out of range, zeros, no traffic)
        constexpr int S = decltype(set)::value;
        const unsigned koff = c < nch ? (unsigned)c * (QP_KC * 2) : OOB_OFF;
#pragma unroll
        for (int i = 0; i < WPT; ++i)
            wreg[S][i] = __builtin_amdgcn_raw_buffer_load_b128(srd_w, wv0 + (unsigned)(i % WSL) * wslab + (unsigned)(i / WSL) * kspan + koff, 0, 0);
#pragma unroll
        for (int j = 0; j < XPT; ++j)
            xreg[S][j] = __builtin_amdgcn_raw_buffer_load_b128(srd_x, (xv[j % XSL] == OOB_OFF || koff == OOB_OFF) ? OOB_OFF : xv[j % XSL] + (unsigned)(j / XSL) * kspan + koff, 0, 0);
    };
    auto park = [&](auto set) {
        constexpr int S = decltype(set)::value;
#pragma unroll
        for (int i = 0; i < WPT; ++i) *reinterpret_cast<u32x4 *>(park_base + (i / WSL) * PLANE + (i % WSL) * 32 * QP_SROW) = wreg[S][i];
#pragma unroll
        for (int j = 0; j < XPT; ++j) *reinterpret_cast<u32x4 *>(park_base + (j / XSL) * PLANE + (TN + (j % XSL) * 32) * QP_SROW) = xreg[S][j];
    };
    typedef std::integral_constant<int, 0> Set0;
    request(Set0{}, 0);
    {   // park K (waits for the K loads only: chunk 0's loads were issued after them); its registers are free before the second set fills
#pragma unroll
        for (int i = 0; i < KCH; ++i) {
            const int c = tid + i * 256;
            if (c < nkchunk) { const int row = c / CPR, ch = c - row * CPR; *reinterpret_cast<u32x4 *>(Kt + row * ROWB + ch * 16) = kreg[i]; }
        }
    }
    for (int c = tid; c < (kt_rows - p.M) * CPR; c += 256) {
        const int row = p.M + c / CPR, ch = c % CPR;
        *reinterpret_cast<u32x4 *>(Kt + row * ROWB + ch * 16) = u32x4{0u, 0u, 0u, 0u};
    }
    static_for<NSETS - 1>([&](auto i) { request(std::integral_constant<int, decltype(i)::value + 1>{}, decltype(i)::value + 1); });
    park(Set0{});
    request(Set0{}, NSETS);
    __syncthreads();
    qp_stamp(p, 1);

    // ---- main loop: acc[nb] (32 channels x 32 tokens) += W[channels][64 k] X[tokens][64 k]^T per chunk, operands read from LDS in MFMA layout
    f32x16 acc[NBW];
#pragma unroll
    for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nb][r] = 0.f;
    const char *a_base = work + kw * PLANE + (cw * NBW * 32 + swap23(l31)) * QP_SROW + hi * 16;
    const char *b_base = work + kw * PLANE + (TN + tw * 32 + l31) * QP_SROW + hi * 16;
    auto chunk_step = [&](auto next_set, int c) {      // compute chunk c (staged), park chunk c + 1 (in `next_set`), request chunk c + 1 + NSETS into that set
        __builtin_amdgcn_sched_barrier(0);
        // operand fragments one k-step ahead of the MFMAs that use them
        V8 af[2][NBW], xf[2];
        auto frags = [&](int ks) {
            xf[ks & 1] = *reinterpret_cast<const V8 *>(b_base + ks * 32);
#pragma unroll
            for (int nb = 0; nb < NBW; ++nb) af[ks & 1][nb] = *reinterpret_cast<const V8 *>(a_base + nb * 32 * QP_SROW + ks * 32);
        };
        frags(0);
#pragma unroll
        for (int ks = 0; ks < QP_KC / 16; ++ks) {
            if (ks + 1 < QP_KC / 16) frags(ks + 1);
#pragma unroll
            for (int nb = 0; nb < NBW; ++nb) acc[nb] = mfma32(af[ks & 1][nb], xf[ks & 1], acc[nb]);
        }
        // issue order of the chunk (hipcc's own choice behind a scheduling barrier is read -> wait -> MFMA, one LDS latency per
        // MFMA): the first k-step's NBW + 1 fragment reads, then one read of the NEXT k-step behind every MFMA, then the last MFMAs
        __builtin_amdgcn_sched_group_barrier(0x100, NBW + 1, 0);
#pragma unroll
        for (int i = 0; i < 3 * (NBW + 1); ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, NBW - 3, 0);
        __syncthreads();              // every wave is done with the staged chunk
        if (c + 1 < nch) {
            park(next_set);
            request(next_set, c + 1 + NSETS);
        }
        __syncthreads();
    };
    for (int c0 = 0; c0 < nch; c0 += NSETS)
        static_for<NSETS>([&](auto i) {
            constexpr int I = decltype(i)::value;
            if (c0 + I < nch) chunk_step(std::integral_constant<int, (I + 1) % NSETS>{}, c0 + I);       // (workgroup-uniform)
        });
    qp_stamp(p, 2);

    // ---- the contraction split over KW waves: partial accumulators meet in LDS (fp32, lane-contiguous 16-byte pieces: conflict-free)
    auto red_store = [&](char *buf) {
#pragma unroll
        for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4)
                *reinterpret_cast<f32x4 *>(buf + ((nb * 4 + r4) * 64 + lane) * 16) = f32x4{acc[nb][r4 * 4], acc[nb][r4 * 4 + 1], acc[nb][r4 * 4 + 2], acc[nb][r4 * 4 + 3]};
    };
    auto red_add = [&](const char *buf) {
#pragma unroll
        for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const f32x4 v = *reinterpret_cast<const f32x4 *>(buf + ((nb * 4 + r4) * 64 + lane) * 16);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[nb][r4 * 4 + j] += v[j];
            }
    };
    if constexpr (KW == 2) {
        const int rid = cw * TW + tw;
        if (kw == 1) red_store(work + rid * RED_BYTES);
        __syncthreads();
        if (kw == 0) red_add(work + rid * RED_BYTES);
        __syncthreads();                     // (the Q tile below aliases the reduction buffers)
    } else if constexpr (KW == 4) {
        if (kw >= 2) red_store(work + (kw - 2) * RED_BYTES);
        __syncthreads();
        if (kw < 2) red_add(work + kw * RED_BYTES);
        __syncthreads();
        if (kw == 1) red_store(work);
        __syncthreads();
        if (kw == 0) red_add(work);
        __syncthreads();
    }

    // ---- Q tile, rounded to T, into LDS: row = token, 16-byte piece (nb, half, hi) = channels 32 nb + 16 half + 8 hi .. + 7
    char *Qt = work;
    if (kw == 0) {
        char *qrow = Qt + (tw * 32 + l31) * ROWB + cw * NBW * 64 + hi * 16;
#pragma unroll
        for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                V8 v;
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = (T)acc[nb][half * 8 + j];
                *reinterpret_cast<V8 *>(qrow + nb * 64 + half * 32) = v;
            }
    }
    __syncthreads();
    qp_stamp(p, 3);

    // ---- (a) statistic partial of the tile: ONE partial PER WAVE (no barrier, no fold across the waves: the attention kernel folds
    // all partials of the image anyway), written before the tile's stores go out -- nothing in this kernel ever waits for a store
    if (p.fields != 0 && gate != 0.f) {                   // workgroup-uniform
        const int HT = TN / p.D;                          // heads in the tile
        const int nkb = (p.M + 31) >> 5;
        float vmax = -INFINITY, vmin = INFINITY;
        double dsum = 0.0, dsq = 0.0;
        const int krow = swap23(l31);
        float neg[16];            // last key block: 0 for the lane's live keys, -inf for its padding keys (register r = key 16 (r >> 3) + 8 hi + (r & 7))
#pragma unroll
        for (int r = 0; r < 16; ++r) neg[r] = key_of(nkb - 1, r, hi) < p.M ? 0.f : -INFINITY;
        for (int u = wave; u < TW * HT; u += 4) {
            const int tu = u % TW, hh = u / TW;
            const bool rvalid = row0 + tu * 32 + l31 < p.N;
            const char *qp = Qt + (tu * 32 + l31) * ROWB + hi * 16;
            const char *kp = Kt + krow * ROWB + hi * 16;
            // (key-block count and, for the 77-token prompt, the slice count are compile-time constants inside: straight-line MFMAs)
            const int nks = (((hh + 1) * p.D - 1) >> 4) - ((hh * p.D) >> 4) + 1;
#define PWW_STAT_UNIT(NKB, KSC) qp_stat_unit<T, NKB, ROWB, KSC>(qp, kp, hh, p.D, neg, hi, rvalid, p.fields, vmax, vmin, dsum, dsq)
            if (nkb == 3 && nks == 3) PWW_STAT_UNIT(3, 3);
            else if (nkb == 3 && nks == 4) PWW_STAT_UNIT(3, 4);
            else if (nkb == 3 && nks == 5) PWW_STAT_UNIT(3, 5);
            else if (nkb == 3 && nks == 10) PWW_STAT_UNIT(3, 10);
            else if (nkb == 1) PWW_STAT_UNIT(1, 0);
            else if (nkb == 2) PWW_STAT_UNIT(2, 0);
            else if (nkb == 3) PWW_STAT_UNIT(3, 0);
            else PWW_STAT_UNIT(4, 0);
#undef PWW_STAT_UNIT
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            vmax = fmaxf(vmax, __shfl_xor(vmax, off));
            vmin = fminf(vmin, __shfl_xor(vmin, off));
            dsum += __shfl_xor(dsum, off);
            dsq += __shfl_xor(dsq, off);
        }
        if (lane == 0) {      // (a wave without a unit writes the neutral element)
            double *out = p.partials + ((long)b * p.nparts + ((long)tile * p.ncg + cg) * 4 + wave) * 4;
            out[0] = (double)vmax; out[1] = (double)vmin; out[2] = dsum; out[3] = dsq;
        }
    }
    qp_stamp(p, 4);

    // ---- (b) the workgroup writes its tile: 16-byte pieces, consecutive threads = consecutive pieces of a row
    for (int c = tid; c < TM * CPR; c += 256) {
        const int row = c / CPR, ch = c - row * CPR;
        if (row0 + row < p.N)
            *reinterpret_cast<u32x4 *>(Qb + (long)(row0 + row) * p.q_sn + ch * 8) = *reinterpret_cast<const u32x4 *>(Qt + row * ROWB + ch * 16);
    }

    qp_stamp(p, 5);
    if (p.timeline && threadIdx.x == 0 && blockIdx.x < p.timeline_wgs) p.timeline[(long)blockIdx.x * TL_SLOTS + 7] = clock64() - tl_c0;
}

// ---- host side ----------------------------------------------------------------------------------------------------------------

struct QprojPlan { int nb, tw, cw, kw; };
constexpr size_t QP_LDS_LIMIT = 160 * 1024 - 1024;

static size_t qproj_lds(const QprojPlan &c, int M) {
    const size_t tn = c.nb * 32, tm = c.tw * 32, rowb = tn * 2 + 16, red = (size_t)(c.nb / c.cw) * 16 * 64 * 4;
    const size_t nred = c.kw == 1 ? 0 : (c.kw == 2 ? c.tw * c.cw : 2);
    size_t work = (size_t)c.kw * (tn + tm) * QP_SROW;
    if (nred * red > work) work = nred * red;
    if (tm * rowb > work) work = tm * rowb;
    return (size_t)(((M + 31) >> 5) << 5) * rowb + work + 16 * sizeof(double);
}

// Tile shape for a problem: the largest tile that still gives the chip ~180 workgroups, else the one with the most workgroups.
// TN = 32 nb channels needs C % TN == 0 and whole heads per tile (TN % D == 0); the contraction must split into 64-element chunks
// per contraction wave (Cin % (64 kw) == 0).
static bool qproj_plan(const pww_qproj_desc_t *d, QprojPlan *out) {
    const int C = d->H * d->D;
    // 160-channel tiles where the heads allow it (d = 40 / 80 / 160: 69 - 90 KB of LDS, two workgroups per CU hide each other's
    // latencies), 320-channel tiles otherwise (d = 64); within a family: the largest tile that fills the chip
    // (a 128 x 320 tile -- <10, 4, 1, 1> -- needs more than 512 registers with two operand sets in flight: not instantiated)
    const QprojPlan cand[] = {{5, 4, 1, 1}, {5, 2, 1, 2}, {5, 1, 1, 4}, {10, 2, 2, 1}, {10, 1, 2, 2}};
    long best_wg = 0;
    bool found = false;
    for (const QprojPlan &c : cand) {
        const int tn = c.nb * 32, tm = c.tw * 32;
        if (C % tn || tn % d->D || d->Cin % (QP_KC * c.kw) || qproj_lds(c, d->M) > QP_LDS_LIMIT) continue;
        const long wgs = (long)d->B * ((d->N + tm - 1) / tm) * (C / tn);
        if (wgs >= 180) { *out = c; return true; }
        if (wgs > best_wg) { best_wg = wgs; *out = c; found = true; }
    }
    return found;
}

int qproj_parts(const pww_qproj_desc_t *d) {
    QprojPlan pl;
    if (!d || d->B <= 0 || d->N <= 0 || d->H <= 0 || d->D <= 0 || d->Cin <= 0 || d->D % 8 || d->M <= 0 || d->M > QP_MAX_KEYS || !qproj_plan(d, &pl)) return 0;
    return ((d->N + pl.tw * 32 - 1) / (pl.tw * 32)) * (d->H * d->D / (pl.nb * 32)) * 4;      // one partial per wave of every tile
}

template <typename T, int NB, int TW, int CW, int KW, int NSETS>
static int launch_qproj(const QprojParams &p, hipStream_t stream) {
    constexpr int TN = NB * 32, TM = TW * 32, ROWB = TN * 2 + 16, RED_BYTES = (NB / CW) * 16 * 64 * 4;
    constexpr int NRED = KW == 1 ? 0 : (KW == 2 ? TW * CW : 2);
    constexpr size_t stage = (size_t)KW * (TN + TM) * QP_SROW;
    constexpr size_t work_a = stage > (size_t)NRED * RED_BYTES ? stage : (size_t)NRED * RED_BYTES;
    constexpr size_t work = work_a > (size_t)TM * ROWB ? work_a : (size_t)TM * ROWB;
    const size_t lds = (size_t)(((p.M + 31) >> 5) << 5) * ROWB + work + 16 * sizeof(double);
    if (lds > QP_LDS_LIMIT) { set_error("qproj_stat: internal error: %zu bytes of LDS", lds); return PWW_EINVAL; }
    auto kern = qproj_stat_kernel<T, NB, TW, CW, KW, NSETS>;
    static thread_local size_t lds_attr[8] = {0, 0, 0, 0, 0, 0, 0, 0};      // per device (hipFuncSetAttribute is per device)
    int dev = 0;
    if (check_hip(hipGetDevice(&dev), "hipGetDevice")) return PWW_EHIP;
    if (lds > 64 * 1024 && (dev >= 8 || lds > lds_attr[dev])) {
        if (check_hip(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds), "hipFuncSetAttribute"))
            return PWW_EHIP;
        if (dev < 8) lds_attr[dev] = lds;
    }
    const long tiles = (long)p.B * p.ntile;
    long grid;
    if (8 % p.ncg == 0) { const int per = 8 / p.ncg; grid = ((tiles + per - 1) / per) * 8; }
    else grid = tiles * p.ncg;
    launch_attn_kernel(kern, dim3((unsigned)grid), dim3(256), lds, stream, p);
    return check_hip(hipGetLastError(), "qproj_stat_kernel launch");
}

int qproj_stat(const void *x, const void *w, void *q, const void *k, const float *gate, const pww_qproj_desc_t *d, int stat_kind,
               double *partials, size_t partials_bytes, hipStream_t stream) {
    if (!x || !w || !q || !k || !d) { set_error("qproj_stat: null argument"); return PWW_EINVAL; }
    if (!arch_ok()) return PWW_ENOTSUP;
    if (d->dtype != PWW_DTYPE_F16 && d->dtype != PWW_DTYPE_BF16) { set_error("qproj_stat: dtype %d unsupported", d->dtype); return PWW_ENOTSUP; }
    if (d->B <= 0 || d->N <= 0 || d->H <= 0 || d->D <= 0 || d->M <= 0 || d->Cin <= 0) { set_error("qproj_stat: empty problem"); return PWW_EINVAL; }
    if (d->D % 8 || d->M > QP_MAX_KEYS) { set_error("qproj_stat: head dim %d must be a multiple of 8 and M = %d at most %d", d->D, d->M, QP_MAX_KEYS); return PWW_ENOTSUP; }
    QprojPlan pl;
    if (!qproj_plan(d, &pl)) {
        set_error("qproj_stat: no tile shape for C = %d (a multiple of 160 or 320 holding whole heads of %d) and Cin = %d (a multiple of 64)", d->H * d->D, d->D, d->Cin);
        return PWW_ENOTSUP;
    }
    const auto mis = [](const void *ptr) { return (reinterpret_cast<uintptr_t>(ptr) & 15) != 0; };
    if (mis(x) || mis(w) || mis(q) || mis(k) || d->x_stride[0] % 8 || d->x_stride[1] % 8 || d->q_stride[0] % 8 || d->q_stride[1] % 8 || d->k_stride[0] % 8 ||
        d->k_stride[1] % 8) {
        set_error("qproj_stat: pointers and strides must keep every row 16-byte aligned");
        return PWW_EINVAL;
    }
    const int C = d->H * d->D;
    if (d->x_stride[1] < d->Cin || d->q_stride[1] < C || d->k_stride[1] < C) { set_error("qproj_stat: row strides shorter than the rows"); return PWW_EINVAL; }
    if (((long)d->N * d->x_stride[1] + d->Cin) * 2 >= (1L << 31) || (long)C * d->Cin * 2 >= (1L << 31)) { set_error("qproj_stat: an image's activations exceed 2 GiB"); return PWW_ENOTSUP; }
    QprojParams p;
    p.x = x; p.w = w; p.k = k; p.q = q; p.gate = gate; p.partials = partials;
    p.B = d->B; p.N = d->N; p.Cin = d->Cin; p.C = C; p.D = d->D; p.M = d->M;
    p.x_sb = d->x_stride[0]; p.x_sn = d->x_stride[1]; p.q_sb = d->q_stride[0]; p.q_sn = d->q_stride[1];
    p.k_sb = d->k_stride[0]; p.k_sm = d->k_stride[1];
    p.timeline = debug_timeline();
    p.timeline_wgs = (unsigned)(debug_timeline_bytes() / (TL_SLOTS * sizeof(unsigned long long)));
    p.ntile = (d->N + pl.tw * 32 - 1) / (pl.tw * 32);
    p.ncg = C / (pl.nb * 32);
    p.nparts = p.ntile * p.ncg * 4;
    switch (stat_kind) {
        case PWW_STAT_NONE: p.fields = 0; break;
        case PWW_STAT_MAX: p.fields = 1; break;
        case PWW_STAT_MIN: p.fields = 2; break;
        case PWW_STAT_ABSMAX: p.fields = 3; break;
        case PWW_STAT_MEAN: p.fields = 4; break;
        case PWW_STAT_STD: p.fields = 12; break;
        case PWW_STAT_ALL: p.fields = 15; break;
        default: set_error("qproj_stat: bad statistic selector %d", stat_kind); return PWW_EINVAL;
    }
    if (p.fields) {
        if (!partials || partials_bytes < (size_t)d->B * p.nparts * 4 * sizeof(double) || (reinterpret_cast<uintptr_t>(partials) & 7)) {
            set_error("qproj_stat: partials buffer missing, misaligned or too small (need %zu bytes)", (size_t)d->B * p.nparts * 4 * sizeof(double));
            return PWW_EINVAL;
        }
    }
    // operand register sets in flight: as many as the register file holds beside the accumulators; the 128 x 160 tile keeps two when the
    // launch has several workgroups per CU anyway (two sets = 2 waves per SIMD = two resident workgroups hiding each other's latencies)
    const bool many = (long)p.B * p.ntile * p.ncg >= 512;
#define PWW_QP(T)                                                                                            \
    if (pl.nb == 10 && pl.tw == 2) return launch_qproj<T, 10, 2, 2, 1, 3>(p, stream);                        \
    if (pl.nb == 10 && pl.tw == 1) return launch_qproj<T, 10, 1, 2, 2, 2>(p, stream);                        \
    if (pl.nb == 5 && pl.tw == 4) return many ? launch_qproj<T, 5, 4, 1, 1, 2>(p, stream) : launch_qproj<T, 5, 4, 1, 1, 4>(p, stream);   \
    if (pl.nb == 5 && pl.tw == 2) return launch_qproj<T, 5, 2, 1, 2, 3>(p, stream);                          \
    return launch_qproj<T, 5, 1, 1, 4, 2>(p, stream);
    if (d->dtype == PWW_DTYPE_F16) { PWW_QP(f16) }
    PWW_QP(bf16)
#undef PWW_QP
}

}  // namespace pww
