// The product path's cross-attention launches since round 5: two SMALL kernels with short dependent chains.
//
//   qk_parts_kernel     partials of the per-image score statistic over a FINISHED Q (paint_with_words/paint_with_words.py:87 + the global
//                       reduction weight_function applies to `qk`: :402-405 qk.max(), runner.py:104, README.md:152 qk.std()) -- for the
//                       layers whose to_q stays the stock GEMM (C = 1280 in SD1.5 / SD2.1: pww_qproj.hip meets too few workgroups there).
//                       One wave per (image, head, 32-row block, 32-key block): 2 KS 16-byte loads per lane, KS MFMAs, one fp64 partial.
//                       No LDS, no barrier, no atomics: load -> MFMA -> reduce -> store.
//   cross_lean_kernel   O = softmax((Q K^T + c[b] w) scale) V over the prompt tokens (:106-116) with c[b] = c0 * stat(fold(partials[b])) * gate[b]:
//                       the pass-2-only form of pww_cross.hip (partials folded at entry, the kernel boundary is the synchronisation), ONE query
//                       block per workgroup.
//
// Why a second kernel instead of the general one (pww_cross.hip, cross_fused_kernel: pass 1, in-kernel hand-off, several blocks per
// workgroup, compact bias, four pass-2 forms): that kernel is 56 - 82 KB of code per instantiation and 310 of its 340 MFMAs (d = 160) sat
// directly behind an `s_waitcnt lgkmcnt(0)` -- at the 2 folded rows of a batch-1 request every launch is ONE dependent chain per wave
// (entry -> loads -> LDS -> scores -> softmax -> PV -> store, 1 wave per SIMD, nothing to overlap with), and the chain paid an LDS zero-fill
// pass with its own barrier, an fp64 fold through LDS with two more barriers and an exposed LDS latency per MFMA: 12.6 us for 1.3 MB of
// traffic (N = 256). Here: every global load of the workgroup is issued before anything is waited for; K / V padding arrives as
// out-of-range zeros instead of a fill pass; every wave folds the partials itself (shuffles only); bias rows are wave-private (no barrier);
// K fragments are requested ahead of the MFMAs (pww_tile.h score_tile); ONE barrier in the kernel; ~15 KB of code.
// Large batches (several query blocks per workgroup pay off: configs 3 / 4) and everything this kernel does not take (compact bias, maps
// wider than 64 columns, M < 64, > 256 + tail partials) keep the general kernel -- same arithmetic, same results to the last few bits.
#include <string.h>
#include "pww_attn_core.h"
#include "pww_cross_tile.h"

namespace pww {

// ------------------------------------------------------------------------------------------------------------------------------------
// qk_parts: statistic partials over a finished Q
// ------------------------------------------------------------------------------------------------------------------------------------
// Kernel arguments in two parts (round 6). HOT: what the wave needs to issue its first loads -- 14 dwords, passed as
// leading scalar arguments so that the dispatcher preloads them into SGPRs (this unit is compiled with -mllvm
// -amdgpu-kernarg-preload-count=16; a by-value struct is never preloaded): the wave does not start with a dependent s_load round trip to a
// cold kernarg segment. COLD (this struct): what is looked at after the MFMAs -- fetched by an s_load that has the loads' latency to land.
struct QkPartsParams {
    const float *gate;        // [B] or null
    double *partials;         // [B][nparts][4]
    int B, H;
    int nparts;               // partials per image = H * nrb * nkb (fine) / H * ceil(nrb / 4) (coarse)
    int n_img;                // images the grid covers (the hinted-in ones first)
    int fields;               // bit 0 max, 1 min, 2 sum, 3 sum of squares
};

// Granularity. FINE: one wave per (head, 32-row block, 32-key block) -- a partial per wave, H * nrb * nkb per image; nothing but
// load -> MFMA -> reduce -> store: for the coarse UNet levels (N <= 576), where the point is to put ~200 independent waves on the chip.
// COARSE (H * nrb * nkb > 256: the consumer folds at most 256 partials from its prologue's load batch, more cost it a round trip per 64):
// one wave per (head, 32-row block) walks the key blocks (the next block's K fragments in flight under the current block's MFMAs), the four
// waves of a workgroup meet in LDS, ONE partial per workgroup: H * ceil(nrb / 4) per image.
// Grid (blockIdx.x: row-block / unit groups, y: head, z: image): no integer division on the way to the first load.
constexpr int QKP_MAX_FINE = 256;

template <typename T, int KS, bool COARSE>
__global__ void __launch_bounds__(256) qk_parts_kernel(const void *q_, const void *k_, int q_sb, int q_sh, int q_sn, int k_sb, int k_sh, int k_sm,
                                                       int N, int M, int D, int blocks, const QkPartsParams p) {
    // (14 dwords is what the dispatcher preloads: 16 user SGPRs less the kernarg pointer) blocks = 32-row blocks << 3 | 32-key blocks (<= 4)
    const int nrb = blocks >> 3, nkb = blocks & 7;
    typedef typename Vec<T>::v8 V8;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int hi = lane >> 5, l31 = lane & 31;
    const int b = blockIdx.z, h = blockIdx.y;
    const T *Qp = reinterpret_cast<const T *>(q_) + (long)b * q_sb + (long)h * q_sh;
    const T *Kp = reinterpret_cast<const T *>(k_) + (long)b * k_sb + (long)h * k_sh;
    const auto srd_q = head_srd(Qp, N, q_sn, D);
    const auto srd_k = head_srd(Kp, M, k_sm, D);
    int rb, kb0;
    if constexpr (COARSE) { rb = blockIdx.x * 4 + wave; kb0 = 0; }
    else {
        const int f = blockIdx.x * 4 + wave;          // (row block, key block), key blocks fastest; nkb <= 4: divisions by constants
        rb = nkb == 3 ? f / 3 : nkb == 2 ? f >> 1 : nkb == 4 ? f >> 2 : f;
        kb0 = f - rb * nkb;
        if (rb >= nrb) return;                        // (wave-uniform; the fine form has no barrier)
    }
    const int qrow = rb * 32 + l31;
    const bool rvalid = qrow < N;                     // (COARSE: a row block past the last one only contributes neutral elements)
    const unsigned k_lane = (unsigned)((long)swap23(l31) * k_sm * 2), k_blk = (unsigned)(32 * k_sm * 2);
    // rows past N / keys past M lie beyond the descriptors: zeros, no memory traffic, no compare on the way to the load
    V8 qf[KS], kf[2][KS];
    load_q_frags_buf<T, KS>(qf, srd_q, (unsigned)((long)qrow * q_sn * 2), hi, D);
    load_q_frags_buf<T, KS>(kf[0], srd_k, k_lane + (unsigned)kb0 * k_blk, hi, D);
    const float gate = p.gate ? p.gate[b] : 1.f;      // (the cold arguments: requested behind the loads above, looked at before the store)
    float vmax = -INFINITY, vmin = INFINITY;
    double dsum = 0.0, dsq = 0.0;
    constexpr int NKB = COARSE ? 4 : 1;
#pragma unroll
    for (int i = 0; i < NKB; ++i) {
        const int kb = kb0 + i;
        if (COARSE && i + 1 < NKB) load_q_frags_buf<T, KS>(kf[(i + 1) & 1], srd_k, k_lane + (unsigned)(kb + 1) * k_blk, hi, D);
        if (!COARSE || kb < nkb) {                  // (wave-uniform)
            f32x16 s;
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) s = mfma32(kf[i & 1][ks], qf[ks], s);
            // register r = key kb * 32 + 16 (r >> 3) + 8 hi + (r & 7) of row qrow. Rows past N / keys past M were loaded as zeros: their
            // scores are exactly 0 and leave the sums alone; the extremes take them out with a select.
            float usum = 0.f, usq = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const bool live = rvalid && kb * 32 + 16 * (r >> 3) + 8 * hi + (r & 7) < M;
                const float x = s[r];
                vmax = fmaxf(vmax, live ? x : -INFINITY);
                vmin = fminf(vmin, live ? x : INFINITY);
                usum += x;
                usq = fmaf(x, x, usq);
            }
            dsum += (double)usum;
            dsq += (double)usq;
        }
    }
    if (p.fields & 1) {
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, off));
    }
    if (p.fields & 2) {
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) vmin = fminf(vmin, __shfl_xor(vmin, off));
    }
    if (p.fields & 4) {
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) dsum += __shfl_xor(dsum, off);
    }
    if (p.fields & 8) {
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) dsq += __shfl_xor(dsq, off);
    }
    if constexpr (COARSE) {
        __shared__ double red[4][4];
        if (lane == 0) { red[wave][0] = (double)vmax; red[wave][1] = (double)vmin; red[wave][2] = dsum; red[wave][3] = dsq; }
        __syncthreads();
        if (threadIdx.x == 0 && gate != 0.f) {
            double m = red[0][0], n = red[0][1], su = red[0][2], sq = red[0][3];
            for (int w = 1; w < 4; ++w) { m = fmax(m, red[w][0]); n = fmin(n, red[w][1]); su += red[w][2]; sq += red[w][3]; }
            double *out = p.partials + ((long)b * p.nparts + (long)h * gridDim.x + blockIdx.x) * 4;
            out[0] = (p.fields & 1) ? m : -INFINITY;
            out[1] = (p.fields & 2) ? n : INFINITY;
            out[2] = (p.fields & 4) ? su : 0.0;
            out[3] = (p.fields & 8) ? sq : 0.0;
        }
    } else if (lane == 0 && gate != 0.f) {       // (a gated-out image's rows are left untouched, like pww_qproj_stat)
        double *out = p.partials + ((long)b * p.nparts + ((long)h * nrb + rb) * nkb + kb0) * 4;
        out[0] = (p.fields & 1) ? (double)vmax : -INFINITY;
        out[1] = (p.fields & 2) ? (double)vmin : INFINITY;
        out[2] = (p.fields & 4) ? dsum : 0.0;
        out[3] = (p.fields & 8) ? dsq : 0.0;
    }
}

static bool qk_parts_coarse(const pww_attn_desc_t *d) {
    return (long)d->H * ((d->N + 31) / 32) * ((d->M + 31) / 32) > QKP_MAX_FINE;
}

int qk_parts_count(const pww_attn_desc_t *d) {
    if (!d || d->B <= 0 || d->H <= 0 || d->N <= 0 || d->M <= 0 || d->M > 128 || d->D <= 0 || d->D % 8 || d->D > PWW_MAX_HEAD_DIM) return 0;
    const long nrb = (d->N + 31) / 32;
    const long n = qk_parts_coarse(d) ? (long)d->H * ((nrb + 3) / 4) : (long)d->H * nrb * ((d->M + 31) / 32);
    return n > 0x7fffffffL ? 0 : (int)n;
}

static int stat_fields(int stat_kind) {
    switch (stat_kind) {
        case PWW_STAT_NONE: return 0;
        case PWW_STAT_MAX: return 1;
        case PWW_STAT_MIN: return 2;
        case PWW_STAT_ABSMAX: return 3;
        case PWW_STAT_MEAN: return 4;
        case PWW_STAT_STD: return 12;
        case PWW_STAT_ALL: return 15;
        default: return -1;
    }
}

int qk_parts(const void *q, const void *k, const float *gate, const pww_attn_desc_t *d, int stat_kind, int gated_images, double *partials,
             size_t partials_bytes, hipStream_t stream) {
    if (!q || !k || !d || !partials) { set_error("qk_parts: null argument"); return PWW_EINVAL; }
    if (!arch_ok()) return PWW_ENOTSUP;
    if (d->dtype != PWW_DTYPE_F16 && d->dtype != PWW_DTYPE_BF16) { set_error("qk_parts: dtype %d unsupported", d->dtype); return PWW_ENOTSUP; }
    const int nparts = qk_parts_count(d);
    if (nparts <= 0) { set_error("qk_parts: unsupported problem (B=%d H=%d N=%d M=%d D=%d; D a multiple of 8, <= %d; M <= 128)", d->B, d->H, d->N, d->M, d->D, PWW_MAX_HEAD_DIM); return PWW_ENOTSUP; }
    const int fields = stat_fields(stat_kind);
    if (fields <= 0) { set_error("qk_parts: bad statistic selector %d", stat_kind); return PWW_EINVAL; }
    if ((reinterpret_cast<uintptr_t>(q) & 15) || (reinterpret_cast<uintptr_t>(k) & 15) || (reinterpret_cast<uintptr_t>(partials) & 15)) {
        set_error("qk_parts: q, k and partials must be 16-byte aligned");
        return PWW_EINVAL;
    }
    for (int i = 0; i < 3; ++i)
        if (d->q_stride[i] % 8 || d->k_stride[i] % 8) { set_error("qk_parts: strides must be multiples of 8 elements"); return PWW_EINVAL; }
    if (d->q_stride[2] < d->D || d->k_stride[2] < d->D || ((long)d->N * d->q_stride[2] + d->D) * 2 >= (1L << 31) || ((long)d->M * d->k_stride[2] + d->D) * 2 >= (1L << 31)) {
        set_error("qk_parts: one head's Q / K extent must be < 2 GiB and rows must not overlap");
        return PWW_EINVAL;
    }
    if (partials_bytes < (size_t)d->B * nparts * 4 * sizeof(double)) { set_error("qk_parts: partials buffer too small (need %zu bytes)", (size_t)d->B * nparts * 4 * sizeof(double)); return PWW_EINVAL; }
    for (int i = 0; i < 3; ++i)
        if (d->q_stride[i] >= (1L << 31) || d->k_stride[i] >= (1L << 31) || d->q_stride[i] < 0 || d->k_stride[i] < 0) { set_error("qk_parts: strides must be below 2^31 elements"); return PWW_ENOTSUP; }
    QkPartsParams p;
    p.gate = gate; p.partials = partials;
    p.B = d->B; p.H = d->H; p.nparts = nparts;
    const int nrb = (d->N + 31) / 32, nkb = (d->M + 31) / 32;
    p.n_img = (gate && gated_images > 0 && gated_images < d->B) ? gated_images : d->B;
    p.fields = fields;
    if (d->H > 65535 || p.n_img > 65535) { set_error("qk_parts: more than 65535 heads or images"); return PWW_ENOTSUP; }
    const bool coarse = qk_parts_coarse(d);
    const dim3 grid(coarse ? (unsigned)((nrb + 3) / 4) : (unsigned)(((long)nrb * nkb + 3) / 4), (unsigned)d->H, (unsigned)p.n_img);
    // the 14 hot dwords first (preloaded into SGPRs by the dispatcher), the cold struct behind them
#define PWW_QKP_ARGS q, k, (int)d->q_stride[0], (int)d->q_stride[1], (int)d->q_stride[2], (int)d->k_stride[0], (int)d->k_stride[1], (int)d->k_stride[2], \
                     (int)d->N, (int)d->M, (int)d->D, (nrb << 3) | nkb, p
#define PWW_QKP1(T, KSV)                                                                                          \
    do {                                                                                                           \
        if (coarse) launch_timed(qk_parts_kernel<T, KSV, true>, grid, dim3(256), 0, stream, PWW_QKP_ARGS);        \
        else launch_timed(qk_parts_kernel<T, KSV, false>, grid, dim3(256), 0, stream, PWW_QKP_ARGS);              \
    } while (0)
#define PWW_QKP(T)                                      \
    do {                                                \
        const int ks = (d->D + 15) / 16;                \
        if (ks <= 3) PWW_QKP1(T, 3);                    \
        else if (ks == 4) PWW_QKP1(T, 4);               \
        else if (ks == 5) PWW_QKP1(T, 5);               \
        else if (ks == 6) PWW_QKP1(T, 6);               \
        else if (ks <= 8) PWW_QKP1(T, 8);               \
        else PWW_QKP1(T, 10);                           \
    } while (0)
    if (d->dtype == PWW_DTYPE_F16) PWW_QKP(f16); else PWW_QKP(bf16);
#undef PWW_QKP
#undef PWW_QKP1
#undef PWW_QKP_ARGS
    return check_hip(hipGetLastError(), "qk_parts_kernel launch");
}

// ------------------------------------------------------------------------------------------------------------------------------------
// cross_lean: pass-2-only cross-attention, one query block per workgroup
// ------------------------------------------------------------------------------------------------------------------------------------
struct LeanParams {
    AttnParams a;             // a.bias_coeff = the row gate [B] (or null)
    const double *parts;      // [B][nparts][4] or null (stat_kind == NONE)
    int nparts;
    double *stats_out;        // optional [B][4]
    int nqb;                  // query blocks per (image, head)
    int tile_stride;          // floats per LDS tile row (16 / 32 / 64: bias_cols rounded up to a power of two)
    int nchunk;               // MULTI: workgroups per (image, head); workgroup c walks the query blocks c, c + nchunk, ...
    int head_major;           // MULTI: the H heads of one (image, chunk) unit adjacent on one XCD (needs B * nchunk % 8 == 0)
};

constexpr int LEAN_PUNROLL = 4;        // partials per lane folded from the prologue's load batch (256 per image); more take the tail loop
constexpr int LEAN_TILE_LOADS = 8;     // 16-byte pieces per lane of a wave's 32 bias rows: 64 columns at most
constexpr int LEAN_ROWS = 2 * KVBLK;   // key rows the LDS image has room for (one stage of two 64-key tiles)

// LDS image: [K rows 0 .. 127][V rows 0 .. 127][bias tile]: the 64-key tile t of K sits at t * KTile::BYTES, of V at 128 rows of K + t * VTile::BYTES
// (row-linear, so a thread's chunk of pass i is its chunk of pass 0 plus a compile-time constant).
//
// The prologue is written for INSTRUCTION COUNT: at one wave per SIMD a wave issues one instruction every ~5 cycles, and round 5's first
// version of this kernel spent 2.9 us (~1500 instructions of per-chunk index arithmetic, validity compares and exec-mask branches) before
// its last load was issued (profiles/r05_timeline_call2.log). Here a thread moves the SAME 16-byte column of consecutive row groups: one
// offset and one LDS address per operand, computed once; pass i adds a uniform step (global side) and an immediate (LDS side); rows past
// M, the head-dim padding and idle threads are out of range of the buffer descriptor (zeros, no memory traffic) -- no compare, no select,
// no branch per chunk. Grid = (query block, head, image): no integer division either.
//
// MULTI (round 6: the batched launches -- 16 folded rows x 4096 tokens = 4096 query blocks -- which round 5 left to the general kernel of
// pww_cross.hip): the same kernel walking SEVERAL query blocks per workgroup, the next block's Q fragments and bias rows in flight under the
// current block's MFMAs; K / V staged and the partials folded once per workgroup. At 16 rows the launch is bound by bytes in flight (94 MB
// in 38 us with the general kernel: 256 VGPRs = 2 waves per SIMD, one block ahead): this form fits 168 registers at d = 40 -- 3 workgroups per CU, as
// many as the LDS holds (d = 64: 2, it spilled at 168) -- and keeps the grid 1-D with the heads of one (image, chunk) unit adjacent on one XCD (they share the lines of every Q / O row:
// profiles/r06_sector_sharing.md).
template <typename T, int KS, int DT, int NW, bool MULTI = false>
__global__ void __launch_bounds__(NW * 64, (DT >= 4 ? 1 : (MULTI && KS <= 3) ? 3 : 2)) cross_lean_kernel(const LeanParams lp) {
    typedef typename Vec<T>::v8 V8;
    typedef KTile<KS> KT;
    typedef VTile<DT> VT;
    constexpr bool RSM = KS * 16 < DT * 32;      // padding channels in the V tile: channel D is a column of ones, the PV MFMAs deliver the row sums
    constexpr int NT = NW * 64;
    constexpr int KRPP = NT / KT::CHK, VRPP = NT / VT::CHK;                   // key rows a pass of the workgroup covers
    constexpr int KPASS = (LEAN_ROWS + KRPP - 1) / KRPP, VPASS = (LEAN_ROWS + VRPP - 1) / VRPP;
    constexpr int K_BYTES = LEAN_ROWS * KT::STRIDE, V_BYTES = LEAN_ROWS * VT::STRIDE;
    const AttnParams &p = lp.a;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *Kl = smem, *Vl = smem + K_BYTES, *tile = smem + K_BYTES + V_BYTES;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hi = lane >> 5, l31 = lane & 31;
    int qb, h, b;
    if constexpr (MULTI) {
        if (lp.head_major) {
            const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
            const int unit = (j / p.H) * 8 + xcd;
            h = j % p.H; b = unit % p.B; qb = unit / p.B;
        } else {
            const int bh = blockIdx.x % (p.B * p.H);
            qb = blockIdx.x / (p.B * p.H); b = bh / p.H; h = bh - b * p.H;
        }
    } else {
        qb = blockIdx.x; h = blockIdx.y; b = blockIdx.z;      // query block c of every head on XCD c % 8 when there are 8 k blocks: the heads share its bias rows
    }
    tl_stamp(p, 0);

    // ---- every global load of the workgroup, before anything is waited for: gate, partials, Q fragments, K / V chunks, bias rows
    const float gate = p.bias_coeff ? p.bias_coeff[b] : 1.f;
    const float c0 = coeff_scalar_of(p);
    const bool maybe_biased = p.bias != nullptr;      // (from the kernel arguments alone: a gated-out image requests its bias rows in vain -- a few KB from L2)
    const bool need_stat = p.stat_kind != PWW_STAT_NONE;
    const bool f_all = lp.stats_out != nullptr;
    const bool f_max = f_all || p.stat_kind == PWW_STAT_MAX || p.stat_kind == PWW_STAT_ABSMAX;
    const bool f_min = f_all || p.stat_kind == PWW_STAT_MIN || p.stat_kind == PWW_STAT_ABSMAX;
    const bool f_sum = f_all || p.stat_kind == PWW_STAT_MEAN || p.stat_kind == PWW_STAT_STD;
    const bool f_sq = f_all || p.stat_kind == PWW_STAT_STD;
    const bool want_lo = need_stat && (f_max || f_min), want_hi = need_stat && (f_sum || f_sq);
    u32x4 plo[LEAN_PUNROLL], phi[LEAN_PUNROLL];
    {
        // (without partials the descriptor covers zero bytes: the loads return zeros without touching memory)
        const unsigned bytes = (need_stat && lp.parts && maybe_biased) ? (unsigned)lp.nparts * 32u : 0u;
        const auto srd_e = __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(lp.parts + (lp.parts ? (long)b * lp.nparts * 4 : 0)), 0, bytes, 0x00020000);
        const unsigned lo0 = want_lo ? (unsigned)lane * 32u : OOB_OFF, hi0 = want_hi ? (unsigned)lane * 32u + 16u : OOB_OFF;
#pragma unroll
        for (int j = 0; j < LEAN_PUNROLL; ++j) {
            plo[j] = __builtin_amdgcn_raw_buffer_load_b128(srd_e, lo0 + (unsigned)(j * 2048), 0, 0);
            phi[j] = __builtin_amdgcn_raw_buffer_load_b128(srd_e, hi0 + (unsigned)(j * 2048), 0, 0);
        }
    }
    const T *Qp = reinterpret_cast<const T *>(p.q) + b * p.q_sb + h * p.q_sh;
    const T *Kp = reinterpret_cast<const T *>(p.k) + b * p.k_sb + h * p.k_sh;
    const T *Vp = reinterpret_cast<const T *>(p.v) + b * p.v_sb + h * p.v_sh;
    T *Op = reinterpret_cast<T *>(p.o) + b * p.o_sb + h * p.o_sh;
    int qrow = (qb * NW + wave) * 32 + l31;
    bool qvalid = qrow < p.N;
    V8 qf[KS];
    const auto srd_q = head_srd(Qp, p.N, p.q_sn, p.D);
    load_q_frags_buf<T, KS>(qf, srd_q, (unsigned)((long)qrow * p.q_sn * 2), hi, p.D);      // (rows past N: beyond the descriptor)

    // K / V: thread -> (row kr of a pass, 16-byte column kc); pass i = rows i * KRPP .. of the head
    u32x4 kreg[KPASS], vreg[VPASS];
    const int kr = tid / KT::CHK, kc = tid - kr * KT::CHK;
    const int vr = tid / VT::CHK, vc = tid - vr * VT::CHK;
    const bool k_act = kr < KRPP, v_act = vr < VRPP;
    {
        const auto srd_k = head_srd(Kp, p.M, p.k_sm, p.D);
        const auto srd_v = head_srd(Vp, p.M, p.v_sm, p.D);
        const unsigned k0 = (k_act && kc * 8 < p.D) ? (unsigned)((kr * p.k_sm + kc * 8) * 2) : OOB_OFF, kstep = (unsigned)(KRPP * p.k_sm * 2);
        const unsigned v0 = (v_act && vc * 8 < p.D) ? (unsigned)((vr * p.v_sm + vc * 8) * 2) : OOB_OFF, vstep = (unsigned)(VRPP * p.v_sm * 2);
#pragma unroll
        for (int i = 0; i < KPASS; ++i) kreg[i] = __builtin_amdgcn_raw_buffer_load_b128(srd_k, k0 + (unsigned)i * kstep, 0, 0);
#pragma unroll
        for (int i = 0; i < VPASS; ++i) vreg[i] = __builtin_amdgcn_raw_buffer_load_b128(srd_v, v0 + (unsigned)i * vstep, 0, 0);
    }
    // bias rows of THIS WAVE's 32 query rows (wave-private piece of the tile: no barrier between its store and its reads): lane -> (row
    // rowl0 of a pass, 16-byte column pc); a pass covers 64 / cpr rows, cpr / 2 passes
    const int cprl = lp.tile_stride == 16 ? 2 : lp.tile_stride == 32 ? 3 : 4;      // log2 of the 16-byte chunks per tile row
    const int pc = lane & ((1 << cprl) - 1), rowl0 = lane >> cprl, tile_passes = 1 << (cprl - 1), tile_rp = 64 >> cprl;
    const bool t_col = pc * 4 < p.bias_cols;
    u32x4 treg[LEAN_TILE_LOADS];
    BiasRef bias;
    if (maybe_biased) {       // (uniform, from the kernel arguments)
        const float *bbase = p.bias + b * p.b_sb + h * p.b_sh;
        const unsigned bytes = (unsigned)((((long)(p.N - 1) * p.b_sn + (long)(p.M - 1) * p.b_sm) + 1) * 4);
        bias.srd = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(bbase), 0, bytes, 0x00020000);
        // (rows past N lie beyond the descriptor for a dense map; a row-broadcast map returns in-range values that nobody stores)
        const unsigned t0 = t_col ? (unsigned)((((long)(qb * NW + wave) * 32 + rowl0) * p.b_sn + pc * 4) * 4) : OOB_OFF, tstep = (unsigned)(tile_rp * p.b_sn * 4);
#pragma unroll
        for (int i = 0; i < LEAN_TILE_LOADS; ++i) treg[i] = __builtin_amdgcn_raw_buffer_load_b128(bias.srd, i < tile_passes ? t0 + (unsigned)i * tstep : OOB_OFF, 0, 0);
    }
    [[maybe_unused]] const unsigned tstep_m = (unsigned)(tile_rp * p.b_sn * 4);
    tl_stamp(p, 6);

    // ---- fold the image's partials: EVERY WAVE folds all of them itself (<= 4 per lane from the batch above, shuffles only -- no LDS, no
    // barrier; the extremes in fp32: a partial's max / min is a float stored as a double). The same order in every wave of every workgroup.
    float coeff = 0.f;
    const bool biased = maybe_biased && gate != 0.f;      // workgroup-uniform
    if (biased) {
        coeff = c0;
        if (need_stat) {
            auto as_double = [](unsigned lo, unsigned hi32) { return __longlong_as_double((long long)(((unsigned long long)hi32 << 32) | lo)); };
            float vmax = -INFINITY, vmin = INFINITY;
            double dsum = 0.0, dsq = 0.0;
#pragma unroll
            for (int j = 0; j < LEAN_PUNROLL; ++j) {
                const bool mine = lane + j * 64 < lp.nparts;
                if (f_max) vmax = fmaxf(vmax, mine ? (float)as_double(plo[j][0], plo[j][1]) : -INFINITY);
                if (f_min) vmin = fminf(vmin, mine ? (float)as_double(plo[j][2], plo[j][3]) : INFINITY);
                if (f_sum) dsum += mine ? as_double(phi[j][0], phi[j][1]) : 0.0;
                if (f_sq) dsq += mine ? as_double(phi[j][2], phi[j][3]) : 0.0;
            }
            for (int i = lane + LEAN_PUNROLL * 64; i < lp.nparts; i += 64) {      // (more than 256 partials per image: rare, not prefetched)
                const double *pp = lp.parts + ((long)b * lp.nparts + i) * 4;
                if (f_max) vmax = fmaxf(vmax, (float)pp[0]);
                if (f_min) vmin = fminf(vmin, (float)pp[1]);
                if (f_sum) dsum += pp[2];
                if (f_sq) dsq += pp[3];
            }
            if (f_max) {
#pragma unroll
                for (int off = 32; off >= 1; off >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, off));
            }
            if (f_min) {
#pragma unroll
                for (int off = 32; off >= 1; off >>= 1) vmin = fminf(vmin, __shfl_xor(vmin, off));
            }
            if (f_sum) {
#pragma unroll
                for (int off = 32; off >= 1; off >>= 1) dsum += __shfl_xor(dsum, off);
            }
            if (f_sq) {
#pragma unroll
                for (int off = 32; off >= 1; off >>= 1) dsq += __shfl_xor(dsq, off);
            }
            const double st[4] = {(double)vmax, (double)vmin, dsum, dsq};
            if (tid == 0 && lp.stats_out && h == 0 && qb == 0) {
                double *so = lp.stats_out + (long)b * 4;
                so[0] = st[0]; so[1] = st[1]; so[2] = st[2]; so[3] = st[3];
            }
            coeff = stat_coefficient(c0, p.stat_kind, st, p.stat_count);
        }
        if (p.bias_coeff) coeff = coeff * gate;
    }
    tl_stamp(p, 3);

    // ---- park K / V (whole 32-key blocks up to M: rows past M arrived as zeros) and the wave's bias rows, ONE barrier
    const int rows = min(LEAN_ROWS, (p.M + 31) & ~31);
    if (k_act) {
        char *kd = Kl + kr * KT::STRIDE + kc * 16;
#pragma unroll
        for (int i = 0; i < KPASS; ++i)
            if (i * KRPP < rows && ((i + 1) * KRPP <= LEAN_ROWS || i * KRPP + kr < LEAN_ROWS))      // (first test uniform; the second is a constant except in the last pass)
                *reinterpret_cast<u32x4 *>(kd + i * KRPP * KT::STRIDE) = kreg[i];
    }
    if (v_act) {
        char *vd = Vl + vr * VT::STRIDE + vc * 16;
        const T one = (T)1.0f;
        unsigned short one_bits;
        __builtin_memcpy(&one_bits, &one, 2);
        const bool v_one = RSM && vc * 8 == p.D;          // first padding chunk: channel D = 1.0 (the softmax denominator's column)
#pragma unroll
        for (int i = 0; i < VPASS; ++i)
            if (i * VRPP < rows && ((i + 1) * VRPP <= LEAN_ROWS || i * VRPP + vr < LEAN_ROWS))
                *reinterpret_cast<u32x4 *>(vd + i * VRPP * VT::STRIDE) = v_one ? u32x4{(unsigned)one_bits, 0u, 0u, 0u} : vreg[i];
    }
    if (biased) {
        if (t_col) {
#pragma unroll
            for (int i = 0; i < LEAN_TILE_LOADS; ++i)
                if (i < tile_passes) {
                    const int row = wave * 32 + rowl0 + i * tile_rp;
                    *reinterpret_cast<u32x4 *>(tile + (long)row * lp.tile_stride * 4 + ((pc ^ tile_swz(row, 1 << cprl)) << 4)) = treg[i];
                }
        }
        bias_ref_tile(bias, tile, wave * 32 + l31, lp.tile_stride, p.bias_cols, hi);
    }
    __syncthreads();
    tl_stamp(p, 1);

    // ---- scores -> (bias) -> softmax -> PV: the FIRST 64-key tile is all live (the host sends M < 64 to the general kernel), sets the row's
    // reference and starts O^T from a zero constant (STEP 1); the second keeps that reference unless a score exceeds it by 2^8 (STEP 2)
    const float c1 = p.scale_log2e;
    for (;;) {
        // MULTI: the NEXT block's Q fragments and bias rows are requested before this block is computed (a block past the last one:
        // beyond the descriptors -- zeros, no traffic)
        [[maybe_unused]] V8 qn[KS];
        constexpr int TLN = 4;            // MULTI takes maps of at most 32 staged columns (the host's rule): 4 pieces per lane
        [[maybe_unused]] u32x4 tn[TLN];
        [[maybe_unused]] const int nblk = qb + lp.nchunk;
        if constexpr (MULTI) {
            // (K / V fragments are the same for every block: left alone, hipcc hoists their LDS reads out of this loop and spills 88 - 136 bytes
            // per lane at the 168 registers three waves per SIMD leave)
            asm volatile("" ::: "memory");
            const int nrow = (nblk * NW + wave) * 32 + l31;
            load_q_frags_buf<T, KS>(qn, srd_q, nblk < lp.nqb ? (unsigned)((long)nrow * p.q_sn * 2) : OOB_OFF, hi, p.D);
            if (biased) {
                const unsigned t0 = (t_col && nblk < lp.nqb) ? (unsigned)((((long)(nblk * NW + wave) * 32 + rowl0) * p.b_sn + pc * 4) * 4) : OOB_OFF;
#pragma unroll
                for (int i = 0; i < TLN; ++i) tn[i] = __builtin_amdgcn_raw_buffer_load_b128(bias.srd, i < tile_passes ? t0 + (unsigned)i * tstep_m : OOB_OFF, 0, 0);
            }
        }
        f32x16 oacc[DT];
        float m_run = -INFINITY, l_run = 0.f;
        if (biased) {
            attn_tile<T, KS, DT, 2, false, RSM, 1>(oacc, m_run, l_run, qf, Kl, Vl, 0, p.M, l31, hi, bias, coeff, c1);
            if (KVBLK < p.M)
                attn_tile<T, KS, DT, 2, true, RSM, 2>(oacc, m_run, l_run, qf, Kl + KT::BYTES, Vl + VT::BYTES, KVBLK, p.M, l31, hi, bias, coeff, c1);
        } else {
            attn_tile<T, KS, DT, 0, false, RSM, 1>(oacc, m_run, l_run, qf, Kl, Vl, 0, p.M, l31, hi, bias, coeff, c1);
            if (KVBLK < p.M)
                attn_tile<T, KS, DT, 0, true, RSM, 2>(oacc, m_run, l_run, qf, Kl + KT::BYTES, Vl + VT::BYTES, KVBLK, p.M, l31, hi, bias, coeff, c1);
        }
        float l_tot;
        if (RSM) {      // row D of O^T: tile D / 32, register (D % 32) / 2, held by the hi == 0 half
            const int rl = p.D & 31, tl = p.D >> 5;
            float lv = 0.f;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                const float c = rl == 0 ? oacc[dt][0] : rl == 8 ? oacc[dt][4] : rl == 16 ? oacc[dt][8] : oacc[dt][12];
                lv = dt == tl ? c : lv;
            }
            const float other = __shfl_xor(lv, 32);
            l_tot = hi ? other : lv;
        } else {
            l_tot = l_run + __shfl_xor(l_run, 32);
        }
        const float inv = 1.f / l_tot;
        store_o_block<T, DT>(Op + (long)(qvalid ? qrow : 0) * p.o_sn, oacc, inv, p.D, hi, qvalid, p.o_wide != 0);
        if constexpr (!MULTI) break;
        else {
            if (nblk >= lp.nqb) break;       // (workgroup-uniform)
            qb = nblk;
            qrow = (qb * NW + wave) * 32 + l31;
            qvalid = qrow < p.N;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) qf[ks] = qn[ks];
            // the wave's own rows of the tile: its reads of the block just computed are behind it in program order (LDS runs in order per wave)
            if (biased && t_col) {
#pragma unroll
                for (int i = 0; i < TLN; ++i)
                    if (i < tile_passes) {
                        const int row = wave * 32 + rowl0 + i * tile_rp;
                        *reinterpret_cast<u32x4 *>(tile + (long)row * lp.tile_stride * 4 + ((pc ^ tile_swz(row, 1 << cprl)) << 4)) = tn[i];
                    }
            }
        }
    }
    tl_stamp(p, 4);
}

int attn_validate(const void *q, const void *k, const void *v, void *o, const float *bias, const pww_attn_desc_t *d);
void attn_fill_params(AttnParams &p, const void *q, const void *k, const void *v, void *o, const float *bias,
                      const float *bias_coeff, const pww_attn_desc_t *d);

static int lean_nw_knob() { return debug_knobs().cross_lean_nw; }

template <typename T, int KS, int DT, int NW>
static int launch_lean(LeanParams lp, hipStream_t stream) {
    constexpr size_t stage = (size_t)LEAN_ROWS * (KTile<KS>::STRIDE + VTile<DT>::STRIDE);
    const size_t lds = stage + (size_t)NW * 32 * lp.tile_stride * 4;
    auto kern = cross_lean_kernel<T, KS, DT, NW>;
    if (lds > 64 * 1024) {
        static thread_local size_t lds_attr[8] = {0, 0, 0, 0, 0, 0, 0, 0};      // per device
        int dev = 0;
        if (check_hip(hipGetDevice(&dev), "hipGetDevice")) return PWW_EHIP;
        if (dev >= 8 || lds > lds_attr[dev]) {
            if (check_hip(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds), "hipFuncSetAttribute"))
                return PWW_EHIP;
            if (dev < 8) lds_attr[dev] = lds;
        }
    }
    lp.nqb = (lp.a.N + NW * 32 - 1) / (NW * 32);
    if (lp.a.H > 65535 || lp.a.B > 65535) { set_error("cross_attn_lean: more than 65535 heads or images"); return PWW_EINVAL; }
    launch_attn_kernel(kern, dim3((unsigned)lp.nqb, (unsigned)lp.a.H, (unsigned)lp.a.B), dim3(NW * 64), lds, stream, lp);
    return check_hip(hipGetLastError(), "cross_lean_kernel launch");
}

// MULTI: several query blocks per workgroup. Workgroups per (image, head) = what is resident at once (by LDS and registers: the occupancy query), at
// most one per block; with B * nchunk a multiple of 8 the heads of a unit share an XCD (a slightly smaller grid is taken for that when it costs
// at most a quarter of the workgroups).
static int lean_device_cus() {
    static thread_local int cus[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    if (dev >= 0 && dev < 8 && cus[dev]) return cus[dev];
    hipDeviceProp_t prop;
    const int n = hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    if (dev >= 0 && dev < 8) cus[dev] = n;
    return n;
}
template <typename T, int KS, int DT, int NW>
static int launch_lean_multi(LeanParams lp, hipStream_t stream) {
    constexpr size_t stage = (size_t)LEAN_ROWS * (KTile<KS>::STRIDE + VTile<DT>::STRIDE);
    const size_t lds = stage + (size_t)NW * 32 * lp.tile_stride * 4;
    auto kern = cross_lean_kernel<T, KS, DT, NW, true>;
    static thread_local size_t lds_attr[8] = {0, 0, 0, 0, 0, 0, 0, 0};      // per device
    static thread_local int per_cu_seen[8][3] = {};                          // per device and tile stride (16 / 32 / 64)
    int dev = 0;
    if (check_hip(hipGetDevice(&dev), "hipGetDevice")) return PWW_EHIP;
    if (lds > 64 * 1024 && (dev >= 8 || lds > lds_attr[dev])) {
        if (check_hip(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds), "hipFuncSetAttribute"))
            return PWW_EHIP;
        if (dev < 8) lds_attr[dev] = lds;
    }
    const int ts = lp.tile_stride == 16 ? 0 : lp.tile_stride == 32 ? 1 : 2;
    int per_cu = dev < 8 ? per_cu_seen[dev][ts] : 0;
    if (per_cu <= 0) {
        if (check_hip(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, NW * 64, lds), "hipOccupancyMaxActiveBlocksPerMultiprocessor")) return PWW_EHIP;
        if (per_cu < 1) per_cu = 1;
        if (per_cu > 4) per_cu = 4;
        if (dev < 8) per_cu_seen[dev][ts] = per_cu;
    }
    lp.nqb = (lp.a.N + NW * 32 - 1) / (NW * 32);
    const long BH = (long)lp.a.B * lp.a.H;
    long nchunk = (long)per_cu * lean_device_cus() / BH;
    if (nchunk < 1) nchunk = 1;
    if (nchunk > lp.nqb) nchunk = lp.nqb;
    lp.head_major = 0;
    if (debug_knobs().cross_head_major) {
        for (long c = nchunk; c >= 1 && 4 * c >= 3 * nchunk; --c)
            if ((lp.a.B * c) % 8 == 0) { nchunk = c; lp.head_major = 1; break; }
    }
    lp.nchunk = (int)nchunk;
    launch_attn_kernel(kern, dim3((unsigned)(BH * nchunk)), dim3(NW * 64), lds, stream, lp);
    return check_hip(hipGetLastError(), "cross_lean_kernel (several blocks per workgroup) launch");
}

template <typename T, int NW> static int dispatch_lean_d(const LeanParams &lp, hipStream_t s) {
    const int D = lp.a.D;
    if (D <= 48) return launch_lean<T, 3, 2, NW>(lp, s);
    if (D <= 64) return launch_lean<T, 4, 2, NW>(lp, s);
    if (D <= 80) return launch_lean<T, 5, 3, NW>(lp, s);
    if (D <= 96) return launch_lean<T, 6, 3, NW>(lp, s);
    if constexpr (NW == 2) { set_error("cross_attn_lean: internal dispatch error"); return PWW_EINVAL; } else {      // D > 96 always gets 4 waves
        if (D <= 128) return launch_lean<T, 8, 4, NW>(lp, s);
        return launch_lean<T, 10, 5, NW>(lp, s);
    }
}

// How many query blocks of 128 rows a launch may have and still take this kernel (one block per workgroup): beyond it the general
// kernel's several-blocks-per-workgroup form amortises the K / V staging and the prologue (configs 3 / 4: 16 folded rows x 4096 tokens).
constexpr long LEAN_MAX_BLOCKS128 = 1024;

// Takes the launch if it is of the shape this kernel was written for; *launched = false hands it back to the general kernel.
int cross_attn_lean(const void *q, const void *k, const void *v, void *o, const float *bias, int stat_kind, float coeff_scalar, const float *gate,
                    const pww_attn_desc_t *d, double *stats_out, const pww_cross_opts_t &op, hipStream_t stream, const double *parts, int nparts,
                    bool *launched) {
    *launched = false;
    const int mode = debug_knobs().cross_lean;        // 0: never, 1: where it fits (default), 2: also for large batches
    if (!mode || !bias || op.bias_compact) return PWW_OK;
    if (d->M < KVBLK || d->M > 2 * KVBLK || d->bias_stride[3] != 1) return PWW_OK;
    if (stat_kind != PWW_STAT_NONE && !parts) return PWW_OK;
    const int m16 = (d->M + 15) & ~15;
    int bias_cols = op.bias_cols > 0 ? ((op.bias_cols + 15) & ~15) : m16;
    if (bias_cols > m16) bias_cols = m16;
    if (bias_cols > 64) return PWW_OK;
    // (the kernel forms offsets of rows just past N / M before the descriptors cut them off: keep them below 2^31)
    if (((long)(d->N + 128) * d->q_stride[2] + d->D) * 2 >= (1L << 31) || (long)(d->N + 128) * d->bias_stride[2] * 4 >= (1L << 31)) return PWW_OK;
    const long blocks128 = (long)d->B * d->H * ((d->N + 127) / 128);
    // beyond LEAN_MAX_BLOCKS128: the several-blocks-per-workgroup form of this kernel for head dims <= 64 (SD1.5's and SD2.1's finest level),
    // the general kernel for the rest (PWW_DEBUG=cross_lean_multi=0: the general kernel, round 5's route)
    const bool multi = blocks128 > LEAN_MAX_BLOCKS128 && mode == 1;
    if (multi && (d->D > 64 || bias_cols > 32 || !debug_knobs().cross_lean_multi)) return PWW_OK;
    LeanParams lp;
    attn_fill_params(lp.a, q, k, v, o, bias, gate, d);
    lp.a.bias_coeff = gate;
    lp.a.stats = nullptr; lp.a.stat_kind = stat_kind; lp.a.stat_count = (double)d->H * d->N * d->M; lp.a.coeff_scalar = coeff_scalar;
    lp.a.coeff_scalar_dev = op.coeff_scalar_dev;
    lp.a.bias_cols = bias_cols;
    lp.parts = stat_kind != PWW_STAT_NONE ? parts : nullptr;
    lp.nparts = stat_kind != PWW_STAT_NONE ? nparts : 0;
    lp.stats_out = stat_kind != PWW_STAT_NONE ? stats_out : nullptr;
    lp.nqb = 0;
    lp.tile_stride = bias_cols <= 16 ? 16 : bias_cols <= 32 ? 32 : 64;
    // waves of 32 rows; 4 per workgroup once that still gives the chip ~a workgroup per CU, else 2 (the widest heads always share their
    // K / V staging between 4: 2 waves would need > 256 staging registers each)
    int nw = (blocks128 >= 192 || d->D > 96) ? 4 : 2;
    if (lean_nw_knob() == 2 && d->D <= 96) nw = 2;
    if (lean_nw_knob() == 4) nw = 4;
    int rc;
    lp.nchunk = 1; lp.head_major = 0;
    if (multi) {
        if (d->dtype == PWW_DTYPE_F16) rc = d->D <= 48 ? launch_lean_multi<f16, 3, 2, 4>(lp, stream) : launch_lean_multi<f16, 4, 2, 4>(lp, stream);
        else rc = d->D <= 48 ? launch_lean_multi<bf16, 3, 2, 4>(lp, stream) : launch_lean_multi<bf16, 4, 2, 4>(lp, stream);
        *launched = rc == PWW_OK;
        return rc;
    }
    if (d->dtype == PWW_DTYPE_F16) rc = nw == 4 ? dispatch_lean_d<f16, 4>(lp, stream) : dispatch_lean_d<f16, 2>(lp, stream);
    else rc = nw == 4 ? dispatch_lean_d<bf16, 4>(lp, stream) : dispatch_lean_d<bf16, 2>(lp, stream);
    *launched = rc == PWW_OK;
    return rc;
}

}  // namespace pww
