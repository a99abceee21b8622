// Kernel and launch templates of the general cross-attention launch (design notes: pww_cross.hip); included by the instantiation
// units (pww_cross_inst.hip, compiled once per storage type and workgroup width so that the build runs them side by side).
#pragma once
#include <string.h>
#include "pww_attn_core.h"
#include "pww_cross_tile.h"

namespace pww {


struct CrossParams {
    AttnParams a;        // a.bias_coeff = the row gate [B] (or null), a.stats unused
    unsigned long long *slots;   // persistent, zero between launches: [B][nqb * H][4] complemented fp64 partials
    unsigned *sync;              // persistent, zero between launches: left[B][H] (workgroups of a head that have left),
                                 // heads_left[B], then the error word
    double *stats_out;   // optional [B][4]: the folded statistics of the gated-in images
    int nqb;             // query blocks per (image, head)
    int nchunk;          // workgroups per (image, head) -- of the first n_gated images when the caller gave that hint
    int nchunk_u;        // workgroups per (image, head) of the images from n_gated on (== nchunk without the hint)
    int n_gated;         // hint: gate[b] != 0 exactly for b < n_gated (0 = unknown): those images carry pass 1, the hand-off and the
                         // bias work -- about three times the time per query block -- and get more, shorter workgroups
    // bias rows of a query block staged in LDS (tile_stride > 0): [NW * 32 rows][tile_stride floats], filled either from the
    // dense map (columns < a.bias_cols; the rows of a block are one contiguous span of the [N, M] map) or from the compact form
    int tile_stride;             // floats per tile row: bias_cols rounded up to a power of two (16 / 32 / 64 / 128; XOR-swizzled chunks: tile_swz); 0 = per-lane global loads
    int tile_nbuf;               // dense form, several query blocks per workgroup: 2 = the next block's rows are loaded (LDS-direct) while
                                 // this block is computed, 1 = one buffer (the second would cost a resident workgroup per CU)
    const float *compact;        // compact bias [B?][N][R] (or null): bias[b][n][col_idx[b?][r]] = compact[b][n][r], every other column zero
    const int *col_idx;          // [B?][R], -1 = unused slot
    int R;
    long c_sb, c_sn, ci_sb;      // compact strides (image, row) and col_idx image stride, in elements
    // statistics partials formed by the PRODUCER of Q (pww_qproj.hip: the to_q GEMM's epilogue): [B][ext_nparts][4] plain fp64
    // { max, min, sum, sum of squares }. When set, the kernel folds them at entry and runs pass 2 only: no pass 1, no slots, no
    // hand-off, no residency requirement, Q read once -- the kernel boundary was the synchronisation.
    const double *ext_part;
    int ext_nparts;
    int head_major;      // workgroup order: the heads of one (image, query chunk) unit adjacent on one XCD (needs a multiple of 8 units)
    int pass2_only;      // host side: the launch has no hand-off (external partials, or no statistic asked for through pww_cross_attn_fwd_parts): any grid is correct
};

constexpr unsigned long long SPIN_LIMIT_TICKS = 100000000ull;   // wall_clock64 runs at 100 MHz: 1 s (the grid is sized to be resident:
                                                                // the limit only bounds the impossible case, e.g. state words left dirty by an aborted launch)
constexpr int RED_BLKS = 8;         // pass 1 folds the per-wave partials of up to 8 query blocks behind one barrier
constexpr int COMPACT_MAX_R = 32;    // compact bias: at most 32 non-zero columns (16 staged values per thread)

// a slot holds ~bits(value): zero = empty (no finite or infinite double has an all-ones bit pattern)
__device__ __forceinline__ void slot_publish(unsigned long long *p, double v) {
    __hip_atomic_store(p, ~(unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned long long slot_read(const unsigned long long *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double slot_value(unsigned long long x) { return __longlong_as_double((long long)~x); }

// acc + x * x with the product rounded before the sum: what the select form `acc += live ? x * x : 0` compiles to here and in
// pww_qk_reduce (the select keeps hipcc from contracting it into an fma); the statistics of the two paths are compared bit for bit.
__device__ __forceinline__ float add_square_unfused(float acc, float x) {
#pragma clang fp contract(off)
    const float sq = x * x;
    return acc + sq;
}
constexpr int WAIT_VMCNT0 = 0x0F70;     // s_waitcnt vmcnt(0) only (gfx9 encoding: expcnt 7, lgkmcnt 15 = "do not wait")

// ---- bias tile staging -------------------------------------------------------------------------------------------------
// A query block's bias rows are ONE contiguous span of the [N, M] fp32 map (rows x 308 bytes for the 77 prompt tokens). Loading
// them per lane (each lane its own row: 32 cache lines per wave-load, repeated by every head's workgroup) was what held the
// batched launch at 13 % of HBM peak; here the workgroup copies the span with coalesced 16-byte loads -- a thread's 4 chunks of a
// 32-column slab: 8 consecutive threads cover 128 contiguous bytes of a row -- into an LDS tile and every lane reads its row from
// there. Only the columns below bias_cols are moved (the map of a prompt is zero past the last region phrase).
template <int NT, typename SRD>
__device__ __forceinline__ void tile_slab_load(u32x4 (&reg)[4], SRD srd, long row0, int N, long b_sn, int slab, int bias_cols, int tid) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int g = tid + i * NT, row = g >> 3, col = slab * 32 + (g & 7) * 4;
        const bool ok = col < bias_cols && row0 + row < N;
        reg[i] = __builtin_amdgcn_raw_buffer_load_b128(srd, ok ? (unsigned)(((row0 + row) * b_sn + col) * 4) : OOB_OFF, 0, 0);
    }
}
template <int NT>
__device__ __forceinline__ void tile_slab_store(const u32x4 (&reg)[4], char *tile, int tile_stride, int slab, int bias_cols, int tid) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int g = tid + i * NT, row = g >> 3, chunk = slab * 8 + (g & 7);
        if (chunk * 4 < bias_cols)
            *reinterpret_cast<u32x4 *>(tile + (long)row * tile_stride * 4 + ((chunk ^ tile_swz(row, tile_stride >> 2)) << 4)) = reg[i];
    }
}
// LDS-direct form of the same copy (several query blocks per workgroup): `buffer_load_dwordx4 ... lds` writes lane-linear
// (wave-uniform base + lane * 16), so the swizzle goes onto the SOURCE address. No staging registers, and the copy of block i + 1
// is in flight while block i is computed. EVERY WAVE COPIES ITS OWN 32 ROWS (a contiguous 32 * cpr * 16-byte piece of the tile, cpr / 2
// copies of 1 KB): nobody else reads them, so the wave's own `s_waitcnt vmcnt` is all the synchronisation there is -- pass 2 has no
// barrier. The tile's physical row is a power of two wide (4 / 8 / 16 chunks for 16 / 32 / 48 columns); copy i of a lane is the
// same physical chunk 64 / cpr rows further down, and the swizzle repeats every 16 rows, i.e. every P = cpr / 4 copies: P source
// offsets per lane (rel[j], or OOB_OFF for the padding chunks of a 48-column tile) describe all of them. Rows past N lie beyond the
// descriptor's range and arrive as zeros.
constexpr int TILE_GLDS_PERIOD_MAX = 4;
__device__ __forceinline__ void tile_glds_plan(unsigned (&rel)[TILE_GLDS_PERIOD_MAX], int cpr, int cols, long b_sn, int wave, int lane) {
#pragma unroll
    for (int j = 0; j < TILE_GLDS_PERIOD_MAX; ++j) {
        const int g = lane + j * 64, rowl = g / cpr, pc = g - rowl * cpr, row = wave * 32 + rowl, c = pc ^ tile_swz(row, cpr);
        rel[j] = (j * 4 < cpr && c * 4 < cols) ? (unsigned)((row * b_sn + c * 4) * 4) : OOB_OFF;
    }
}
// The copies are issued as inline assembly ON PURPOSE: hipcc orders every later LDS read behind an LDS-direct load it knows about
// (it cannot tell the K / V fragment reads and the other tile buffer from the copy's destination) and would wait for the copy at
// the first `ds_read` of the block -- the latency this scheme exists to hide. Untracked, the copies only ever make hipcc's own counted
// waits conservative (vmcnt retires in order); their completion is waited for explicitly (s_waitcnt vmcnt(0) + barrier) before
// the tile is read. M0 (the LDS destination) is saved and restored inside the statement.
__device__ __forceinline__ void glds16(const u32x4 srd, unsigned lds_dst, unsigned voff) {
    unsigned keep;
    __asm__ volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(voff), "s"(lds_dst), "s"(srd) : "memory");
}
template <int CNT, int P>
__device__ __forceinline__ void tile_glds_n(const unsigned (&rel)[TILE_GLDS_PERIOD_MAX], unsigned base, unsigned step16, const u32x4 srd, unsigned lds_dst) {
#pragma unroll
    for (int i = 0; i < CNT; ++i) glds16(srd, lds_dst + (unsigned)(i * 1024), base + rel[i % P] + (unsigned)(i / P) * step16);
}
// (one straight-line sequence per width: 2 / 4 / 8 copies per lane)
__device__ __forceinline__ void tile_glds(const unsigned (&rel)[TILE_GLDS_PERIOD_MAX], const u32x4 srd, const char *tile, long row0, long b_sn, int cpr, int wave) {
    typedef __attribute__((address_space(3))) const char *lds_cp;
    const unsigned base = (unsigned)(row0 * b_sn * 4);                   // (OOB_OFF + anything below 2^31 stays out of range)
    const unsigned step16 = (unsigned)(16 * b_sn * 4);                   // the swizzle's period: 16 rows
    const unsigned lds_dst = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_cp)tile + (unsigned)(wave * 32 * cpr * 16));   // the wave's own rows (an SGPR)
    if (cpr == 4) tile_glds_n<2, 1>(rel, base, step16, srd, lds_dst);
    else if (cpr == 8) tile_glds_n<4, 2>(rel, base, step16, srd, lds_dst);
    else tile_glds_n<8, 4>(rel, base, step16, srd, lds_dst);
}
// compact form: rows x R contiguous floats of the block, scattered to their columns (the tile was zeroed once: columns outside
// col_idx stay zero for the whole launch)
// Thread mapping: two threads per tile row (the tile has NT / 2 rows), thread t moves slots r = (t & 1) + 2 i of row t >> 1 --
// every address is a per-thread base plus an immediate, no index arithmetic is kept in registers. col_idx sits in LDS (cidx_lds,
// COMPACT_MAX_R ints, -1 = unused) so the scatter reads its column right before the store.
template <int NT>
__device__ __forceinline__ void compact_load(u32x4 (&reg)[4], const float *cbase, long row0, int N, long c_sn, int R, int tid) {
    const long row = row0 + (tid >> 1);
    const float *src = cbase + row * c_sn + (tid & 1);
    const bool row_ok = row < N;
#pragma unroll
    for (int i = 0; i < COMPACT_MAX_R / 2; ++i) {
        const float v = (row_ok && (tid & 1) + 2 * i < R) ? src[2 * i] : 0.f;
        reg[i >> 2][i & 3] = __float_as_uint(v);
    }
}
template <int NT>
__device__ __forceinline__ void compact_store(const u32x4 (&reg)[4], char *tile, int tile_stride, const int *cidx_lds, int R, int tid) {
    const int row = tid >> 1;
    char *trow = tile + (long)row * tile_stride * 4;
    const int swz = tile_swz(row, tile_stride >> 2);
    const int *ci = cidx_lds + (tid & 1);
#pragma unroll
    for (int i = 0; i < COMPACT_MAX_R / 2; ++i) {
        if ((tid & 1) + 2 * i < R) {
            const int col = ci[2 * i];
            if (col >= 0) *reinterpret_cast<float *>(trow + (((col >> 2) ^ swz) << 4) + (col & 3) * 4) = __uint_as_float(reg[i >> 2][i & 3]);
        }
    }
}

// request / park one query block's bias rows (slab 0 of the dense form, or the compact values)
template <int NT, typename SRD>
__device__ __forceinline__ void tile_request(u32x4 (&reg)[4], bool compact, const float *cbase, SRD srd, long row0, int N, long c_sn, long b_sn,
                                             int R, int bias_cols, int tid) {
    if (compact) compact_load<NT>(reg, cbase, row0, N, c_sn, R, tid);
    else tile_slab_load<NT>(reg, srd, row0, N, b_sn, 0, bias_cols, tid);
}
template <int NT, typename SRD>
__device__ __forceinline__ void tile_park(u32x4 (&reg)[4], bool compact, char *tile, int tile_stride, const int *cidx, SRD srd, long row0, int N,
                                          long b_sn, int R, int bias_cols, int tid) {
    if (compact) { compact_store<NT>(reg, tile, tile_stride, cidx, R, tid); return; }
    tile_slab_store<NT>(reg, tile, tile_stride, 0, bias_cols, tid);
    const int nslab = (bias_cols + 31) >> 5;
    for (int slab = 1; slab < nslab; ++slab) {     // maps wider than 32 columns: the further slabs are not prefetched
        tile_slab_load<NT>(reg, srd, row0, N, b_sn, slab, bias_cols, tid);
        tile_slab_store<NT>(reg, tile, tile_stride, slab, bias_cols, tid);
    }
}

template <typename T, int KS, int DT, int NW, bool SINGLE, bool COMPACT>
__global__ void __launch_bounds__(NW * 64, (DT >= 4 ? 1 : 2)) cross_fused_kernel(const CrossParams cp) {
    typedef typename Vec<T>::v8 V8;
    typedef KTile<KS> KT;
    typedef VTile<DT> VT;
    constexpr int NSUB = 2;
    // the head dim (in (16 (KS - 1), 16 KS], a multiple of 8) leaves padding channels in the V tile (d = 40 -> 64, 80 -> 96): channel D is
    // a column of ones and the row sums come out of the PV MFMAs (no per-score adds; the sum of the ROUNDED P, as in pww_attn.hip)
    constexpr bool RSM = KS * 16 < DT * 32;
    constexpr int NT = NW * 64;
    constexpr int SUB_BYTES = KT::BYTES + VT::BYTES;
    constexpr int STAGE_BYTES = NSUB * SUB_BYTES;
    constexpr int KPT = (NSUB * KT::NCHUNK + NT - 1) / NT;
    constexpr int VPT = (NSUB * VT::NCHUNK + NT - 1) / NT;
    const AttnParams &p = cp.a;

    extern __shared__ __attribute__((aligned(16))) char smem[];   // [K|V stage][fin: NW x 4 f64][final: 4 f64][red: 2 x RED_BLKS x NW x 4 f32][flag][col_idx][bias tile(s)]
    double *fin = reinterpret_cast<double *>(smem + STAGE_BYTES);
    double *final_st = fin + NW * 4;
    float *red = reinterpret_cast<float *>(final_st + 4);          // pass 1: [2 groups][RED_BLKS blocks][NW waves][4] per-wave partials
    [[maybe_unused]] volatile int *ok_flag = reinterpret_cast<volatile int *>(red + 2 * RED_BLKS * NW * 4);
    int *cidx_lds = reinterpret_cast<int *>(red + 2 * RED_BLKS * NW * 4) + 4;             // COMPACT_MAX_R ints
    char *tile = reinterpret_cast<char *>(cidx_lds + COMPACT_MAX_R);
    const int tile_bytes = NW * 32 * cp.tile_stride * 4;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int hi = lane >> 5, l31 = lane & 31;
    const int BH = p.B * p.H;
    tl_stamp(p, 0);
    const unsigned long long tl_c0 = p.timeline ? clock64() : 0ull;
    // Workgroup -> (image, head, query chunk). The hardware deals workgroup i to XCD i % 8, and the only bytes two workgroups of this
    // kernel share are the bias rows of a query block (the same [rows, 77] fp32 tile for all H heads of an image). With the plain
    // order (heads fastest) every XCD's L2 ends up fetching the whole map (measured: 11.9 MB fetched per launch for 6.7 MB of
    // distinct bytes); dealing query chunk c to XCD c % 8 for every head fetches each bias tile once.
    int bh, chunk, nchunk = cp.nchunk;
    if (cp.head_major) {
        // (round 6) ... and the H heads of one (image, query chunk) unit ADJACENT on one XCD, for every grid the host may choose: besides
        // the unit's bias rows they share 32-byte sectors of Q and O where a head's slice is not a whole number of sectors (d = 40: 80
        // bytes -- 96 moved per 80 with the heads on eight XCDs, which is what 16 images x 4 chunks got: 107.5 MB for 86.7,
        // profiles/r04_pmc_cross_route.json). XCD x takes the units x, x + 8, ...; the hinted-in images' units come first.
        const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
        const int h_ = j % p.H, unit = (j / p.H) * 8 + xcd;
        const int first = cp.n_gated * cp.nchunk;
        int b_;
        if (unit < first) { b_ = unit % cp.n_gated; chunk = unit / cp.n_gated; }
        else { const int i = unit - first, ub = p.B - cp.n_gated; b_ = cp.n_gated + i % ub; chunk = i / ub; nchunk = cp.nchunk_u; }
        bh = b_ * p.H + h_;
    } else if (cp.n_gated > 0) {
        // two classes of images (see CrossParams::n_gated): the workgroups of the hinted-in images come first
        const int gh = cp.n_gated * p.H, first = gh * cp.nchunk;
        if ((int)blockIdx.x < first) { bh = blockIdx.x % gh; chunk = blockIdx.x / gh; }
        else { const int i = blockIdx.x - first, uh = BH - gh; bh = gh + i % uh; chunk = i / uh; nchunk = cp.nchunk_u; }
    } else if ((cp.nchunk & 7) == 0) {
        const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
        bh = j % BH;
        chunk = (j / BH) * 8 + xcd;
    } else {
        bh = blockIdx.x % BH;
        chunk = blockIdx.x / BH;
    }
    const int b = bh / p.H, h = bh - b * p.H;

    // The row gate is the first global word this workgroup touches (a cold first access costs microseconds): it is REQUESTED here
    // and looked at after K and V are staged, so the prologue's loads -- Q, K, V, the bias rows -- do not queue up behind it.
    // Until then everything is decided from the kernel arguments alone (`maybe`: a gated-out image stages its bias rows in vain).
    const float gate = p.bias_coeff ? p.bias_coeff[b] : 1.f;
    const bool maybe_biased = COMPACT || p.bias != nullptr;
    const bool use_tile = maybe_biased && cp.tile_stride > 0;
    const bool use_compact = COMPACT && use_tile;     // (the host instantiates COMPACT exactly when cp.compact is set: the compact
                                                      // form's staging registers stay out of the dense kernels)

    const T *Qp = reinterpret_cast<const T *>(p.q) + b * p.q_sb + h * p.q_sh;
    const T *Kp = reinterpret_cast<const T *>(p.k) + b * p.k_sb + h * p.k_sh;
    const T *Vp = reinterpret_cast<const T *>(p.v) + b * p.v_sb + h * p.v_sh;
    T *Op = reinterpret_cast<T *>(p.o) + b * p.o_sb + h * p.o_sh;

    // first query block of this workgroup: its Q fragments are requested before anything else (SINGLE: they stay in
    // registers for both passes; otherwise every iteration requests the NEXT block's fragments before it computes)
    V8 qf[KS];
    const auto srd_q = head_srd(Qp, p.N, p.q_sn, p.D);
    auto request_q = [&](V8 (&dst)[KS], int blk) {       // the fragments of query block `blk` (zeros past the last block / row)
        const int nrow = (blk * NW + wave) * 32 + l31;
        load_q_frags_buf<T, KS>(dst, srd_q, blk < cp.nqb && nrow < p.N ? (unsigned)((long)nrow * p.q_sn * 2) : OOB_OFF, hi, p.D);
    };
    request_q(qf, chunk);

    BiasRef bias;
    u32x4 bias_srd4 = {0u, 0u, 0u, 0u};
    if (use_tile) bias_ref_tile(bias, tile, wave * 32 + l31, cp.tile_stride, p.bias_cols, hi);
    // dense form, several blocks per workgroup: LDS-direct copies (no staging registers), double-buffered when cp.tile_nbuf == 2
    const bool use_glds = !SINGLE && use_tile && !use_compact;
    const int tcpr = cp.tile_stride >> 2;      // physical 16-byte chunks per tile row (4 / 8 / 16 / 32)
    unsigned trel[TILE_GLDS_PERIOD_MAX] = {0u, 0u, 0u, 0u};
    if (use_glds) tile_glds_plan(trel, tcpr, p.bias_cols, p.b_sn, wave, lane);
    const float *cbase = nullptr;
    const int *cidx = nullptr;
    if (maybe_biased && !use_compact) {
        const float *bbase = p.bias + b * p.b_sb + h * p.b_sh;
        const unsigned bytes = (unsigned)((((long)(p.N - 1) * p.b_sn + (long)(p.M - 1) * p.b_sm) + 1) * 4);
        bias.srd = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(bbase), 0, bytes, 0x00020000);
        const unsigned long long ba = (unsigned long long)bbase;       // the same descriptor as four words, for the LDS-direct copies
        bias_srd4 = u32x4{(unsigned)ba, (unsigned)(ba >> 32) & 0xffffu, bytes, 0x00020000u};
        bias.key_stride = (unsigned)(p.b_sm * 4);
        bias.unit = p.b_sm == 1;
    }
    if (use_compact) {
        cbase = cp.compact + b * cp.c_sb;
        if (tid < COMPACT_MAX_R) cidx_lds[tid] = tid < cp.R ? cp.col_idx[b * cp.ci_sb + tid] : -1;     // (visible after the prologue's barrier)
        cidx = cidx_lds;
    }
    u32x4 treg[4];                          // staging registers of the bias tile
    u32x4 ext_lo = {0u, 0u, 0u, 0u}, ext_hi = {0u, 0u, 0u, 0u};     // external partials: this thread's first one (max, min | sum, sum of squares)

    // K and V of this head -> LDS (rows past M and the head-dim padding are zeros)
    for (int i = tid * 16; i < STAGE_BYTES; i += NT * 16) *reinterpret_cast<u32x4 *>(smem + i) = u32x4{0u, 0u, 0u, 0u};
    if (use_compact)
        for (int i = tid * 16; i < tile_bytes; i += NT * 16) *reinterpret_cast<u32x4 *>(tile + i) = u32x4{0u, 0u, 0u, 0u};
    {
        StagePlan<KPT, VPT> plan;
        make_plan<T, KS, DT, NT, NSUB, KPT, VPT>(plan, tid, p.D, p.k_sm, p.v_sm);
        const auto srd_k = head_srd(Kp, p.M, p.k_sm, p.D);
        const auto srd_v = head_srd(Vp, p.M, p.v_sm, p.D);
        u32x4 kreg[KPT];
        u32x4 vreg[VPT];
        stage_load(kreg, vreg, plan, srd_k, srd_v, 0u, 0u);
        // external partials: thread t's first partial travels with K and V too (folded below, once the gate is known). UNCONDITIONAL
        // buffer loads (a load under an `if` would make everything in flight wait at the join): without partials the descriptor covers
        // zero bytes and the loads return zeros without touching memory
        {
            const unsigned ext_bytes = cp.ext_part ? (unsigned)cp.ext_nparts * 32u : 0u;
            const auto srd_e = __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(cp.ext_part + (cp.ext_part ? (long)b * cp.ext_nparts * 4 : 0)), 0, ext_bytes, 0x00020000);
            const unsigned eo = tid < cp.ext_nparts ? (unsigned)tid * 32u : OOB_OFF;
            ext_lo = __builtin_amdgcn_raw_buffer_load_b128(srd_e, eo, 0, 0);
            ext_hi = __builtin_amdgcn_raw_buffer_load_b128(srd_e, eo + 16u, 0, 0);
        }
        // SINGLE: the (only) block's bias rows travel with K and V (one latency for all three)
        if (SINGLE && use_tile) tile_request<NT>(treg, use_compact, cbase, bias.srd, (long)chunk * NW * 32, p.N, cp.c_sn, p.b_sn, cp.R, p.bias_cols, tid);
        tl_stamp(p, 6);
        __syncthreads();                      // the zero fill is complete
        stage_store<DT, KPT, VPT>(kreg, vreg, plan, smem);
        if (RSM) {      // column D of every V row is one: the PV MFMA accumulates the softmax denominator in row D of O^T
            const T one = (T)1.0f;
            for (int i = tid; i < NSUB * KVBLK; i += NT)
                *reinterpret_cast<T *>(smem + (i >> 6) * SUB_BYTES + KT::BYTES + (i & 63) * VT::STRIDE + p.D * 2) = one;
        }
        if (SINGLE && use_tile) tile_park<NT>(treg, use_compact, tile, cp.tile_stride, cidx, bias.srd, (long)chunk * NW * 32, p.N, p.b_sn, cp.R, p.bias_cols, tid);
    }
    __syncthreads();
    tl_stamp(p, 1);
    const bool biased = maybe_biased && gate != 0.f;      // workgroup-uniform
    const bool need_stat = biased && p.stat_kind != PWW_STAT_NONE;

    float coeff = 0.f;
    [[maybe_unused]] unsigned depart_prev = 0u;     // lane 0 of the workgroup: how many workgroups of the image had left before this one
    // which of { max, min, sum, sum of squares } this launch has to know: all four when the caller wants the statistics back,
    // else only what the selected statistic is made of -- the others are neither reduced nor polled (a quarter of the hand-off's
    // slot traffic for qk.max())
    const bool f_all = cp.stats_out != nullptr;
    [[maybe_unused]] const bool f_max = f_all || p.stat_kind == PWW_STAT_MAX || p.stat_kind == PWW_STAT_ABSMAX;
    [[maybe_unused]] const bool f_min = f_all || p.stat_kind == PWW_STAT_MIN || p.stat_kind == PWW_STAT_ABSMAX;
    [[maybe_unused]] const bool f_sum = f_all || p.stat_kind == PWW_STAT_MEAN || p.stat_kind == PWW_STAT_STD;
    [[maybe_unused]] const bool f_sq = f_all || p.stat_kind == PWW_STAT_STD;
    if (need_stat && cp.ext_part) {
        // ---- the statistic's partials came with Q (pww_qproj.hip): fold the image's partials -- a few hundred doubles, requested with K / V
        const bool mine = tid < cp.ext_nparts;
        auto as_double = [](unsigned lo, unsigned hi32) { return __longlong_as_double((long long)(((unsigned long long)hi32 << 32) | lo)); };
        double dmax = mine ? as_double(ext_lo[0], ext_lo[1]) : -INFINITY, dmin = mine ? as_double(ext_lo[2], ext_lo[3]) : INFINITY;
        double dsum = mine ? as_double(ext_hi[0], ext_hi[1]) : 0.0, dsq = mine ? as_double(ext_hi[2], ext_hi[3]) : 0.0;
        for (int i = tid + NT; i < cp.ext_nparts; i += NT) {
            const double *pp = cp.ext_part + ((long)b * cp.ext_nparts + i) * 4;
            dmax = fmax(dmax, pp[0]); dmin = fmin(dmin, pp[1]); dsum += pp[2]; dsq += pp[3];
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            dmax = fmax(dmax, __shfl_xor(dmax, off));
            dmin = fmin(dmin, __shfl_xor(dmin, off));
            dsum += __shfl_xor(dsum, off);
            dsq += __shfl_xor(dsq, off);
        }
        if (lane == 0) { fin[wave * 4 + 0] = dmax; fin[wave * 4 + 1] = dmin; fin[wave * 4 + 2] = dsum; fin[wave * 4 + 3] = dsq; }
        __syncthreads();
        for (int w = 0; w < NW; ++w) {
            if (w == 0) { dmax = fin[0]; dmin = fin[1]; dsum = fin[2]; dsq = fin[3]; }
            else { dmax = fmax(dmax, fin[w * 4 + 0]); dmin = fmin(dmin, fin[w * 4 + 1]); dsum += fin[w * 4 + 2]; dsq += fin[w * 4 + 3]; }
        }
        if (tid == 0 && cp.stats_out && h == 0 && chunk == 0) {
            double *st = cp.stats_out + (long)b * 4;
            st[0] = dmax; st[1] = dmin; st[2] = dsum; st[3] = dsq;
        }
        const double st[4] = {dmax, dmin, dsum, dsq};
        coeff = stat_coefficient(coeff_scalar_of(p), p.stat_kind, st, p.stat_count);
        if (p.bias_coeff) coeff = coeff * gate;
        tl_stamp(p, 3);
        if constexpr (!SINGLE)
            if (use_glds) tile_glds(trel, bias_srd4, tile, (long)chunk * NW * 32, p.b_sn, tcpr, wave);
    }
#if PWW_EXPERIMENTS       // round 3's in-launch statistic (pass 1 + workgroup hand-off): libpww_hip_experiments.so only; the product's launches fold partials (above)
    else if (need_stat) {
        // Everything up to the folded statistic is the launch's critical path (every workgroup of the image waits for the slowest
        // pass 1): these waves go ahead of the co-resident workgroups that are already in pass 2 (unconditional rows have no pass 1).
        __builtin_amdgcn_s_setprio(3);
        // ---- pass 1: per query block (max, min, sum, sum of squares) of the raw scores, as pww_qk_reduce computes them.
        // Several blocks per workgroup: the Q fragments run THREE blocks ahead (a ring of named register sets: q1, q2, q3 -- pass 1
        // holds no accumulators, the registers are there), and the per-block partials are folded in GROUPS of up to RED_BLKS blocks:
        // every wave parks its four values per block in LDS, one barrier per group, then thread t folds the group's t-th block over
        // the waves in fp64 (the order of pww_qk_reduce) and publishes it -- no barrier and no serial one-thread stretch per block.
        V8 q1[KS], q2[KS], q3[KS];
        if constexpr (!SINGLE) {
            request_q(q1, chunk + nchunk);
            request_q(q2, chunk + 2 * nchunk);
            request_q(q3, chunk + 3 * nchunk);
        }
        auto fold_group = [&](int grp, int nblk) {       // blocks grp * RED_BLKS ... + nblk - 1 of this workgroup (counted from its first)
            __syncthreads();
            if (tid < nblk) {
                const float *rp = red + ((grp & 1) * RED_BLKS + tid) * NW * 4;
                double dmax = -INFINITY, dmin = INFINITY, dsum = 0.0, dsq = 0.0;
                for (int w = 0; w < NW; ++w) {
                    dmax = fmax(dmax, (double)rp[w * 4 + 0]);
                    dmin = fmin(dmin, (double)rp[w * 4 + 1]);
                    dsum += (double)rp[w * 4 + 2];
                    dsq += (double)rp[w * 4 + 3];
                }
                const int qb = chunk + (grp * RED_BLKS + tid) * nchunk;
                unsigned long long *slot = cp.slots + ((long)b * cp.nqb * p.H + (long)qb * p.H + h) * 4;
                slot_publish(slot + 0, dmax); slot_publish(slot + 1, dmin);
                slot_publish(slot + 2, dsum); slot_publish(slot + 3, dsq);
            }
        };
        // Extremes only (qk.max() -- the shipped weight functions -- / min / absmax, nobody asked for the statistics back): max and min are
        // idempotent, so the workgroup keeps ONE running per-lane extreme over all its blocks, reduces it across lanes and waves once, and
        // publishes that value into every one of its blocks' slots -- the image's folded maximum is the same number, and the per-block
        // cross-lane reduction (six dependent shuffles), the LDS hand-over and the fold leave the loop. Sums keep the per-block partials
        // (their fold order is what makes the statistics bit-identical to pww_qk_reduce).
        const bool lazy_ext = !f_sum;
        float vmax = -INFINITY, vmin = INFINITY, vsum = 0.f, vsq = 0.f;
        int it = 0;
        for (int qb = chunk; qb < cp.nqb; qb += nchunk, ++it) {
            const int qrow = (qb * NW + wave) * 32 + l31;
            const bool qvalid = qrow < p.N;
            const bool rows_full = (qb * NW + wave) * 32 + 32 <= p.N;      // wave-uniform: every row of the wave is a real row
            if (!lazy_ext) { vmax = -INFINITY; vmin = INFINITY; vsum = 0.f; vsq = 0.f; }
#pragma unroll
            for (int sub = 0; sub < NSUB; ++sub) {
                const int key0 = sub * KVBLK;
                if (key0 < p.M) {
                    f32x16 s[2];
                    score_tile<T, KS>(s, qf, smem + sub * SUB_BYTES, key0, p.M, l31, hi);
#pragma unroll
                    for (int kb = 0; kb < 2; ++kb) {
                        if (key0 + kb * 32 < p.M) {         // (a 32-key block past M holds no live key)
                            if (rows_full && key0 + kb * 32 + 32 <= p.M) {
                                // every score of the block is live (all but the last key block of all but the last query block): no selects
                                if (f_max) {      // (four-element groups: a dependent chain of 4 per block instead of 16)
#pragma unroll
                                    for (int r = 0; r < 16; r += 4) vmax = fmaxf(vmax, fmaxf(fmaxf(s[kb][r], s[kb][r + 1]), fmaxf(s[kb][r + 2], s[kb][r + 3])));
                                }
                                if (f_min) {
#pragma unroll
                                    for (int r = 0; r < 16; r += 4) vmin = fminf(vmin, fminf(fminf(s[kb][r], s[kb][r + 1]), fminf(s[kb][r + 2], s[kb][r + 3])));
                                }
                                if (f_sum) {
#pragma unroll
                                    for (int r = 0; r < 16; ++r) {       // (product and sum rounded separately, as in the select form -- and in pww_qk_reduce)
                                        const float x = s[kb][r];
                                        vsum += x;
                                        vsq = add_square_unfused(vsq, x);
                                    }
                                }
                            } else {
                                if (f_max || f_min) {
#pragma unroll
                                    for (int r = 0; r < 16; ++r) {
                                        const bool live = qvalid && (key0 + key_of(kb, r, hi) < p.M);
                                        const float x = s[kb][r];
                                        vmax = fmaxf(vmax, live ? x : -INFINITY);
                                        vmin = fminf(vmin, live ? x : INFINITY);
                                    }
                                }
                                if (f_sum) {
#pragma unroll
                                    for (int r = 0; r < 16; ++r) {
                                        const bool live = qvalid && (key0 + key_of(kb, r, hi) < p.M);
                                        const float x = s[kb][r];
                                        vsum += live ? x : 0.f;
                                        vsq += live ? x * x : 0.f;
                                    }
                                }
                            }
                        }
                    }
                }
            }
            if (!lazy_ext) {
                if (f_max) {
#pragma unroll
                    for (int off = 32; off >= 1; off >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, off));
                }
                if (f_min) {
#pragma unroll
                    for (int off = 32; off >= 1; off >>= 1) vmin = fminf(vmin, __shfl_xor(vmin, off));
                }
#pragma unroll
                for (int off = 32; off >= 1; off >>= 1) { vsum += __shfl_xor(vsum, off); vsq += __shfl_xor(vsq, off); }
                // (the groups alternate between two buffers: thread t still reads group g while the waves already park group g + 1)
                if (lane == 0) {
                    float *rp = red + ((((it / RED_BLKS) & 1) * RED_BLKS + it % RED_BLKS) * NW + wave) * 4;
                    rp[0] = vmax; rp[1] = vmin; rp[2] = vsum; rp[3] = vsq;
                }
                if (it % RED_BLKS == RED_BLKS - 1) fold_group(it / RED_BLKS, RED_BLKS);
            }
            if constexpr (!SINGLE) {
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) { qf[ks] = q1[ks]; q1[ks] = q2[ks]; q2[ks] = q3[ks]; }
                request_q(q3, qb + 4 * nchunk);
            }
        }
        if (!lazy_ext) {
            if (it % RED_BLKS) fold_group(it / RED_BLKS, it % RED_BLKS);
        } else {
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) { vmax = fmaxf(vmax, __shfl_xor(vmax, off)); vmin = fminf(vmin, __shfl_xor(vmin, off)); }
            if (lane == 0) { red[wave * 4 + 0] = vmax; red[wave * 4 + 1] = vmin; }
            __syncthreads();
            double dmax = -INFINITY, dmin = INFINITY;
            for (int w = 0; w < NW; ++w) { dmax = fmax(dmax, (double)red[w * 4 + 0]); dmin = fmin(dmin, (double)red[w * 4 + 1]); }
            for (int t = tid; t < it; t += NT) {        // one slot set per block of this workgroup, all with the workgroup's extremes
                unsigned long long *slot = cp.slots + ((long)b * cp.nqb * p.H + (long)(chunk + t * nchunk) * p.H + h) * 4;
                slot_publish(slot + 0, dmax); slot_publish(slot + 1, dmin);
                slot_publish(slot + 2, 0.0); slot_publish(slot + 3, 0.0);
            }
        }
        tl_stamp(p, 2);
        if constexpr (!SINGLE) {   // pass 2 starts over at the first block: its fragments and its bias rows are requested before the hand-off
            __builtin_amdgcn_s_waitcnt(WAIT_VMCNT0);     // (nothing is in flight any more -- the ring's last requests lie past the last block -- and hipcc should know)
            request_q(qf, chunk);
            if (use_glds) tile_glds(trel, bias_srd4, tile, (long)chunk * NW * 32, p.b_sn, tcpr, wave);
        }
        // ---- hand-off + fold: every workgroup folds the image's partials itself (the order of pww_qk_reduce's
        // last-arriver fold), re-reading them until none is empty
        {
            const int bpi = cp.nqb * p.H;
            const unsigned long long *base = cp.slots + (long)b * bpi * 4;
            double dmax, dmin, dsum, dsq;
            const unsigned long long t0 = wall_clock64();
            bool complete;
            for (;;) {
                dmax = -INFINITY; dmin = INFINITY; dsum = 0.0; dsq = 0.0;
                complete = true;
                for (int i = tid; i < bpi; i += NT) {
                    // (a block's four words are published together; any one of them tells that the block has arrived)
                    const unsigned long long one = ~0ull;
                    const unsigned long long x0 = f_max ? slot_read(base + i * 4 + 0) : one, x1 = f_min ? slot_read(base + i * 4 + 1) : one;
                    const unsigned long long x2 = f_sum ? slot_read(base + i * 4 + 2) : one, x3 = f_sq ? slot_read(base + i * 4 + 3) : one;
                    complete = complete && x0 != 0ull && x1 != 0ull && x2 != 0ull && x3 != 0ull;
                    if (f_max) dmax = fmax(dmax, slot_value(x0));
                    if (f_min) dmin = fmin(dmin, slot_value(x1));
                    if (f_sum) dsum += slot_value(x2);
                    if (f_sq) dsq += slot_value(x3);
                }
                const bool expired = wall_clock64() - t0 > SPIN_LIMIT_TICKS;
                if (__syncthreads_and(complete || expired)) break;
                __builtin_amdgcn_s_sleep(1);
            }
            const int ok = __syncthreads_and(complete);
            if (tid == 0) {
                // leaving is counted per head (a few dozen workgroups per word, not hundreds); the returned count is only
                // looked at when the workgroup is done
                if (ok) depart_prev = __hip_atomic_fetch_add(cp.sync + b * p.H + h, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                else __hip_atomic_store(cp.sync + p.B * p.H + p.B, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // error word
                *ok_flag = ok;
            }
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) {
                dmax = fmax(dmax, __shfl_xor(dmax, off));
                dmin = fmin(dmin, __shfl_xor(dmin, off));
                dsum += __shfl_xor(dsum, off);
                dsq += __shfl_xor(dsq, off);
            }
            if (lane == 0) { fin[wave * 4 + 0] = dmax; fin[wave * 4 + 1] = dmin; fin[wave * 4 + 2] = dsum; fin[wave * 4 + 3] = dsq; }
            __syncthreads();
            if (tid == 0) {
                for (int w = 1; w < NW; ++w) {
                    dmax = fmax(dmax, fin[w * 4 + 0]); dmin = fmin(dmin, fin[w * 4 + 1]);
                    dsum += fin[w * 4 + 2]; dsq += fin[w * 4 + 3];
                }
                final_st[0] = dmax; final_st[1] = dmin; final_st[2] = dsum; final_st[3] = dsq;
                if (cp.stats_out && h == 0 && chunk == 0) {
                    double *st = cp.stats_out + (long)b * 4;
                    st[0] = dmax; st[1] = dmin; st[2] = dsum; st[3] = dsq;
                }
            }
            __syncthreads();
            const double st[4] = {final_st[0], final_st[1], final_st[2], final_st[3]};
            coeff = stat_coefficient(coeff_scalar_of(p), p.stat_kind, st, p.stat_count);
            if (p.bias_coeff) coeff = coeff * gate;
            if (!*ok_flag) coeff = __builtin_nanf("");     // a hand-off that timed out must not look like a result
        }
        __builtin_amdgcn_s_setprio(0);
        tl_stamp(p, 3);
    }
#endif
    else if (biased) {
        coeff = coeff_scalar_of(p);
        if (p.bias_coeff) coeff = coeff * gate;
        if constexpr (!SINGLE)
            if (use_glds) tile_glds(trel, bias_srd4, tile, (long)chunk * NW * 32, p.b_sn, tcpr, wave);
    }

    // ---- pass 2: bias -> softmax -> PV per query block
    // Dense bias rows with several blocks per workgroup (use_glds): every wave copied ITS rows of block i into LDS (LDS-direct) while
    // it computed block i - 1, and waits for those copies itself (vmcnt) at the end of block i - 1, before it issues that block's output
    // stores (a later wait would also wait for the stores). No barrier: the waves of a workgroup drift apart freely. With a single
    // buffer (tile_nbuf == 1) a wave can only start the copy when it is done with the previous block: its latency shows.
    const float c1 = p.scale_log2e;
    const bool glds_on = !SINGLE && use_glds && biased;          // workgroup-uniform
    const bool two_buf = glds_on && cp.tile_nbuf == 2;
    if constexpr (!SINGLE) {
        if (glds_on) __builtin_amdgcn_s_waitcnt(WAIT_VMCNT0);     // the first block's rows (requested before the hand-off) have landed
    }
    V8 qn[KS];
    int it2 = 0;
    for (int qb = chunk; qb < cp.nqb; qb += nchunk, ++it2) {
        const int qrow = (qb * NW + wave) * 32 + l31;
        const bool qvalid = qrow < p.N;
        const char *cur_tile = tile + (two_buf && (it2 & 1) ? tile_bytes : 0);
        if constexpr (!SINGLE) {
            if (glds_on && !two_buf && it2 > 0) tile_glds(trel, bias_srd4, tile, (long)qb * NW * 32, p.b_sn, tcpr, wave);
            if (two_buf && qb + nchunk < cp.nqb)
                tile_glds(trel, bias_srd4, tile + ((it2 & 1) ? 0 : tile_bytes), (long)(qb + nchunk) * NW * 32, p.b_sn, tcpr, wave);
            request_q(qn, qb + nchunk);
            // compact form: this block's values are requested now, they land under the first sub-tile's score MFMAs
            if (biased && use_compact) tile_request<NT>(treg, true, cbase, bias.srd, (long)qb * NW * 32, p.N, cp.c_sn, p.b_sn, cp.R, p.bias_cols, tid);
        }
        if (biased && !use_tile) bias.row_off = qvalid ? (unsigned)((long)qrow * p.b_sn * 4) : OOB_OFF;
        // one key stage (M <= 128): the FIRST 64-key tile sets the row's reference, starts O^T from a zero constant and has nothing to
        // rescale (STEP 1); the second one (13 live keys of the 77 prompt tokens) keeps that reference unless a score of it exceeds it by
        // 2^8 (STEP 2: with 32 rows per wave the exact online step rescaled O^T in 997 of 1000 waves, for a handful of keys)
        f32x16 oacc[DT];
        float m_run = -INFINITY, l_run = 0.f;
        if (biased && use_tile) {
            f32x16 s0[2];
            score_tile<T, KS>(s0, qf, smem, 0, p.M, l31, hi);
            if constexpr (!SINGLE) {
                if (use_compact) {
                    __syncthreads();            // every wave is done reading the previous block's rows
                    tile_park<NT>(treg, true, tile, cp.tile_stride, cidx, bias.srd, (long)qb * NW * 32, p.N, p.b_sn, cp.R, p.bias_cols, tid);
                    __syncthreads();
                } else if (!two_buf && it2 > 0) {
                    __builtin_amdgcn_s_waitcnt(WAIT_VMCNT0);
                }
                if (use_glds) bias_ref_tile(bias, cur_tile, wave * 32 + l31, cp.tile_stride, p.bias_cols, hi);
            }
            // (with the 77 prompt tokens the first 64 keys are all live: no per-score key compare there)
            if (p.M >= KVBLK) attn_tile_sm_pv<T, KS, DT, 2, false, RSM, 1>(s0, oacc, m_run, l_run, smem + KT::BYTES, 0, p.M, l31, hi, bias, coeff, c1);
            else attn_tile_sm_pv<T, KS, DT, 2, true, RSM, 1>(s0, oacc, m_run, l_run, smem + KT::BYTES, 0, p.M, l31, hi, bias, coeff, c1);
            if (KVBLK < p.M)
                attn_tile<T, KS, DT, 2, true, RSM, 2>(oacc, m_run, l_run, qf, smem + SUB_BYTES, smem + SUB_BYTES + KT::BYTES, KVBLK, p.M, l31, hi, bias, coeff, c1);
        } else {
            static_assert(NSUB == 2, "one key stage of two 64-key tiles");
            const char *Ks1 = smem + SUB_BYTES;
            if (biased) {
                // bias rows read per lane from global memory (maps wider than the LDS tile, or a strided key axis): the fall-back form, kept
                // on the general online step in ONE instantiation (a second one makes hipcc hoist 32 strided offsets into scratch)
#pragma unroll
                for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) oacc[dt][r] = 0.f;
#pragma unroll
                for (int sub = 0; sub < NSUB; ++sub) {
                    const char *Ks = smem + sub * SUB_BYTES;
                    if (sub * KVBLK < p.M) attn_tile<T, KS, DT, 1, true, RSM>(oacc, m_run, l_run, qf, Ks, Ks + KT::BYTES, sub * KVBLK, p.M, l31, hi, bias, coeff, c1);
                }
            } else {
                // (a full first tile -- the 77 prompt tokens -- needs no key compares)
                if (KVBLK <= p.M) attn_tile<T, KS, DT, 0, false, RSM, 1>(oacc, m_run, l_run, qf, smem, smem + KT::BYTES, 0, p.M, l31, hi, bias, coeff, c1);
                else attn_tile<T, KS, DT, 0, true, RSM, 1>(oacc, m_run, l_run, qf, smem, smem + KT::BYTES, 0, p.M, l31, hi, bias, coeff, c1);
                if (KVBLK < p.M) attn_tile<T, KS, DT, 0, true, RSM, 2>(oacc, m_run, l_run, qf, Ks1, Ks1 + KT::BYTES, KVBLK, p.M, l31, hi, bias, coeff, c1);
            }
        }
        if constexpr (!SINGLE) {
            // the next block's fragments and bias rows have had this block's compute to arrive: wait for them HERE, before the output
            // stores are issued
            if (two_buf) __builtin_amdgcn_s_waitcnt(WAIT_VMCNT0);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) qf[ks] = qn[ks];
        }
        float l_tot;
        if (RSM) {      // row D of O^T: tile D / 32, register (D % 32) / 2 (D is a multiple of 8 below 32 DT), held by the hi == 0 half
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
    }
    tl_stamp(p, 4);

    // ---- the last workgroup of an image to leave puts the image's state words back to zero. (Every other workgroup of
    // the image has finished reading the slots: a workgroup leaves only after its fold. Last of its head -> bumps the
    // image's heads_left word; last of those -> everybody is out.)
#if PWW_EXPERIMENTS
    if (need_stat && !cp.ext_part) {
        if (tid == 0) {
            int last = 0;
            if (*ok_flag && depart_prev == (unsigned)nchunk - 1u) {
                const unsigned prev = __hip_atomic_fetch_add(cp.sync + p.B * p.H + b, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                last = prev == (unsigned)p.H - 1u;
            }
            *ok_flag = last;
        }
        __syncthreads();
        if (*ok_flag) {
            const int nslot = cp.nqb * p.H * 4;
            unsigned long long *base = cp.slots + (long)b * nslot;
            for (int i = tid; i < nslot; i += NT) __hip_atomic_store(base + i, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            for (int i = tid; i < p.H; i += NT) __hip_atomic_store(cp.sync + b * p.H + i, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (tid == 0) __hip_atomic_store(cp.sync + p.B * p.H + b, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
#endif
    tl_stamp(p, 5);
    tl_cycles(p, tl_c0);
}

// ---- host side -------------------------------------------------------------------------------

int attn_fwd(const void *q, const void *k, const void *v, void *o, const float *bias, const float *bias_coeff,
             const pww_attn_desc_t *d, hipStream_t stream, const double *stats, int stat_kind, double stat_count,
             float coeff_scalar, const float *coeff_scalar_dev);
int attn_validate(const void *q, const void *k, const void *v, void *o, const float *bias, const pww_attn_desc_t *d);
void attn_fill_params(AttnParams &p, const void *q, const void *k, const void *v, void *o, const float *bias,
                      const float *bias_coeff, const pww_attn_desc_t *d);
int qk_reduce(const void *q, const void *k, const pww_attn_desc_t *d, double *stats, void *workspace, size_t workspace_bytes,
              hipStream_t stream);
size_t qk_reduce_workspace_bytes(const pww_attn_desc_t *d);
bool attn_wide_groups(const pww_attn_desc_t *d);
int cross_attn_lean(const void *q, const void *k, const void *v, void *o, const float *bias, int stat_kind, float coeff_scalar, const float *gate,
                    const pww_attn_desc_t *d, double *stats_out, const pww_cross_opts_t &op, hipStream_t stream, const double *parts, int nparts,
                    bool *launched);

static int current_device() { int dev = 0; return hipGetDevice(&dev) == hipSuccess ? dev : -1; }

static int device_cus() {      // per device (a thread may drive several GPUs)
    static thread_local int cus = 0, cus_dev = -1;
    const int dev = current_device();
    if (!cus || dev != cus_dev) {
        hipDeviceProp_t prop;
        if (dev < 0 || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0;
        cus = prop.multiProcessorCount;
        cus_dev = dev;
    }
    return cus;
}

static int fused_wg_cap() {   // PWW_DEBUG=cross_wg_per_cu=n: upper bound on the resident workgroups per CU the fused launch counts on (A/B testing)
    static int cap = -1;
    if (cap < 0) { cap = debug_knobs().cross_wg_per_cu; if (cap < 1) cap = 1; }
    return cap;
}
static int fused_assume_resident() {   // PWW_DEBUG=cross_assume_resident=n: TEST HOOK -- count on n workgroups per CU whatever the occupancy query says
    static int n = -1;                 // (tests/test_round3_gpu.py drives the hand-off's time-out / error-word path with it)
    if (n < 0) { n = debug_knobs().cross_assume_resident; if (n < 0) n = 0; }
    return n;
}

// PWW_DEBUG=cross_bias_lds=n (A/B testing): 0 = per-lane global bias loads as in round 2; 1 = LDS tile only in launches with one query block per
// workgroup (register-staged), several blocks per workgroup keep the per-lane loads; 2 (default) = LDS tile everywhere (several blocks
// per workgroup: LDS-direct copies)
static int bias_tile_mode() {
    static int mode = -2;
    if (mode == -2) { mode = debug_knobs().cross_bias_lds; }
    return mode;
}
static int gate_balance_weight() {   // PWW_DEBUG=cross_gate_weight=n: cost of a gated-in image's query block in units of a gated-out one's (default 2 -- measured best of 2 / 3 / 4; 1 = ignore the hint)
    static int w = -1;
    if (w < 0) { w = debug_knobs().cross_gate_weight; if (w < 1) w = 1; }
    return w;
}
static int tile_nbuf_mode() {   // PWW_DEBUG=cross_tile_nbuf=1: never double-buffer the bias tile (A/B testing); default 2: when it costs no residency
    static int mode = -1;
    if (mode < 0) { mode = debug_knobs().cross_tile_nbuf; if (mode != 1) mode = 2; }
    return mode;
}

template <typename T, int KS, int DT, int NW, bool COMPACT>
static int launch_cross(CrossParams cp, hipStream_t stream, bool *launched) {
    constexpr size_t lds_fixed = 2 * (KTile<KS>::BYTES + VTile<DT>::BYTES) + NW * 4 * 8 + 4 * 8 + 2 * RED_BLKS * NW * 4 * 4 + 16 + COMPACT_MAX_R * 4;
    const size_t tile_bytes = (size_t)NW * 32 * cp.tile_stride * 4;
    auto k_single = cross_fused_kernel<T, KS, DT, NW, true, COMPACT>;
    auto k_multi = cross_fused_kernel<T, KS, DT, NW, false, COMPACT>;
    // resident workgroups per CU for an LDS size (the tile width is a run-time value): asked once per size
    static thread_local size_t lds_seen[16];
    static thread_local int per_cu_seen[16];
    static thread_local int n_seen = 0;
    static thread_local size_t lds_attr = 0;
    static thread_local int seen_dev = -1;       // the answers and the function attribute are per DEVICE: start over when the thread switched GPUs
    auto resident = [&](size_t lds, int *per_cu_out) -> int {
        const int dev = current_device();
        if (dev != seen_dev) { n_seen = 0; lds_attr = 0; seen_dev = dev; }
        for (int i = 0; i < n_seen; ++i) if (lds_seen[i] == lds) { *per_cu_out = per_cu_seen[i]; return PWW_OK; }
        // (the kernels also own a few hundred bytes of static LDS: stay clear of the 160 KiB a workgroup can have)
        if (lds > 156 * 1024) { *per_cu_out = 0; return PWW_OK; }
        if (lds > 64 * 1024 && lds > lds_attr) {
            for (auto kern : {k_single, k_multi})
                if (check_hip(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds),
                              "hipFuncSetAttribute"))
                    return PWW_EHIP;
            lds_attr = lds;
        }
        int n1 = 0, n2 = 0;
        if (check_hip(hipOccupancyMaxActiveBlocksPerMultiprocessor(&n1, k_single, NW * 64, lds), "hipOccupancyMaxActiveBlocksPerMultiprocessor") ||
            check_hip(hipOccupancyMaxActiveBlocksPerMultiprocessor(&n2, k_multi, NW * 64, lds), "hipOccupancyMaxActiveBlocksPerMultiprocessor"))
            return PWW_EHIP;
        // every workgroup of the launch must be resident at once (the hand-off spins on partials of workgroups that have to be
        // running). The occupancy answer can be one high near an SGPR edge when it says 7 or 8 (MI355X_MICROARCH.md); this
        // kernel's answers are LDS / VGPR bound (<= 4) and taken as they are, up to 4.
        int per_cu = n1 < n2 ? n1 : n2;
        if (per_cu >= 7) per_cu -= 1;
        if (per_cu > fused_wg_cap()) per_cu = fused_wg_cap();
        if (fused_assume_resident()) per_cu = fused_assume_resident();
        if (n_seen < 16) { lds_seen[n_seen] = lds; per_cu_seen[n_seen] = per_cu; ++n_seen; }
        *per_cu_out = per_cu;
        return PWW_OK;
    };
    size_t lds = lds_fixed + tile_bytes;
    int per_cu = 0;
    if (int rc = resident(lds, &per_cu)) return rc;
    const AttnParams &p = cp.a;
    const long BH = (long)p.B * p.H;
    cp.nqb = (p.N + NW * 32 - 1) / (NW * 32);
    const long cap = (long)per_cu * device_cus();
    const bool ext = cp.pass2_only != 0;      // partials came with Q (or none are needed): nothing waits for another workgroup, any grid is correct
    if (cap < BH && !ext) { *launched = false; return PWW_OK; }      // cannot be made resident: the caller takes the two-launch path
    long nchunk = cap / BH;
    if (nchunk < 1) nchunk = 1;
    if (nchunk > cp.nqb) nchunk = cp.nqb;
    cp.nchunk = cp.nchunk_u = (int)nchunk;
    const int hint = cp.n_gated;
    cp.n_gated = 0;
    if (hint > 0 && hint < p.B && nchunk < cp.nqb && gate_balance_weight() > 1) {
        // The caller told which images carry the statistic and the bias (the conditional rows of a CFG-folded batch). A query block
        // of theirs costs about three times one of the others (pass 1 + pass 2 with bias, vs pass 2 alone -- measured 7.2 vs 2.6 us),
        // and a workgroup's lifetime is what the launch lasts: give those images c workgroups per head and the others u so that both
        // kinds finish together, within the resident capacity (g * c + (B - g) * u <= cap / H).
        const long per_head = cap / p.H, g = hint, rest = p.B - hint;
        // cost of a gated-in query block in HALVES of a gated-out one: pass 1 + pass 2 with bias = 2 x (fused), pass 2 with bias alone
        // = 1.5 x (external partials: 3.5 vs 2.6 us measured per block in round 3); + the hand-off (fused only)
        const int w2 = ext ? 3 : 2 * gate_balance_weight();
        const long extra2 = ext ? 0 : 4;
        long best_c = nchunk, best_u = nchunk, best_cost = -1;
        for (long c = 1; c <= cp.nqb; ++c) {
            long u = (per_head - g * c) / rest;
            if (u < 1) break;
            if (u > cp.nqb) u = cp.nqb;
            const long bc = (cp.nqb + c - 1) / c, bu = (cp.nqb + u - 1) / u;
            const long cost = bc * w2 + extra2 > 2 * bu ? bc * w2 + extra2 : 2 * bu;
            if (best_cost < 0 || cost < best_cost) { best_cost = cost; best_c = c; best_u = u; }
        }
        if (best_c != best_u) { cp.n_gated = hint; cp.nchunk = (int)best_c; cp.nchunk_u = (int)best_u; }
    }
    const long n_wgs = cp.n_gated ? (long)p.H * (cp.n_gated * (long)cp.nchunk + (p.B - cp.n_gated) * (long)cp.nchunk_u) : BH * nchunk;
    cp.head_major = ((n_wgs / p.H) % 8 == 0 && debug_knobs().cross_head_major) ? 1 : 0;
    const bool single = cp.nchunk == cp.nqb && cp.nchunk_u == cp.nqb;
    cp.tile_nbuf = 1;
    if (!COMPACT && !single && cp.tile_stride > 0 && bias_tile_mode() == 1) {
        // (A/B switch: per-lane loads in multi-block launches; smaller LDS, so the residency answer above still holds)
        lds -= tile_bytes;
        cp.tile_stride = 0;
    }
    if (!single && cp.tile_stride > 0 && !cp.compact && tile_nbuf_mode() == 2) {
        // several blocks per workgroup, dense tile: a second buffer lets the next block's rows load under this block's compute --
        // taken when it does not cost a resident workgroup per CU
        int per_cu2 = 0;
        if (int rc = resident(lds + tile_bytes, &per_cu2)) return rc;
        if (per_cu2 >= per_cu) { cp.tile_nbuf = 2; lds += tile_bytes; }
    }
    const dim3 grid((unsigned)n_wgs);
    if (single) launch_attn_kernel(k_single, grid, dim3(NW * 64), lds, stream, cp);
    else launch_attn_kernel(k_multi, grid, dim3(NW * 64), lds, stream, cp);
    *launched = true;
    return check_hip(hipGetLastError(), "cross_fused_kernel launch");
}

template <typename T, int NW, bool COMPACT> static int dispatch_cross_d(const CrossParams &cp, hipStream_t s, bool *launched) {
    const int D = cp.a.D;
    if (D <= 48) return launch_cross<T, 3, 2, NW, COMPACT>(cp, s, launched);
    if (D <= 64) return launch_cross<T, 4, 2, NW, COMPACT>(cp, s, launched);
    if (D <= 80) return launch_cross<T, 5, 3, NW, COMPACT>(cp, s, launched);
    if (D <= 96) return launch_cross<T, 6, 3, NW, COMPACT>(cp, s, launched);
    if constexpr (NW == 2) { set_error("cross_attn_fused: internal dispatch error"); return PWW_EINVAL; } else {
        if (D <= 128) return launch_cross<T, 8, 4, NW, COMPACT>(cp, s, launched);
        return launch_cross<T, 10, 5, NW, COMPACT>(cp, s, launched);
    }
}
template <typename T, int NW> static int dispatch_cross_form(const CrossParams &cp, hipStream_t s, bool *launched) {
#if PWW_EXPERIMENTS      // the compact bias form (measured 0.5 - 6 us slower per launch than the dense tile: opt-in since round 3) doubles the instantiations
    if (cp.compact) return dispatch_cross_d<T, NW, true>(cp, s, launched);
#endif
    return dispatch_cross_d<T, NW, false>(cp, s, launched);
}

}  // namespace pww
