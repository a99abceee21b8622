// Device-side building blocks shared by the fused attention kernels (pww_attn.hip: general flash-style kernel;
// pww_cross.hip: single-stage cross-attention with the score statistic formed in the same launch): LDS tile
// geometry, buffer-load staging, the bias reference, and one 64-key tile of scores -> (bias) -> online softmax -> PV.
#pragma once
#include <stdlib.h>
#include "pww_tile.h"

namespace pww {

struct AttnParams {
    const void *q, *k, *v;
    void *o;
    const float *bias;
    const float *bias_coeff;
    int B, H, N, M, D;
    long q_sb, q_sh, q_sn;
    long k_sb, k_sh, k_sm;
    long v_sb, v_sh, v_sm;
    long o_sb, o_sh, o_sn;
    long b_sb, b_sh, b_sn, b_sm;
    float scale_log2e;  // scale * log2(e): softmax runs in the exp2 domain
    const double *stats;   // optional [B][4] from qk_reduce: the per-image coefficient is formed here
    int stat_kind;
    double stat_count;
    float coeff_scalar;
    const float *coeff_scalar_dev;   // optional device word that REPLACES coeff_scalar (read when the kernel runs: one captured
                                     // hipGraph serves every denoise step, the host rewrites the word before each replay)
    int bias_cols;                   // columns >= bias_cols of the bias map are zero (multiple of 16, <= M rounded up); 0 = unknown
    int o_wide;                      // 1: rows of O are 16-byte aligned (o strides % 8 == 0): 16-byte epilogue stores
    int pair_major;                  // 1: workgroups of one (image, head) pair run on ONE XCD, pairs dealt round-robin to the XCDs; 2 / 4: groups of 2 / 4 adjacent heads per XCD
    unsigned long long *timeline;    // debug: per-workgroup phase time stamps (pww_debug_timeline), normally null
    unsigned timeline_wgs;           // workgroups the debug buffer has room for
    // folded-reference self-attention (d = 40): the magnitude guard's limit on |row maximum| in exp2 units (FoldLimit<T>), the first-stage row
    // sum below which an f16 workgroup leaves the range-free mode (0: never, < 0: lazy from the start), and three debug counters
    // { range-free, lazy, exact } workgroups (pww_debug_path_counts), normally null
    float fold_limit, hot_sum;
    unsigned *path_counts;
};

// Magnitude guard of the folded-reference d = 40 kernel (FoldLimit<T>, pww_attn_kernel.h), exp2 units. Round 6, from the sweep of
// tools/diag_hot_f16.py (profiles/r06_hot_logits.md: error against fp64 on sampled rows, N = 4096, scaled-logit std 1 ... 8): f16 0.8e-3 of max|O| at
// row maxima of 22 natural units, 1.2e-3 at 33, 1.55e-3 at 36 (bar 2e-3) -> 48 (= 33 natural units; was 36 = 25: every workgroup of a
// std-5 input took the exact path, after a complete fast pass); bf16 1.05e-2 at 35 natural units, 1.1 - 1.6e-2 at 39 - 41 (bar 1.6e-2) ->
// 56 (= 39; was 72 = 50, extrapolated from two points in round 2).
constexpr float FOLD_LIMIT_BF16 = 56.f, FOLD_LIMIT_F16 = 48.f;

// the Python scalar c0 * g(sigma) of the weight function: baked into the launch, or read from a device word
__device__ __forceinline__ float coeff_scalar_of(const AttnParams &p) {
    return p.coeff_scalar_dev ? *p.coeff_scalar_dev : p.coeff_scalar;
}

// Workgroup -> ((image, head) pair, query block). Default: pairs fastest -- consecutive workgroups (= consecutive XCDs) work on
// different pairs, every pair's query blocks are spread over all XCDs and each XCD's L2 streams the K/V of every pair it meets.
// That is fine while all pairs' K/V fit the L2s together (B = 2: 10.5 MB, traffic 1.2x algorithmic); with 128 pairs (16 folded
// rows) each XCD runs 32 workgroups of 16 different pairs at a time and re-fetches 655 KB of K/V per pair and query block from the
// Infinity Cache (5.4x algorithmic, profiles/r03_traffic.json). pair_major: XCD x (= blockIdx & 7) takes the pairs x, x + 8, ...
// one after the other, all query blocks of a pair together -- a pair's K/V is fetched into one L2 once.
__device__ __forceinline__ void wg_to_pair_block(const AttnParams &p, int nqb, int &bh, int &qb) {
    const int BH = p.B * p.H;
    if (p.pair_major >= 2) {
        // Head dims whose slices are not whole 128-byte lines (d = 40: 80 bytes of a 640-byte row): an L2 read request moves the WHOLE line
        // (tools/ubench_linefill.cpp: a 32-, 64- or 80-byte touch costs what 128 bytes cost), and with every head on its own XCD the eight
        // L2s fetch a row's 5 lines 12 times -- 2.4x on the read side, which rounds 2 - 5 read as 1.2x because FETCH_SIZE tallies a
        // request at 64 bytes (profiles/r06_sector_sharing.md). Here XCD x takes GROUPS of G = 2 (4) adjacent heads -- 8 (6) line fills
        // per row -- the heads of a group alternating query block by query block; a group's K / V (1.3 / 2.6 MB at N = 4096) fit its L2.
        const int lg = p.pair_major == 4 ? 2 : 1, G = 1 << lg;
        const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
        const int t = j % (G * nqb);
        qb = t >> lg;
        bh = (((j / (G * nqb)) * 8 + xcd) << lg) + (t & (G - 1));
    } else if (p.pair_major) {
        const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
        qb = j % nqb;
        bh = (j / nqb) * 8 + xcd;
    } else {
        bh = blockIdx.x % BH;
        qb = blockIdx.x / BH;
    }
}

// debug time stamps (100 MHz wall clock), [workgroup][TL_SLOTS]
constexpr int TL_SLOTS = 8;
// slot 7: shader cycles (s_memtime) between kernel entry and exit of the workgroup's first wave: with the wall-clock stamps it
// gives the clock the launch actually ran at
__device__ __forceinline__ void tl_cycles(const AttnParams &p, unsigned long long c0) {
    if (p.timeline && threadIdx.x == 0 && blockIdx.x < p.timeline_wgs) p.timeline[(long)blockIdx.x * TL_SLOTS + 7] = clock64() - c0;
}
__device__ __forceinline__ void tl_stamp(const AttnParams &p, int slot) {
    if (p.timeline && threadIdx.x == 0 && blockIdx.x < p.timeline_wgs) p.timeline[(long)blockIdx.x * TL_SLOTS + slot] = wall_clock64();
}

// c * stat(st), the statistic selected from one image's { max, min, sum, sum of squares } (fp64) and rounded to fp32 first
__device__ __forceinline__ float stat_coefficient(float c, int kind, const double *st, double count) {
    double v;
    switch (kind) {
        case PWW_STAT_MAX: v = st[0]; break;
        case PWW_STAT_MIN: v = st[1]; break;
        case PWW_STAT_MEAN: v = st[2] / count; break;
        case PWW_STAT_ABSMAX: v = fmax(fabs(st[0]), fabs(st[1])); break;
        default: {   // PWW_STAT_STD: unbiased, from sum and sum of squares in fp64
            const double n = count;
            const double var = (st[3] - st[2] * st[2] / n) / (n > 1.0 ? n - 1.0 : 1.0);
            v = var > 0.0 ? var : 0.0;
        }
    }
    float f = (float)v;
    if (kind == PWW_STAT_STD) f = sqrtf(f);
    return c * f;
}

// c[b] = coeff_scalar * stat(stats[b]) * gate[b]   (fp32 products in this order: what the host's elementwise ops gave)
__device__ __forceinline__ float bias_coefficient(const AttnParams &p, int b) {
    float c = coeff_scalar_of(p);
    if (p.stat_kind != PWW_STAT_NONE) c = stat_coefficient(c, p.stat_kind, p.stats + (long)b * 4, p.stat_count);
    if (p.bias_coeff) c = c * p.bias_coeff[b];
    return c;
}

// V tile as staged in LDS: ROW-MAJOR like K, [64 keys][DT*32 d], and transposed by the READ: gfx950's
// ds_read_b64_tr_b16 hands lane i of a 16-lane group column i of a 4-row x 16-column block whose rows are supplied,
// four 8-byte pieces each, by lanes 4r..4r+3 (probed on hardware: tools/tr_probe.cpp). So the PV MFMA's A operand
// (V^T[d = lane & 31][8 consecutive keys]) is two such reads, and staging V costs what staging K costs: 16-byte
// global loads and 16-byte LDS stores, no per-element transposition work (16 v_perm + 8 ds_write_b64 per 4x8 block
// before, carried by 3 of 8 waves -- the stragglers every barrier waited for; profiles/r01_attn_phases.md 2b).
// Row stride = 64 or 192 (mod 256) bytes, so the four rows a 32-lane half reads in one LDS cycle (64 bytes each)
// fall into disjoint bank ranges.
template <int DT> struct VTile {
    static constexpr int COLS = DT * 32;               // head dim padded to the MFMA M granularity
    static constexpr int CHK = COLS / 8;               // 16-byte chunks per key row (those with d < D are staged)
    static constexpr int STRIDE = (DT & 1) ? DT * 64 : DT * 64 + 64;   // bytes per key row: 64 / 192 / 192 / 320 / 320
    static constexpr int BYTES = KVBLK * STRIDE;
    static constexpr int NCHUNK = KVBLK * CHK;
};

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

// byte offset, inside a V tile, of the 8-byte piece lane l supplies for keys 0..3 / d-tile 0 of its fragment
template <int DT> __device__ __forceinline__ int vfrag_lane_off(int lane) {
    return ((lane >> 5) * 8 + ((lane & 15) >> 2)) * VTile<DT>::STRIDE + (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;
}

// A-operand fragment of the PV MFMA: V[key0 .. key0+7][d = dt*32 + (lane & 31)] with key0 = kb*32 + k2*16 + 8*(lane>>5).
// `vl` = tile base + vfrag_lane_off(lane); the rest are compile-time immediates.
template <typename T, int DT>
__device__ __forceinline__ typename Vec<T>::v8 load_vfrag(const char *vl, int kb, int k2, int dt) {
    typedef __attribute__((address_space(3))) s16x4 *lds_p;
    const char *a = vl + (kb * 32 + k2 * 16) * VTile<DT>::STRIDE + dt * 64;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(a));
    const s16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(a + 4 * VTile<DT>::STRIDE));
    const s16x8 v = __builtin_shufflevector(lo, hi4, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(typename Vec<T>::v8, v);
}

// Loop-invariant part of a thread's share of the K/V staging: which 16-byte chunks it moves and where
// they land in an LDS stage buffer. A stage = NSUB sub-tiles of 64 keys, each sub-tile = [K tile | Vt tile].
//
// Global reads are BUFFER loads: a per-(b,h) resource descriptor (base = first key row of this head,
// num_records = last valid byte + 1), a loop-invariant per-thread 32-bit byte offset, plus the stage's byte
// offset. That keeps the hot loop free of 64-bit address arithmetic (v_mul_lo / v_mad_u64 are multi-cycle
// VOP3 ops: ~70 of them per stage before), and rows past the last key are OUT OF RANGE for the descriptor,
// so the hardware returns zeros for them -- no clamping, no selects, no special tail path. Idle threads and
// head-dim padding chunks carry an offset >= 2^31 (never in range; the host checks the extent is < 2^31).
constexpr unsigned OOB_OFF = 0x80000000u;

template <int KPT, int VPT> struct StagePlan {
    unsigned k_off[KPT];
    int k_lds[KPT];
    bool k_ok[KPT];
    unsigned v_off[VPT];
    int v_lds[VPT];
    bool v_ok[VPT];
};

template <typename T, int KS, int DT, int NT, int NSUB, int KPT, int VPT>
__device__ __forceinline__ void make_plan(StagePlan<KPT, VPT> &pl, int tid, int D, long k_sm, long v_sm) {
    typedef KTile<KS> KT;
    typedef VTile<DT> VT;
    constexpr int SUB_BYTES = KT::BYTES + VT::BYTES;
#pragma unroll
    for (int i = 0; i < KPT; ++i) {
        const int c = tid + i * NT;
        const int key = c / KT::CHK, ch = c - key * KT::CHK;       // key in [0, 64*NSUB)
        pl.k_ok[i] = c < NSUB * KT::NCHUNK && ch * 8 < D;            // padding chunks stay at their initial zeros
        pl.k_off[i] = pl.k_ok[i] ? (unsigned)((key * k_sm + ch * 8) * 2) : OOB_OFF;
        pl.k_lds[i] = (key >> 6) * SUB_BYTES + (key & 63) * KT::STRIDE + ch * 16;
    }
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
        const int c = tid + i * NT;
        const int key = c / VT::CHK, ch = c - key * VT::CHK;
        pl.v_ok[i] = c < NSUB * VT::NCHUNK && ch * 8 < D;            // padding columns are initialised once, never staged
        pl.v_off[i] = pl.v_ok[i] ? (unsigned)((key * v_sm + ch * 8) * 2) : OOB_OFF;
        pl.v_lds[i] = (key >> 6) * SUB_BYTES + KT::BYTES + (key & 63) * VT::STRIDE + ch * 16;
    }
}

// Issue the global loads of the stage whose first key row sits `k_stage_off` / `v_stage_off` bytes into the
// head's K / V. Nothing here consumes the loaded registers: the loads stay in flight across the compute of
// the current stage. (Unconditional on purpose: a load inside an `if` gets an s_waitcnt vmcnt(0) behind it.)
template <typename SRD, int KPT, int VPT>
__device__ __forceinline__ void stage_load(u32x4 (&kreg)[KPT], u32x4 (&vreg)[VPT], const StagePlan<KPT, VPT> &pl,
                                           SRD srd_k, SRD srd_v, unsigned k_stage_off, unsigned v_stage_off) {
#pragma unroll
    for (int i = 0; i < KPT; ++i) kreg[i] = __builtin_amdgcn_raw_buffer_load_b128(srd_k, pl.k_off[i] + k_stage_off, 0, 0);
#pragma unroll
    for (int i = 0; i < VPT; ++i) vreg[i] = __builtin_amdgcn_raw_buffer_load_b128(srd_v, pl.v_off[i] + v_stage_off, 0, 0);
}

// Registers -> LDS stage buffer (rows past the last key arrive as zeros from the buffer load).
template <int DT, int KPT, int VPT>
__device__ __forceinline__ void stage_store(const u32x4 (&kreg)[KPT], const u32x4 (&vreg)[VPT],
                                            const StagePlan<KPT, VPT> &pl, char *buf) {
#pragma unroll
    for (int i = 0; i < KPT; ++i)
        if (pl.k_ok[i]) *reinterpret_cast<u32x4 *>(buf + pl.k_lds[i]) = kreg[i];
#pragma unroll
    for (int i = 0; i < VPT; ++i)
        if (pl.v_ok[i]) *reinterpret_cast<u32x4 *>(buf + pl.v_lds[i]) = vreg[i];
}

// Resource descriptor over one head's K or V rows: [base, base + (M-1)*row_stride + D) elements of T.
template <typename T>
__device__ __forceinline__ auto head_srd(const T *base, int M, long row_stride, int D) {
    const unsigned bytes = (unsigned)(((long)(M - 1) * row_stride + D) * 2);
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<T *>(base), 0, bytes, 0x00020000);
}

// max over both half-waves of a per-lane value (lanes l and l^32 hold the two halves of one query row)
__device__ __forceinline__ float xhalf_max(float x) {
#if __has_builtin(__builtin_amdgcn_permlane32_swap)
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
#else
    return fmaxf(x, __shfl_xor(x, 32));
#endif
}

// Epilogue: normalise and write one 32-row block of O. Register r of tile dt is d = dt*32 + (r&3) + 8*(r>>2) + 4*hi of the lane's
// row: the two half-waves hold ADJACENT 8-byte pieces of the same row. `wide`: pair the column groups (g, g+1) with
// v_permlane32_swap -- afterwards the lower half holds 16 contiguous bytes at d = 8g, the upper half the next 16 -- and write one
// 16-byte store per pair instead of two 8-byte ones (the store tail of a row-per-lane epilogue is issue-bound, not bandwidth-bound:
// cdna_hip_programming.md T21). Needs 16-byte aligned rows (o strides % 8 == 0), else the 8-byte form. All 64 lanes must call
// (cross-lane exchange); rows that are not valid only skip the store.
template <typename T, int DT>
__device__ __forceinline__ void store_o_block(T *orow, const f32x16 (&oacc)[DT], float inv, int D, int hi, bool qvalid, bool wide) {
    typedef typename Vec<T>::v4 V4;
    if (wide) {
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
#pragma unroll
            for (int gp = 0; gp < 4; gp += 2) {
                if (dt * 32 + gp * 8 < D) {              // wave-uniform: the pair holds at least one stored column group
                    V4 a, b;
#pragma unroll
                    for (int j = 0; j < 4; ++j) { a[j] = (T)(oacc[dt][gp * 4 + j] * inv); b[j] = (T)(oacc[dt][gp * 4 + 4 + j] * inv); }
                    const u32x2 ua = __builtin_bit_cast(u32x2, a), ub = __builtin_bit_cast(u32x2, b);
                    const auto x = __builtin_amdgcn_permlane32_swap(ua[0], ub[0], false, false);
                    const auto y = __builtin_amdgcn_permlane32_swap(ua[1], ub[1], false, false);
                    const u32x4 out = {x[0], y[0], x[1], y[1]};
                    const int d = dt * 32 + 8 * (gp + hi);
                    if (qvalid && d < D) *reinterpret_cast<u32x4 *>(orow + d) = out;
                }
            }
        }
    } else if (qvalid) {
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int d = dt * 32 + g * 8 + hi * 4;
                if (d < D) {
                    V4 out;
#pragma unroll
                    for (int j = 0; j < 4; ++j) out[j] = (T)(oacc[dt][g * 4 + j] * inv);
                    *reinterpret_cast<V4 *>(orow + d) = out;
                }
            }
        }
    }
}

// Waves per SIMD the register allocator must leave room for (2 => <= 256 VGPRs+AGPRs). The bias
// variants (32 extra loads in flight per tile) and the widest heads keep the whole 512-entry file.
template <int DT, int NW, bool HAS_BIAS> struct MinWaves {
    static constexpr int value = (DT >= 4 || (NW == 2 && DT >= 3) || (HAS_BIAS && DT >= 3)) ? 1 : 2;   // (an 8-wave workgroup needs 2 waves per SIMD to fit at all)
};

// One KV tile: scores -> (bias) -> online softmax -> PV. MASKED tiles (only the last one can be)
// additionally kill keys >= M; full tiles skip every key compare.
// ROWSUM_MFMA: the head dim is not a multiple of 32, so the V tile has padding columns; column D holds
// ones and the PV MFMA accumulates the softmax denominator there for free (no per-element adds).
// Bias addressing for one lane: a buffer descriptor over this (b, h) slice of the bias, the byte offset of the
// lane's query row (>= 2^31, i.e. out of range -> zeros, for rows past N) and the key stride in bytes. With a unit
// key stride (the PwW [N, 77] maps) every bias load is `row offset + immediate`: one VGPR for all 32 loads of a tile.
struct BiasRef {
    __amdgpu_buffer_rsrc_t srd;
    unsigned row_off;
    unsigned key_stride;   // bytes
    bool unit;             // key_stride == 4
    // BIAS == 2: the query block's bias rows sit in LDS (pww_cross.hip stages the contiguous [rows, cols] span with coalesced
    // 16-byte loads): columns >= lds_cols (a multiple of 16) are known to be zero. The tile is UNPADDED (row stride = a power of two
    // >= lds_cols floats: what an LDS-direct load writes, lane-linear) and XOR-swizzled in 16-byte chunks -- physical chunk = chunk ^
    // tile_swz(row) -- so that the lanes of a wave, each reading the same chunk of its own row, hit different banks. For the
    // reading lane the swizzle is a constant: lds_p0 / lds_p1 = row start + ((2 hi) ^ (swz & 3)) * 16 and its ^1 neighbour (the
    // lane's two chunks of a 16-key group), lds_hi16 = (swz & ~3) * 16 is XORed onto the group's chunk base.
    const char *lds_p0, *lds_p1;
    unsigned lds_hi16;
    int lds_cols;
};

// swizzle of the bias tile: cpr = 16-byte chunks per tile row (a multiple of 4). Rows whose starts fall on the same banks
// (row stride 64 / 192 bytes mod 256: every 4th row; 128: every 2nd; 0: every row) get different chunk permutations: any 16
// consecutive rows reading the same logical chunk touch 16 different 4-bank groups.
__device__ __forceinline__ int tile_swz(int row, int cpr) {
    const int t = (cpr & 4) ? 2 : (cpr & 8) ? 3 : 4;
    return (row >> (4 - t)) & ((1 << t) - 1);
}
__device__ __forceinline__ void bias_ref_tile(BiasRef &b, const char *tile, int row, int stride, int cols, int hi) {
    const int swz = tile_swz(row, stride >> 2);
    const char *rowp = tile + (long)row * stride * 4;
    b.lds_p0 = rowp + (((2 * hi) ^ (swz & 3)) << 4);
    b.lds_p1 = rowp + (((2 * hi + 1) ^ (swz & 3)) << 4);
    b.lds_hi16 = (unsigned)(swz & ~3) << 4;
    b.lds_cols = cols;
}

template <typename T, int KS, int DT, int HAS_BIAS, bool MASKED, bool ROWSUM_MFMA, int STEP = 0, int ONLY = -1>
__device__ __forceinline__ void attn_tile_sm_pv(f32x16 (&s)[2], f32x16 (&oacc)[DT], float &m_run, float &l_run, const char *Vs,
                                                int key0, int M, int l31, int hi, const BiasRef &bias,
                                                float coeff, float c1);

// ONLY = 0 / 1: only that 32-key block of the 64-key tile takes part (half-tile key groups); -1: the whole tile.
template <typename T, int KS, int DT, int HAS_BIAS, bool MASKED, bool ROWSUM_MFMA, int STEP = 0, int ONLY = -1>
__device__ __forceinline__ void attn_tile(f32x16 (&oacc)[DT], float &m_run, float &l_run,
                                          const typename Vec<T>::v8 (&qf)[KS], const char *Ks, const char *Vs,
                                          int key0, int M, int l31, int hi, const BiasRef &bias,
                                          float coeff, float c1) {
    f32x16 s[2];
    score_tile<T, KS, ONLY>(s, qf, Ks, key0, MASKED ? M : 0x7fffffff, l31, hi);
    attn_tile_sm_pv<T, KS, DT, HAS_BIAS, MASKED, ROWSUM_MFMA, STEP, ONLY>(s, oacc, m_run, l_run, Vs, key0, M, l31, hi, bias, coeff, c1);
}

// Headroom of the lazy step (STEP 2), in exp2 units: a later tile keeps the row's running reference while no score of it exceeds the
// reference by more than this -- P <= 2^8 is exact in f16 and bf16 alike, and O^T / the row sum accumulate in fp32.
constexpr float LAZY_HEADROOM = 8.f;

// scores (already in `s`) -> (bias) -> online softmax -> PV against the V tile at Vs
// HAS_BIAS: 0 none, 1 bias rows read from global memory (per-lane buffer loads), 2 bias rows read from the LDS tile
// STEP: 0 the general online step; 1 the FIRST tile of a row (m_run = -inf, l_run = 0, O^T = 0 on entry: nothing to rescale, and the
//       first MFMA of every channel tile starts from a zero constant -- the caller need not clear O^T); 2 a LATER tile with a lazy
//       reference: the running maximum is only raised (and O^T rescaled) when some row of the wave would otherwise produce a P above
//       2^LAZY_HEADROOM -- any reference gives the same softmax, it only has to keep P inside the storage type's range.
template <typename T, int KS, int DT, int HAS_BIAS, bool MASKED, bool ROWSUM_MFMA, int STEP, int ONLY>
__device__ __forceinline__ void attn_tile_sm_pv(f32x16 (&s)[2], f32x16 (&oacc)[DT], float &m_run, float &l_run, const char *Vs,
                                                int key0, int M, int l31, int hi, const BiasRef &bias,
                                                float coeff, float c1) {
    typedef typename Vec<T>::v8 V8;
    const char *vl = Vs + vfrag_lane_off<DT>(hi * 32 + l31);

    // raw-domain logits x = s + c*bias (scale > 0, so the row max commutes with the scaling)
    // (a 4-way max tree and packed v_pk_fma_f32 for the exp arguments were measured: neutral to -3%)
    // a 32-key block entirely past M (the second block of a ragged tail: keys 96..127 of the 77 prompt tokens) takes no part in
    // anything below -- wave-uniform, and its P^T fragments are never multiplied (the PV loop skips the same blocks)
    const bool live0 = ONLY != 1 && (!MASKED || key0 < M), live1 = ONLY != 0 && (!MASKED || key0 + 32 < M);
    float tmax = -INFINITY;
    if (HAS_BIAS == 2) {
        // the lane's 8 consecutive keys of each (block, half) are 32 contiguous bytes of its row in the LDS tile; a 16-key
        // group past the last non-zero column of the map (wave-uniform test) gets no loads and no multiply-adds at all
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            if (kb == 0 ? live0 : live1) {
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    const int col0 = key0 + kb * 32 + 16 * g;
                    float m4[2] = {-INFINITY, -INFINITY};        // two short max chains per group instead of one long one
                    if (col0 < bias.lds_cols) {
                        const unsigned grp = ((unsigned)col0 * 4u) ^ bias.lds_hi16;      // (col0 / 4) chunks * 16 bytes, swizzled
                        const f32x4 b0 = *reinterpret_cast<const f32x4 *>(bias.lds_p0 + grp);
                        const f32x4 b1 = *reinterpret_cast<const f32x4 *>(bias.lds_p1 + grp);
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            const int r = g * 8 + j;
                            float x = fmaf(j < 4 ? b0[j] : b1[j - 4], coeff, s[kb][r]);
                            if (MASKED) x = key0 + key_of(kb, r, hi) < M ? x : -INFINITY;
                            s[kb][r] = x;
                            m4[j & 1] = fmaxf(m4[j & 1], x);
                        }
                    } else {
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            const int r = g * 8 + j;
                            float x = s[kb][r];
                            if (MASKED) x = key0 + key_of(kb, r, hi) < M ? x : -INFINITY;
                            s[kb][r] = x;
                            m4[j & 1] = fmaxf(m4[j & 1], x);
                        }
                    }
                    tmax = fmaxf(tmax, fmaxf(m4[0], m4[1]));
                }
            }
        }
    } else if (HAS_BIAS && bias.unit) {
        // unit key stride (the PwW [N, 77] maps): a lane's 8 consecutive keys of each (block, half) are 32 contiguous
        // bytes of its bias row -> two 16-byte loads instead of eight 4-byte ones (the row stride makes every lane hit
        // its own cache line either way, so the texture-address work per tile drops 4x). Dword-aligned only, which
        // buffer loads allow; the range check is per dword, so a row tail never zeroes its in-range neighbours.
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            if (kb == 0 ? live0 : live1) {
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    const unsigned off = bias.row_off + (unsigned)(key0 + kb * 32 + 16 * g + 8 * hi) * 4u;
                    const u32x4 b0 = __builtin_amdgcn_raw_buffer_load_b128(bias.srd, off, 0, 0);
                    const u32x4 b1 = __builtin_amdgcn_raw_buffer_load_b128(bias.srd, off + 16u, 0, 0);
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int r = g * 8 + j;
                        const float bv = __builtin_bit_cast(float, j < 4 ? b0[j] : b1[j - 4]);
                        float x = fmaf(bv, coeff, s[kb][r]);
                        if (MASKED) x = key0 + key_of(kb, r, hi) < M ? x : -INFINITY;
                        s[kb][r] = x;
                        tmax = fmaxf(tmax, x);
                    }
                }
            }
        }
    } else {
    unsigned key_stride = HAS_BIAS ? bias.key_stride : 0u;
    // (opaque to the optimiser: otherwise the 32 products key * stride of this rarely taken form are hoisted out of the caller's block
    // loop and live -- or spill -- across the hot forms)
    if (HAS_BIAS) asm volatile("" : "+s"(key_stride));
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
        if (kb == 0 ? live0 : live1) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = key0 + key_of(kb, r, hi);
            float x = s[kb][r];
            if (HAS_BIAS) {   // keys >= M of a ragged tile read a neighbouring (in-range) value; they are masked below
                const unsigned off = bias.row_off + (unsigned)key * key_stride;
                const float bv = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(bias.srd, off, 0, 0));
                x = fmaf(bv, coeff, x);
            }
            if (MASKED) x = key < M ? x : -INFINITY;
            s[kb][r] = x;
            tmax = fmaxf(tmax, x);
        }
        }
    }
    }
    tmax = xhalf_max(tmax);
    float m_new;
    if (STEP == 1) {
        m_new = tmax;                         // finite: key 0 of the row is live
        m_run = m_new;
    } else {
        m_new = fmaxf(m_run, tmax);           // finite: key0 < M, so at least one key of the tile is live
        const bool keep = STEP == 2 ? __all((tmax - m_run) * c1 <= LAZY_HEADROOM) : __all(m_new == m_run);
        if (!keep) {                          // (STEP 0: an exact skip -- alpha == 1 for every row of the wave)
            const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c1);
            if (!ROWSUM_MFMA) l_run *= alpha;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r) oacc[dt][r] *= alpha;
            m_run = m_new;
        } else if (STEP == 2) {
            m_new = m_run;
        }
    }
    const float mc = -m_new * c1;
    float psum = 0.f;
    V8 pf[2][2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
        if (kb == 0 ? live0 : live1) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float pv = __builtin_amdgcn_exp2f(fmaf(s[kb][r], c1, mc));   // exp2((x - m) * scale * log2 e)
                if (!ROWSUM_MFMA) psum += pv;
                pf[kb][r >> 3][r & 7] = (T)pv;
            }
        }
    }
    if (!ROWSUM_MFMA) l_run = STEP == 1 ? psum : l_run + psum;

    // O^T[d][row] += V^T[d][key] * P^T[key][row]   (V^T fragments come out of the transpose read)
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
        if ((ONLY < 0 || ONLY == kb) && (!MASKED || key0 + kb * 32 < M)) {
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2) {
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) {
                    const V8 vf = load_vfrag<T, DT>(vl, kb, k2, dt);
                    if (STEP == 1 && kb == 0 && k2 == 0) {      // (block 0 of the first tile is always live)
                        f32x16 zero;
#pragma unroll
                        for (int r = 0; r < 16; ++r) zero[r] = 0.f;
                        oacc[dt] = mfma32(vf, pf[kb][k2], zero);
                    } else {
                        oacc[dt] = mfma32(vf, pf[kb][k2], oacc[dt]);
                    }
                }
            }
        }
    }
}

// ---- range-free tile (bf16, no bias) ------------------------------------------------------------------------------------
// A bf16 P keeps its 8 significant bits at any magnitude and O^T / the row sum accumulate in fp32, so the softmax
// reference does not have to follow the running maximum: it is fixed by the FIRST tile of the row (its maximum plus
// 2^RF_HEADROOM), and every later tile is  P = exp2(s * c1 + mc)  with no maximum, no compare and no O^T rescale. A
// later score more than ~120 binary orders above that first maximum would overflow exp2: the kernels check the row
// sums / accumulators once at the end and recompute the workgroup's rows with the exact online softmax in that case.
// Headroom of the range-free reference above the first tile's maximum, in binary orders. bf16 (8 exponent bits): 2^8 -- P never
// leaves the range whatever follows within ~120 orders. f16 (5 exponent bits, finite to 2^16): NONE -- P = 1 at the first tile's
// maximum, later scores may rise 16 orders (11 natural units) above it before P overflows to inf (caught by the final check: the
// workgroup then takes the exact path), and everything more than 2^-24 below the reference is flushed: at most 4096 * 2^-24 = 2.4e-4
// of a row sum that is >= 1 (O^T and the sum accumulate in fp32).
template <typename T> struct RfHeadroom { static constexpr float value = 8.f; };
template <> struct RfHeadroom<f16> { static constexpr float value = 0.f; };

// The f16 range-free reference has only 16 binary orders of room above it, and the first key stage (in self-attention: the first
// 128 tokens = the top two rows of the image) says little about where a query's largest logits are. Trained self-attention keeps them
// around the query's OWN token, so for M == N the reference is floored by the row's self-logit q_row . k_row -- one 3-fragment load and
// a 24-term dot product per lane, once per kernel; the key stages stay in natural order (all query blocks of a head walk the keys in
// lockstep: that is what lets one L2 fill serve 16 workgroups -- visiting the diagonal stage first instead was measured: 2.6x the
// fabric traffic, profiles/r03_traffic.json). Returns the raw dot product over the lane's fragment halves, summed over both halves.
template <typename T, int KS>
__device__ __forceinline__ float self_logit(const typename Vec<T>::v8 (&qf)[KS], const T *krow_ptr, bool valid, int hi, int D) {
    typename Vec<T>::v8 kd[KS];
    load_q_frags<T, KS>(kd, krow_ptr, valid, hi, D);
    float acc = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc = fmaf((float)qf[ks][j], (float)kd[ks][j], acc);
    return acc + __shfl_xor(acc, 32);
}

template <typename T, int KS, int DT, bool MASKED, bool ROWSUM_MFMA>
__device__ __forceinline__ void attn_tile_rf(f32x16 (&oacc)[DT], float &mc, float &l_run, bool first,
                                             const typename Vec<T>::v8 (&qf)[KS], const char *Ks, const char *Vs,
                                             int key0, int M, int l31, int hi, float c1, float ref_floor = -INFINITY) {
    typedef typename Vec<T>::v8 V8;
    f32x16 s[2];
    score_tile<T, KS>(s, qf, Ks, key0, MASKED ? M : 0x7fffffff, l31, hi);
    if (MASKED) {
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) s[kb][r] = key0 + key_of(kb, r, hi) < M ? s[kb][r] : -INFINITY;
    }
    if (first) {
        asm volatile("; attn_tile_rf: reference from the first tile" ::: "memory");    // keep this a branch (no if-conversion into every tile)
        float tmax = -INFINITY;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) tmax = fmaxf(tmax, s[kb][r]);
        tmax = fmaxf(xhalf_max(tmax), ref_floor);    // finite: key0 < M, at least one live key (ref_floor: the row's self-logit, f16)
        mc = -(tmax * c1) - RfHeadroom<T>::value;
    }
    const char *vl = Vs + vfrag_lane_off<DT>(hi * 32 + l31);
    float psum = 0.f;
    V8 pf[2][2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float pv = __builtin_amdgcn_exp2f(fmaf(s[kb][r], c1, mc));
            if (!ROWSUM_MFMA) psum += pv;
            pf[kb][r >> 3][r & 7] = (T)pv;
        }
    }
    if (!ROWSUM_MFMA) l_run += psum;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
        if (!MASKED || key0 + kb * 32 < M) {
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2) {
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) {
                    const V8 vf = load_vfrag<T, DT>(vl, kb, k2, dt);
                    oacc[dt] = mfma32(vf, pf[kb][k2], oacc[dt]);
                }
            }
        }
    }
}

}  // namespace pww
