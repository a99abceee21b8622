// Kernels and launch templates of the fused attention forward (see pww_attn.hip for the design notes); included by the per-dtype
// instantiation units (pww_attn_inst.hip, compiled once per storage type so that the build runs them side by side).
#pragma once
#include "pww_attn_core.h"

namespace pww {


// The exact path behind the range-free modes (RangeFree<T>, below): the workgroup's rows again, from key 0, with the plain
// online softmax (running maximum in the raw-score domain; row sum from the ones column of V when ROWSUM_MFMA, else in
// l_run). Q is re-read unscaled and its fragment columns >= D are zero, so the ones the folded kernel keeps in column D of
// the K tile contribute nothing.
template <typename T> struct RangeFree { static constexpr bool value = false; };
template <> struct RangeFree<bf16> { static constexpr bool value = true; };
template <> struct RangeFree<f16> { static constexpr bool value = true; };     // round 3: without headroom (RfHeadroom<f16>), diagonal stage first

template <typename T, int KS, int DT, int NSUB, bool ROWSUM_MFMA, int KPT, int VPT, typename SRD>
__device__ __forceinline__ void exact_rows(f32x16 (&oacc)[DT], float &l_run, const AttnParams &p, const T *qrow_ptr, bool qvalid, char *smem,
                                        const StagePlan<KPT, VPT> &plan, SRD srd_k, SRD srd_v, unsigned k_step, unsigned v_step,
                                        int l31, int hi, bool active = true) {
    // `active` (wave-uniform): this wave has a row to redo. A wave without one keeps its finished O^T and only helps staging K / V (round 6:
    // at the 2 folded rows of the headline ONE workgroup on this path is the launch's tail -- 8 waves recomputing 256 rows for one bad row).
    typedef typename Vec<T>::v8 V8;
    typedef KTile<KS> KT;
    typedef VTile<DT> VT;
    constexpr int SUB_BYTES = KT::BYTES + VT::BYTES;
    constexpr int STAGE_BYTES = NSUB * SUB_BYTES;
    constexpr int STAGE_KEYS = NSUB * KVBLK;
    V8 qf[KS];
    load_q_frags<T, KS>(qf, qrow_ptr, qvalid && active, hi, p.D);
    if (active) {
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) oacc[dt][r] = 0.f;
        l_run = 0.f;
    }
    float m_run = -INFINITY;
    BiasRef bias;
    const float c1 = p.scale_log2e;
    const int nstage = (p.M + STAGE_KEYS - 1) / STAGE_KEYS, nfull = p.M / STAGE_KEYS;
    u32x4 kreg[KPT];
    u32x4 vreg[VPT];
    __syncthreads();                                   // every wave is done with the stage buffers
    stage_load(kreg, vreg, plan, srd_k, srd_v, 0u, 0u);
    stage_store<DT, KPT, VPT>(kreg, vreg, plan, smem);
    stage_load(kreg, vreg, plan, srd_k, srd_v, k_step, v_step);
    __syncthreads();
    int st = 0;
    for (; st < nfull; ++st) {
        char *cur = smem + (st & 1) * STAGE_BYTES;
        stage_store<DT, KPT, VPT>(kreg, vreg, plan, smem + ((st & 1) ^ 1) * STAGE_BYTES);
        stage_load(kreg, vreg, plan, srd_k, srd_v, (unsigned)(st + 2) * k_step, (unsigned)(st + 2) * v_step);
        if (active) {
#pragma unroll
            for (int sub = 0; sub < NSUB; ++sub)
                attn_tile<T, KS, DT, false, false, ROWSUM_MFMA>(oacc, m_run, l_run, qf, cur + sub * SUB_BYTES, cur + sub * SUB_BYTES + KT::BYTES,
                                                         st * STAGE_KEYS + sub * KVBLK, p.M, l31, hi, bias, 1.f, c1);
        }
        __syncthreads();
    }
    if (st < nstage) {
        char *cur = smem + (st & 1) * STAGE_BYTES;
#pragma unroll
        for (int sub = 0; sub < NSUB; ++sub) {
            const int key0 = st * STAGE_KEYS + sub * KVBLK;
            if (key0 < p.M && active)
                attn_tile<T, KS, DT, false, true, ROWSUM_MFMA>(oacc, m_run, l_run, qf, cur + sub * SUB_BYTES, cur + sub * SUB_BYTES + KT::BYTES,
                                                        key0, p.M, l31, hi, bias, 1.f, c1);
        }
    }
}

// KG = key groups: with KG > 1 the workgroup has NW * KG waves; wave (rg, kg) owns query rows of row group rg and,
// in every stage of KG sub-tiles, only sub-tile kg -- i.e. the KEYS of a stage are split over wave groups. This is
// how a launch with few query rows (B = 2: two 32-row waves per SIMD) still fills 3 waves per SIMD; the KG partial
// (m, O^T) states of a row group are merged once at the end through the (then free) LDS stage buffers.
// RF = range-free softmax (pww_attn_core.h: attn_tile_rf; bf16 without bias, KG == 1 only): no running maximum in the loop,
// one range check at the end, exact_rows as the fallback.
template <typename T, int KS, int DT, int NW, int NSUB, int KG, bool HAS_BIAS, bool ROWSUM_MFMA, bool RF = false>
__global__ void __launch_bounds__(NW * KG * 64, (KG > 1 ? (NW * KG >= 12 ? 3 : 2) : MinWaves<DT, NW, HAS_BIAS>::value)) attn_fwd_kernel(const AttnParams p) {
    static_assert(!RF || (KG == 1 && !HAS_BIAS && RangeFree<T>::value), "range-free mode: no bias, no key split");
    typedef typename Vec<T>::v8 V8;
    typedef KTile<KS> KT;
    typedef VTile<DT> VT;
    // HALF (round 6): KG = 2 NSUB key groups, each owns one 32-key BLOCK of a stage (half a sub-tile): twice the waves of the plain key
    // split on the same stage buffers -- two waves per SIMD whose dependent chains (LDS -> MFMA -> max -> exp -> MFMA) the hardware
    // interleaves -- for launches that otherwise run ONE wave per SIMD (SD1.5 N = 1024 d = 80 at 2 folded rows: 8 tiles per wave at
    // ~2800 cycles each for ~700 cycles of MFMA)
    constexpr bool HALF = KG > 1 && KG == 2 * NSUB;
    static_assert(KG == 1 || NSUB == KG || HALF, "key-split workgroups process one sub-tile (or one 32-key block of it) per key group");
    constexpr int NT = NW * KG * 64;
    constexpr int SUB_BYTES = KT::BYTES + VT::BYTES;     // one 64-key sub-tile: K rows, then V rows
    constexpr int STAGE_BYTES = NSUB * SUB_BYTES;
    constexpr int STAGE_KEYS = NSUB * KVBLK;
    constexpr int KPT = (NSUB * KT::NCHUNK + NT - 1) / NT;
    constexpr int VPT = (NSUB * VT::NCHUNK + NT - 1) / NT;

    extern __shared__ __attribute__((aligned(16))) char smem[];   // two stage buffers (double buffering)

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int rg = KG > 1 ? wave % NW : wave, kg = KG > 1 ? wave / NW : 0;
    const int hi = lane >> 5, l31 = lane & 31;
    int bh, qb;
    wg_to_pair_block(p, (p.N + NW * 32 - 1) / (NW * 32), bh, qb);
    const int b = bh / p.H, h = bh - b * p.H;

    const T *Qp = reinterpret_cast<const T *>(p.q) + b * p.q_sb + h * p.q_sh;
    const T *Kp = reinterpret_cast<const T *>(p.k) + b * p.k_sb + h * p.k_sh;
    const T *Vp = reinterpret_cast<const T *>(p.v) + b * p.v_sb + h * p.v_sh;
    T *Op = reinterpret_cast<T *>(p.o) + b * p.o_sb + h * p.o_sh;

    const int qrow = (qb * NW + rg) * 32 + l31;
    const bool qvalid = qrow < p.N;
    tl_stamp(p, 0);
    const unsigned long long tl_c0 = p.timeline ? clock64() : 0ull;

    V8 qf[KS];
    load_q_frags<T, KS>(qf, Qp + (long)qrow * p.q_sn, qvalid, hi, p.D);

    BiasRef bias;
    float coeff = 1.f;
    if (HAS_BIAS) {   // descriptor over this (b, h) slice; the host guarantees its extent is < 2^31 bytes
        const float *bbase = p.bias + b * p.b_sb + h * p.b_sh;
        const unsigned bytes = (unsigned)((((long)(p.N - 1) * p.b_sn + (long)(p.M - 1) * p.b_sm) + 1) * 4);
        bias.srd = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(bbase), 0, bytes, 0x00020000);
        bias.row_off = qvalid ? (unsigned)((long)qrow * p.b_sn * 4) : OOB_OFF;
        bias.key_stride = (unsigned)(p.b_sm * 4);
        bias.unit = p.b_sm == 1;
        coeff = bias_coefficient(p, b);
    }
    const float c1 = p.scale_log2e;

    f32x16 oacc[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[dt][r] = 0.f;
    float m_run = -INFINITY;  // running row max of the raw logits, identical in both half-waves
    float l_run = 0.f;        // running row sum (VALU path only), PARTIAL per half-wave
    float mc = 0.f;           // RF: -(reference * c1) - headroom, set by the row's first tile

    StagePlan<KPT, VPT> plan;
    make_plan<T, KS, DT, NT, NSUB, KPT, VPT>(plan, tid, p.D, p.k_sm, p.v_sm);
    const auto srd_k = head_srd(Kp, p.M, p.k_sm, p.D);
    const auto srd_v = head_srd(Vp, p.M, p.v_sm, p.D);
    const unsigned k_step = (unsigned)(STAGE_KEYS * p.k_sm * 2), v_step = (unsigned)(STAGE_KEYS * p.v_sm * 2);   // bytes per stage
    u32x4 kreg[KPT];
    u32x4 vreg[VPT];
    const int nstage = (p.M + STAGE_KEYS - 1) / STAGE_KEYS;
    const int nfull = p.M / STAGE_KEYS;                 // stages without any key >= M
    // prologue: the first stage's global loads are issued BEFORE anything touches LDS (round 6: they used to wait behind a zero fill of
    // both buffers and its barrier -- 94 KB of LDS stores at d = 80 for 8 KB of padding)
    stage_load(kreg, vreg, plan, srd_k, srd_v, 0u, 0u);
    // Head-dim padding is never staged: the 16-byte chunks past D of every K / V row of both buffers are zeroed once, here, and with
    // ROWSUM_MFMA column D of every V row is one (the PV MFMA then accumulates the softmax denominator in O^T row D). Only the padding
    // chunks are written -- the staged chunks (c * 8 < D) are disjoint from them, so no barrier separates this from the first stage_store;
    // head dims without padding (64, 96, 128, 160) write nothing.
    {
        const int c0 = p.D >> 3;
        constexpr int NROW = 2 * NSUB * KVBLK;
        const T one = (T)1.0f;
        unsigned short one_bits;
        __builtin_memcpy(&one_bits, &one, 2);
        for (int c = c0; c < KT::CHK; ++c)
            for (int r = tid; r < NROW; r += NT) *reinterpret_cast<u32x4 *>(smem + (r >> 6) * SUB_BYTES + (r & 63) * KT::STRIDE + c * 16) = u32x4{0u, 0u, 0u, 0u};
        for (int c = c0; c < VT::CHK; ++c)
            for (int r = tid; r < NROW; r += NT)
                *reinterpret_cast<u32x4 *>(smem + (r >> 6) * SUB_BYTES + KT::BYTES + (r & 63) * VT::STRIDE + c * 16) = (ROWSUM_MFMA && c == c0) ? u32x4{(unsigned)one_bits, 0u, 0u, 0u} : u32x4{0u, 0u, 0u, 0u};
    }
    // f16 range-free mode: the reference is floored by the row's self-logit (self_logit, pww_attn_core.h); raw-score domain here
    float ref_floor = -INFINITY;
    if constexpr (RF && RfHeadroom<T>::value == 0.f) {
        if (p.M == p.N) { const float sl = self_logit<T, KS>(qf, Kp + (long)qrow * p.k_sm, qvalid, hi, p.D); ref_floor = qvalid ? sl : -INFINITY; }
    }

    // first stage -> buffer 0, second stage -> registers
    stage_store<DT, KPT, VPT>(kreg, vreg, plan, smem);
    if (nstage > 1) stage_load(kreg, vreg, plan, srd_k, srd_v, k_step, v_step);
    __syncthreads();
    tl_stamp(p, 1);

    int st = 0;
    for (; st < nfull; ++st) {   // full stages: no key masking anywhere; ONE barrier per stage
        char *cur = smem + (st & 1) * STAGE_BYTES;
        char *nxt = smem + ((st & 1) ^ 1) * STAGE_BYTES;
        // registers hold stage st+1: park it in the other buffer (its readers finished before the last
        // barrier), then re-use the registers for stage st+2, whose loads fly during this stage's compute
        if (st + 1 < nstage) stage_store<DT, KPT, VPT>(kreg, vreg, plan, nxt);
        if (st + 2 < nstage) stage_load(kreg, vreg, plan, srd_k, srd_v, (unsigned)(st + 2) * k_step, (unsigned)(st + 2) * v_step);
        if constexpr (HALF) {
            const char *sb = cur + (kg >> 1) * SUB_BYTES;       // (wave-uniform: kg comes from the wave index)
            const int key0 = st * STAGE_KEYS + (kg >> 1) * KVBLK;
            if (kg & 1) attn_tile<T, KS, DT, HAS_BIAS, false, ROWSUM_MFMA, 0, 1>(oacc, m_run, l_run, qf, sb, sb + KT::BYTES, key0, p.M, l31, hi, bias, coeff, c1);
            else attn_tile<T, KS, DT, HAS_BIAS, false, ROWSUM_MFMA, 0, 0>(oacc, m_run, l_run, qf, sb, sb + KT::BYTES, key0, p.M, l31, hi, bias, coeff, c1);
        } else if constexpr (KG > 1) {
            attn_tile<T, KS, DT, HAS_BIAS, false, ROWSUM_MFMA>(oacc, m_run, l_run, qf, cur + kg * SUB_BYTES,
                                                               cur + kg * SUB_BYTES + KT::BYTES, st * STAGE_KEYS + kg * KVBLK,
                                                               p.M, l31, hi, bias, coeff, c1);
        } else if constexpr (RF) {
#pragma unroll
            for (int sub = 0; sub < NSUB; ++sub)
                attn_tile_rf<T, KS, DT, false, ROWSUM_MFMA>(oacc, mc, l_run, st == 0 && sub == 0, qf, cur + sub * SUB_BYTES,
                                                            cur + sub * SUB_BYTES + KT::BYTES, st * STAGE_KEYS + sub * KVBLK, p.M, l31, hi, c1, ref_floor);
        } else {
#pragma unroll
            for (int sub = 0; sub < NSUB; ++sub)
                attn_tile<T, KS, DT, HAS_BIAS, false, ROWSUM_MFMA>(oacc, m_run, l_run, qf, cur + sub * SUB_BYTES,
                                                                   cur + sub * SUB_BYTES + KT::BYTES, st * STAGE_KEYS + sub * KVBLK,
                                                                   p.M, l31, hi, bias, coeff, c1);
        }
        __syncthreads();
    }
    if (st < nstage) {           // ragged tail stage (already in LDS: stored by the prologue or the last iteration)
        char *cur = smem + (st & 1) * STAGE_BYTES;
        if constexpr (HALF) {
            const char *sb = cur + (kg >> 1) * SUB_BYTES;
            const int key0 = st * STAGE_KEYS + (kg >> 1) * KVBLK;
            if (key0 + (kg & 1) * 32 < p.M) {       // (a block past the last key: this key group saw nothing -- m = -inf, merged as such)
                if (kg & 1) attn_tile<T, KS, DT, HAS_BIAS, true, ROWSUM_MFMA, 0, 1>(oacc, m_run, l_run, qf, sb, sb + KT::BYTES, key0, p.M, l31, hi, bias, coeff, c1);
                else attn_tile<T, KS, DT, HAS_BIAS, true, ROWSUM_MFMA, 0, 0>(oacc, m_run, l_run, qf, sb, sb + KT::BYTES, key0, p.M, l31, hi, bias, coeff, c1);
            }
        } else if constexpr (KG > 1) {
            const int key0 = st * STAGE_KEYS + kg * KVBLK;
            if (key0 < p.M)
                attn_tile<T, KS, DT, HAS_BIAS, true, ROWSUM_MFMA>(oacc, m_run, l_run, qf, cur + kg * SUB_BYTES,
                                                                  cur + kg * SUB_BYTES + KT::BYTES, key0, p.M, l31, hi, bias, coeff, c1);
        } else {
#pragma unroll
            for (int sub = 0; sub < NSUB; ++sub) {
                const int key0 = st * STAGE_KEYS + sub * KVBLK;
                if (key0 < p.M) {
                    if constexpr (RF)
                        attn_tile_rf<T, KS, DT, true, ROWSUM_MFMA>(oacc, mc, l_run, key0 == 0, qf, cur + sub * SUB_BYTES,
                                                                   cur + sub * SUB_BYTES + KT::BYTES, key0, p.M, l31, hi, c1, ref_floor);
                    else
                        attn_tile<T, KS, DT, HAS_BIAS, true, ROWSUM_MFMA>(oacc, m_run, l_run, qf, cur + sub * SUB_BYTES,
                                                                          cur + sub * SUB_BYTES + KT::BYTES, key0, p.M, l31, hi, bias, coeff, c1);
                }
            }
        }
    }

    tl_stamp(p, 2);
    if constexpr (KG > 1) {
        // merge the KG partial softmax states of each row group: key groups 1.. publish (m, l, O^T) in LDS,
        // key group 0 folds them:  m = max m_i,  O = sum_i O_i * 2^((m_i - m) c1)  (the row-sum row included)
        constexpr int REC = (DT * 16 + 2) * 64;           // floats per published wave state
        float *xch = reinterpret_cast<float *>(smem);
        __syncthreads();                                   // every wave is done with the stage buffers
        if (kg > 0) {
            float *rec = xch + ((kg - 1) * NW + rg) * REC;
            rec[lane] = m_run;
            rec[64 + lane] = l_run;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r) rec[(2 + dt * 16 + r) * 64 + lane] = oacc[dt][r];
        }
        __syncthreads();
        if (kg > 0) return;
#pragma unroll
        for (int g = 1; g < KG; ++g) {
            const float *rec = xch + ((g - 1) * NW + rg) * REC;
            const float m_o = rec[lane];
            const float m_n = fmaxf(m_run, m_o);
            // a key group that saw no key (m = -inf, only possible for groups > 0 in a short sequence) contributes nothing
            const float a_s = m_run == -INFINITY ? 0.f : __builtin_amdgcn_exp2f((m_run - m_n) * c1);
            const float a_o = m_o == -INFINITY ? 0.f : __builtin_amdgcn_exp2f((m_o - m_n) * c1);
            l_run = l_run * a_s + rec[64 + lane] * a_o;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r) oacc[dt][r] = oacc[dt][r] * a_s + rec[(2 + dt * 16 + r) * 64 + lane] * a_o;
            m_run = m_n;
        }
    }

    // softmax denominator
    auto row_sum = [&]() -> float {
        if (ROWSUM_MFMA) {   // row D of O^T: tile D / 32, register (D % 32) / 2, held by the hi == 0 half
            const int rl = p.D & 31, tl = p.D >> 5;
            float lv = 0.f;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                const float c = rl == 8 ? oacc[dt][4] : rl == 16 ? oacc[dt][8] : oacc[dt][12];
                lv = dt == tl ? c : lv;
            }
            const float other = __shfl_xor(lv, 32);
            return hi ? other : lv;
        }
        return l_run + __shfl_xor(l_run, 32);
    };
    float l_tot = row_sum();
    if constexpr (RF) {
        // range check of the range-free mode: finite, positive row sums and finite accumulators mean no exp2 overflowed and
        // no row vanished; otherwise the workgroup redoes its rows exactly
        float asum = 0.f;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) asum += fabsf(oacc[dt][r]);
        const bool bad = qvalid && !(l_tot > 0.f && l_tot < 3.0e38f && asum < 3.0e38f);
        if (__syncthreads_or(bad)) {
            exact_rows<T, KS, DT, NSUB, ROWSUM_MFMA, KPT, VPT>(oacc, l_run, p, Qp + (long)qrow * p.q_sn, qvalid, smem, plan, srd_k, srd_v,
                                                                k_step, v_step, l31, hi, __any(bad));      // (only the waves that hold such a row recompute)
            l_tot = row_sum();
        }
    }
    // epilogue: normalise and write O[row][d]; register r of tile dt is d = dt*32 + (r&3) + 8*(r>>2) + 4*hi
    const float inv = 1.f / l_tot;
    store_o_block<T, DT>(Op + (long)(qvalid ? qrow : 0) * p.o_sn, oacc, inv, p.D, hi, qvalid, p.o_wide != 0);
    tl_stamp(p, 3);
    tl_cycles(p, tl_c0);
}

#if PWW_EXPERIMENTS     // (measured slower than the double-buffered form: libpww_hip_experiments.so only, behind PWW_DEBUG=attn_ksplit1=1)
// ---- key-split workgroups for the SMALL launches (round 5): NW row groups x KG key groups, ONE stage buffer --------------------------
// Self-attention at the coarse UNet levels (SD1.5: N = 1024 d = 80, N = 256 d = 160 at 2 folded rows) has 128 - 512 wave tasks of 32 rows
// for 1024 SIMDs: with the double-buffered form above (two 128-key stage buffers = 94 KB at d = 80) a workgroup cannot hold more than
// 2 key groups, every wave walks 8 tiles at one wave per SIMD, and a tile costs ~2800 cycles there (~700 of them MFMA: nothing overlaps
// the latency chain LDS -> MFMA -> max -> exp -> MFMA of a single wave; profiles/r05_timeline_call2.log). Here the stage buffer is
// SINGLE (KG x 64 keys of K and V), which makes room for KG = 4 at d = 80 (2 at d = 160): every SIMD runs two waves of different key
// groups, each walks a QUARTER of the keys, and the next stage's global loads fly under the current stage's compute (registers, parked
// between two barriers). The KG partial softmax states of a row group are merged once at the end through the then free stage buffer.
// Staging as in pww_cross_lean.hip: a thread moves one 16-byte column of consecutive row groups (one offset per operand; a pass adds a
// uniform step), rows past M and the head-dim padding are out of the descriptor's range (zeros: no fill pass), grid = (query block, head, image).
// Row sums on the vector ALU (RSM = false in every instantiation): the ones-column form measured WRONG here for head dims with D % 32 == 24
// (88, 120, 152: the denominator came out as channel D - 8's accumulator, tools/diag_rowsum_probe.py) and is not worth a second look at 4 tiles per wave.
template <typename T, int KS, int DT, int NW, int KG, bool RSM>
__global__ void __launch_bounds__(NW * KG * 64, (NW * KG >= 8 ? 2 : 1)) attn_ksplit1_kernel(const AttnParams p) {
    typedef typename Vec<T>::v8 V8;
    typedef KTile<KS> KT;
    typedef VTile<DT> VT;
    constexpr int NT = NW * KG * 64;
    constexpr int SROWS = KG * KVBLK;                                          // key rows per stage
    constexpr int KRPP = NT / KT::CHK, VRPP = NT / VT::CHK;                    // key rows a pass of the workgroup covers
    constexpr int KPASS = (SROWS + KRPP - 1) / KRPP, VPASS = (SROWS + VRPP - 1) / VRPP;
    constexpr int K_BYTES = SROWS * KT::STRIDE;

    extern __shared__ __attribute__((aligned(16))) char smem[];               // [K rows of the stage][V rows of the stage]; the merge records afterwards
    char *Kl = smem, *Vl = smem + K_BYTES;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rg = wave % NW, kg = wave / NW;
    const int hi = lane >> 5, l31 = lane & 31;
    const int qb = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
    tl_stamp(p, 0);

    const T *Qp = reinterpret_cast<const T *>(p.q) + b * p.q_sb + h * p.q_sh;
    const T *Kp = reinterpret_cast<const T *>(p.k) + b * p.k_sb + h * p.k_sh;
    const T *Vp = reinterpret_cast<const T *>(p.v) + b * p.v_sb + h * p.v_sh;
    T *Op = reinterpret_cast<T *>(p.o) + b * p.o_sb + h * p.o_sh;
    const int qrow = (qb * NW + rg) * 32 + l31;
    const bool qvalid = qrow < p.N;

    const auto srd_k = head_srd(Kp, p.M, p.k_sm, p.D);
    const auto srd_v = head_srd(Vp, p.M, p.v_sm, p.D);
    const int kr = tid / KT::CHK, kc = tid - kr * KT::CHK;
    const int vr = tid / VT::CHK, vc = tid - vr * VT::CHK;
    const bool k_act = kr < KRPP, v_act = vr < VRPP;
    const unsigned k0 = (k_act && kc * 8 < p.D) ? (unsigned)((kr * p.k_sm + kc * 8) * 2) : OOB_OFF, kstep = (unsigned)(KRPP * p.k_sm * 2);
    const unsigned v0 = (v_act && vc * 8 < p.D) ? (unsigned)((vr * p.v_sm + vc * 8) * 2) : OOB_OFF, vstep = (unsigned)(VRPP * p.v_sm * 2);
    const unsigned k_stage = (unsigned)(SROWS * p.k_sm * 2), v_stage = (unsigned)(SROWS * p.v_sm * 2);
    u32x4 kreg[KPASS], vreg[VPASS];
    auto request = [&](int st) {      // stage st -> registers (rows past M: out of range, zeros, no traffic; the host bounds the extent below 2^31)
#pragma unroll
        for (int i = 0; i < KPASS; ++i) kreg[i] = __builtin_amdgcn_raw_buffer_load_b128(srd_k, k0 + (unsigned)st * k_stage + (unsigned)i * kstep, 0, 0);
#pragma unroll
        for (int i = 0; i < VPASS; ++i) vreg[i] = __builtin_amdgcn_raw_buffer_load_b128(srd_v, v0 + (unsigned)st * v_stage + (unsigned)i * vstep, 0, 0);
    };
    const T one = (T)1.0f;
    unsigned short one_bits;
    __builtin_memcpy(&one_bits, &one, 2);
    const bool v_one = RSM && vc * 8 == p.D;          // first padding chunk of a V row: channel D = 1.0 (the softmax denominator's column)
    auto park = [&]() {
        if (k_act) {
            char *kd = Kl + kr * KT::STRIDE + kc * 16;
#pragma unroll
            for (int i = 0; i < KPASS; ++i)
                if ((i + 1) * KRPP <= SROWS || i * KRPP + kr < SROWS) *reinterpret_cast<u32x4 *>(kd + i * KRPP * KT::STRIDE) = kreg[i];
        }
        if (v_act) {
            char *vd = Vl + vr * VT::STRIDE + vc * 16;
#pragma unroll
            for (int i = 0; i < VPASS; ++i)
                if ((i + 1) * VRPP <= SROWS || i * VRPP + vr < SROWS) *reinterpret_cast<u32x4 *>(vd + i * VRPP * VT::STRIDE) = v_one ? u32x4{(unsigned)one_bits, 0u, 0u, 0u} : vreg[i];
        }
    };

    request(0);
    V8 qf[KS];
    load_q_frags<T, KS>(qf, Qp + (long)(qvalid ? qrow : 0) * p.q_sn, true, hi, p.D);      // (rows past N compute on row 0's values and are never stored)
    f32x16 oacc[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[dt][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;
    const float c1 = p.scale_log2e;
    BiasRef bias;
    const int nstage = (p.M + SROWS - 1) / SROWS;
    park();
    __syncthreads();
    tl_stamp(p, 1);
    for (int st = 0; st < nstage; ++st) {
        request(st + 1);          // (past the last stage: out of range -- issued anyway: a load under a branch is waited for at the join)
        const int key0 = st * SROWS + kg * KVBLK;
        const char *Ks = Kl + kg * KT::BYTES, *Vs = Vl + kg * VT::BYTES;
        if (key0 + KVBLK <= p.M) attn_tile<T, KS, DT, 0, false, RSM>(oacc, m_run, l_run, qf, Ks, Vs, key0, p.M, l31, hi, bias, 1.f, c1);
        else if (key0 < p.M) attn_tile<T, KS, DT, 0, true, RSM>(oacc, m_run, l_run, qf, Ks, Vs, key0, p.M, l31, hi, bias, 1.f, c1);
        __syncthreads();          // every wave is done reading the stage
        if (st + 1 < nstage) {    // (workgroup-uniform)
            park();
            __syncthreads();
        }
    }
    tl_stamp(p, 2);

    // merge the KG partial softmax states of each row group: key groups 1.. publish (m, l, O^T) in LDS, key group 0 folds them
    constexpr int REC = (DT * 16 + 2) * 64;           // floats per published wave state
    float *xch = reinterpret_cast<float *>(smem);
    if (kg > 0) {
        float *rec = xch + ((kg - 1) * NW + rg) * REC;
        rec[lane] = m_run;
        rec[64 + lane] = l_run;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) rec[(2 + dt * 16 + r) * 64 + lane] = oacc[dt][r];
    }
    __syncthreads();
    if (kg > 0) return;
#pragma unroll
    for (int g = 1; g < KG; ++g) {
        const float *rec = xch + ((g - 1) * NW + rg) * REC;
        const float m_o = rec[lane];
        const float m_n = fmaxf(m_run, m_o);
        // a key group that saw no key (m = -inf: a short sequence) contributes nothing
        const float a_s = m_run == -INFINITY ? 0.f : __builtin_amdgcn_exp2f((m_run - m_n) * c1);
        const float a_o = m_o == -INFINITY ? 0.f : __builtin_amdgcn_exp2f((m_o - m_n) * c1);
        l_run = l_run * a_s + rec[64 + lane] * a_o;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) oacc[dt][r] = oacc[dt][r] * a_s + rec[(2 + dt * 16 + r) * 64 + lane] * a_o;
        m_run = m_n;
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
    tl_stamp(p, 3);
}

#endif  // PWW_EXPERIMENTS

// ---- folded-reference variant (head dims with D % 16 == 8: SD1.x's d = 40) -----------------------------
// A stage of this kernel costs about the SUM of its LDS, MFMA and VALU times (profiles/r01_attn_phases.md), so the
// way to make d = 40 faster is to remove work from one of the three. This variant removes VALU work per score:
//   * Q is pre-multiplied by scale*log2(e) once, and the head-dim padding column d = D of the K tile holds 1.0
//     while the same column of the lane's Q fragment holds -m_ref: the score MFMA then delivers
//     x = (q.k) scale log2(e) - m_ref directly -- no per-score subtract/multiply (32 v_fma per 64-key tile).
//   * m_ref is a LAZY reference, not the exact running max: it is only raised (and O^T rescaled) when a score
//     exceeds it by more than 2^FOLD_TAU; until then P = exp2(x) <= 2^FOLD_TAU is harmless in f16/bf16 and the
//     final division by the row sum (accumulated from the same P by the ones column of V) makes the result
//     independent of the reference. m_ref is always exactly representable in T, so the folded column is exact.
//   * both 64-key sub-tiles of a stage are scored before one joint max / (rare) re-reference, and all MFMA operand
//     fragments of the stage are requested from LDS up front.
//   * bf16 ("range-free" mode, RF): a bf16 P keeps its 8 significant bits at ANY magnitude, and O^T / the row sum
//     accumulate in fp32, so the reference never has to follow the running maximum at all: it is set ONCE, from the
//     first stage (its row maximum plus 2^FOLD_HEADROOM), and the 41 max / compare instructions per stage -- a quarter
//     of the loop's VALU work, sitting in a phase where the matrix pipe idles -- disappear (69.6 -> 64.0 us at
//     N = 4096, B = 2). The only thing that can go wrong is range: a later score more than ~120 binary orders above the
//     first stage's maximum would overflow exp2. That cannot be excluded for arbitrary inputs, so the row sums are
//     checked at the end and a workgroup that sees a non-finite (or zero) sum recomputes its rows with the exact online
//     softmax (exact_rows above): always correct, fast for every input whose logits span less than e^83.
constexpr float FOLD_TAU = 6.f;

// MAGNITUDE GUARD. The one approximation of this variant is the extra rounding of Q * scale * log2(e) to T: a relative error of
// 2^-9 (bf16) / 2^-12 (f16) on every term of a score, i.e. an absolute logit error that grows LINEARLY with the magnitude of the
// logits that carry the softmax weight -- measured 1.1e-2 of max|O| (bf16, bar 1.6e-2) and 1.3e-3 .. 2.3e-3 (f16, bar 2e-3) at
// logit maxima of 40 - 45 natural units (tests/test_round2_gpu.py::test_hot_logits), negligible at |logit| < 10. The kernel
// therefore bounds the row maximum it has seen -- exactly, in the exp2 domain: after the first stage (early exit: the whole
// workgroup goes straight to the exact path) and at the end (m_ref + log2(row sum) >= the true row maximum) -- and a workgroup
// with a row beyond the limit (AttnParams::fold_limit) does not keep the folded scale for it. Limits (round 6, re-measured on a sweep of
// scaled-logit std 1 ... 8, profiles/r06_hot_logits.md; pww_attn_core.h FOLD_LIMIT_*): f16 48 exp2 units (= 33 natural units: 1.2e-3 of
// max|O| there, bar 2e-3; rounds 2 - 5: 36), bf16 56 (= 39: 1.1 - 1.6e-2, bar 1.6e-2; rounds 2 - 5: 72, extrapolated from two points).
// What "does not keep the folded scale" means:
//   f16, 8-wave workgroups: in the lazy-reference loop a row whose reference comes within FOLD_TAU + 1 of the limit takes the workgroup to
//        the EXACT-SCALE loop (Q unscaled, raw-domain reference, P = exp2(x c1)) -- before the error is made, no second pass;
//   everything else (bf16, whose range-free loop keeps no running maximum; the 4- and 2-wave forms; a FIRST key stage already past the
//        limit): the waves that hold such a row recompute it with the exact-scale online softmax (exact_rows) after the pass.

template <typename T, int KS>
__device__ __forceinline__ void fold_set_ref(typename Vec<T>::v8 (&qf)[KS], float mref, int hi, int D) {
    const T v = (T)(-mref);           // exact: mref is a T value
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)    // column D = k-step D/16, upper half (D % 16 == 8), element 0
        if (ks == (D >> 4) && hi) qf[ks][0] = v;
}

// Raise the reference of the rows that need it (all rows on the very first tile, where m_ref is still 0 and may be
// far ABOVE the scores as well), shift the pending scores and rescale O^T accordingly. Wave-uniform, rare.
// `cs` / `tau`: 1 and FOLD_TAU when the scores are in the exp2 domain (Q pre-multiplied by scale log2 e); c1 and FOLD_TAU / c1 when they are RAW
// (the exact-scale mode of the f16 kernel: P = exp2(x c1), the reference lives in the raw domain too).
template <typename T, int KS, int DT, int NS>
__device__ __forceinline__ void fold_rereference(f32x16 (&s)[NS][2], f32x16 (&oacc)[DT], float &mref,
                                                 typename Vec<T>::v8 (&qf)[KS], float tmax, bool first, int hi, int D, float cs = 1.f, float tau = FOLD_TAU) {
    // This path must stay a BRANCH: without a side effect in it hipcc if-converts the whole body into the hot loop
    // (64 v_sub_f32 per stage with delta = 0 -- the per-score VALU op the folded reference exists to remove).
    asm volatile("; fold_rereference: rare path" ::: "memory");
    const float mnew = (first || tmax > tau) ? (float)(T)(mref + tmax) : mref;
    const float delta = mnew - mref;                                   // exact in fp32
    const float alpha = first ? 1.f : __builtin_amdgcn_exp2f(-delta * cs);  // O^T is still zero on the first tile
#pragma unroll
    for (int n = 0; n < NS; ++n)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) s[n][kb][r] -= delta;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[dt][r] *= alpha;
    mref = mnew;
    fold_set_ref<T, KS>(qf, mref, hi, D);
}

// P = exp2(x) -> T, then O^T += V^T P^T for one 64-key sub-tile
template <typename T, int DT, bool MASKED, bool RAW = false>
__device__ __forceinline__ void fold_exp_pv(const f32x16 (&s)[2], f32x16 (&oacc)[DT], const char *Vs, int key0, int M,
                                            int l31, int hi, float c1 = 1.f) {
    typedef typename Vec<T>::v8 V8;
    const char *vl = Vs + vfrag_lane_off<DT>(hi * 32 + l31);
    V8 pf[2][2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) pf[kb][r >> 3][r & 7] = (T)__builtin_amdgcn_exp2f(RAW ? s[kb][r] * c1 : s[kb][r]);
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

// max of 32 scores and m: four independent chains (a dependent VALU op issues only every ~12 cycles for one wave,
// so one serial v_max3 chain over a stage's 64 scores costs ~390 cycles by itself -- profiles/r01_attn_phases.md)
__device__ __forceinline__ float max32(const f32x16 (&s)[2], float m) {
    float m0 = m, m1 = s[0][0], m2 = s[1][0], m3 = s[1][8];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        m0 = fmaxf(m0, s[0][r]);
        m1 = fmaxf(m1, s[0][8 + r]);
        m2 = fmaxf(m2, s[1][r]);
        m3 = fmaxf(m3, s[1][8 + r]);
    }
    return fmaxf(fmaxf(m0, m1), fmaxf(m2, m3));
}

// one (possibly ragged) 64-key sub-tile
template <typename T, int KS, int DT, bool MASKED, bool RFMODE, bool RAW = false>
__device__ __forceinline__ void fold_tile(f32x16 (&oacc)[DT], float &mref, bool first, typename Vec<T>::v8 (&qf)[KS],
                                          const char *Ks, const char *Vs, int key0, int M, int l31, int hi, int D, float ref_floor, float c1 = 1.f) {
    static_assert(!(RAW && RFMODE), "the exact-scale mode follows the running maximum");
    const float cs = RAW ? c1 : 1.f, tau = RAW ? FOLD_TAU / c1 : FOLD_TAU;
    f32x16 s[1][2];
    score_tile<T, KS>(s[0], qf, Ks, key0, MASKED ? M : 0x7fffffff, l31, hi);
    if (MASKED) {
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) s[0][kb][r] = key0 + key_of(kb, r, hi) < M ? s[0][kb][r] : -INFINITY;
    }
    if constexpr (RFMODE) {
        if (first) fold_rereference<T, KS, DT, 1>(s, oacc, mref, qf, fmaxf(xhalf_max(max32(s[0], -INFINITY)), ref_floor) + RfHeadroom<T>::value, true, hi, D);
    } else {
        const float tmax = xhalf_max(max32(s[0], -INFINITY));   // finite: key0 < M
        if (first || !__all(tmax <= tau)) fold_rereference<T, KS, DT, 1>(s, oacc, mref, qf, tmax, first, hi, D, cs, tau);
    }
    fold_exp_pv<T, DT, MASKED, RAW>(s[0], oacc, Vs, key0, M, l31, hi, c1);
}

// MFMA operand fragments of one 64-key sub-tile, requested from LDS ahead of their use: a ds_read_b128 issued
// right before its MFMA exposes the LDS latency (100+ cycles with 8 waves queueing) on every k-step.
template <typename T, int KS>
__device__ __forceinline__ void load_kfrags(typename Vec<T>::v8 (&kf)[2][KS], const char *Ks, int l31, int hi) {
    typedef typename Vec<T>::v8 V8;
    typedef KTile<KS> KT;
    const char *base = Ks + swap23(l31) * KT::STRIDE + hi * 16;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) kf[kb][ks] = *reinterpret_cast<const V8 *>(base + kb * 32 * KT::STRIDE + ks * 32);
}

template <typename T, int DT>
__device__ __forceinline__ void load_vfrags(typename Vec<T>::v8 (&vf)[2][2][DT], const char *Vs, int l31, int hi) {
    const char *vl = Vs + vfrag_lane_off<DT>(hi * 32 + l31);
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) vf[kb][k2][dt] = load_vfrag<T, DT>(vl, kb, k2, dt);
}

template <typename T, int KS>
__device__ __forceinline__ void score_frags(f32x16 (&s)[2], const typename Vec<T>::v8 (&kf)[2][KS],
                                            const typename Vec<T>::v8 (&qf)[KS]) {
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) acc = mfma32(kf[kb][ks], qf[ks], acc);
        s[kb] = acc;
    }
}

// O^T += V^T P^T with both operands in registers
template <typename T, int DT>
__device__ __forceinline__ void pv_frags(const typename Vec<T>::v8 (&pf)[2][2], f32x16 (&oacc)[DT],
                                         const typename Vec<T>::v8 (&vf)[2][2][DT]) {
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) oacc[dt] = mfma32(vf[kb][k2][dt], pf[kb][k2], oacc[dt]);
}

template <typename T, bool RAW = false>
__device__ __forceinline__ void exp_tile(typename Vec<T>::v8 (&pf)[2][2], const f32x16 (&s)[2], float c1 = 1.f) {
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) pf[kb][r >> 3][r & 7] = (T)__builtin_amdgcn_exp2f(RAW ? s[kb][r] * c1 : s[kb][r]);
}

// one full 128-key stage: all K fragments and the first sub-tile's V fragments are requested up front, both
// sub-tiles are scored, one joint reference check, then exp / PV per sub-tile
// RFMODE: the reference is fixed by the first stage (range-free); else it follows the running maximum lazily (FOLD_TAU)
// RAW (f16, round 6): Q is NOT pre-scaled -- the MFMA delivers raw score minus the raw-domain reference, P = exp2(x c1): one multiply per
// score more than the folded scale, and no rounding of Q scale log2 e any more: the form a workgroup continues in when a row's logits
// pass the magnitude guard (it used to recompute all its rows on the general online-softmax path after the complete fast pass).
template <typename T, int KS, int DT, int SUB_BYTES, bool RFMODE, bool RAW = false>
__device__ __forceinline__ void fold_stage2(f32x16 (&oacc)[DT], float &mref, bool first, typename Vec<T>::v8 (&qf)[KS],
                                            const char *cur, int key0, int l31, int hi, int D, float ref_floor, float c1 = 1.f) {
    static_assert(!(RAW && RFMODE), "the exact-scale mode follows the running maximum");
    const float cs = RAW ? c1 : 1.f, tau = RAW ? FOLD_TAU / c1 : FOLD_TAU;
    typedef typename Vec<T>::v8 V8;
    typedef KTile<KS> KT;
    V8 k0[2][KS], k1[2][KS], v0[2][2][DT], v1[2][2][DT];
    load_kfrags<T, KS>(k0, cur, l31, hi);
    load_kfrags<T, KS>(k1, cur + SUB_BYTES, l31, hi);
    load_vfrags<T, DT>(v0, cur + KT::BYTES, l31, hi);
    __builtin_amdgcn_sched_barrier(0);          // keep the requests ahead of the MFMAs (hipcc sinks them otherwise)
    f32x16 s[2][2];
    score_frags<T, KS>(s[0], k0, qf);
    score_frags<T, KS>(s[1], k1, qf);
    load_vfrags<T, DT>(v1, cur + SUB_BYTES + KT::BYTES, l31, hi);   // lands during the max / check below
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (RFMODE) {    // reference set once, from the first stage (see the header comment)
        if (first) fold_rereference<T, KS, DT, 2>(s, oacc, mref, qf, fmaxf(xhalf_max(max32(s[1], max32(s[0], -INFINITY))), ref_floor) + RfHeadroom<T>::value, true, hi, D);
    } else {
        const float tmax = xhalf_max(max32(s[1], max32(s[0], -INFINITY)));
        if (first || !__all(tmax <= tau)) fold_rereference<T, KS, DT, 2>(s, oacc, mref, qf, tmax, first, hi, D, cs, tau);
    }
    V8 pf[2][2];
    exp_tile<T, RAW>(pf, s[0], c1);
    pv_frags<T, DT>(pf, oacc, v0);
    exp_tile<T, RAW>(pf, s[1], c1);
    pv_frags<T, DT>(pf, oacc, v1);
}

template <typename T, int KS, int DT, int NW>
__global__ void __launch_bounds__(NW * 64, (NW == 2 ? 1 : MinWaves<DT, NW, false>::value)) attn_fwd_fold_kernel(const AttnParams p) {
    typedef typename Vec<T>::v8 V8;
    typedef KTile<KS> KT;
    typedef VTile<DT> VT;
    constexpr int NSUB = 2;
    constexpr int NT = NW * 64;
    constexpr int SUB_BYTES = KT::BYTES + VT::BYTES;
    constexpr int STAGE_BYTES = NSUB * SUB_BYTES;
    constexpr int STAGE_KEYS = NSUB * KVBLK;
    constexpr int KPT = (NSUB * KT::NCHUNK + NT - 1) / NT;
    constexpr int VPT = (NSUB * VT::NCHUNK + NT - 1) / NT;

    extern __shared__ __attribute__((aligned(16))) char smem[];   // two stage buffers (double buffering)

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int hi = lane >> 5, l31 = lane & 31;
    int bh, qb;
    wg_to_pair_block(p, (p.N + NW * 32 - 1) / (NW * 32), bh, qb);
    const int b = bh / p.H, h = bh - b * p.H;

    const T *Qp = reinterpret_cast<const T *>(p.q) + b * p.q_sb + h * p.q_sh;
    const T *Kp = reinterpret_cast<const T *>(p.k) + b * p.k_sb + h * p.k_sh;
    const T *Vp = reinterpret_cast<const T *>(p.v) + b * p.v_sb + h * p.v_sh;
    T *Op = reinterpret_cast<T *>(p.o) + b * p.o_sb + h * p.o_sh;

    const int qrow = (qb * NW + wave) * 32 + l31;
    const bool qvalid = qrow < p.N;

    V8 qf[KS];
    load_q_frags<T, KS>(qf, Qp + (long)qrow * p.q_sn, qvalid, hi, p.D);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)          // scores come out of the MFMA in the exp2 domain
#pragma unroll
        for (int j = 0; j < 8; ++j) qf[ks][j] = (T)((float)qf[ks][j] * p.scale_log2e);

    f32x16 oacc[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[dt][r] = 0.f;
    // static priority for the later-dispatched half of an 8-wave workgroup: it otherwise loses every VALU
    // arbitration against its older SIMD partner (guide: "two waves per SIMD"); measured +1 %
    if (NW == 8 && wave >= 4) __builtin_amdgcn_s_setprio(1);
    float mref = 0.f;      // softmax reference of the lane's row (exp2 domain), identical in both half-waves
    bool first = true;     // no tile processed yet: m_ref not established
    bool early = false;    // magnitude guard tripped after the first stage: skip the fast path
    // f16 only: range-free until the first stage shows a hot row, lazily following the running maximum from then on (see the loops). (bf16 was
    // given the same switch and lost it again the same day: its range-free loop is 20 % faster than its lazy one -- 423 against 514 us at 16
    // rows, where f16 pays 2 - 3 % -- and the benchmark's own rows fell under the threshold: 60.8 -> 73.9 us on the dominant launch.)
    constexpr bool CAN_SWITCH = RangeFree<T>::value && RfHeadroom<T>::value == 0.f;
    bool lazy = false;                               // (p.hot_sum < 0: lazy from the second stage on whatever the rows look like -- A/B)
    // third mode: lazy reference with the EXACT scale (rows past the magnitude guard), see the loops. In the 8-wave workgroups of the large
    // launches only: the 4-wave form has no registers left for a third loop (12 bytes of scratch), its rows past the guard keep exact_rows.
    constexpr bool CAN_RAW = CAN_SWITCH && NW >= 8;
    bool raw = false, redo0 = false;
    auto tl_sum = [&]() -> float {     // the lane's share of row D of O^T (the running softmax denominator: the ones column of V)
        const int rl_ = p.D & 31, tl_ = p.D >> 5;
        float lv = 0.f;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            const float c = rl_ == 8 ? oacc[dt][4] : oacc[dt][12];
            lv = dt == tl_ ? c : lv;
        }
        return lv;
    };
    tl_stamp(p, 0);
    const unsigned long long tl_c0 = p.timeline ? clock64() : 0ull;

    StagePlan<KPT, VPT> plan;
    make_plan<T, KS, DT, NT, NSUB, KPT, VPT>(plan, tid, p.D, p.k_sm, p.v_sm);
    const auto srd_k = head_srd(Kp, p.M, p.k_sm, p.D);
    const auto srd_v = head_srd(Vp, p.M, p.v_sm, p.D);
    const unsigned k_step = (unsigned)(STAGE_KEYS * p.k_sm * 2), v_step = (unsigned)(STAGE_KEYS * p.v_sm * 2);
    u32x4 kreg[KPT];
    u32x4 vreg[VPT];
    const int nstage = (p.M + STAGE_KEYS - 1) / STAGE_KEYS;
    const int nfull = p.M / STAGE_KEYS;

    // padding is never staged: the 16-byte chunks past D of every K / V row of both buffers are written once, here -- zeros, with column D
    // of every V row = one (softmax denominator from the PV MFMA) and column D of every K row = one (the folded -m_ref term of the score
    // MFMA). Only those chunks (disjoint from what stage_store writes: no barrier in between; round 5 zeroed all 78 KB behind a barrier).
    // (The first stage's loads are issued AFTER this, unlike in attn_fwd_kernel: with them in flight the 4-wave f16 form spilled 12 bytes.)
    {
        constexpr int NROW = 2 * NSUB * KVBLK;
        const int c0 = p.D >> 3;
        const T one = (T)1.0f;
        unsigned short one_bits;
        __builtin_memcpy(&one_bits, &one, 2);
        const u32x4 z = {0u, 0u, 0u, 0u}, o = {(unsigned)one_bits, 0u, 0u, 0u};
        for (int c = c0; c < KT::CHK; ++c)
            for (int r = tid; r < NROW; r += NT) *reinterpret_cast<u32x4 *>(smem + (r >> 6) * SUB_BYTES + (r & 63) * KT::STRIDE + c * 16) = c == c0 ? o : z;
        for (int c = c0; c < VT::CHK; ++c)
            for (int r = tid; r < NROW; r += NT) *reinterpret_cast<u32x4 *>(smem + (r >> 6) * SUB_BYTES + KT::BYTES + (r & 63) * VT::STRIDE + c * 16) = c == c0 ? o : z;
    }
    // f16 range-free mode: the reference is floored by the row's self-logit (self_logit, pww_attn_core.h; qf is pre-scaled: exp2 domain)
    float ref_floor = -INFINITY;
    if constexpr (RangeFree<T>::value && RfHeadroom<T>::value == 0.f) {
        if (p.M == p.N) { const float sl = self_logit<T, KS>(qf, Kp + (long)qrow * p.k_sm, qvalid, hi, p.D); ref_floor = qvalid ? sl : -INFINITY; }
    }

    stage_load(kreg, vreg, plan, srd_k, srd_v, 0u, 0u);
    stage_store<DT, KPT, VPT>(kreg, vreg, plan, smem);
    stage_load(kreg, vreg, plan, srd_k, srd_v, k_step, v_step);     // past the last key: zeros (out of range)
    __syncthreads();

    // Full stages; ONE barrier per stage. Store / load are unconditional (stages past the end read zeros and land in a buffer nobody reads).
    // Stage 0 is written out in front of the loop: it sets the reference and carries the guards, the loop behind it is the steady state and
    // nothing else (hipcc peeled the first iteration of the one-loop form for bf16 and did not for f16, whose loop kept the first stage's
    // branches, 10 - 13 waits and ~140 more instructions in its body: 458 against 407 us at 16 rows).
    int st = 0;
    bool steady = false;
    if (nfull > 0) {
        stage_store<DT, KPT, VPT>(kreg, vreg, plan, smem + STAGE_BYTES);
        stage_load(kreg, vreg, plan, srd_k, srd_v, 2u * k_step, 2u * v_step);
        fold_stage2<T, KS, DT, SUB_BYTES, RangeFree<T>::value>(oacc, mref, first, qf, smem, 0, l31, hi, p.D, ref_floor);
        first = false;
        // magnitude guard, early form: the first stage's row maximum (m_ref minus the range-free headroom)
        const float m0 = mref - (RangeFree<T>::value ? RfHeadroom<T>::value : 0.f);
        if (__syncthreads_or(qvalid && !(fabsf(m0) <= p.fold_limit))) {
            // a FIRST stage already past the limit: the 8-wave f16 form starts over on the exact scale (stage 0 is still in its buffer, stage 1
            // parked, stage 2 in registers: one stage computed twice); everything else recomputes on exact_rows after the loop
            if constexpr (CAN_RAW) { raw = true; redo0 = true; } else { early = true; }
        } else {
            st = 1;
            steady = true;
            if constexpr (CAN_SWITCH) {
                // HOT ROWS (round 6). The f16 range-free reference has 16 binary orders of room above the first stage's maximum: a row whose
                // logits spread over more than that (scaled-logit std >= ~3: what trained SD layers produce) overflows P to inf at some later
                // key, and the whole workgroup used to redo its rows on the exact path (733 us instead of 484 at 16 rows, scaled-logit std 4).
                // The first stage says which rows those are: its row sum relative to its own maximum is the effective number of keys that
                // carry the softmax -- ~128 e^(sigma^2 / 2 - 2.6 sigma) of the stage's 128 for logits of std sigma: 40 at 0.5, 16 at 1, 5 at 3 --
                // and a WAVE with a row below p.hot_sum (8) follows the running maximum lazily from here on: the SECOND loop below (the
                // FOLD_TAU reference of the non-range-free form: one max per score and stage, +2 % on that wave; P can no longer
                // overflow). Two loops, not one loop with two bodies: that form spilled 350 - 480 bytes per lane.
                // (row D of O^T sits in the hi == 0 half; the bf16 reference sits 2^RfHeadroom above the stage's maximum: the sum is taken relative to the maximum)
                const float l0 = __shfl(tl_sum(), l31) * __builtin_amdgcn_exp2f(RfHeadroom<T>::value);
                // (8 waves: per WAVE, like the switch to the third loop below -- the guard's vote above was this stage's barrier; the 4-wave
                // form keeps round 6's first version, a workgroup-wide vote: the per-wave form spilled 24 bytes there)
                if constexpr (CAN_RAW) lazy = __any((qvalid && l0 < p.hot_sum) || p.hot_sum < 0.f) != 0;
                else lazy = __syncthreads_or((qvalid && l0 < p.hot_sum) || p.hot_sum < 0.f) != 0;
                steady = !lazy;
            }
        }
    }
    if (steady) {
        for (; st < nfull; ++st) {
            char *cur = smem + (st & 1) * STAGE_BYTES;
            char *nxt = smem + ((st & 1) ^ 1) * STAGE_BYTES;
            stage_store<DT, KPT, VPT>(kreg, vreg, plan, nxt);
            stage_load(kreg, vreg, plan, srd_k, srd_v, (unsigned)(st + 2) * k_step, (unsigned)(st + 2) * v_step);
            fold_stage2<T, KS, DT, SUB_BYTES, RangeFree<T>::value>(oacc, mref, false, qf, cur, st * STAGE_KEYS, l31, hi, p.D, ref_floor);
            __syncthreads();
        }
    }
    if constexpr (CAN_SWITCH) {
        if (lazy && !early && !raw) {
            // second loop: lazy reference, folded scale. A row whose reference comes within FOLD_TAU + 1 of the magnitude guard's limit (the
            // folded scale's rounding error grows with the logits: FOLD_LIMIT_F16) takes its WAVE to the third loop -- BEFORE the error is
            // made, not after the pass.
            const float lim_sw = p.fold_limit - FOLD_TAU - 1.f;
            for (; st < nfull; ++st) {
                char *cur = smem + (st & 1) * STAGE_BYTES;
                char *nxt = smem + ((st & 1) ^ 1) * STAGE_BYTES;
                stage_store<DT, KPT, VPT>(kreg, vreg, plan, nxt);
                stage_load(kreg, vreg, plan, srd_k, srd_v, (unsigned)(st + 2) * k_step, (unsigned)(st + 2) * v_step);
                fold_stage2<T, KS, DT, SUB_BYTES, false>(oacc, mref, false, qf, cur, st * STAGE_KEYS, l31, hi, p.D, ref_floor);
                if constexpr (CAN_RAW) {
                    // per WAVE: the modes differ in a wave's own arithmetic only (its Q fragments, its reference), every loop keeps the stage
                    // protocol -- one barrier per stage -- so waves of one workgroup may sit in different loops. (A workgroup-wide vote here,
                    // __syncthreads_or = a reduction through LDS behind two barriers, cost every lazy stage ~10 %: config 3's dominant launch
                    // 484 -> 530 us when the third loop first shipped.)
                    const bool hot = __any(qvalid && !(fabsf(mref) <= lim_sw));
                    __syncthreads();
                    if (hot) { raw = true; ++st; break; }
                } else {
                    __syncthreads();
                }
            }
        }
        if (CAN_RAW && raw) {
            // third loop: lazy reference, EXACT scale -- Q unscaled again (re-read: a few KB), the reference moved to the raw domain (a T value
            // r with r c1 as close to the old reference as T allows; O^T rescaled by the difference), P = exp2(x c1). What was accumulated so far
            // came from logits below the limit (inside the accuracy the guard stands for); everything larger is computed without the rounding of
            // Q scale log2 e. No magnitude limit from here on, and no second pass (round 5: exact_rows after the complete fast pass, 2 x the time).
            load_q_frags<T, KS>(qf, Qp + (long)qrow * p.q_sn, qvalid, hi, p.D);
            const float c1 = p.scale_log2e;
            if (redo0) {                    // (from the first-stage guard: nothing of the folded pass is kept)
#pragma unroll
                for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                    for (int i = 0; i < 16; ++i) oacc[dt][i] = 0.f;
                mref = 0.f;
                fold_set_ref<T, KS>(qf, mref, hi, p.D);
                fold_stage2<T, KS, DT, SUB_BYTES, false, true>(oacc, mref, true, qf, smem, 0, l31, hi, p.D, ref_floor, c1);
                __syncthreads();            // every wave is done with buffer 0 before iteration 1 parks stage 2 there
                st = 1;
            } else {
                const float r = (float)(T)(mref / c1);
                const float alpha = __builtin_amdgcn_exp2f(mref - r * c1);
#pragma unroll
                for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                    for (int i = 0; i < 16; ++i) oacc[dt][i] *= alpha;
                mref = r;                   // RAW domain from here on
                fold_set_ref<T, KS>(qf, mref, hi, p.D);
            }
            for (; st < nfull; ++st) {
                char *cur = smem + (st & 1) * STAGE_BYTES;
                char *nxt = smem + ((st & 1) ^ 1) * STAGE_BYTES;
                stage_store<DT, KPT, VPT>(kreg, vreg, plan, nxt);
                stage_load(kreg, vreg, plan, srd_k, srd_v, (unsigned)(st + 2) * k_step, (unsigned)(st + 2) * v_step);
                fold_stage2<T, KS, DT, SUB_BYTES, false, true>(oacc, mref, false, qf, cur, st * STAGE_KEYS, l31, hi, p.D, ref_floor, c1);
                __syncthreads();
            }
        }
    }
    if (st < nstage && !early) {           // ragged tail stage (already in LDS)
        char *cur = smem + (st & 1) * STAGE_BYTES;
#pragma unroll
        for (int sub = 0; sub < NSUB; ++sub) {
            const int key0 = st * STAGE_KEYS + sub * KVBLK;
            if (key0 < p.M) {
                if (CAN_RAW && raw)
                    fold_tile<T, KS, DT, true, false, true>(oacc, mref, first, qf, cur + sub * SUB_BYTES, cur + sub * SUB_BYTES + KT::BYTES, key0, p.M, l31, hi, p.D, ref_floor, p.scale_log2e);
                else if (CAN_SWITCH ? lazy : !RangeFree<T>::value)
                    fold_tile<T, KS, DT, true, false>(oacc, mref, first, qf, cur + sub * SUB_BYTES, cur + sub * SUB_BYTES + KT::BYTES, key0, p.M, l31, hi, p.D, ref_floor);
                else
                    fold_tile<T, KS, DT, true, true>(oacc, mref, first, qf, cur + sub * SUB_BYTES, cur + sub * SUB_BYTES + KT::BYTES, key0, p.M, l31, hi, p.D, ref_floor);
                first = false;
            }
        }
    }
    tl_stamp(p, 2);

    // softmax denominator: row D of O^T (tile D / 32, register (D % 32) / 2, held by the hi == 0 half)
    const int rl = p.D & 31, tl = p.D >> 5;
    float lsum;
    {
        float lv = 0.f;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            const float c = rl == 8 ? oacc[dt][4] : oacc[dt][12];
            lv = dt == tl ? c : lv;
        }
        const float other = __shfl_xor(lv, 32);
        lsum = hi ? other : lv;
    }
    {
        // Range check (range-free bf16 mode): a finite, positive row sum means no exp2 overflowed and no row vanished (and the
        // accumulated O^T itself: a P just below the float range times |V| > 1 overflows the product, not the sum).
        // Magnitude guard, final form (every dtype): m_ref + log2(row sum) bounds the row maximum from above (by at most log2 M).
        float asum = 0.f;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) asum += fabsf(oacc[dt][r]);     // inf or NaN anywhere makes the comparison below false
        float est_max = mref + __builtin_amdgcn_logf(lsum);              // v_log_f32 = log2
        if (!RangeFree<T>::value || lazy) est_max = fminf(est_max, mref + FOLD_TAU);       // the lazy reference is never more than 2^FOLD_TAU below a score
        if (raw) est_max = 0.f;                                          // exact scale: no magnitude limit (mref is a raw-domain value there)
        const bool bad = qvalid && !(lsum > 0.f && lsum < 3.0e38f && asum < 3.0e38f && fabsf(est_max) <= p.fold_limit);
        const bool redo = early || __syncthreads_or(bad);
        if (p.path_counts) {      // debug: which path this workgroup took (exact-scale: any of its waves; the extra vote only runs when somebody counts)
            const bool raw_any = __syncthreads_or(raw) != 0, lazy_any = __syncthreads_or(lazy) != 0;
            if (threadIdx.x == 0) atomicAdd(p.path_counts + (redo ? 2 : raw_any ? 3 : lazy_any ? 1 : 0), 1u);
        }
        if (redo) {
            float l_unused;
            // after a COMPLETE fast pass only the waves that hold a bad row recompute (the others keep their results and help staging);
            // the early exit left every wave without a result
            const bool wave_redo = early || __any(bad);
            exact_rows<T, KS, DT, NSUB, true, KPT, VPT>(oacc, l_unused, p, Qp + (long)qrow * p.q_sn, qvalid, smem, plan, srd_k, srd_v, k_step, v_step, l31, hi, wave_redo);
            float lv = 0.f;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                const float c = rl == 8 ? oacc[dt][4] : oacc[dt][12];
                lv = dt == tl ? c : lv;
            }
            const float other = __shfl_xor(lv, 32);
            lsum = hi ? other : lv;
        }
    }
    const float inv = 1.f / lsum;
    store_o_block<T, DT>(Op + (long)(qvalid ? qrow : 0) * p.o_sn, oacc, inv, p.D, hi, qvalid, p.o_wide != 0);
    tl_stamp(p, 3);
    tl_cycles(p, tl_c0);
}

// ---- host dispatch ---------------------------------------------------------------------------

#if PWW_EXPERIMENTS
static int rf_mode() {   // PWW_DEBUG=attn_rf=0: bf16 self-attention with the running-maximum softmax instead of the range-free one (A/B testing)
    static int mode = -2;
    if (mode == -2) { mode = debug_knobs().attn_rf; }
    return mode;
}
#endif

template <typename T, int KS, int DT, int NW, bool HAS_BIAS, bool ROWSUM_MFMA>
static int launch_attn_rs(const AttnParams &p, hipStream_t stream) {
    // two 64-key sub-tiles per stage (one barrier per 128 keys) while the double buffer stays small enough
    // for two workgroups per CU; the widest heads use single sub-tile stages
    constexpr int SUBB = KTile<KS>::BYTES + VTile<DT>::BYTES;
    constexpr int NSUB = (2 * 2 * SUBB <= 80 * 1024) ? 2 : 1;   // (256-key stages for NW == 8 measured no faster)
    constexpr size_t lds = 2 * NSUB * (KTile<KS>::BYTES + VTile<DT>::BYTES);
    const int qblocks = (p.N + NW * 32 - 1) / (NW * 32);
    const dim3 grid((unsigned)(qblocks * p.B * p.H));
    constexpr bool CAN_RF = !HAS_BIAS && RangeFree<T>::value && NW >= 4;     // (2-wave workgroups: small, latency-bound launches)
#if PWW_EXPERIMENTS      // PWW_DEBUG=attn_rf=0: the running-maximum form of the same launches (A/B; a second instantiation per shape class)
    auto kern = (CAN_RF && rf_mode() == 1) ? attn_fwd_kernel<T, KS, DT, NW, NSUB, 1, HAS_BIAS, ROWSUM_MFMA, CAN_RF>
                                           : attn_fwd_kernel<T, KS, DT, NW, NSUB, 1, HAS_BIAS, ROWSUM_MFMA, false>;
#else
    auto kern = attn_fwd_kernel<T, KS, DT, NW, NSUB, 1, HAS_BIAS, ROWSUM_MFMA, CAN_RF>;
#endif
    if (lds > 64 * 1024) {
        static thread_local bool done = false;
        if (!done) {
            if (check_hip(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds),
                          "hipFuncSetAttribute"))
                return PWW_EHIP;
            done = true;
        }
    }
    launch_attn_kernel(kern, grid, dim3(NW * 64), lds, stream, p);
    return check_hip(hipGetLastError(), "attn_fwd_kernel launch");
}

// key-split workgroups: NW row groups x KG key groups, KG*64-key stages (self-attention launches too small to give
// every SIMD a wave otherwise): d <= 64 uses 4 x 3 = 12 waves, d = 80/96 uses 2 x 2 = 4 waves
// NSUB = KG: every key group owns a 64-key sub-tile of a stage; NSUB = KG / 2 (round 6): a 32-key block of one
template <typename T, int KS, int DT, int NW, int KG, bool ROWSUM_MFMA, int NSUB = KG>
static int launch_attn_ksplit(const AttnParams &p, hipStream_t stream) {
    constexpr size_t stage = NSUB * (KTile<KS>::BYTES + VTile<DT>::BYTES);
    constexpr size_t merge = (size_t)(KG - 1) * NW * (DT * 16 + 2) * 64 * sizeof(float);
    constexpr size_t lds = 2 * stage > merge ? 2 * stage : merge;
    static_assert(lds <= 160 * 1024, "stage buffers / merge records of the key-split workgroup");
    const int qblocks = (p.N + NW * 32 - 1) / (NW * 32);
    auto kern = attn_fwd_kernel<T, KS, DT, NW, NSUB, KG, false, ROWSUM_MFMA>;
    static thread_local bool done = false;
    if (!done) {
        if (check_hip(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds),
                      "hipFuncSetAttribute"))
            return PWW_EHIP;
        done = true;
    }
    launch_attn_kernel(kern, dim3((unsigned)(qblocks * p.B * p.H)), dim3(NW * KG * 64), lds, stream, p);
    return check_hip(hipGetLastError(), "attn_fwd_kernel<key-split> launch");
}

#if PWW_EXPERIMENTS
// the single-buffer kernel requests one stage past the last key (out of range by construction): keep every offset it forms far below 2^31
static bool ksplit1_extent_ok(const AttnParams &p) {
    return (long)(p.M + 512) * p.k_sm * 2 < (1L << 30) && (long)(p.M + 512) * p.v_sm * 2 < (1L << 30);
}

template <typename T, int KS, int DT, int NW, int KG, bool RSM>
static int launch_attn_ksplit1(const AttnParams &p, hipStream_t stream) {
    constexpr size_t stage = (size_t)KG * KVBLK * (KTile<KS>::STRIDE + VTile<DT>::STRIDE);
    constexpr size_t merge = (size_t)(KG - 1) * NW * (DT * 16 + 2) * 64 * sizeof(float);
    constexpr size_t lds = stage > merge ? stage : merge;
    static_assert(lds <= 158 * 1024, "stage buffer of the key-split kernel");
    const int qblocks = (p.N + NW * 32 - 1) / (NW * 32);
    if (p.H > 65535 || p.B > 65535) { set_error("attn_fwd: more than 65535 heads or images"); return PWW_EINVAL; }
    auto kern = attn_ksplit1_kernel<T, KS, DT, NW, KG, RSM>;
    static thread_local bool done = false;
    if (!done) {
        if (check_hip(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds), "hipFuncSetAttribute"))
            return PWW_EHIP;
        done = true;
    }
    launch_attn_kernel(kern, dim3((unsigned)qblocks, (unsigned)p.H, (unsigned)p.B), dim3(NW * KG * 64), lds, stream, p);
    return check_hip(hipGetLastError(), "attn_ksplit1_kernel launch");
}

#endif  // PWW_EXPERIMENTS

static int ksplit_mode() {   // PWW_DEBUG=attn_ksplit=0|1 (A/B testing); default on
    static int mode = -2;
    if (mode == -2) { mode = debug_knobs().attn_ksplit; }
    return mode;
}

static int fold_mode() {   // PWW_DEBUG=attn_fold=n (A/B testing only -- accuracy is guarded in the kernel): 0 = never, 1 = bf16 and f16 (default), 2 = bf16 only
    static int mode = -2;
    if (mode == -2) { mode = debug_knobs().attn_fold; }
    return mode;
}

template <typename T, int KS, int DT, int NW>
static int launch_attn_fold(const AttnParams &p, hipStream_t stream) {
    constexpr size_t lds = 2 * 2 * (KTile<KS>::BYTES + VTile<DT>::BYTES);
    const int qblocks = (p.N + NW * 32 - 1) / (NW * 32);
    auto kern = attn_fwd_fold_kernel<T, KS, DT, NW>;
    if (lds > 64 * 1024) {
        static thread_local bool done = false;
        if (!done) {
            if (check_hip(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds),
                          "hipFuncSetAttribute"))
                return PWW_EHIP;
            done = true;
        }
    }
    launch_attn_kernel(kern, dim3((unsigned)(qblocks * p.B * p.H)), dim3(NW * 64), lds, stream, p);
    return check_hip(hipGetLastError(), "attn_fwd_fold_kernel launch");
}

template <typename T, int KS, int DT, int NW, bool HAS_BIAS>
static int launch_attn(const AttnParams &p, hipStream_t stream) {
    if constexpr (!HAS_BIAS && KS == 3 && DT == 2) {
        // d = 40 (and 8, 24): a free head-dim padding column in the K tile -> folded-reference softmax
        // (also beats the key-split variant below at B = 1, N = 4096: 47 vs 52 us)
        // The variant rounds Q * scale * log2(e) to T once more; the error that costs grows with the logit magnitude, and the
        // kernel itself sends every workgroup whose rows exceed FoldLimit<T> through the exact-scale path (magnitude guard).
        const bool fold_ok = fold_mode() == 1 || (fold_mode() == 2 && !__is_same(T, f16));
        if ((p.D & 15) == 8 && fold_ok) return launch_attn_fold<T, KS, DT, NW>(p, stream);
    }
    if constexpr (!HAS_BIAS && DT <= 2 && NW == 4) {
        // at most one 4-wave workgroup per CU (1 wave/SIMD) and a long key sequence: split the keys over 3 wave
        // groups -> 3 waves/SIMD. (Measured N=4096 d=40: B=1 55.8 -> 51.6 us; with two workgroups per CU, B=2,
        // the split LOSES, 101 -> 114 us, so it is limited to the under-filled case.)
        const long wgs = (long)((p.N + 127) / 128) * p.B * p.H;
        if (ksplit_mode() == 1 && wgs <= 256 && p.M >= 1024) {
            if ((p.D & 31) != 0) return launch_attn_ksplit<T, KS, DT, 4, 3, true>(p, stream);
            return launch_attn_ksplit<T, KS, DT, 4, 3, false>(p, stream);
        }
    }
    if constexpr (!HAS_BIAS && DT == 3 && NW == 2) {
        // d = 80 / 96 with at most one 2-wave workgroup per CU (SD1.5 N = 1024 at B <= 2): two key groups -> a wave on
        // every SIMD and half the serial stage count
        const long wgs = (long)((p.N + 63) / 64) * p.B * p.H;
#if PWW_EXPERIMENTS
        if (ksplit_mode() == 1 && debug_knobs().attn_ksplit1 && wgs <= 256 && p.M >= 512 && ksplit1_extent_ok(p)) {
            // (round 5) 2 row groups x 4 key groups on ONE stage buffer: two waves per SIMD, a quarter of the keys per wave
            return launch_attn_ksplit1<T, KS, DT, 2, 4, false>(p, stream);
        }
#endif
        if (ksplit_mode() == 1 && wgs <= 256 && p.M >= 512) {
#if PWW_EXPERIMENTS
            if (debug_knobs().attn_ksplit_nw == 4) {      // A/B: 4 row groups x 2 key groups (half the workgroups, half the K / V staging traffic, 2 waves per SIMD on half the CUs)
                if ((p.D & 31) != 0) return launch_attn_ksplit<T, KS, DT, 4, 2, true>(p, stream);
                return launch_attn_ksplit<T, KS, DT, 4, 2, false>(p, stream);
            }
#endif
#if PWW_EXPERIMENTS
            if (debug_knobs().attn_ksplit_half) {
                // (round 6, measured a TIE: profiles/r06_small_attn.md) 2 row groups x 4 key groups of HALF a sub-tile each on the same 128-key
                // stages: 8 waves, two per SIMD -- the key loop gets 7 % shorter (9.6 against 10.3 us), the merge of four partial states 0.8 us longer
                if ((p.D & 31) != 0) return launch_attn_ksplit<T, KS, DT, 2, 4, true, 2>(p, stream);
                return launch_attn_ksplit<T, KS, DT, 2, 4, false, 2>(p, stream);
            }
#endif
            if ((p.D & 31) != 0) return launch_attn_ksplit<T, KS, DT, 2, 2, true>(p, stream);
            return launch_attn_ksplit<T, KS, DT, 2, 2, false>(p, stream);
        }
    }
#if PWW_EXPERIMENTS
    if constexpr (!HAS_BIAS && DT >= 4 && NW == 4) {
        // (round 6, A/B) the widest heads at the coarse levels (SD1.5 N = 256 d = 160: 32 workgroups of 4 waves, 4 stages of one 64-key tile): 4 row
        // groups x 2 key groups of a 32-key block each on the same double-buffered stages -- 8 waves, half the dependent chain per stage
        const long wgs128 = (long)((p.N + 127) / 128) * p.B * p.H;
        if (ksplit_mode() == 1 && debug_knobs().attn_ksplit_half && wgs128 <= 256 && p.M >= 128) {
            if ((p.D & 31) != 0) return launch_attn_ksplit<T, KS, DT, 4, 2, true, 1>(p, stream);
            return launch_attn_ksplit<T, KS, DT, 4, 2, false, 1>(p, stream);
        }
    }
    if constexpr (!HAS_BIAS && DT >= 4 && NW == 4) {
        // (round 5) the widest heads at the coarsest levels (SD1.5 N = 256 d = 160: 32 workgroups of 4 waves walking 4 tiles each): 64-row
        // workgroups of 2 row groups x 2 key groups -- twice the workgroups, half the tiles per wave
        const long wgs = (long)((p.N + 63) / 64) * p.B * p.H;
        if (ksplit_mode() == 1 && debug_knobs().attn_ksplit1 && wgs <= 256 && p.M >= 128 && ksplit1_extent_ok(p)) {
            return launch_attn_ksplit1<T, KS, DT, 2, 2, false>(p, stream);
        }
    }
#endif
    // head dims with padding columns in the V tile get the row sum from the MFMA (self-attention path)
    if constexpr (!HAS_BIAS) {
        if ((p.D & 31) != 0) return launch_attn_rs<T, KS, DT, NW, false, true>(p, stream);
    }
    return launch_attn_rs<T, KS, DT, NW, HAS_BIAS, false>(p, stream);
}

template <typename T, int NW, bool HAS_BIAS> static int dispatch_d(const AttnParams &p, hipStream_t s) {
    const int D = p.D;
    if (D <= 48) return launch_attn<T, 3, 2, NW, HAS_BIAS>(p, s);
    if (D <= 64) return launch_attn<T, 4, 2, NW, HAS_BIAS>(p, s);
    if constexpr (NW == 8) { set_error("attn_fwd: internal dispatch error"); return PWW_EINVAL; } else {
    if (D <= 80) return launch_attn<T, 5, 3, NW, HAS_BIAS>(p, s);
    if (D <= 96) return launch_attn<T, 6, 3, NW, HAS_BIAS>(p, s);
    if constexpr (NW == 2) { set_error("attn_fwd: internal dispatch error"); return PWW_EINVAL; } else {   // D > 96 always gets 4 waves
    if (D <= 128) return launch_attn<T, 8, 4, NW, HAS_BIAS>(p, s);
    return launch_attn<T, 10, 5, NW, HAS_BIAS>(p, s);
    }
    }
}

// 4-wave workgroups for big launches and always for the widest heads (2-wave groups would need > 512 registers per
// lane for their share of the d=160 K/V staging). The biased attention kernels, the score reduction and the fused
// cross-attention kernel all follow this one rule, so their per-workgroup partial statistics have the same granularity.
static bool wide_groups(int B, int H, int N, int D) {
    const long rows32 = (long)((N + 31) / 32) * B * H;
    return (rows32 >= 4 * 256 && N >= 128) || D > 96;
}

template <typename T> static int dispatch_nw(const AttnParams &p, hipStream_t s) {
    // Fewer waves per workgroup when the problem is too small to give every CU a 4-wave block.
    const long rows32 = (long)((p.N + 31) / 32) * p.B * p.H;  // 32-row wave tasks
    const bool wide = wide_groups(p.B, p.H, p.N, p.D);
    if (p.bias) return wide ? dispatch_d<T, 4, true>(p, s) : dispatch_d<T, 2, true>(p, s);
    // 8-wave workgroups (256 query rows per K/V stage) once they still give every CU a workgroup
    static int nw8 = -1;
    if (nw8 < 0) { nw8 = debug_knobs().attn_nw8; }
    if (nw8 && rows32 >= 8 * 256 && p.N >= 256 && p.D <= 64) return dispatch_d<T, 8, false>(p, s);
    return wide ? dispatch_d<T, 4, false>(p, s) : dispatch_d<T, 2, false>(p, s);
}

}  // namespace pww
