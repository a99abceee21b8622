// Cross-attention of the C = 320 layers WITH the output projection in the same launch:
//     out = to_out[0]( merge_heads( softmax((Q K^T + c[b] w) scale) V ) ) [+ residual]        paint_with_words.py:106-123
// One workgroup owns 128 query rows of one image and ALL heads (the projection contracts over the merged head axis, so whoever applies it
// must hold every head's O of its rows). O never exists in HBM; the results equal `linear(pww_cross_attn_fwd_parts(...), W, b)` bit for
// bit on every shape of tests/test_round5_gpu.py (O is rounded to the storage type where the two-launch path stores it, the 320-term dot
// products accumulate in fp32 in channel order, bias added in fp32, ONE rounding; the optional residual is a second rounding).
//
// What it costs, by construction: heads are sequential inside a workgroup, so a launch has B * ceil(N / 128) workgroups instead of
// B * H * ceil(N / 128), and the O image (84 KB) + K / V / bias tile fill the CU's LDS: one workgroup of four waves per CU. Measured
// (profiles/r05_to_out_epilogue.md; N = 4096, 8 x 40, back to back): 34 us per workgroup round -- prologue 4.7, eight heads 20.6
// (2.6 each: one wave per SIMD, nothing overlaps the softmax), projection 6.5, stores 2.1 -- against 18.3 (small kernel 7.9 + stock GEMM
// 8.5) at the 2 folded rows of a batch-1 request, 37.9 against 41.2 at 8 rows, 71.2 against 72.7 at 16, 139 against 136 at 32; config 3
// end to end 6.08 against 6.10 images/s. Not the default anywhere: PWW_FUSE_TO_OUT=1 (pww_hip/attention.py) takes it where it fits.
#include <string.h>
#include "pww_attn_core.h"
#include "pww_cross_tile.h"

namespace pww {

struct OutParams {
    AttnParams a;             // a.o unused; a.bias_coeff = the row gate [B] (or null)
    const double *parts;      // [B][nparts][4] or null (stat_kind == NONE)
    int nparts;
    double *stats_out;        // optional [B][4]
    int tile_stride;          // floats per LDS bias-tile row (16 / 32 / 64)
    const void *w;            // [C][H * D] row-major nn.Linear weight (storage type of q)
    const void *wb;           // [C] or null
    const void *res;          // [B][N][C] or null
    void *out;                // [B][N][C]
    long w_sr;                // row stride of w (elements)
    long out_sb, out_sn, res_sb, res_sn;
};

constexpr int OUT_NW = 4;                  // waves of 32 query rows
constexpr int OUT_ROWS = 2 * KVBLK;        // key rows the LDS image has room for (M <= 128)
constexpr int OUT_TILE_LOADS = 8;          // 16-byte pieces per lane of a wave's 32 bias rows: 64 columns at most

// A buffer descriptor whose words are FORCED into scalar registers: hipcc otherwise forms `(M - 1) * stride` with the vector multiply it
// already has for the per-thread offsets and wraps every load in a waterfall loop (readfirstlane / compare / branch per load).
template <typename T>
__device__ __forceinline__ auto uniform_srd(const T *base, unsigned bytes) {
    const unsigned long long a = reinterpret_cast<unsigned long long>(base);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<T *>(((unsigned long long)hi << 32) | lo), 0, __builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}

// Two phases, so that neither needs more than the 256 architectural registers (beyond them hipcc selects the accumulator-file form of
// every MFMA and copies each score / O^T tile to and from it for the softmax: measured 40 us per workgroup in the first, one-phase
// version of this kernel):
//   phase 1  the heads one after the other, exactly the small kernel's tile code (pww_cross_lean.hip); the next head's Q / K / V in flight
//            under the current head's MFMAs; the normalised O_h goes to the workgroup's [128 rows][C] LDS image (rounded to the storage
//            type: what the two-launch path stores) -- rows are wave-private, no barrier for them.
//   phase 2  out^T[c][row] = sum_k W[c][k] O[row][k]: a plain 128 x 320 x 320 GEMM, W in 32-channel chunks through two LDS buffers (they
//            overlay phase 1's K / V / bias tile), ONE barrier per chunk, 10 accumulator tiles of 32 output channels per wave.
// LDS: [phase 1: K 128 rows | V 128 rows | bias tile]  overlaid by  [phase 2: W chunk buffers 0, 1]; [O image]; [bias vector]
constexpr int OUT_WCH = 32;                              // channels of a W chunk
constexpr int OUT_WSTRIDE = OUT_WCH * 2 + 16;            // bytes per W row in LDS (odd multiple of 16: conflict-free 16-byte fragment reads)

template <typename T, int KS, int MT>
__global__ void __launch_bounds__(OUT_NW * 64, 2) cross_out_kernel(const OutParams op) {
    typedef typename Vec<T>::v8 V8;
    typedef KTile<KS> KT;
    constexpr int DT = 2;
    typedef VTile<DT> VT;
    constexpr bool RSM = KS * 16 < DT * 32;
    constexpr int NT = OUT_NW * 64, C = MT * 32;
    constexpr int KRPP = NT / KT::CHK, VRPP = NT / VT::CHK;
    constexpr int KPASS = (OUT_ROWS + KRPP - 1) / KRPP, VPASS = (OUT_ROWS + VRPP - 1) / VRPP;
    constexpr int K_BYTES = OUT_ROWS * KT::STRIDE, V_BYTES = OUT_ROWS * VT::STRIDE;
    constexpr int WBUF = C * OUT_WSTRIDE, WPASS = C / (NT / 4), NCHUNK = C / OUT_WCH;      // a pass of the workgroup = 64 rows x 4 pieces of 16 bytes
    constexpr int OSTRIDE = C * 2 + 16, O_BYTES = OUT_NW * 32 * OSTRIDE, WB_BYTES = C * 2;
    const AttnParams &p = op.a;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tile_bytes = OUT_NW * 32 * op.tile_stride * 4;
    const int a_bytes = max(K_BYTES + V_BYTES + tile_bytes, 2 * WBUF);
    char *Kl = smem, *Vl = smem + K_BYTES, *tile = smem + K_BYTES + V_BYTES, *Wl = smem, *Ol = smem + a_bytes, *wbl = Ol + O_BYTES;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hi = lane >> 5, l31 = lane & 31;
    const int qb = blockIdx.x, b = blockIdx.y;
    tl_stamp(p, 0);

    // ---- prologue: gate, partials, the wave's bias rows, the projection's bias vector, head 0's operands -- all requested before any wait
    const float gate = p.bias_coeff ? p.bias_coeff[b] : 1.f;
    const float c0 = coeff_scalar_of(p);
    const bool maybe_biased = p.bias != nullptr;
    const bool need_stat = p.stat_kind != PWW_STAT_NONE;
    const PartsWant want(p.stat_kind, op.stats_out != nullptr);
    PartsRegs pr;
    parts_request(pr, op.parts, op.nparts, b, need_stat && maybe_biased, want, lane);

    const int qrow = (qb * OUT_NW + wave) * 32 + l31;
    const bool qvalid = qrow < p.N;
    const unsigned q_off = (unsigned)((long)qrow * p.q_sn * 2);
    const int kr = tid / KT::CHK, kc = tid - kr * KT::CHK;
    const int vr = tid / VT::CHK, vc = tid - vr * VT::CHK;
    const bool k_act = kr < KRPP, v_act = vr < VRPP;
    const unsigned k0 = (k_act && kc * 8 < p.D) ? (unsigned)((kr * p.k_sm + kc * 8) * 2) : OOB_OFF, kstep = (unsigned)(KRPP * p.k_sm * 2);
    const unsigned v0 = (v_act && vc * 8 < p.D) ? (unsigned)((vr * p.v_sm + vc * 8) * 2) : OOB_OFF, vstep = (unsigned)(VRPP * p.v_sm * 2);
    const unsigned q_bytes = (unsigned)(((long)(p.N - 1) * p.q_sn + p.D) * 2), k_bytes = (unsigned)(((long)(p.M - 1) * p.k_sm + p.D) * 2),
                   v_bytes = (unsigned)(((long)(p.M - 1) * p.v_sm + p.D) * 2);
    const T *Qb = reinterpret_cast<const T *>(p.q) + b * p.q_sb;
    const T *Kb = reinterpret_cast<const T *>(p.k) + b * p.k_sb;
    const T *Vb = reinterpret_cast<const T *>(p.v) + b * p.v_sb;

    V8 qn[KS];
    u32x4 kreg[KPASS], vreg[VPASS];
    // the operands of head h: Q fragments of the wave's rows, K / V chunks (rows past M and the head-dim padding lie beyond the
    // descriptors: zeros, no memory traffic, no compare)
    auto request = [&](int h) {
        load_q_frags_buf<T, KS>(qn, uniform_srd(Qb + h * p.q_sh, q_bytes), q_off, hi, p.D);
        const auto srd_k = uniform_srd(Kb + h * p.k_sh, k_bytes);
        const auto srd_v = uniform_srd(Vb + h * p.v_sh, v_bytes);
#pragma unroll
        for (int i = 0; i < KPASS; ++i) kreg[i] = __builtin_amdgcn_raw_buffer_load_b128(srd_k, k0 + (unsigned)i * kstep, 0, 0);
#pragma unroll
        for (int i = 0; i < VPASS; ++i) vreg[i] = __builtin_amdgcn_raw_buffer_load_b128(srd_v, v0 + (unsigned)i * vstep, 0, 0);
    };
    request(0);

    // W chunk c = channels 32 c .. 32 c + 31 of every output row: thread -> (row tid / 4 of a 64-row pass, 16-byte piece tid % 4)
    u32x4 wreg[WPASS];
    const auto srd_w = uniform_srd(reinterpret_cast<const T *>(op.w), (unsigned)((long)C * op.w_sr * 2));
    const unsigned w0 = (unsigned)(((tid >> 2) * op.w_sr + (tid & 3) * 8) * 2), wstep = (unsigned)(64 * op.w_sr * 2);
    char *wd = Wl + (tid >> 2) * OUT_WSTRIDE + (tid & 3) * 16;
    auto request_w = [&](int c) {
#pragma unroll
        for (int i = 0; i < WPASS; ++i) wreg[i] = __builtin_amdgcn_raw_buffer_load_b128(srd_w, w0 + (unsigned)i * wstep + (unsigned)(c * OUT_WCH * 2), 0, 0);
    };

    // bias rows of THIS WAVE's 32 query rows (the map is shared by the heads: staged once; wave-private, no barrier needed for it)
    const int cprl = op.tile_stride == 16 ? 2 : op.tile_stride == 32 ? 3 : 4;
    const int pc = lane & ((1 << cprl) - 1), rowl0 = lane >> cprl, tile_passes = 1 << (cprl - 1), tile_rp = 64 >> cprl;
    const bool t_col = pc * 4 < p.bias_cols;
    u32x4 treg[OUT_TILE_LOADS];
    BiasRef bias;
    if (maybe_biased) {
        const float *bbase = p.bias + b * p.b_sb;
        const unsigned bytes = (unsigned)((((long)(p.N - 1) * p.b_sn + (long)(p.M - 1) * p.b_sm) + 1) * 4);
        bias.srd = uniform_srd(bbase, bytes);
        const unsigned t0 = t_col ? (unsigned)((((long)(qb * OUT_NW + wave) * 32 + rowl0) * p.b_sn + pc * 4) * 4) : OOB_OFF, tstep = (unsigned)(tile_rp * p.b_sn * 4);
#pragma unroll
        for (int i = 0; i < OUT_TILE_LOADS; ++i) treg[i] = __builtin_amdgcn_raw_buffer_load_b128(bias.srd, i < tile_passes ? t0 + (unsigned)i * tstep : OOB_OFF, 0, 0);
    }
    u32x4 wbreg = {0u, 0u, 0u, 0u};
    if (op.wb) wbreg = __builtin_amdgcn_raw_buffer_load_b128(uniform_srd(reinterpret_cast<const T *>(op.wb), (unsigned)WB_BYTES), tid < MT * 4 ? (unsigned)tid * 16u : OOB_OFF, 0, 0);
    tl_stamp(p, 6);

    float coeff = 0.f;
    const bool biased = maybe_biased && gate != 0.f;      // workgroup-uniform
    if (biased) {
        coeff = c0;
        if (need_stat) {
            double st[4];
            parts_fold(st, pr, op.parts, op.nparts, b, want, lane);
            if (tid == 0 && op.stats_out && qb == 0) {
                double *so = op.stats_out + (long)b * 4;
                so[0] = st[0]; so[1] = st[1]; so[2] = st[2]; so[3] = st[3];
            }
            coeff = stat_coefficient(c0, p.stat_kind, st, p.stat_count);
        }
        if (p.bias_coeff) coeff = coeff * gate;
    }
    tl_stamp(p, 3);
    if (tid < MT * 4) *reinterpret_cast<u32x4 *>(wbl + tid * 16) = wbreg;      // (zeros without a bias vector; read after phase 2's barriers)
    if (biased) {
        if (t_col) {
#pragma unroll
            for (int i = 0; i < OUT_TILE_LOADS; ++i)
                if (i < tile_passes) {
                    const int row = wave * 32 + rowl0 + i * tile_rp;
                    *reinterpret_cast<u32x4 *>(tile + (long)row * op.tile_stride * 4 + ((pc ^ tile_swz(row, 1 << cprl)) << 4)) = treg[i];
                }
        }
        bias_ref_tile(bias, tile, wave * 32 + l31, op.tile_stride, p.bias_cols, hi);
    }

    // ---- phase 1: the heads
    const int rows = min(OUT_ROWS, (p.M + 31) & ~31);
    char *kd = Kl + kr * KT::STRIDE + kc * 16, *vd = Vl + vr * VT::STRIDE + vc * 16;
    const T one = (T)1.0f;
    unsigned short one_bits;
    __builtin_memcpy(&one_bits, &one, 2);
    const bool v_one = RSM && vc * 8 == p.D;              // first padding chunk of V: channel D = 1.0 (the softmax denominator's column)
    const float c1 = p.scale_log2e;
    T *orow_l = reinterpret_cast<T *>(Ol + (wave * 32 + l31) * OSTRIDE);      // the lane's row of the O image

    for (int h = 0; h < p.H; ++h) {
        if (h) __syncthreads();          // every wave is done with head h - 1's K / V
        if (k_act) {
#pragma unroll
            for (int i = 0; i < KPASS; ++i)
                if (i * KRPP < rows && ((i + 1) * KRPP <= OUT_ROWS || i * KRPP + kr < OUT_ROWS)) *reinterpret_cast<u32x4 *>(kd + i * KRPP * KT::STRIDE) = kreg[i];
        }
        if (v_act) {
#pragma unroll
            for (int i = 0; i < VPASS; ++i)
                if (i * VRPP < rows && ((i + 1) * VRPP <= OUT_ROWS || i * VRPP + vr < OUT_ROWS))
                    *reinterpret_cast<u32x4 *>(vd + i * VRPP * VT::STRIDE) = v_one ? u32x4{(unsigned)one_bits, 0u, 0u, 0u} : vreg[i];
        }
        V8 qf[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) qf[ks] = qn[ks];
        __syncthreads();
        if (h + 1 < p.H) request(h + 1);      // in flight under this head's MFMAs
        else request_w(0);                    // ... and the projection's first W chunk under the last head's
        if (h == 0) tl_stamp(p, 1);

        f32x16 oacc[DT];
        float m_run = -INFINITY, l_run = 0.f;
        if (biased) {
            attn_tile<T, KS, DT, 2, false, RSM, 1>(oacc, m_run, l_run, qf, Kl, Vl, 0, p.M, l31, hi, bias, coeff, c1);
            if (KVBLK < p.M) attn_tile<T, KS, DT, 2, true, RSM, 2>(oacc, m_run, l_run, qf, Kl + KT::BYTES, Vl + VT::BYTES, KVBLK, p.M, l31, hi, bias, coeff, c1);
        } else {
            attn_tile<T, KS, DT, 0, false, RSM, 1>(oacc, m_run, l_run, qf, Kl, Vl, 0, p.M, l31, hi, bias, coeff, c1);
            if (KVBLK < p.M) attn_tile<T, KS, DT, 0, true, RSM, 2>(oacc, m_run, l_run, qf, Kl + KT::BYTES, Vl + VT::BYTES, KVBLK, p.M, l31, hi, bias, coeff, c1);
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
        // O_h -> the lane's row of the O image, channels h D .. h D + D - 1 (16-byte pieces: D % 8 == 0)
        store_o_block<T, DT>(orow_l + h * p.D, oacc, 1.f / l_tot, p.D, hi, true, true);
    }
    tl_stamp(p, 2);

    // ---- phase 2: out^T = W O^T
    __syncthreads();                     // K / V / bias tile are dead: the W buffers take their place
#pragma unroll
    for (int i = 0; i < WPASS; ++i) *reinterpret_cast<u32x4 *>(wd + i * 64 * OUT_WSTRIDE) = wreg[i];
    __syncthreads();
    f32x16 acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mt][r] = 0.f;
    const char *wl = Wl + swap23(l31) * OUT_WSTRIDE + hi * 16;      // A fragments: row = output channel, like a key row of the score tile
    const char *ol = reinterpret_cast<const char *>(orow_l) + hi * 16;      // B fragments: the lane's own row, channels 16 ks + 8 hi ..
    for (int c = 0; c < NCHUNK; ++c) {
        if (c + 1 < NCHUNK) request_w(c + 1);
        const char *wc = wl + (c & 1) * WBUF;
        V8 of[2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) of[ks] = *reinterpret_cast<const V8 *>(ol + (c * 2 + ks) * 32);
#pragma unroll
        for (int mt = 0; mt < MT; mt += 2) {
            V8 wf[2][2];
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) wf[u][ks] = *reinterpret_cast<const V8 *>(wc + (mt + u) * 32 * OUT_WSTRIDE + ks * 32);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int u = 0; u < 2; ++u) acc[mt + u] = mfma32(wf[u][ks], of[ks], acc[mt + u]);
        }
        if (c + 1 < NCHUNK) {
            char *dst = wd + ((c + 1) & 1) * WBUF;
#pragma unroll
            for (int i = 0; i < WPASS; ++i) *reinterpret_cast<u32x4 *>(dst + i * 64 * OUT_WSTRIDE) = wreg[i];
            __syncthreads();
        }
    }
    tl_stamp(p, 5);

    // ---- epilogue: register r of tile mt = output channel mt * 32 + 16 (r >> 3) + 8 hi + (r & 7) of the lane's row: 16-byte pieces
    T *orow = reinterpret_cast<T *>(op.out) + b * op.out_sb + (long)(qvalid ? qrow : 0) * op.out_sn;
    const T *rrow = op.res ? reinterpret_cast<const T *>(op.res) + b * op.res_sb + (long)(qvalid ? qrow : 0) * op.res_sn : nullptr;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int ch = mt * 32 + 16 * j + 8 * hi;
            const V8 bv = *reinterpret_cast<const V8 *>(wbl + ch * 2);
            V8 o8;
#pragma unroll
            for (int i = 0; i < 8; ++i) o8[i] = (T)(acc[mt][j * 8 + i] + (float)bv[i]);
            if (rrow) {
                const V8 rv = qvalid ? *reinterpret_cast<const V8 *>(rrow + ch) : zero8<V8>();
#pragma unroll
                for (int i = 0; i < 8; ++i) o8[i] = (T)((float)o8[i] + (float)rv[i]);
            }
            if (qvalid) *reinterpret_cast<V8 *>(orow + ch) = o8;
        }
    }
    tl_stamp(p, 4);
}

int attn_validate(const void *q, const void *k, const void *v, void *o, const float *bias, const pww_attn_desc_t *d);
void attn_fill_params(AttnParams &p, const void *q, const void *k, const void *v, void *o, const float *bias,
                      const float *bias_coeff, const pww_attn_desc_t *d);

constexpr int OUT_MT = 10;      // 320 output channels: the cross-attention layers of the finest UNet level (SD1.5: 8 heads x 40, SD2.1: 5 x 64)

// 1 if pww_cross_attn_fwd_parts_out takes a problem of this shape (the host asks before it decides the route)
int cross_attn_out_supported(const pww_attn_desc_t *d, int C, int bias_cols) {
    if (!d || C != OUT_MT * 32 || (long)d->H * d->D != C || d->D % 8 || d->D > 64) return 0;
    if (d->M < KVBLK || d->M > 2 * KVBLK) return 0;
    if (d->dtype != PWW_DTYPE_F16 && d->dtype != PWW_DTYPE_BF16) return 0;
    if (d->bias_stride[3] != 1 || d->bias_stride[1] != 0) return 0;      // dense rows, one map for all heads
    const int m16 = (d->M + 15) & ~15;
    int bc = bias_cols > 0 ? ((bias_cols + 15) & ~15) : m16;
    if (bc > m16) bc = m16;
    if (bc > 64) return 0;
    if (((long)(d->N + 128) * d->q_stride[2] + d->D) * 2 >= (1L << 31) || (long)(d->N + 128) * d->bias_stride[2] * 4 >= (1L << 31)) return 0;
    return 1;
}

template <typename T, int KS>
static int launch_out(const OutParams &op, hipStream_t stream) {
    typedef KTile<KS> KT;
    constexpr size_t C = OUT_MT * 32;
    const size_t phase1 = (size_t)OUT_ROWS * (KT::STRIDE + VTile<2>::STRIDE) + (size_t)OUT_NW * 32 * op.tile_stride * 4, phase2 = 2 * C * OUT_WSTRIDE;
    const size_t lds = (phase1 > phase2 ? phase1 : phase2) + (size_t)OUT_NW * 32 * (C * 2 + 16) + C * 2;
    auto kern = cross_out_kernel<T, KS, OUT_MT>;
    static thread_local size_t lds_attr[8] = {0, 0, 0, 0, 0, 0, 0, 0};      // per device
    int dev = 0;
    if (check_hip(hipGetDevice(&dev), "hipGetDevice")) return PWW_EHIP;
    if (dev >= 8 || lds > lds_attr[dev]) {
        if (check_hip(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds), "hipFuncSetAttribute")) return PWW_EHIP;
        if (dev < 8) lds_attr[dev] = lds;
    }
    const int nqb = (op.a.N + OUT_NW * 32 - 1) / (OUT_NW * 32);
    if (op.a.B > 65535) { set_error("cross_attn_out: more than 65535 images"); return PWW_EINVAL; }
    launch_attn_kernel(kern, dim3((unsigned)nqb, (unsigned)op.a.B), dim3(OUT_NW * 64), lds, stream, op);
    return check_hip(hipGetLastError(), "cross_out_kernel launch");
}

int cross_attn_out(const void *q, const void *k, const void *v, void *out, const float *bias, int stat_kind, float coeff_scalar, const float *gate,
                   const pww_attn_desc_t *d, const double *parts, int nparts, double *stats_out, const pww_cross_opts_t *opts, const void *w,
                   const void *w_bias, const void *residual, const int64_t *residual_stride, hipStream_t stream) {
    if (!w || !out) { set_error("cross_attn_out: null weight or output"); return PWW_EINVAL; }
    if (int rc = attn_validate(q, k, v, out, bias, d)) return rc;
    if (!bias) { set_error("cross_attn_out: needs the bias map (a bias-free layer keeps pww_attn_fwd + the stock GEMM)"); return PWW_ENOTSUP; }
    if (stat_kind < PWW_STAT_NONE || stat_kind > PWW_STAT_ABSMAX) { set_error("cross_attn_out: bad statistic selector %d", stat_kind); return PWW_EINVAL; }
    if (stat_kind != PWW_STAT_NONE && !parts) { set_error("cross_attn_out: null partials"); return PWW_EINVAL; }
    // the same rules as pww_cross_attn_fwd_parts: the partials are addressed through a 32-bit buffer descriptor of nparts * 32 bytes
    if (stat_kind != PWW_STAT_NONE && (nparts < 1 || (reinterpret_cast<uintptr_t>(parts) & 15) || (long)nparts * 32 >= (1L << 31))) {
        set_error("cross_attn_out: partials must be 16-byte aligned, 1 <= nparts (got %d)", nparts);
        return PWW_EINVAL;
    }
    pww_cross_opts_t o;
    memset(&o, 0, sizeof(o));
    if (opts) {
        if (opts->size < 16 || opts->size > sizeof(o)) { set_error("cross_attn_out: pww_cross_opts_t.size = %u is not a known layout", opts->size); return PWW_EINVAL; }
        memcpy(&o, opts, opts->size);
    }
    if (o.bias_compact) { set_error("cross_attn_out: the compact bias form is not taken here"); return PWW_ENOTSUP; }
    const int C = d->H * d->D;
    if (!cross_attn_out_supported(d, C, o.bias_cols)) {
        set_error("cross_attn_out: unsupported problem (H * D = %d must be 320 with D <= 64, 64 <= M <= 128, dense bias rows shared by the heads, <= 64 bias columns)", C);
        return PWW_ENOTSUP;
    }
    if ((reinterpret_cast<uintptr_t>(w) & 15) || (w_bias && (reinterpret_cast<uintptr_t>(w_bias) & 15)) || (residual && (reinterpret_cast<uintptr_t>(residual) & 15))) {
        set_error("cross_attn_out: w, w_bias and residual must be 16-byte aligned");
        return PWW_EINVAL;
    }
    if (d->o_stride[0] % 8 || d->o_stride[2] % 8 || d->o_stride[2] < C) { set_error("cross_attn_out: output strides (o_stride[0], o_stride[2]) must be multiples of 8 elements, rows >= %d apart", C); return PWW_EINVAL; }
    if (residual && (!residual_stride || residual_stride[0] % 8 || residual_stride[1] % 8)) { set_error("cross_attn_out: residual strides must be multiples of 8 elements"); return PWW_EINVAL; }
    OutParams op;
    attn_fill_params(op.a, q, k, v, out, bias, gate, d);
    op.a.bias_coeff = gate;
    op.a.stats = nullptr; op.a.stat_kind = stat_kind; op.a.stat_count = (double)d->H * d->N * d->M; op.a.coeff_scalar = coeff_scalar;
    op.a.coeff_scalar_dev = o.coeff_scalar_dev;
    const int m16 = (d->M + 15) & ~15;
    int bias_cols = o.bias_cols > 0 ? ((o.bias_cols + 15) & ~15) : m16;
    if (bias_cols > m16) bias_cols = m16;
    op.a.bias_cols = bias_cols;
    op.parts = stat_kind != PWW_STAT_NONE ? parts : nullptr;
    op.nparts = stat_kind != PWW_STAT_NONE ? nparts : 0;
    op.stats_out = stat_kind != PWW_STAT_NONE ? stats_out : nullptr;
    op.tile_stride = bias_cols <= 16 ? 16 : bias_cols <= 32 ? 32 : 64;
    op.w = w; op.wb = w_bias; op.res = residual; op.out = out;
    op.w_sr = C;
    op.out_sb = d->o_stride[0]; op.out_sn = d->o_stride[2];
    op.res_sb = residual ? residual_stride[0] : 0; op.res_sn = residual ? residual_stride[1] : 0;
    if (d->dtype == PWW_DTYPE_F16) return d->D <= 48 ? launch_out<f16, 3>(op, stream) : launch_out<f16, 4>(op, stream);
    return d->D <= 48 ? launch_out<bf16, 3>(op, stream) : launch_out<bf16, 4>(op, stream);
}

}  // namespace pww
