// Fused attention forward for the Paint-with-Words hot path on gfx950 (MI355X):
//     O = softmax((Q K^T + c[b] * bias) * scale) V
// One kernel covers both attention flavours of the reference's inj_forward
// (paint_with_words/paint_with_words.py:83-118): self-attention (M = N, no bias) and
// cross-attention over the 77 prompt tokens with the per-token mask bias added BEFORE the
// 1/sqrt(D) scaling (:112). Flash-style: the [N, M] score tensor the reference materialises four
// times (:87, :112, :114, :116) never leaves registers.
//
// Design (see pww_tile.h for the lane geometry):
//   * workgroup = NW waves, each wave owns 32 query rows; all waves share the K/V tile in LDS.
//   * scores are computed transposed (S^T = K Q^T) so a lane owns one query row: the row max/sum
//     need a single cross-half exchange, the rescale factor is lane-local, and the exponentiated
//     P^T registers are already in MFMA B-operand order for O^T = V^T P^T.
//   * V is stored transposed in LDS (Vt[d][key]) by the staging pass, so the A operand of the PV
//     MFMA is one ds_read_b128 per lane; K keeps its row-major image. Both row strides are padded
//     by 16 B, which makes every ds_read_b128 lane group hit 16 distinct bank slots.
//   * global->LDS staging is split (issue the next tile's loads before computing the current
//     tile, write them to LDS after the barrier), hiding HBM/L2 latency under the MFMAs.
//   * blockIdx.x enumerates (b, h) fastest: with the observed block -> XCD round-robin every XCD's
//     L2 keeps the K/V of a fixed subset of heads while all query blocks of those heads stream by.
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
};

template <int DT> struct VTile {
    static constexpr int ROWS = DT * 32;               // head dim padded to the MFMA M granularity
    static constexpr int STRIDE = KVBLK * 2 + 16;      // bytes per d-row (64 keys + pad)
    static constexpr int BYTES = ROWS * STRIDE;
    static constexpr int NUNIT = (KVBLK / 4) * (ROWS / 8);  // (4 keys x 8 d) transpose units
};

// One transpose unit: 4 consecutive keys x 8 consecutive d, i.e. four 16-byte global loads.
template <typename T, int DT, int NT, int VPT>
__device__ __forceinline__ void vtile_load(uint4 (&vreg)[VPT][4], const T *Vp, long v_sm, int key0,
                                           int M, int D, int tid) {
    typedef VTile<DT> VT;
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
        const int u = tid + i * NT;
        const int kg = u & 15, dc = u >> 4;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            uint4 val = make_uint4(0, 0, 0, 0);
            const int gk = key0 + kg * 4 + kk;
            if (u < VT::NUNIT && gk < M && dc * 8 < D)
                val = *reinterpret_cast<const uint4 *>(Vp + (long)gk * v_sm + dc * 8);
            vreg[i][kk] = val;
        }
    }
}

__device__ __forceinline__ uint32_t half_of(const uint4 &v, int j) {
    const uint32_t w = (j >> 1) == 0 ? v.x : (j >> 1) == 1 ? v.y : (j >> 1) == 2 ? v.z : v.w;
    return (j & 1) ? (w >> 16) : (w & 0xffffu);
}

template <int DT, int NT, int VPT>
__device__ __forceinline__ void vtile_store(const uint4 (&vreg)[VPT][4], char *Vs, int tid) {
    typedef VTile<DT> VT;
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
        const int u = tid + i * NT;
        if (u < VT::NUNIT) {
            const int kg = u & 15, dc = u >> 4;
            char *dst = Vs + (dc * 8) * VT::STRIDE + kg * 8;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                uint2 w;
                w.x = half_of(vreg[i][0], j) | (half_of(vreg[i][1], j) << 16);
                w.y = half_of(vreg[i][2], j) | (half_of(vreg[i][3], j) << 16);
                *reinterpret_cast<uint2 *>(dst + j * VT::STRIDE) = w;
            }
        }
    }
}

template <typename T, int KS, int DT, int NW>
__global__ void __launch_bounds__(NW * 64) attn_fwd_kernel(const AttnParams p) {
    typedef typename Vec<T>::v8 V8;
    typedef typename Vec<T>::v4 V4;
    typedef KTile<KS> KT;
    typedef VTile<DT> VT;
    constexpr int NT = NW * 64;
    constexpr int KPT = (KT::NCHUNK + NT - 1) / NT;
    constexpr int VPT = (VT::NUNIT + NT - 1) / NT;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *Ks = smem;
    char *Vs = smem + KT::BYTES;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int hi = lane >> 5, l31 = lane & 31;
    const int BH = p.B * p.H;
    const int bh = blockIdx.x % BH, qb = blockIdx.x / BH;
    const int b = bh / p.H, h = bh - b * p.H;

    const T *Qp = reinterpret_cast<const T *>(p.q) + b * p.q_sb + h * p.q_sh;
    const T *Kp = reinterpret_cast<const T *>(p.k) + b * p.k_sb + h * p.k_sh;
    const T *Vp = reinterpret_cast<const T *>(p.v) + b * p.v_sb + h * p.v_sh;
    T *Op = reinterpret_cast<T *>(p.o) + b * p.o_sb + h * p.o_sh;

    const int qrow = (qb * NW + wave) * 32 + l31;
    const bool qvalid = qrow < p.N;

    V8 qf[KS];
    load_q_frags<T, KS>(qf, Qp + (long)qrow * p.q_sn, qvalid, hi, p.D);

    const bool has_bias = p.bias != nullptr;
    const float *bias_row = nullptr;
    float coeff = 1.f;
    if (has_bias) {
        bias_row = p.bias + b * p.b_sb + h * p.b_sh + (long)qrow * p.b_sn;
        if (p.bias_coeff) coeff = p.bias_coeff[b];
    }
    const float c1 = p.scale_log2e;
    const float cb = coeff * c1;

    f32x16 oacc[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[dt][r] = 0.f;
    float m_run = -INFINITY;  // running row max (log2 domain), identical in both half-waves
    float l_run = 0.f;        // running row sum, PARTIAL per half-wave (combined in the epilogue)

    uint4 kreg[KPT];
    uint4 vreg[VPT][4];
    const int ntiles = (p.M + KVBLK - 1) / KVBLK;
    ktile_load<T, KS, NT, KPT>(kreg, Kp, p.k_sm, 0, p.M, p.D, tid);
    vtile_load<T, DT, NT, VPT>(vreg, Vp, p.v_sm, 0, p.M, p.D, tid);

    for (int t = 0; t < ntiles; ++t) {
        const int key0 = t * KVBLK;
        __syncthreads();  // every wave is done reading the previous tile
        ktile_store<KS, NT, KPT>(kreg, Ks, tid);
        vtile_store<DT, NT, VPT>(vreg, Vs, tid);
        __syncthreads();
        if (t + 1 < ntiles) {  // in flight while this tile is computed
            ktile_load<T, KS, NT, KPT>(kreg, Kp, p.k_sm, key0 + KVBLK, p.M, p.D, tid);
            vtile_load<T, DT, NT, VPT>(vreg, Vp, p.v_sm, key0 + KVBLK, p.M, p.D, tid);
        }

        f32x16 s[2];
        score_tile<T, KS>(s, qf, Ks, key0, p.M, l31, hi);

        // logits in the log2 domain: (s + c*bias) * scale * log2(e); keys past M -> -inf
        float tmax = -INFINITY;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = key0 + key_of(kb, r, hi);
                float x = s[kb][r] * c1;
                if (has_bias) {
                    float bv = 0.f;
                    if (qvalid && key < p.M) bv = bias_row[(long)key * p.b_sm];
                    x = fmaf(bv, cb, x);
                }
                x = key < p.M ? x : -INFINITY;
                s[kb][r] = x;
                tmax = fmaxf(tmax, x);
            }
        }
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
        const float m_new = fmaxf(m_run, tmax);  // finite: key0 < M so at least one key is live
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
        m_run = m_new;

        float psum = 0.f;
        V8 pf[2][2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float pv = __builtin_amdgcn_exp2f(s[kb][r] - m_new);
                psum += pv;
                pf[kb][r >> 3][r & 7] = (T)pv;
            }
        }
        l_run = l_run * alpha + psum;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) oacc[dt][r] *= alpha;

        // O^T[d][row] += Vt[d][key] * P^T[key][row]
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            if (key0 + kb * 32 < p.M) {
#pragma unroll
                for (int k2 = 0; k2 < 2; ++k2) {
                    const char *vbase = Vs + l31 * VT::STRIDE + (kb * 32 + k2 * 16 + hi * 8) * 2;
#pragma unroll
                    for (int dt = 0; dt < DT; ++dt) {
                        const V8 vf = *reinterpret_cast<const V8 *>(vbase + dt * 32 * VT::STRIDE);
                        oacc[dt] = mfma32(vf, pf[kb][k2], oacc[dt]);
                    }
                }
            }
        }
    }

    // epilogue: normalise and write O[row][d]; register r of tile dt is d = dt*32 + (r&3) + 8*(r>>2) + 4*hi
    const float l_tot = l_run + __shfl_xor(l_run, 32);
    const float inv = 1.f / l_tot;
    if (qvalid) {
        T *orow = Op + (long)qrow * p.o_sn;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int d = dt * 32 + g * 8 + hi * 4;
                if (d < p.D) {
                    V4 out;
#pragma unroll
                    for (int j = 0; j < 4; ++j) out[j] = (T)(oacc[dt][g * 4 + j] * inv);
                    *reinterpret_cast<V4 *>(orow + d) = out;
                }
            }
        }
    }
}

// ---- host dispatch ---------------------------------------------------------------------------

template <typename T, int KS, int DT, int NW>
static int launch_attn(const AttnParams &p, hipStream_t stream) {
    constexpr size_t lds = KTile<KS>::BYTES + VTile<DT>::BYTES;
    const int qblocks = (p.N + NW * 32 - 1) / (NW * 32);
    const dim3 grid((unsigned)(qblocks * p.B * p.H));
    auto kern = attn_fwd_kernel<T, KS, DT, NW>;
    if (lds > 48 * 1024) {
        static thread_local bool done = false;
        if (!done) {
            if (check_hip(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds),
                          "hipFuncSetAttribute"))
                return PWW_EHIP;
            done = true;
        }
    }
    hipLaunchKernelGGL(kern, grid, dim3(NW * 64), lds, stream, p);
    return check_hip(hipGetLastError(), "attn_fwd_kernel launch");
}

template <typename T, int NW> static int dispatch_d(const AttnParams &p, hipStream_t s) {
    const int D = p.D;
    if (D <= 48) return launch_attn<T, 3, 2, NW>(p, s);
    if (D <= 64) return launch_attn<T, 4, 2, NW>(p, s);
    if (D <= 80) return launch_attn<T, 5, 3, NW>(p, s);
    if (D <= 96) return launch_attn<T, 6, 3, NW>(p, s);
    if (D <= 128) return launch_attn<T, 8, 4, NW>(p, s);
    return launch_attn<T, 10, 5, NW>(p, s);
}

template <typename T> static int dispatch_nw(const AttnParams &p, hipStream_t s) {
    // Fewer waves per workgroup when the problem is too small to give every CU a 4-wave block.
    const long rows32 = (long)((p.N + 31) / 32) * p.B * p.H;  // 32-row wave tasks
    if (rows32 >= 4 * 256 && p.N >= 128) return dispatch_d<T, 4>(p, s);
    return dispatch_d<T, 2>(p, s);
}

static bool aligned16(const void *ptr) { return (reinterpret_cast<uintptr_t>(ptr) & 15) == 0; }

int attn_fwd(const void *q, const void *k, const void *v, void *o, const float *bias,
             const float *bias_coeff, const pww_attn_desc_t *d, hipStream_t stream) {
    if (!d || !q || !k || !v || !o) { set_error("attn_fwd: null argument"); return PWW_EINVAL; }
    if (d->B <= 0 || d->H <= 0 || d->N <= 0 || d->M <= 0 || d->D <= 0) {
        set_error("attn_fwd: non-positive dimension (B=%d H=%d N=%d M=%d D=%d)", d->B, d->H, d->N, d->M, d->D);
        return PWW_EINVAL;
    }
    if (d->D % 8 != 0 || d->D > PWW_MAX_HEAD_DIM) {
        set_error("attn_fwd: head dim %d unsupported (multiple of 8, <= %d)", d->D, PWW_MAX_HEAD_DIM);
        return PWW_ENOTSUP;
    }
    if (d->dtype != PWW_DTYPE_F16 && d->dtype != PWW_DTYPE_BF16) {
        set_error("attn_fwd: dtype %d unsupported", d->dtype);
        return PWW_ENOTSUP;
    }
    if (!aligned16(q) || !aligned16(k) || !aligned16(v) || !aligned16(o)) {
        set_error("attn_fwd: q/k/v/o must be 16-byte aligned");
        return PWW_EINVAL;
    }
    for (int i = 0; i < 3; ++i) {
        if (d->q_stride[i] % 8 || d->k_stride[i] % 8 || d->v_stride[i] % 8 || d->o_stride[i] % 4) {
            set_error("attn_fwd: strides must be multiples of 8 elements (o: 4)");
            return PWW_EINVAL;
        }
    }
    if ((long)d->B * d->H * ((d->N + 31) / 32) > 0x7fffffffL) { set_error("attn_fwd: grid too large"); return PWW_EINVAL; }
    if (!arch_ok()) return PWW_ENOTSUP;

    AttnParams p;
    p.q = q; p.k = k; p.v = v; p.o = o;
    p.bias = bias; p.bias_coeff = bias ? bias_coeff : nullptr;
    p.B = d->B; p.H = d->H; p.N = d->N; p.M = d->M; p.D = d->D;
    p.q_sb = d->q_stride[0]; p.q_sh = d->q_stride[1]; p.q_sn = d->q_stride[2];
    p.k_sb = d->k_stride[0]; p.k_sh = d->k_stride[1]; p.k_sm = d->k_stride[2];
    p.v_sb = d->v_stride[0]; p.v_sh = d->v_stride[1]; p.v_sm = d->v_stride[2];
    p.o_sb = d->o_stride[0]; p.o_sh = d->o_stride[1]; p.o_sn = d->o_stride[2];
    p.b_sb = d->bias_stride[0]; p.b_sh = d->bias_stride[1]; p.b_sn = d->bias_stride[2]; p.b_sm = d->bias_stride[3];
    p.scale_log2e = d->scale * 1.4426950408889634f;
    return d->dtype == PWW_DTYPE_F16 ? dispatch_nw<f16>(p, stream) : dispatch_nw<bf16>(p, stream);
}

}  // namespace pww
