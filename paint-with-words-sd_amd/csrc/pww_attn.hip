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
//   * K and V both keep their row-major image in LDS (16-byte global loads, 16-byte LDS stores, no staging
//     arithmetic); the PV MFMA's A operand V^T[d][8 keys] comes out of the hardware transpose read
//     ds_read_b64_tr_b16 (two per fragment). Row strides are chosen so that every lane group of a ds_read_b128
//     (K) / every 32-lane half of a transpose read (V) hits distinct banks.
//   * global->LDS staging is split (issue the next tile's loads before computing the current
//     tile, write them to LDS after the barrier), hiding HBM/L2 latency under the MFMAs.
//   * blockIdx.x enumerates (b, h) fastest: with the observed block -> XCD round-robin every XCD's
//     L2 keeps the K/V of a fixed subset of heads while all query blocks of those heads stream by.
#include "pww_attn_core.h"

namespace pww {

// kernels + dispatch: pww_attn_kernel.h, instantiated per storage type in pww_attn_inst.hip
int attn_dispatch_f16(const AttnParams &p, hipStream_t s);
int attn_dispatch_bf16(const AttnParams &p, hipStream_t s);
bool attn_wide_groups_rule(int B, int H, int N, int D);
unsigned *debug_path_counts();

bool attn_wide_groups_dims(int B, int H, int N, int D) { return attn_wide_groups_rule(B, H, N, D); }
bool attn_wide_groups(const pww_attn_desc_t *d) { return attn_wide_groups_rule(d->B, d->H, d->N, d->D); }

static int pair_major_min() {   // PWW_DEBUG=attn_pair_major=n: pair-major workgroup order for launches with at least n (image, head) pairs (0 = never; default 16)
    static int n = -2;
    if (n == -2) { n = debug_knobs().attn_pair_major; if (n <= 0) n = 0x7fffffff; }
    return n;
}

static int wide_store_mode() {   // PWW_DEBUG=attn_wide_store=0: 8-byte epilogue stores as in round 2 (A/B testing)
    static int mode = -2;
    if (mode == -2) { mode = debug_knobs().attn_wide_store; }
    return mode;
}

static bool aligned16(const void *ptr) { return (reinterpret_cast<uintptr_t>(ptr) & 15) == 0; }

// Shape / stride / alignment rules shared by every attention entry point.
int attn_validate(const void *q, const void *k, const void *v, void *o, const float *bias, const pww_attn_desc_t *d) {
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
    if (((long)d->M * d->k_stride[2] + d->D) * 2 >= (1L << 31) || ((long)d->M * d->v_stride[2] + d->D) * 2 >= (1L << 31) ||
        d->k_stride[2] < d->D || d->v_stride[2] < d->D) {
        set_error("attn_fwd: one head's K/V extent must be < 2 GiB and rows must not overlap (row stride >= D)");
        return PWW_EINVAL;
    }
    if (bias) {
        // the kernels address one (image, head) slice of the bias through a 32-bit buffer descriptor: offsets are
        // unsigned, the slice must stay below 2 GiB, and the pointer must be dword-aligned
        const int64_t *bs = d->bias_stride;
        if (bs[0] < 0 || bs[1] < 0 || bs[2] < 0 || bs[3] < 0) { set_error("attn_fwd: negative bias stride"); return PWW_EINVAL; }
        if (((long)(d->N - 1) * bs[2] + (long)(d->M - 1) * bs[3] + 1) * 4 >= (1L << 31)) {
            set_error("attn_fwd: one (image, head) slice of the bias must span < 2 GiB (strides %ld, %ld for N=%d, M=%d)",
                      (long)bs[2], (long)bs[3], d->N, d->M);
            return PWW_EINVAL;
        }
        if (reinterpret_cast<uintptr_t>(bias) & 3) { set_error("attn_fwd: bias must be 4-byte aligned"); return PWW_EINVAL; }
    }
    if (!(d->scale > 0.f)) { set_error("attn_fwd: scale must be positive (got %g)", (double)d->scale); return PWW_EINVAL; }
    if (!arch_ok()) return PWW_ENOTSUP;
    return PWW_OK;
}

void attn_fill_params(AttnParams &p, const void *q, const void *k, const void *v, void *o, const float *bias,
                      const float *bias_coeff, const pww_attn_desc_t *d) {
    p.q = q; p.k = k; p.v = v; p.o = o;
    p.bias = bias; p.bias_coeff = bias ? bias_coeff : nullptr;
    p.B = d->B; p.H = d->H; p.N = d->N; p.M = d->M; p.D = d->D;
    p.q_sb = d->q_stride[0]; p.q_sh = d->q_stride[1]; p.q_sn = d->q_stride[2];
    p.k_sb = d->k_stride[0]; p.k_sh = d->k_stride[1]; p.k_sm = d->k_stride[2];
    p.v_sb = d->v_stride[0]; p.v_sh = d->v_stride[1]; p.v_sm = d->v_stride[2];
    p.o_sb = d->o_stride[0]; p.o_sh = d->o_stride[1]; p.o_sn = d->o_stride[2];
    p.b_sb = d->bias_stride[0]; p.b_sh = d->bias_stride[1]; p.b_sn = d->bias_stride[2]; p.b_sm = d->bias_stride[3];
    p.scale_log2e = d->scale * 1.4426950408889634f;
    p.stats = nullptr; p.stat_kind = PWW_STAT_NONE; p.stat_count = 1.0; p.coeff_scalar = 1.f;
    p.coeff_scalar_dev = nullptr; p.bias_cols = 0; p.timeline = debug_timeline();
    p.pair_major = ((d->B * d->H) % 8 == 0 && d->B * d->H >= pair_major_min() && !bias) ? 1 : 0;
    // d = 40: a head's 80-byte slice shares its 128-byte lines with its neighbours: keep groups of adjacent heads on one XCD (wg_to_pair_block;
    // contiguous heads only: h stride = D elements in q, k, v and o). PWW_DEBUG=attn_head_pairs=0: round 5's order, 4: groups of four (A/B)
    {
        const int G = debug_knobs().attn_head_pairs, BH = d->B * d->H;
        const bool contiguous = d->q_stride[1] == d->D && d->k_stride[1] == d->D && d->v_stride[1] == d->D && d->o_stride[1] == d->D;
        if (p.pair_major && (d->D * 2) % 128 != 0 && contiguous && G >= 2) {
            if (G >= 4 && d->H % 4 == 0 && BH % 32 == 0) p.pair_major = 4;
            else if (d->H % 2 == 0 && BH % 16 == 0) p.pair_major = 2;
        }
    }
    p.o_wide = (d->o_stride[0] % 8 == 0 && d->o_stride[1] % 8 == 0 && d->o_stride[2] % 8 == 0 && wide_store_mode()) ? 1 : 0;
    p.timeline_wgs = (unsigned)(debug_timeline_bytes() / (TL_SLOTS * sizeof(unsigned long long)));
    // folded-reference d = 40 kernel (pww_attn_kernel.h): FoldLimit of the storage type (PWW_DEBUG=attn_fold_limit_f16=n overrides the f16 one, A/B),
    // the hot-row threshold of the f16 range-free mode (PWW_DEBUG=attn_hot_sum=n; 0 = never switch: round 5's behaviour, -1 = lazy from the start)
    const DebugKnobs &kn = debug_knobs();
    p.fold_limit = d->dtype == PWW_DTYPE_F16 ? (kn.attn_fold_limit_f16 > 0 ? (float)kn.attn_fold_limit_f16 : FOLD_LIMIT_F16) : FOLD_LIMIT_BF16;
    p.hot_sum = (float)kn.attn_hot_sum;
    p.path_counts = debug_path_counts();
}

int attn_fwd(const void *q, const void *k, const void *v, void *o, const float *bias,
             const float *bias_coeff, const pww_attn_desc_t *d, hipStream_t stream,
             const double *stats, int stat_kind, double stat_count, float coeff_scalar, const float *coeff_scalar_dev) {
    if (int rc = attn_validate(q, k, v, o, bias, d)) return rc;
    AttnParams p;
    attn_fill_params(p, q, k, v, o, bias, bias_coeff, d);
    if (stat_kind < PWW_STAT_NONE || stat_kind > PWW_STAT_ABSMAX || (stat_kind != PWW_STAT_NONE && (!stats || !bias))) {
        set_error("attn_fwd: bad statistic selector %d (or missing stats / bias pointer)", stat_kind);
        return PWW_EINVAL;
    }
    p.stats = bias ? stats : nullptr; p.stat_kind = bias ? stat_kind : PWW_STAT_NONE; p.stat_count = stat_count;
    p.coeff_scalar = coeff_scalar;
    p.coeff_scalar_dev = bias ? coeff_scalar_dev : nullptr;
    return d->dtype == PWW_DTYPE_F16 ? attn_dispatch_f16(p, stream) : attn_dispatch_bf16(p, stream);
}

}  // namespace pww
