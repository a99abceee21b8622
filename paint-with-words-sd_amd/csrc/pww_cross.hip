// Cross-attention over the prompt tokens (M <= 128 keys: one K/V stage) with the per-image score statistic formed
// IN THE SAME LAUNCH:
//     O = softmax((Q K^T + c[b] * w) * scale) V,     c[b] = c0 * stat_b(Q K^T) * gate[b]
// This is the whole of the reference's cond branch of inj_forward (paint_with_words/paint_with_words.py:87-116) for the
// shipped weight functions  c0 * w * g(sigma) * qk.max() / qk.std()  (:402-405, runner.py:104, README.md:152): the
// global reduction over heads x rows x keys that weight_function applies to `qk`, the bias add before the 1/sqrt(d)
// scaling (:112), softmax and PV. Before, the statistic needed its own launches (arrival-counter init + pww_qk_reduce)
// and the attention kernel re-read Q and K 12 us later.
//
// Structure (workgroup = NW waves x 32 query rows, one (image, head) and a strided set of its query blocks):
//   1. K and V of the head (M <= 128 rows) are staged into LDS once.
//   2. every query block's score tile is computed on the MFMA and reduced to (max, min, sum, sum of squares); one
//      fp64 partial per query block goes to the workspace -- same granularity, same arithmetic order as
//      pww_qk_reduce, so the statistic is bit-identical to the two-launch path.
//   3. hand-off between the workgroups of an image: the partial IS the flag (cdna_hip_programming.md Guideline 16,
//      form R2 -- 8-byte granules, agent-scope atomics on both sides, no fence, no counter on the critical path).
//      A slot holds the bitwise complement of the fp64 partial, so an all-zero slot means "not written yet"; every
//      workgroup folds the image's partials itself (the fold order of pww_qk_reduce's last arriver) and simply repeats
//      the fold's loads until none of them is empty (bounded by a wall-clock limit). A same-address arrival counter was
//      measured first: 256-512 workgroups bumping and polling one word cost 12 us per launch -- as much as the
//      attention itself. Leaving is counted (one returning atomic per workgroup, issued after the fold and consumed at
//      the very end), and the last workgroup to leave zeroes the image's slots and the counter again: the state
//      words need zeroing once, not per call (no memset node in front of every launch, also under hipGraph replay).
//      Images whose gate is 0 (the unconditional rows of a CFG-folded batch) take no part in any of this.
//   4. bias -> online softmax -> PV -> store, exactly the tile code of the general kernel (pww_attn_core.h).
// The launch is sized so that every workgroup is resident at once (the host clamps the grid to the occupancy
// answer, at most two workgroups per CU, and gives each workgroup several query blocks when the batch is large);
// a launch that could not be made resident takes the two-launch path instead -- same result, bit for bit.
#include "pww_cross_kernel.h"

namespace pww {

// instantiations: pww_cross_inst.hip (one unit per storage type x workgroup width)
int cross_dispatch_f16_nw2(const CrossParams &cp, hipStream_t s, bool *launched);
int cross_dispatch_f16_nw4(const CrossParams &cp, hipStream_t s, bool *launched);
int cross_dispatch_bf16_nw2(const CrossParams &cp, hipStream_t s, bool *launched);
int cross_dispatch_bf16_nw4(const CrossParams &cp, hipStream_t s, bool *launched);

// scratch of the two-launch path (pww_qk_reduce's workspace followed by its [B][4] statistics): only touched when the
// launch cannot be made resident
size_t cross_fused_workspace_bytes(const pww_attn_desc_t *d) {
    if (!d || d->B <= 0 || d->H <= 0 || d->N <= 0) return 0;
    return ((qk_reduce_workspace_bytes(d) + 63) & ~(size_t)63) + (size_t)d->B * 4 * sizeof(double) + 64;
}

// persistent state words: left[B][H], heads_left[B], error word (padded to 8 bytes), then the partial slots
static size_t state_sync_bytes(const pww_attn_desc_t *d) { return (((size_t)d->B * d->H + d->B + 1) * sizeof(unsigned) + 7) & ~(size_t)7; }

size_t cross_fused_state_bytes(const pww_attn_desc_t *d) {
    if (!d || d->B <= 0 || d->H <= 0 || d->N <= 0) return 0;
    return state_sync_bytes(d) + (size_t)d->B * ((d->N + 63) / 64) * d->H * 4 * sizeof(unsigned long long);
}

int cross_attn_fused(const void *q, const void *k, const void *v, void *o, const float *bias, int stat_kind, float coeff_scalar,
                     const float *gate, const pww_attn_desc_t *d, double *stats_out, void *state, size_t state_bytes,
                     void *workspace, size_t workspace_bytes, const pww_cross_opts_t *opts, hipStream_t stream,
                     const double *ext_part, int ext_nparts, bool pass2_only) {
    if (int rc = attn_validate(q, k, v, o, bias, d)) return rc;
    const bool ext = pass2_only;
#if !PWW_EXPERIMENTS
    if (!ext && stat_kind != PWW_STAT_NONE) {
        set_error("cross_attn_fused: the in-launch statistic (round 3's hand-off form) lives in libpww_hip_experiments.so; this library folds partials (pww_cross_attn_fwd_parts)");
        return PWW_ENOTSUP;
    }
#endif
    if (ext && !ext_part && stat_kind != PWW_STAT_NONE) { set_error("cross_attn_parts: a statistic needs its partials"); return PWW_EINVAL; }
    if (ext && ext_part && (ext_nparts < 1 || (reinterpret_cast<uintptr_t>(ext_part) & 15) || (long)ext_nparts * 32 >= (1L << 31))) {
        set_error("cross_attn_parts: partials must be 16-byte aligned, 1 <= nparts (got %d)", ext_nparts);
        return PWW_EINVAL;
    }
    pww_cross_opts_t op;
    memset(&op, 0, sizeof(op));
    if (opts) {
        if (opts->size < 16 || opts->size > sizeof(op)) { set_error("cross_attn_fused: pww_cross_opts_t.size = %u is not a known layout", opts->size); return PWW_EINVAL; }
        memcpy(&op, opts, opts->size);
    }
#if !PWW_EXPERIMENTS
    // the compact form of the map is an experiments-library form: with the dense map at hand (callers pass both) this library uses that
    if (op.bias_compact && bias) { op.bias_compact = nullptr; op.col_idx = nullptr; op.R = 0; }
    else if (op.bias_compact) { set_error("cross_attn_parts: the compact bias form lives in libpww_hip_experiments.so; pass the dense map"); return PWW_ENOTSUP; }
#endif
    const bool compact = op.bias_compact != nullptr;
    if (!bias && !compact) { set_error("cross_attn_fused: bias map required (dense, compact or both)"); return PWW_EINVAL; }
    if (d->M > 2 * KVBLK) { set_error("cross_attn_fused: at most %d keys (got %d)", 2 * KVBLK, d->M); return PWW_ENOTSUP; }
    if (stat_kind < PWW_STAT_NONE || stat_kind > PWW_STAT_ABSMAX) { set_error("cross_attn_fused: bad statistic selector %d", stat_kind); return PWW_EINVAL; }
    if (!ext && (!state || state_bytes < cross_fused_state_bytes(d) || (reinterpret_cast<uintptr_t>(state) & 7))) {
        set_error("cross_attn_fused: state buffer missing, misaligned or too small (need %zu bytes, 8-byte aligned)", cross_fused_state_bytes(d));
        return PWW_EINVAL;
    }
    if (!ext && (!workspace || workspace_bytes < cross_fused_workspace_bytes(d) || (reinterpret_cast<uintptr_t>(workspace) & 7))) {
        set_error("cross_attn_fused: workspace missing, misaligned or too small (need %zu bytes, 8-byte aligned)", cross_fused_workspace_bytes(d));
        return PWW_EINVAL;
    }
    const int m16 = (d->M + 15) & ~15;
    int bias_cols = op.bias_cols > 0 ? ((op.bias_cols + 15) & ~15) : m16;
    if (bias_cols > m16) bias_cols = m16;
    if (compact) {
        if (!op.col_idx || op.R < 1 || op.R > COMPACT_MAX_R) { set_error("cross_attn_fused: compact bias needs col_idx and 1 <= R <= %d (got %d)", COMPACT_MAX_R, op.R); return PWW_EINVAL; }
        if ((reinterpret_cast<uintptr_t>(op.bias_compact) & 3) || (reinterpret_cast<uintptr_t>(op.col_idx) & 3) || op.compact_stride[0] < 0 ||
            op.compact_stride[1] < op.R || op.col_idx_stride < 0) {
            set_error("cross_attn_fused: compact bias must be 4-byte aligned with row stride >= R and non-negative image strides");
            return PWW_EINVAL;
        }
    }
    if (ext || stat_kind == PWW_STAT_NONE) {
        // nothing in this launch waits for another workgroup: the small one-block-per-workgroup kernel (pww_cross_lean.hip) where it fits
        bool lean = false;
        if (int rc = cross_attn_lean(q, k, v, o, bias, stat_kind, coeff_scalar, gate, d, stats_out, op, stream, ext_part, ext_nparts, &lean)) return rc;
        if (lean) return PWW_OK;
    }
    CrossParams cp;
    attn_fill_params(cp.a, q, k, v, o, bias, gate, d);
    cp.a.bias_coeff = gate;     // (attn_fill_params drops the coefficient pointer when the dense map is absent)
    cp.a.stats = nullptr; cp.a.stat_kind = stat_kind; cp.a.stat_count = (double)d->H * d->N * d->M; cp.a.coeff_scalar = coeff_scalar;
    cp.a.coeff_scalar_dev = op.coeff_scalar_dev;
    cp.a.bias_cols = bias_cols;
    cp.sync = ext ? nullptr : reinterpret_cast<unsigned *>(state);
    cp.slots = ext ? nullptr : reinterpret_cast<unsigned long long *>(reinterpret_cast<char *>(state) + state_sync_bytes(d));
    cp.ext_part = ext ? ext_part : nullptr; cp.ext_nparts = (ext && ext_part) ? ext_nparts : 0; cp.pass2_only = ext ? 1 : 0;
    cp.stats_out = stats_out;
    cp.nqb = cp.nchunk = cp.nchunk_u = 0;
    cp.n_gated = op.gated_images > 0 && op.gated_images < d->B ? op.gated_images : 0;
    // The LDS tile needs unit key stride (dense form) or the compact form, and it pays when the map is NARROW: measured (MI355X,
    // N = 4096, d = 40) 16 folded rows 73.6 us with a 32-column tile vs 81.8 us with per-lane loads, but 126 us with all 80 columns
    // staged (82 KB of LDS: one workgroup per CU, two slabs fetched without prefetch) -- so a dense map without a column bound of
    // at most 48 keeps the per-lane loads (wider tiles would cost a resident workgroup per CU). PWW_DEBUG=cross_bias_lds=0: per-lane loads (A/B).
    const bool tile_ok = compact || (d->bias_stride[3] == 1 && bias_tile_mode() != 0 && bias_cols <= 48);
    int tile_stride = 16;
    while (tile_stride < bias_cols) tile_stride *= 2;
    cp.tile_stride = tile_ok ? tile_stride : 0;
    cp.tile_nbuf = 1;
    cp.compact = compact ? op.bias_compact : nullptr;
    cp.col_idx = op.col_idx; cp.R = op.R;
    cp.c_sb = op.compact_stride[0]; cp.c_sn = op.compact_stride[1]; cp.ci_sb = op.col_idx_stride;
    bool launched = false;
    const bool wide = attn_wide_groups(d);
    int rc;
    if (d->dtype == PWW_DTYPE_F16) rc = wide ? cross_dispatch_f16_nw4(cp, stream, &launched) : cross_dispatch_f16_nw2(cp, stream, &launched);
    else rc = wide ? cross_dispatch_bf16_nw4(cp, stream, &launched) : cross_dispatch_bf16_nw2(cp, stream, &launched);
    if (rc || launched) return rc;
    if (ext) { set_error("cross_attn_parts: internal error (launch with external partials not issued)"); return PWW_EINVAL; }
    // more (image, head) pairs than resident workgroups: statistic and attention as two launches, same arithmetic
    if (!bias) { set_error("cross_attn_fused: this launch cannot be made resident (B*H = %d) and the two-launch path needs the dense bias map", d->B * d->H); return PWW_ENOTSUP; }
    char *ws = reinterpret_cast<char *>(workspace);
    const size_t red_bytes = (qk_reduce_workspace_bytes(d) + 63) & ~(size_t)63;
    double *stats = stats_out ? stats_out : reinterpret_cast<double *>(ws + red_bytes);
    if (stat_kind != PWW_STAT_NONE)
        if (int rc2 = qk_reduce(q, k, d, stats, ws, red_bytes, stream)) return rc2;
    return attn_fwd(q, k, v, o, bias, gate, d, stream, stat_kind != PWW_STAT_NONE ? stats : nullptr, stat_kind,
                    (double)d->H * d->N * d->M, coeff_scalar, op.coeff_scalar_dev);
}

}  // namespace pww
