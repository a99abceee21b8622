/*
 * pww_hip.h -- C ABI of libpww_hip.so: the MI355X (gfx950 / CDNA4) kernels behind the
 * Paint-with-Words attention hot path.
 *
 * The reference (cloneofsimo/paint-with-words-sd) has no native code and therefore no FFI; the
 * entry points below are what a binding for its hot path would call. Each one names the reference
 * lines it replaces (paths relative to the reference repo root):
 *
 *   pww_self_attn_fwd    paint_with_words/paint_with_words.py:83-118 with context=None
 *                        (head split, QK^T, +0.0, *scale, softmax, PV, head merge)
 *   pww_cross_attn_fwd   paint_with_words/paint_with_words.py:83-118 with a context dict
 *                        (same, plus the `+ cross_attention_weight` term of :112)
 *   pww_qk_reduce        the global reductions weight_function applies to `qk`
 *                        (paint_with_words.py:402-405 qk.max(); README.md:152 qk.std())
 *   pww_cross_attn_fwd_stat  pww_cross_attn_fwd with `c0 * g(sigma) * reduce(qk)` formed in the kernel
 *   pww_mask_build       paint_with_words/paint_with_words.py:207-276
 *                        (_image_context_seperator + _tokens_img_attention_weight +
 *                         _img_importance_flatten for ratios 8/16/32/64)
 *   pww_cfg_combine      paint_with_words/paint_with_words.py:501-503 (classifier-free guidance)
 *
 * Conventions
 *   - All tensor arguments are DEVICE pointers owned by the caller. The library never allocates,
 *     frees or retains device memory. Host-side state: a thread-local error string, per-thread caches of launch
 *     attributes (occupancy answers), and -- only while the measurement aids at the end of this file are in use -- a
 *     mutex-guarded pool of at most PWW_PROFILE_SLOTS timing event pairs and one debug pointer; none of it is touched by
 *     the entry points of the path unless a thread armed a timing slot.
 *   - `stream` is a hipStream_t passed as void* (0 = the null stream). Every function only
 *     enqueues work on that stream and returns; none synchronises, so all of them are legal
 *     inside hipGraph stream capture.
 *   - Return value: 0 on success, a negative errno-style code on failure
 *     (PWW_EINVAL bad shape/stride/alignment, PWW_ENOTSUP unsupported head-dim/dtype/arch,
 *     PWW_EHIP a HIP runtime error). pww_last_error() describes the last failure on the calling
 *     thread. No C++ exception crosses this boundary.
 *   - Strides are in ELEMENTS of the tensor's dtype. The innermost (head-dim / key) stride is 1.
 */
#ifndef PWW_HIP_H
#define PWW_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PWW_VERSION 126 /* 0.1.26: the forms that were measured and not made a default moved to libpww_hip_experiments.so (section "experiments" below: pww_cross_attn_fwd_fused[_ex], pww_cross_fused_*_bytes, pww_cross_attn_fwd_parts_out, pww_cross_attn_out_supported); pww_has_experiments; 0.1.25: pww_cross_attn_fwd_parts_out / pww_cross_attn_out_supported (the C = 320 cross-attention layers with to_out + bias [+ residual] in the attention launch); 0.1.24: pww_qk_parts / pww_qk_parts_count (statistic partials over a finished Q; pww_cross_attn_fwd_parts takes the small one-block-per-workgroup kernel where it fits); 0.1.23: pww_group_norm_fwd / pww_group_norm_workspace_bytes, pww_add_layer_norm, pww_geglu, pww_bias_residual (norms and elementwise glue of the blocks that call the attention path); 0.1.22: pww_qproj_stat / pww_qproj_parts / pww_cross_attn_fwd_parts (score statistic formed in the to_q GEMM's epilogue), pww_mask_build_f32_levels, PWW_STAT_ALL; 0.1.21: pww_cross_opts_t.gated_images (was padding); 0.1.20: + pww_cross_attn_fwd_fused_ex / pww_cross_attn_fwd_stat_ex (pww_cross_opts_t: device-side coefficient word, bias column bound,
                           compact bias), pww_debug_timeline (0.1.11: pww_profile_*; 0.1.10: fused cross-attention, blur, resize, inpaint prep) */

#define PWW_OK 0
#define PWW_EINVAL (-22)
#define PWW_ENOTSUP (-95)
#define PWW_EHIP (-5)

/* 2-byte storage types of Q/K/V/O. Accumulation and softmax are always fp32. */
#define PWW_DTYPE_F16 0
#define PWW_DTYPE_BF16 1

/* Largest supported head dim (SD1.x uses 40/80/160, SD2.x 64). D must be a multiple of 8. */
#define PWW_MAX_HEAD_DIM 160

/*
 * Shape/stride descriptor of one attention call. Tensors are addressed as
 *   Q[b][h][n][d] = q + b*q_stride[0] + h*q_stride[1] + n*q_stride[2] + d      (n < N, d < D)
 *   K[b][h][m][d], V[b][h][m][d]   likewise with m < M
 *   O[b][h][n][d]                  likewise
 * so the diffusers layout [B, tokens, H*D] is read/written in place (stride {tokens*H*D, D, H*D}):
 * reshape_heads_to_batch_dim / reshape_batch_dim_to_heads (paint_with_words.py:83-85,:118) never
 * materialise. Base pointers and all strides must keep every row 16-byte aligned
 * (pointer % 16 == 0, stride % 8 == 0).
 */
typedef struct pww_attn_desc {
    int32_t dtype;        /* PWW_DTYPE_* */
    int32_t B, H, N, M, D;
    int64_t q_stride[3];  /* b, h, n */
    int64_t k_stride[3];  /* b, h, m */
    int64_t v_stride[3];  /* b, h, m */
    int64_t o_stride[3];  /* b, h, n */
    float scale;          /* CrossAttention.scale = D^-0.5, applied AFTER the bias add (:112) */
    /* bias[b][h][n][m] = bias + b*bs[0] + h*bs[1] + n*bs[2] + m*bs[3]  (fp32 elements; a stride
       of 0 broadcasts that axis). Ignored when the bias pointer is NULL. */
    int64_t bias_stride[4];
} pww_attn_desc_t;

/* ABI version (PWW_VERSION of the built library). */
int pww_version(void);
/* 1 in libpww_hip_experiments.so (section "experiments" at the end of this file), 0 in libpww_hip.so (version >= 126). */
int pww_has_experiments(void);

/* Description of the last error on this thread ("" if none). */
const char *pww_last_error(void);

/* Name of the HIP device the library would launch on, e.g. "gfx950". Writes at most n bytes. */
int pww_device_arch(char *buf, size_t n);

/*
 * O = softmax((Q K^T) * scale) V, flash-style (no [N,M] score tensor in HBM).
 * Replaces paint_with_words.py:83-118 for context=None (attn1) and for the unconditional pass
 * (`WEIGHT_FUNCTION = lambda ...: 0.0`, :491-494).
 */
int pww_self_attn_fwd(const void *q, const void *k, const void *v, void *o,
                      const pww_attn_desc_t *desc, void *stream);

/*
 * O = softmax((Q K^T + c[b] * bias) * scale) V.
 *   bias        fp32, addressed through desc->bias_stride; NULL = no bias.
 *   bias_coeff  fp32 [B] device array of per-image coefficients c[b] (NULL = 1.0). Lets the host
 *               pass the constant `w` map (paint_with_words.py:94) plus a device scalar such as
 *               0.4*log(1+sigma)*qk.max() without materialising their product.
 * The bias is added to the RAW scores before the 1/sqrt(D) scaling, exactly as :112 does.
 */
int pww_cross_attn_fwd(const void *q, const void *k, const void *v, void *o,
                       const float *bias, const float *bias_coeff,
                       const pww_attn_desc_t *desc, void *stream);

/* Which statistic of pww_qk_reduce's stats[b][4] scales the bias in pww_cross_attn_fwd_stat. */
#define PWW_STAT_NONE 0   /* 1.0 */
#define PWW_STAT_MAX 1    /* qk.max()                       (paint_with_words.py:402-405, runner.py:104) */
#define PWW_STAT_MIN 2    /* qk.min() */
#define PWW_STAT_MEAN 3   /* qk.mean()  = sum / count */
#define PWW_STAT_STD 4    /* qk.std()   = unbiased, like torch.std (README.md:152) */
#define PWW_STAT_ABSMAX 5 /* qk.abs().max() */
#define PWW_STAT_ALL 6    /* pww_qproj_stat only: form all four fields of the partials */

/*
 * pww_cross_attn_fwd with the per-image bias coefficient formed INSIDE the kernel:
 *   c[b] = coeff_scalar * stat(stats[b]) * (gate ? gate[b] : 1)
 * i.e. the whole of `c0 * g(sigma) * qk.max()` (or .std() ...) times the CFG row gate, without the three or four
 * [B]-sized elementwise launches the host would otherwise spend per cross-attention layer and step.
 *   stats       double [B][4] as written by pww_qk_reduce for the same q/k (device pointer; NULL only with
 *               stat_kind == PWW_STAT_NONE)
 *   stat_count  number of score elements per image (H*N*M), used by MEAN / STD
 *   gate        fp32 [B] device array or NULL
 * The products are taken in fp32 in the order written above (the same values the host path produces).
 */
int pww_cross_attn_fwd_stat(const void *q, const void *k, const void *v, void *o, const float *bias,
                            const double *stats, int32_t stat_kind, double stat_count, float coeff_scalar,
                            const float *gate, const pww_attn_desc_t *desc, void *stream);

/*
 * Optional arguments of the *_ex cross-attention entry points. Zero-initialise, then set `size = sizeof(pww_cross_opts_t)`
 * (fields added by later versions are appended; a library accepts every size it knows).
 *   coeff_scalar_dev  device word that REPLACES the by-value `coeff_scalar` argument; it is read when the kernel RUNS, so one
 *                     captured hipGraph serves every denoise step: the host rewrites the word (c0 * g(sigma_i) of
 *                     paint_with_words.py:402-405 / runner.py:104) before each replay. NULL = use the by-value argument.
 *   bias_cols         the caller's promise that columns >= bias_cols of the bias map are zero (the map of a prompt is zero past
 *                     its last region phrase, paint_with_words.py:257-268): only those columns are moved and added. 0 = unknown.
 *                     A hint (this one, bias_compact, gated_images) selects another form of the same kernel -- the same mathematics in
 *                     another rounding order (first-tile / lazy softmax steps): outputs may differ from the unhinted call's in the last
 *                     bit or two of the storage type; the hinted forms agree with each other bit for bit.
 *   bias_compact      compact bias: fp32 [B?][N][R] with bias[b][h][n][col_idx[b?][r]] = bias_compact[b][n][r] and
 *                     every other column zero; addressed  bias_compact + b*compact_stride[0] + n*compact_stride[1] + r.
 *                     5 - 17 of the 77 columns of a Paint-with-Words map are non-zero. R <= 32. When given it is what the fused
 *                     kernel reads; the dense `bias` argument is then only used by the two-launch fall-back and may be NULL
 *                     (a launch that cannot be made resident then fails with PWW_ENOTSUP).
 *   col_idx           int32 [B?][R] (image stride col_idx_stride, 0 = shared): the column of compact slot r, or -1 = unused.
 *                     bias_cols must cover max(col_idx) + 1 (0 = M).
 *   gated_images      (version >= 121; the field was padding before) the caller's hint that gate[b] != 0 exactly for b < gated_images
 *                     -- a CFG-folded batch [conditional rows; unconditional rows] (paint_with_words.py:479-489 as one call). The
 *                     gated-in images carry the statistic, the hand-off and the bias work; the launch gives them more, shorter
 *                     workgroups so that all workgroups finish together. 0 = unknown. A wrong hint costs time, never correctness:
 *                     the kernel still reads gate[].
 */
typedef struct pww_cross_opts {
    uint32_t size;
    int32_t bias_cols;
    const float *coeff_scalar_dev;
    const float *bias_compact;
    const int32_t *col_idx;
    int32_t R;
    int32_t gated_images;
    int64_t compact_stride[2];   /* image, row (elements) */
    int64_t col_idx_stride;      /* image (elements) */
} pww_cross_opts_t;

/* pww_cross_attn_fwd_stat with the optional arguments above (opts == NULL: identical to the plain form): reads coeff_scalar_dev only; the
   other fields concern the LDS bias tile of pww_cross_attn_fwd_parts (and of pww_cross_attn_fwd_fused_ex, section "experiments"). */
int pww_cross_attn_fwd_stat_ex(const void *q, const void *k, const void *v, void *o, const float *bias,
                               const double *stats, int32_t stat_kind, double stat_count, float coeff_scalar,
                               const float *gate, const pww_attn_desc_t *desc, const pww_cross_opts_t *opts, void *stream);

/*
 * Query projection of a cross-attention layer WITH the score statistic's partials (version >= 122):
 *     Q = X W^T                         paint_with_words.py:76   query = self.to_q(hidden_states)   (bias-free Linear)
 *     partials[b][i] = { max, min, sum, sum of squares } of Q_tile K_h^T over tile i's heads, rows and the M prompt keys
 *                                       paint_with_words.py:87 + the global reduction weight_function applies to it
 *                                       (:402-405 qk.max(); runner.py:104; README.md:152 qk.std())
 * The statistic must exist before any softmax of the image can start (:106 -> :112). Formed here -- in the epilogue of the GEMM
 * that produces Q, on the rounded Q tile it still holds -- the boundary between this launch and the attention launch is the
 * synchronisation: pww_cross_attn_fwd_parts folds the partials at entry and reads Q once (no in-kernel hand-off, no residency
 * requirement, no time-out path).
 *   x         [B][N][Cin]   addressed x + b*x_stride[0] + n*x_stride[1] + c
 *   w         [H*D][Cin]    row-major nn.Linear weight, contiguous
 *   q (out)   [B][N][H*D]   addressed q + b*q_stride[0] + n*q_stride[1] + c
 *   k         [B?][M][H*D]  the layer's projected prompt keys (k_stride[0] = 0: one prompt shared by every image), M <= 128
 *   gate      fp32 [B] or NULL: images with gate[b] == 0 (unconditional rows of a CFG-folded batch) get Q but no partials
 *   stat_kind PWW_STAT_*: only the fields that statistic is made of are formed (PWW_STAT_ALL: all four, PWW_STAT_NONE: none)
 *   partials  double [B][pww_qproj_parts(desc)][4], 16-byte aligned; rows of gated-out images are left untouched
 * Supported: H*D a multiple of 320 (or 160) that holds whole heads, Cin a multiple of 64, D a multiple of 8; anything else
 * returns PWW_ENOTSUP and pww_qproj_parts() returns 0 (the caller keeps its own projection and pww_cross_attn_fwd_fused).
 * Statistics are those of the ROUNDED Q (what the attention kernel reads), accumulated in fp32 per 48-score lane slice and in fp64
 * from there: within 1e-6 (relative to the largest score) of pww_qk_reduce on the same Q.
 */
typedef struct pww_qproj_desc {
    int32_t dtype;        /* PWW_DTYPE_* of x, w, q, k */
    int32_t B, N, Cin, H, D, M;
    int64_t x_stride[2];  /* b, n */
    int64_t q_stride[2];  /* b, n */
    int64_t k_stride[2];  /* b, m */
} pww_qproj_desc_t;

int pww_qproj_stat(const void *x, const void *w, void *q, const void *k, const float *gate, const pww_qproj_desc_t *desc,
                   int32_t stat_kind, double *partials, size_t partials_bytes, void *stream);
int32_t pww_qproj_parts(const pww_qproj_desc_t *desc);

/*
 * pww_cross_attn_fwd_fused_ex with the statistic's partials supplied by pww_qproj_stat instead of formed in the launch:
 *   c[b] = coeff_scalar * stat(fold(partials[b][0 .. nparts-1])) * gate[b]
 * `stats_out` (optional, double [B][4]) receives the folded fields that were formed. No state or workspace buffers: nothing in this
 * launch waits for another workgroup.
 */
int pww_cross_attn_fwd_parts(const void *q, const void *k, const void *v, void *o, const float *bias, int32_t stat_kind,
                             float coeff_scalar, const float *gate, const pww_attn_desc_t *desc, const double *partials,
                             int32_t nparts, double *stats_out, const pww_cross_opts_t *opts, void *stream);

/*
 * The statistic's partials over a FINISHED Q (version >= 124): what pww_qproj_stat's epilogue forms, for the layers whose to_q stays the
 * stock GEMM (SD1.5 / SD2.1 at C = 1280: the projection kernel meets too few workgroups there). paint_with_words.py:87 + the global
 * reduction weight_function applies to `qk` (:402-405 qk.max(); runner.py:104; README.md:152 qk.std()). One wave per (image, head, 32-row
 * block, 32-key block) and a partial per wave -- loads -> MFMAs -> one fp64 partial, no LDS, no barrier -- while that gives at most 256
 * partials per image; beyond, one wave per (head, 32-row block) walks the key blocks and a workgroup's four waves share one partial.
 * No atomics, nothing waits for another workgroup.
 *   q, k      as described by the attention descriptor: only dtype, B / H / N / M / D and the q / k strides are read; M <= 128
 *   gate      fp32 [B] or NULL: images with gate[b] == 0 get no partials (their rows of `partials` are left untouched)
 *   stat_kind PWW_STAT_* (PWW_STAT_ALL: all four fields): only the fields that statistic is made of are formed, the others hold the
 *             neutral element
 *   gated_images  the caller's hint that gate[b] != 0 exactly for b < gated_images (0 = unknown): the grid then covers those images only
 *   partials  double [B][pww_qk_parts_count(desc)][4] = { max, min, sum, sum of squares }, 16-byte aligned: the input of
 *             pww_cross_attn_fwd_parts
 * Sums accumulate in fp32 over a lane's 16 scores and in fp64 from there (within 1e-6, relative, of pww_qk_reduce).
 */
int pww_qk_parts(const void *q, const void *k, const float *gate, const pww_attn_desc_t *desc, int32_t stat_kind, int32_t gated_images,
                 double *partials, size_t partials_bytes, void *stream);
int32_t pww_qk_parts_count(const pww_attn_desc_t *desc);

/*
 * GroupNorm of the UNet blocks that call the attention path, fused with the elementwise neighbours those callers put around it
 * (diffusers==0.10.0 -- requirements.txt:1 of the reference, source not under the reference tree --
 * ResnetBlock2D.forward `conv1(silu(norm1(x)))`, `h + time_emb_proj(silu(temb))[:, :, None, None]` -> `conv2(silu(norm2(h)))`, and
 * Transformer2DModel.forward `norm(hidden_states)` in front of proj_in and the patched CrossAttention modules):
 *   y[b, c, p] = act( (h - mean[b, g]) * rstd[b, g] * gamma[c] + beta[c] ),   h = x[b, c, p] + pre_c[c] + add_bc[b, c],   g = c / (C / G),
 *   mean / rstd over the C / G channels x HW positions of the group (biased variance, rstd = 1 / sqrt(var + eps)), act = identity | SiLU.
 * Rounding points are those of the stock sequence on tensors of the storage type: h is rounded to it before it is normalised, the normalised
 * value before the activation. Statistics accumulate in fp64.
 *   x, y      [B, C, H, W] of `dtype` in memory format `layout`: PWW_LAYOUT_NCHW (contiguous) or PWW_LAYOUT_NHWC (torch.channels_last:
 *             element (b, c, p) at (b * HW + p) * C + c); y has x's layout; x == y is allowed (in place)
 *   pre_c     [C] of `dtype` or NULL: a per-channel addend applied (and rounded) BEFORE add_bc -- the bias of the convolution that produced x,
 *             handed over instead of being added by a launch of its own
 *   add_bc    [B, C] of `dtype` (row b at add_bc + b * add_stride), or NULL
 *   gamma, beta   [C] of `dtype`, or NULL (1 / 0)
 *   workspace caller-owned scratch, at least pww_group_norm_workspace_bytes(desc) bytes, 16-byte aligned, contents need not be
 *             initialised (fp64 partial sums handed from the first launch to the second).
 * One launch when a group fits one workgroup's registers (the small feature maps), else two (moments, apply) on `stream`; no atomics,
 * results bitwise repeatable. Requirements: C % G == 0, C % 8 == 0, HW % 8 == 0, G <= 32, C <= 4096 for NHWC; PWW_ENOTSUP otherwise.
 */
#define PWW_LAYOUT_NCHW 0
#define PWW_LAYOUT_NHWC 1
#define PWW_ACT_NONE 0
#define PWW_ACT_SILU 1
typedef struct pww_gn_desc {
    int32_t dtype;        /* PWW_DTYPE_* */
    int32_t layout;       /* PWW_LAYOUT_* */
    int32_t B, C, HW, G;
    float eps;
    int32_t act;          /* PWW_ACT_* */
    int32_t add_stride;   /* elements between two images' rows of add_bc; 0 = C (contiguous). A multiple of 8. */
    int32_t _pad;
} pww_gn_desc_t;

size_t pww_group_norm_workspace_bytes(const pww_gn_desc_t *desc);
int pww_group_norm_fwd(const void *x, const void *pre_c, const void *add_bc, const void *gamma, const void *beta, void *y,
                       const pww_gn_desc_t *desc, void *workspace, size_t workspace_bytes, void *stream);

/*
 * Elementwise glue of the same blocks (row a17; diffusers 0.10.0 BasicTransformerBlock / FeedForward / ResnetBlock2D), each ONE launch with
 * the stock sequence's rounding points on tensors of the storage type T:
 *   pww_add_layer_norm   s = T(a + x);  y = T(LayerNorm_C(s) * gamma + beta)     `attn(norm(h)) + h` and the block's NEXT norm in one launch.
 *                        a == NULL (then s == NULL): plain LayerNorm of x. Rows of C channels (a multiple of 8, <= 2048), row strides in
 *                        elements (0 = C); statistics in fp32, two passes over the row in registers (mean, centred squares).
 *   pww_geglu            y[m, 0..D) = T(h[m, 0..D) * T(gelu(h[m, D..2D))))        FeedForward's GEGLU after its projection (erf form of gelu)
 *   pww_bias_residual    y = T(r + T(v + bias[c]))  over [B, C, H, W] in layout    ResnetBlock2D's `input + conv2(h)` with conv2's bias handed
 *                        over as an operand instead of a launch of its own. y may alias r or v.
 */
typedef struct pww_ln_desc {
    int32_t dtype;        /* PWW_DTYPE_* */
    int32_t C;
    int64_t rows;
    int64_t a_stride, x_stride, s_stride, y_stride;
    float eps;
    int32_t _pad;
} pww_ln_desc_t;
int pww_add_layer_norm(const void *a, const void *x, const void *gamma, const void *beta, void *s, void *y, const pww_ln_desc_t *desc, void *stream);
/* (version >= 124) pww_add_layer_norm whose sum output also carries a per-channel addend: s = T(T(a + x) + post_bias[c]), y = LayerNorm(T(a + x)) as
   before. The caller adds the NEXT GEMM's output to s with that GEMM's C operand (beta = 1) and hands the GEMM's bias over here -- the `ff(norm3(h)) + h`
   of diffusers' BasicTransformerBlock then costs no add launch (one rounding of the bias moves from the GEMM's epilogue to this sum). a, s required. */
int pww_add_layer_norm_bias(const void *a, const void *x, const void *gamma, const void *beta, const void *post_bias, void *s, void *y,
                            const pww_ln_desc_t *desc, void *stream);
int pww_geglu(const void *h, void *y, int64_t rows, int32_t D, int64_t h_stride, int64_t y_stride, int32_t dtype, void *stream);
int pww_bias_residual(const void *r, const void *v, const void *bias, void *y, int32_t B, int32_t C, int32_t HW, int32_t layout, int32_t dtype, void *stream);

/*
 * Per-image global statistics of the raw score tensor S = Q K^T over all heads, rows and keys
 * (what weight_function reduces: qk.max(), qk.min(), qk.mean(), qk.std()).
 *   stats      double [B][4] = { max, min, sum, sum of squares } per image b (fully overwritten).
 *   workspace  caller-owned device scratch of at least pww_workspace_bytes(desc) bytes, 8-byte
 *              aligned; contents need not be initialised (per-workgroup partials + arrival counters:
 *              the last workgroup of each image folds the partials, no same-address atomics).
 * Only q/k strides, B/H/N/M/D and dtype of the descriptor are read.
 */
int pww_qk_reduce(const void *q, const void *k, const pww_attn_desc_t *desc,
                  double *stats, void *workspace, size_t workspace_bytes, void *stream);

/*
 * Region table entry for pww_mask_build: one (color -> strength) pair of color_context
 * (paint_with_words.py:218-238).
 */
typedef struct pww_region {
    uint8_t r, g, b, _pad;
    float strength;
} pww_region_t;

/*
 * RGB color map -> the four per-resolution token weight maps of paint_with_words.py:346-357.
 *   rgb       uint8 [H][W][3] device image
 *   regions   device array [R]
 *   col_ptr   int32 [T+1] device, CSR over the T prompt positions (T = 77)
 *   col_reg   int32 [col_ptr[T]] device: for column t the region ordinals whose phrase covers
 *             prompt position t, in the reference's accumulation order (:257-268)
 *   out[i]    fp32 [round(H/r_i) * round(W/r_i)][T] device, r = {8,16,32,64}; any may be NULL
 * Each output element is the sum, in list order, of the bilinear(align_corners=True) downsample
 * of `strength * (pixel == color)` (:38-45, :231-236). Every element is written (zeros included).
 */
int pww_mask_build(const uint8_t *rgb, int32_t H, int32_t W,
                   const pww_region_t *regions, int32_t R,
                   const int32_t *col_ptr, const int32_t *col_reg, int32_t T,
                   float *out8, float *out16, float *out32, float *out64, void *stream);

/* One output of pww_mask_build at an arbitrary integer `ratio` (ratio 1 gives the reference's
   CROSS_ATTENTION_WEIGHT_ORIG map of paint_with_words.py:343-345, shape [H*W][T]). */
int pww_mask_build_rgb(const uint8_t *rgb, int32_t H, int32_t W,
                       const pww_region_t *regions, int32_t R,
                       const int32_t *col_ptr, const int32_t *col_reg, int32_t T,
                       int32_t ratio, float *out, void *stream);

/* Same accumulation from R float32 masks [R][H][W] (already strength-scaled and optionally
   blurred, paint_with_words.py:236,:307-312) instead of an RGB image; `ratio` selects one output. */
int pww_mask_build_f32(const float *masks, int32_t H, int32_t W, int32_t R,
                       const int32_t *col_ptr, const int32_t *col_reg, int32_t T,
                       int32_t ratio, float *out, void *stream);

/* The four maps of pww_mask_build (ratios 8 / 16 / 32 / 64) from float masks, ONE launch (version >= 122). Any output may be NULL. */
int pww_mask_build_f32_levels(const float *masks, int32_t H, int32_t W, int32_t R, const int32_t *col_ptr, const int32_t *col_reg,
                              int32_t T, float *out8, float *out16, float *out32, float *out64, void *stream);

/*
 * CROSS_ATTENTION_WEIGHT_ORIG [H][W][T] -> [n_tokens][T]: the reference's fallback when a layer's token count has no
 * pre-computed map (paint_with_words.py:96-101; image sizes that are not multiples of 64): bilinear
 * (align_corners=True) to (oh, ow) = floor(H / r), floor(W / r) with r = sqrt(H*W / n_tokens) -- computed by the
 * caller in double precision, as torch does -- then 1-D nearest over the flattened oh*ow pixels.
 */
int pww_resize_tokens(const float *orig, int32_t H, int32_t W, int32_t T, int32_t oh, int32_t ow,
                      int32_t n_tokens, float *out, void *stream);

/*
 * out = GaussianBlur(in) for one float mask [H][W] (paint_with_words.py:307-312: torchvision GaussianBlur(39x39,
 * sigma), reflect padding). `weights` = the normalised 1-D kernel, `ksize` floats on the device (odd, ksize/2 < H, W);
 * `tmp` = H*W doubles of scratch. Two 1-D passes with fp64 accumulation. in == out is allowed.
 */
int pww_gauss_blur(const float *in, float *out, int32_t H, int32_t W, const float *weights, int32_t ksize,
                   double *tmp, void *stream);

/*
 * Inpainting inputs from the PIL-side arrays (paint_with_words_inpaint.py:92-106, :115):
 *   mask_out      [H][W]     1.0 where mask/255 >= 0.5 else 0.0        (NULL = skip)
 *   masked_image  [3][H][W]  (rgb / 127.5 - 1) * (mask < 0.5)
 *   mask_lat      [h][w]     mask_out at latent resolution, nearest     (NULL = skip)
 */
int pww_inpaint_prep(const uint8_t *rgb, const uint8_t *mask, int32_t H, int32_t W, int32_t h, int32_t w,
                     float *mask_out, float *masked_image, float *mask_lat, void *stream);

/*
 * out = uncond + g * (cond - uncond)  (paint_with_words.py:501-503), fp32 math, n elements of
 * `dtype` in, fp32 out.
 */
int pww_cfg_combine(const void *cond, const void *uncond, float guidance, float *out,
                    int64_t n, int32_t dtype, void *stream);

/*
 * dst[i] = values[i], i < n <= 64: `values` is a HOST array that is copied into the kernel arguments at the call, so the
 * store is an ordinary asynchronous launch on `stream` -- no pinned staging buffer to keep alive, no synchronous copy. This is
 * how the per-step scalars c0 * g(sigma_i) of the weight function (paint_with_words.py:479-482) reach the device words that
 * pww_cross_opts_t.coeff_scalar_dev names, between two replays of the captured UNet graph.
 */
int pww_store_f32(float *dst, const float *values, int32_t n, void *stream);

/* Device workspace pww_qk_reduce needs from the caller for this problem (the attention entry points
   need none). */
size_t pww_workspace_bytes(const pww_attn_desc_t *desc);

/*
 * Kernel-only timing of attention launches (measurement aid; no counterpart in the reference). pww_profile_arm() reserves a
 * timing slot (its index >= 0 is returned; negative = error) and arms the calling thread: the NEXT self-/cross-attention kernel
 * this thread launches through the library stamps the dispatch's own start and end device timestamps into the slot
 * (hipExtLaunchKernelGGL) -- the duration rocprofv3 reports for that kernel, without host launch latency and without the gap
 * between dispatches that an event pair recorded around a launch includes. pww_profile_elapsed_us() waits for that kernel and
 * returns the duration; pww_profile_reset() releases all slots. While the calling thread's stream is capturing a hipGraph an armed
 * slot is ignored (the launch is an ordinary captured node and the slot stays unused).
 */
#define PWW_PROFILE_SLOTS 4096   /* slots are recycled: arming slot 4096 + i re-uses the event pair of slot i */
int pww_profile_arm(void);
int pww_profile_elapsed_us(int32_t slot, float *microseconds);
void pww_profile_reset(void);

/* Measurement aid (version >= 126): FOUR uint32 counters in device memory (zeroed by the caller) that every workgroup of the d = 40
 * folded-reference self-attention kernel bumps at its end -- [0] finished on the range-free fast path, [1] finished on the lazy-reference
 * path (an fp16 workgroup whose first key stage showed a hot row), [2] recomputed its rows on the exact path (overflow or magnitude guard),
 * [3] finished on the lazy-reference path with the EXACT scale (fp16: a row's logits passed the magnitude guard on the way; no second pass).
 * NULL switches it off (the default). Process-wide, like pww_debug_timeline; bench.py reports the counts of its hot-logit rows. */
void pww_debug_path_counts(void *device_buffer);

/*
 * Phase time stamps of the attention kernels (debug / measurement aid). After pww_debug_timeline(buf, bytes) every attention
 * launch of the PROCESS whose grid fits writes 8 uint64 wall-clock stamps (100 MHz) per workgroup to buf[workgroup][8]:
 * [0] kernel entry, [1] K/V staged, then per kernel -- fused cross-attention: [2] partials published, [3] statistic folded,
 * [4] outputs stored, [5] exit; self-attention: [2] key loop done, [3] outputs stored. buf == NULL switches it off (the default).
 */
void pww_debug_timeline(void *device_buffer, size_t bytes);

/* ================================================================ experiments ================================================================
 * Entry points of libpww_hip_experiments.so ONLY (version >= 126; `pww_has_experiments()` returns 1 there, 0 in libpww_hip.so). That library
 * is the product library compiled with -DPWW_EXPERIMENTS=1 (python paint-with-words-sd_amd/build.py --experiments): everything above, plus
 * the forms that were built, measured on MI355X and NOT made a default -- kept for the tests and the A/B tools, never loaded by the product:
 *   - round 3's in-launch score statistic (pass 1 + device-scope hand-off between workgroups, spin limit, residency requirement):
 *     pww_cross_attn_fwd_fused[_ex], pww_cross_fused_state_bytes, pww_cross_fused_workspace_bytes -- superseded by partials formed outside the
 *     launch (pww_qproj_stat / pww_qk_parts + pww_cross_attn_fwd_parts: no launch waits for another workgroup);
 *   - round 5's attention + to_out launch: pww_cross_attn_fwd_parts_out, pww_cross_attn_out_supported -- bit-identical to the two-launch
 *     route, slower at batch 1 and a tie at 16 rows (profiles/r05_to_out_epilogue.md);
 *   - kernels behind PWW_DEBUG knobs that lost their A/B (single-buffered key-split self-attention, running-maximum forms of the range-free
 *     launches, 4 x 2 key-split workgroups).
 * In libpww_hip.so these symbols do not exist. */

/*
 * pww_qk_reduce + pww_cross_attn_fwd_stat as ONE launch, for cross-attention over at most 128 keys (the 77 prompt
 * tokens): the cond branch of inj_forward (paint_with_words.py:87-116) for weight functions of the form
 * c0 * w * g(sigma) * reduce(qk). Every workgroup reduces its score tiles and publishes one partial per query block
 * in `state`; the workgroups of an image re-read those slots until none is empty, fold them (agent-scope atomics on
 * both sides, no fence, no counter on the critical path) and carry on with bias -> softmax -> PV. Output and
 * statistics are bit-identical to the two-launch path.
 *   bias        fp32 map, required (desc->bias_stride)
 *   stat_kind   PWW_STAT_*; PWW_STAT_NONE gives c[b] = coeff_scalar * gate[b] with no hand-off at all
 *   gate        fp32 [B] device array or NULL; images with gate[b] == 0 (the unconditional rows of a CFG-folded batch)
 *               get no bias and take no part in the reduction
 *   stats_out   optional double [B][4] (device): { max, min, sum, sum of squares } of the gated-in images
 *   state       device buffer of pww_cross_fused_state_bytes(desc) bytes, 8-byte aligned, owned by the caller and
 *               ZERO before the first call. The kernel leaves it zero again (the last workgroup of an image to leave
 *               clears the image's words), so one buffer serves any number of calls -- of any shape it is large enough
 *               for -- issued on ONE stream, hipGraph replays included; calls that may overlap on different streams
 *               need a buffer each. Word B*H + B is an error flag: it becomes 1 if a hand-off ever timed out (1 s;
 *               the affected outputs are NaN) -- re-zero the buffer then.
 *   workspace   caller-owned scratch, pww_cross_fused_workspace_bytes(desc) bytes, 8-byte aligned, uninitialised;
 *               only touched by the two-launch path below
 * The launch is sized to be fully resident (<= 2 workgroups per CU, several query blocks per workgroup for large
 * batches); if B*H alone exceeds that, the call issues the two launches instead.
 */
int pww_cross_attn_fwd_fused(const void *q, const void *k, const void *v, void *o, const float *bias,
                             int32_t stat_kind, float coeff_scalar, const float *gate, const pww_attn_desc_t *desc,
                             double *stats_out, void *state, size_t state_bytes, void *workspace, size_t workspace_bytes,
                             void *stream);
size_t pww_cross_fused_state_bytes(const pww_attn_desc_t *desc);
size_t pww_cross_fused_workspace_bytes(const pww_attn_desc_t *desc);

int pww_cross_attn_fwd_fused_ex(const void *q, const void *k, const void *v, void *o, const float *bias,
                                int32_t stat_kind, float coeff_scalar, const float *gate, const pww_attn_desc_t *desc,
                                double *stats_out, void *state, size_t state_bytes, void *workspace, size_t workspace_bytes,
                                const pww_cross_opts_t *opts, void *stream);

/*
 * pww_cross_attn_fwd_parts WITH the layer's output projection in the launch (version >= 125):
 *     out[b][n][:] = to_out[0]( merge_heads(O) )[b][n][:] (+ residual[b][n][:])        paint_with_words.py:118-123
 *       = sum over heads h of  W[:, h*D .. h*D+D) . O_h[b][n][:]  + w_bias
 * One workgroup owns 128 query rows of an image and walks ALL heads (O_h is rounded to the storage type where the two-launch path stores
 * it, then multiplied from registers; the sum over heads and the bias add run in fp32, one rounding; the residual add is a second
 * rounding, like the stock `linear(...) + residual`). O is never written. Differences to pww_cross_attn_fwd_parts + a library GEMM: the
 * fp32 summation order of the 320-term dot products (results agree to the last bit or two of the storage type).
 *   out        [B][N][C]   addressed out + b*desc->o_stride[0] + n*desc->o_stride[2] + c   (o_stride[1] is not read), 16-byte aligned rows
 *   w          [C][H*D]    row-major nn.Linear weight (contiguous), storage type of q;  w_bias [C] same type, or NULL
 *   residual   [B][N][C]   same type or NULL;  residual_stride = { b, n } in elements (read only with a residual)
 *   bias       required, dense rows (bias_stride[3] == 1) shared by the heads (bias_stride[1] == 0), at most 64 non-zero columns
 *              (opts->bias_cols or M <= 64)
 * Supported (pww_cross_attn_out_supported returns 1): C = H*D = 320 with D <= 64 (SD1.5: 8 x 40, SD2.1: 5 x 64), 64 <= M <= 128; anything
 * else returns PWW_ENOTSUP and the caller keeps pww_cross_attn_fwd_parts + its own GEMM. A launch has B * ceil(N / 128) workgroups (heads
 * are sequential inside one): worth taking from ~8 images per launch on, measured slower below (profiles/r05_to_out_epilogue.md).
 */
int pww_cross_attn_fwd_parts_out(const void *q, const void *k, const void *v, void *out, const float *bias, int32_t stat_kind, float coeff_scalar,
                                 const float *gate, const pww_attn_desc_t *desc, const double *partials, int32_t nparts, double *stats_out,
                                 const pww_cross_opts_t *opts, const void *w, const void *w_bias, const void *residual,
                                 const int64_t *residual_stride, void *stream);
int32_t pww_cross_attn_out_supported(const pww_attn_desc_t *desc, int32_t c_out, int32_t bias_cols);

#ifdef __cplusplus
}
#endif
#endif /* PWW_HIP_H */
