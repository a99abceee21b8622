// Mask preparation: RGB color map -> per-resolution per-token weight maps
// (paint_with_words/paint_with_words.py:207-276), and the classifier-free-guidance combine
// (:501-503). Both are HBM-bound streaming kernels: coalesced loads, LDS only to share the
// per-pixel region values among the threads that write the 77 token columns.
//
// Semantics restated from the reference (NOT copied; see oracle/pww_oracle.py for the CPU twin):
//   region mask_r[y][x] = strength_r if rgb[y][x] == color_r else 0               (:231-236)
//   down_r = bilinear(mask_r, size=(Hr, Wr), align_corners=True)                  (:38-45)
//        src = dst * (in-1)/(out-1); i0 = floor(src); i1 = i0 + (i0 < in-1); l1 = src - i0
//        v = (1-ly) * ((1-lx) p00 + lx p01) + ly * ((1-lx) p10 + lx p11)          fp32, no FMA
//   out[pix][t] = sum over the regions listed for prompt position t, in list order (:257-268)
#include "pww_common.h"

namespace pww {

constexpr int MASK_PIX = 64;      // output pixels per workgroup
constexpr int MASK_THREADS = 256;
constexpr int MASK_MAX_R = 64;    // regions per call (color_context entries)

struct RgbTap {
    const uint8_t *rgb; int W; const pww_region_t *regions;
    __device__ __forceinline__ float operator()(int r, int y, int x) const {
        const uint8_t *px = rgb + ((long)y * W + x) * 3;
        const pww_region_t reg = regions[r];
        return (px[0] == reg.r && px[1] == reg.g && px[2] == reg.b) ? reg.strength : 0.f;
    }
};
struct F32Tap {
    const float *masks; int H, W;
    __device__ __forceinline__ float operator()(int r, int y, int x) const {
        return masks[((long)r * H + y) * W + x];
    }
};

// One launch builds up to four resolutions (the 8 / 16 / 32 / 64 maps of one request were four launches of 10 - 13 us each --
// launch latency, not bytes): workgroup -> (level, 64-pixel block) through the levels' block offsets. The per-element arithmetic is
// unchanged, so every level is bit-identical to a launch of its own.
struct MaskLevels {
    int n;
    int Hr[4], Wr[4];
    int blk0[5];          // first workgroup of level i (blk0[n] = grid size)
    float *out[4];
};

template <typename Tap>
__global__ void __launch_bounds__(MASK_THREADS)
mask_build_kernel(Tap tap, int H, int W, MaskLevels lv, int R, const int32_t *col_ptr, const int32_t *col_reg, int T) {
    __shared__ float vals[MASK_PIX][MASK_MAX_R + 1];
    const int tid = threadIdx.x;
    int level = 0;
#pragma unroll
    for (int i = 1; i < 4; ++i)
        if (i < lv.n && (int)blockIdx.x >= lv.blk0[i]) level = i;
    const int Hr = lv.Hr[level], Wr = lv.Wr[level];
    float *out = lv.out[level];
    const int pix0 = ((int)blockIdx.x - lv.blk0[level]) * MASK_PIX;
    const int npix = Hr * Wr;
    // ATen: area_pixel_compute_scale(align_corners=True) = (in - 1) / (out - 1) in fp32, 0 if out == 1
    const float sy = Hr > 1 ? __fdiv_rn((float)(H - 1), (float)(Hr - 1)) : 0.f;
    const float sx = Wr > 1 ? __fdiv_rn((float)(W - 1), (float)(Wr - 1)) : 0.f;

    for (int i = tid; i < MASK_PIX * R; i += MASK_THREADS) {
        const int lp = i % MASK_PIX, r = i / MASK_PIX;
        const int pix = pix0 + lp;
        float v = 0.f;
        if (pix < npix) {
            const int oy = pix / Wr, ox = pix - oy * Wr;
            const float fy = __fmul_rn(sy, (float)oy), fx = __fmul_rn(sx, (float)ox);
            const int y0 = (int)fy, x0 = (int)fx;
            const int y1 = y0 + (y0 < H - 1 ? 1 : 0), x1 = x0 + (x0 < W - 1 ? 1 : 0);
            const float ly = fminf(fmaxf(__fsub_rn(fy, (float)y0), 0.f), 1.f);
            const float lx = fminf(fmaxf(__fsub_rn(fx, (float)x0), 0.f), 1.f);
            const float hy = __fsub_rn(1.f, ly), hx = __fsub_rn(1.f, lx);
            const float top = __fadd_rn(__fmul_rn(tap(r, y0, x0), hx), __fmul_rn(tap(r, y0, x1), lx));
            const float bot = __fadd_rn(__fmul_rn(tap(r, y1, x0), hx), __fmul_rn(tap(r, y1, x1), lx));
            v = __fadd_rn(__fmul_rn(top, hy), __fmul_rn(bot, ly));
        }
        vals[lp][r] = v;
    }
    __syncthreads();
    const int nlocal = min(MASK_PIX, npix - pix0);
    for (int i = tid; i < nlocal * T; i += MASK_THREADS) {
        const int lp = i / T, t = i - lp * T;
        float acc = 0.f;
        for (int e = col_ptr[t]; e < col_ptr[t + 1]; ++e) acc = __fadd_rn(acc, vals[lp][col_reg[e]]);
        out[(long)pix0 * T + i] = acc;  // [pix][T] row-major: consecutive threads -> consecutive floats
    }
}

// Round-half-up division result used by the reference for the per-ratio sizes
// (always_round, paint_with_words.py:18-26): for x >= 0 this is floor(x + 0.5).
static int round_div(int a, int ratio) { return (int)((2L * a + ratio) / (2L * ratio)); }

template <typename Tap>
static int launch_mask_levels(Tap tap, int H, int W, int n, const int *ratios, float *const *outs, int R, const int32_t *col_ptr,
                              const int32_t *col_reg, int T, hipStream_t stream) {
    MaskLevels lv;
    lv.n = 0;
    int blocks = 0;
    for (int i = 0; i < n; ++i) {
        if (!outs[i]) continue;
        const int Hr = round_div(H, ratios[i]), Wr = round_div(W, ratios[i]);
        if (Hr <= 0 || Wr <= 0) { set_error("mask_build: image %dx%d too small for ratio %d", H, W, ratios[i]); return PWW_EINVAL; }
        lv.Hr[lv.n] = Hr; lv.Wr[lv.n] = Wr; lv.out[lv.n] = outs[i]; lv.blk0[lv.n] = blocks;
        blocks += (Hr * Wr + MASK_PIX - 1) / MASK_PIX;
        ++lv.n;
    }
    if (!lv.n) return PWW_OK;
    for (int i = lv.n; i < 4; ++i) { lv.Hr[i] = lv.Wr[i] = 1; lv.out[i] = nullptr; }
    for (int i = lv.n; i < 5; ++i) lv.blk0[i] = blocks;
    launch_timed(mask_build_kernel<Tap>, dim3(blocks), dim3(MASK_THREADS), 0, stream, tap, H, W, lv, R, col_ptr, col_reg, T);      // (armable: bench.py times it kernel-only)
    return check_hip(hipGetLastError(), "mask_build_kernel launch");
}

template <typename Tap>
static int launch_mask(Tap tap, int H, int W, int ratio, int R, const int32_t *col_ptr,
                       const int32_t *col_reg, int T, float *out, hipStream_t stream) {
    return launch_mask_levels(tap, H, W, 1, &ratio, &out, R, col_ptr, col_reg, T, stream);
}

static int check_mask_args(const void *src, int H, int W, int R, const int32_t *col_ptr, const int32_t *col_reg, int T) {
    if (!src || !col_ptr || !col_reg) { set_error("mask_build: null argument"); return PWW_EINVAL; }
    if (H <= 0 || W <= 0 || T <= 0) { set_error("mask_build: bad size H=%d W=%d T=%d", H, W, T); return PWW_EINVAL; }
    if (R <= 0 || R > MASK_MAX_R) { set_error("mask_build: region count %d outside 1..%d", R, MASK_MAX_R); return PWW_ENOTSUP; }
    if (!arch_ok()) return PWW_ENOTSUP;
    return PWW_OK;
}

int mask_build(const uint8_t *rgb, int H, int W, const pww_region_t *regions, int R, const int32_t *col_ptr,
               const int32_t *col_reg, int T, float *out8, float *out16, float *out32, float *out64,
               hipStream_t stream) {
    if (int rc = check_mask_args(rgb, H, W, R, col_ptr, col_reg, T)) return rc;
    if (!regions) { set_error("mask_build: null regions"); return PWW_EINVAL; }
    RgbTap tap{rgb, W, regions};
    float *outs[4] = {out8, out16, out32, out64};
    const int ratios[4] = {8, 16, 32, 64};
    return launch_mask_levels(tap, H, W, 4, ratios, outs, R, col_ptr, col_reg, T, stream);      // ONE launch for the four maps
}

int mask_build_f32_levels(const float *masks, int H, int W, int R, const int32_t *col_ptr, const int32_t *col_reg, int T,
                          float *out8, float *out16, float *out32, float *out64, hipStream_t stream) {
    if (int rc = check_mask_args(masks, H, W, R, col_ptr, col_reg, T)) return rc;
    F32Tap tap{masks, H, W};
    float *outs[4] = {out8, out16, out32, out64};
    const int ratios[4] = {8, 16, 32, 64};
    return launch_mask_levels(tap, H, W, 4, ratios, outs, R, col_ptr, col_reg, T, stream);
}

int mask_build_rgb(const uint8_t *rgb, int H, int W, const pww_region_t *regions, int R, const int32_t *col_ptr,
                   const int32_t *col_reg, int T, int ratio, float *out, hipStream_t stream) {
    if (int rc = check_mask_args(rgb, H, W, R, col_ptr, col_reg, T)) return rc;
    if (!regions || !out || ratio <= 0) { set_error("mask_build_rgb: bad regions/out/ratio"); return PWW_EINVAL; }
    RgbTap tap{rgb, W, regions};
    return launch_mask(tap, H, W, ratio, R, col_ptr, col_reg, T, out, stream);
}

int mask_build_f32(const float *masks, int H, int W, int R, const int32_t *col_ptr, const int32_t *col_reg, int T,
                   int ratio, float *out, hipStream_t stream) {
    if (int rc = check_mask_args(masks, H, W, R, col_ptr, col_reg, T)) return rc;
    if (!out || ratio <= 0) { set_error("mask_build_f32: bad out/ratio"); return PWW_EINVAL; }
    F32Tap tap{masks, H, W};
    return launch_mask(tap, H, W, ratio, R, col_ptr, col_reg, T, out, stream);
}

// ---- CROSS_ATTENTION_WEIGHT_ORIG -> [n_tokens, T]: the reference's fallback for token counts without a per-resolution
// key (paint_with_words.py:96-101): bilinear(align_corners=True) to (oh, ow) = floor(size / sqrt(H*W/n)), flattened
// row-major, then 1-D nearest to n_tokens (src = floor(dst * n_in / n_out) in fp32, ATen's nearest). One thread per
// output element; consecutive threads read consecutive prompt positions of the same four taps (coalesced).
__global__ void __launch_bounds__(256) resize_tokens_kernel(const float *orig, int H, int W, int T, int oh, int ow, int n_tokens, float *out) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)n_tokens * T) return;
    const int n = (int)(i / T), t = (int)(i - (long)n * T);
    const int n_in = oh * ow;
    const float nscale = __fdiv_rn((float)n_in, (float)n_tokens);
    int src = (int)floorf(__fmul_rn((float)n, nscale));
    src = src < n_in - 1 ? src : n_in - 1;
    const int oy = src / ow, ox = src - oy * ow;
    const float sy = oh > 1 ? __fdiv_rn((float)(H - 1), (float)(oh - 1)) : 0.f;
    const float sx = ow > 1 ? __fdiv_rn((float)(W - 1), (float)(ow - 1)) : 0.f;
    const float fy = __fmul_rn(sy, (float)oy), fx = __fmul_rn(sx, (float)ox);
    int y0 = (int)fy, x0 = (int)fx;
    y0 = y0 < H - 1 ? y0 : H - 1; x0 = x0 < W - 1 ? x0 : W - 1;
    const int y1 = y0 + (y0 < H - 1 ? 1 : 0), x1 = x0 + (x0 < W - 1 ? 1 : 0);
    const float ly = fminf(fmaxf(__fsub_rn(fy, (float)y0), 0.f), 1.f);
    const float lx = fminf(fmaxf(__fsub_rn(fx, (float)x0), 0.f), 1.f);
    const float hy = __fsub_rn(1.f, ly), hx = __fsub_rn(1.f, lx);
    auto tap = [&](int y, int x) { return orig[((long)y * W + x) * T + t]; };
    const float top = __fadd_rn(__fmul_rn(tap(y0, x0), hx), __fmul_rn(tap(y0, x1), lx));
    const float bot = __fadd_rn(__fmul_rn(tap(y1, x0), hx), __fmul_rn(tap(y1, x1), lx));
    out[i] = __fadd_rn(__fmul_rn(top, hy), __fmul_rn(bot, ly));
}

int resize_tokens(const float *orig, int H, int W, int T, int oh, int ow, int n_tokens, float *out, hipStream_t stream) {
    if (!orig || !out) { set_error("resize_tokens: null argument"); return PWW_EINVAL; }
    if (H <= 0 || W <= 0 || T <= 0 || oh <= 0 || ow <= 0 || n_tokens <= 0 || oh > H || ow > W) {
        set_error("resize_tokens: bad size H=%d W=%d T=%d oh=%d ow=%d n=%d", H, W, T, oh, ow, n_tokens);
        return PWW_EINVAL;
    }
    if (!arch_ok()) return PWW_ENOTSUP;
    const long total = (long)n_tokens * T;
    hipLaunchKernelGGL(resize_tokens_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, orig, H, W, T, oh, ow, n_tokens, out);
    return check_hip(hipGetLastError(), "resize_tokens_kernel launch");
}

// ---- per-region Gaussian blur of a float mask (paint_with_words.py:307-312: torchvision GaussianBlur(39x39, sigma),
// reflect padding). The 2-D kernel is the outer product of one normalised 1-D kernel, so two 1-D passes give the same
// map; accumulation in fp64 (the taps are summed in ascending order, deterministic) keeps the result within fp32
// rounding of the exact convolution whatever order the reference's conv2d sums its 1521 taps in.
__device__ __forceinline__ int reflect_index(int i, int n) { return i < 0 ? -i : (i >= n ? 2 * (n - 1) - i : i); }

__global__ void __launch_bounds__(256) blur_rows_kernel(const float *in, int H, int W, const float *k, int ks, double *tmp) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)H * W) return;
    const int y = (int)(i / W), x = (int)(i - (long)y * W), half = ks / 2;
    const float *row = in + (long)y * W;
    double acc = 0.0;
    for (int j = 0; j < ks; ++j) acc += (double)k[j] * (double)row[reflect_index(x + j - half, W)];
    tmp[i] = acc;
}

__global__ void __launch_bounds__(256) blur_cols_kernel(const double *tmp, int H, int W, const float *k, int ks, float *out) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)H * W) return;
    const int y = (int)(i / W), x = (int)(i - (long)y * W), half = ks / 2;
    double acc = 0.0;
    for (int j = 0; j < ks; ++j) acc += (double)k[j] * tmp[(long)reflect_index(y + j - half, H) * W + x];
    out[i] = (float)acc;
}

int gauss_blur(const float *in, float *out, int H, int W, const float *weights, int ksize, double *tmp, hipStream_t stream) {
    if (!in || !out || !weights || !tmp) { set_error("gauss_blur: null argument"); return PWW_EINVAL; }
    if (H <= 0 || W <= 0 || ksize <= 0 || (ksize & 1) == 0 || ksize / 2 >= H || ksize / 2 >= W) {
        set_error("gauss_blur: bad size H=%d W=%d ksize=%d (odd, reflect padding needs ksize/2 < H, W)", H, W, ksize);
        return PWW_EINVAL;
    }
    if (!arch_ok()) return PWW_ENOTSUP;
    const unsigned blocks = (unsigned)(((long)H * W + 255) / 256);
    hipLaunchKernelGGL(blur_rows_kernel, dim3(blocks), dim3(256), 0, stream, in, H, W, weights, ksize, tmp);
    hipLaunchKernelGGL(blur_cols_kernel, dim3(blocks), dim3(256), 0, stream, tmp, H, W, weights, ksize, out);
    return check_hip(hipGetLastError(), "blur kernels launch");
}

// ---- inpainting inputs (paint_with_words_inpaint.py:92-106 PIL branch, :115): init image uint8 [H,W,3] -> [-1,1]
// planar, mask uint8 [H,W] -> {0,1} at threshold 0.5, masked_image = image * (mask < 0.5), and the mask at latent
// resolution (nearest: src = floor(dst * in/out) in fp32) -- one pass over the pixels instead of a dozen torch ops.
__global__ void __launch_bounds__(256) inpaint_prep_kernel(const uint8_t *rgb, const uint8_t *mask, int H, int W, int h, int w,
                                                            float *mask_out, float *masked, float *mask_lat) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long npx = (long)H * W;
    if (i < npx) {
        const float m = __fdiv_rn((float)mask[i], 255.0f) >= 0.5f ? 1.f : 0.f;
        const float keep = m < 0.5f ? 1.f : 0.f;
        if (mask_out) mask_out[i] = m;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float v = __fsub_rn(__fdiv_rn((float)rgb[i * 3 + c], 127.5f), 1.0f);
            masked[c * npx + i] = __fmul_rn(v, keep);
        }
    }
    if (mask_lat && i < (long)h * w) {
        const int y = (int)(i / w), x = (int)(i - (long)y * w);
        int sy = (int)floorf(__fmul_rn((float)y, __fdiv_rn((float)H, (float)h)));
        int sx = (int)floorf(__fmul_rn((float)x, __fdiv_rn((float)W, (float)w)));
        sy = sy < H - 1 ? sy : H - 1; sx = sx < W - 1 ? sx : W - 1;
        mask_lat[i] = __fdiv_rn((float)mask[(long)sy * W + sx], 255.0f) >= 0.5f ? 1.f : 0.f;
    }
}

int inpaint_prep(const uint8_t *rgb, const uint8_t *mask, int H, int W, int h, int w, float *mask_out, float *masked,
                 float *mask_lat, hipStream_t stream) {
    if (!rgb || !mask || !masked) { set_error("inpaint_prep: null argument"); return PWW_EINVAL; }
    if (H <= 0 || W <= 0 || (mask_lat && (h <= 0 || w <= 0 || h > H || w > W))) { set_error("inpaint_prep: bad size"); return PWW_EINVAL; }
    if (!arch_ok()) return PWW_ENOTSUP;
    hipLaunchKernelGGL(inpaint_prep_kernel, dim3((unsigned)(((long)H * W + 255) / 256)), dim3(256), 0, stream, rgb, mask, H, W, h, w,
                       mask_out, masked, mask_lat);
    return check_hip(hipGetLastError(), "inpaint_prep_kernel launch");
}

// ---- classifier-free guidance combine ---------------------------------------------------------
template <typename T>
__global__ void cfg_combine_kernel(const T *cond, const T *uncond, float g, float *out, long n) {
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const float c = (float)cond[i], u = (float)uncond[i];
        out[i] = __fadd_rn(u, __fmul_rn(g, __fsub_rn(c, u)));
    }
}

int cfg_combine(const void *cond, const void *uncond, float g, float *out, long n, int dtype, hipStream_t stream) {
    if (!cond || !uncond || !out || n <= 0) { set_error("cfg_combine: bad argument"); return PWW_EINVAL; }
    if (!arch_ok()) return PWW_ENOTSUP;
    const int threads = 256;
    const int blocks = (int)((n + threads - 1) / threads < 2048 ? (n + threads - 1) / threads : 2048);
    if (dtype == PWW_DTYPE_F16)
        launch_timed(cfg_combine_kernel<f16>, dim3(blocks), dim3(threads), 0, stream, (const f16 *)cond, (const f16 *)uncond, g, out, n);
    else if (dtype == PWW_DTYPE_BF16)
        launch_timed(cfg_combine_kernel<bf16>, dim3(blocks), dim3(threads), 0, stream, (const bf16 *)cond, (const bf16 *)uncond, g, out, n);
    else { set_error("cfg_combine: dtype %d unsupported", dtype); return PWW_ENOTSUP; }
    return check_hip(hipGetLastError(), "cfg_combine_kernel launch");
}

// ---- pww_store_f32: up to 64 host floats, carried in the kernel arguments, written to device words
struct StoreArgs { float v[64]; float *dst; int n; };
__global__ void store_f32_kernel(const StoreArgs a) {
    if ((int)threadIdx.x < a.n) a.dst[threadIdx.x] = a.v[threadIdx.x];
}
int store_f32(float *dst, const float *values, int n, hipStream_t stream) {
    if (!dst || !values || n < 1 || n > 64) { set_error("pww_store_f32: need 1 <= n <= 64 values and a destination (n = %d)", n); return PWW_EINVAL; }
    if (!arch_ok()) return PWW_ENOTSUP;
    StoreArgs a;
    for (int i = 0; i < 64; ++i) a.v[i] = i < n ? values[i] : 0.f;
    a.dst = dst; a.n = n;
    hipLaunchKernelGGL(store_f32_kernel, dim3(1), dim3(64), 0, stream, a);
    return check_hip(hipGetLastError(), "store_f32_kernel launch");
}

}  // namespace pww
