// Elementwise glue of the blocks that call the attention path (diffusers==0.10.0 BasicTransformerBlock /
// FeedForward(GEGLU) / ResnetBlock2D / Transformer2DModel, pinned by the reference's requirements.txt:1, not under /root/reference). At 2
// folded rows each of these is a 5 - 13 us launch of the stock stack for 1 - 10 MB of traffic; the kernels here merge neighbours, keep the
// stock sequence's rounding points on tensors of the storage type T, and move 16 bytes per lane and access:
//   add_layer_norm:  s = T(a + x);  y = T(LayerNorm_C(s) * gamma + beta)      BasicTransformerBlock: `attn(norm(h)) + h` followed by the next norm
//   geglu:           y = T(x * T(gelu(gate))),  [x | gate] = the two halves of the projection's last dim                       FeedForward.net[0]
//   bias_residual:   y = T(r + T(v + bias[c]))                                ResnetBlock2D: `input + conv2(h)` with conv2's bias as an operand
// (LayerNorm statistics: two passes over the row held in registers, fp32, like ATen's vectorized_layer_norm_kernel: mean, then the centred
// sum of squares; one wave per row, shuffles only.)
#include "pww_common.h"

namespace pww {

namespace {

template <typename T> __device__ __forceinline__ float round_to(float v) { return (float)(T)v; }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

// ---- add + LayerNorm: one wave per row of C channels (C a multiple of 8, <= 64 * 8 * LN_MAXK) ---------------------------------------
constexpr int LN_MAXK = 4;          // 16-byte chunks per lane: C <= 2048
struct LnParams {
    const void *a, *x, *gamma, *beta;
    const void *post_bias;      // ADD only, optional [C]: s = T(T(a + x) + post_bias) -- the bias of the GEMM whose output will be added to s next
                                // (pww_add_layer_norm_bias); y is the LayerNorm of T(a + x), without it
    void *s, *y;
    long rows;
    int C;
    long a_stride, x_stride, s_stride, y_stride;      // elements between rows
    float eps;
};

// (hot / cold kernel arguments as in pww_norm.hip: the leading 13 dwords are preloaded into SGPRs by the dispatcher)
struct LnCold { const void *beta, *post_bias; void *s, *y; long s_stride, y_stride; float eps; };
#define LN_KARGS const void *a_, const void *x_, const void *gamma_, long rows_, long a_stride_, long x_stride_, int C_, const LnCold cold_
#define LN_UNPACK const LnParams p = {a_, x_, gamma_, cold_.beta, cold_.post_bias, cold_.s, cold_.y, rows_, C_, a_stride_, x_stride_, cold_.s_stride, cold_.y_stride, cold_.eps};
#define LN_LARGS(p) p.a, p.x, p.gamma, p.rows, p.a_stride, p.x_stride, p.C, LnCold{p.beta, p.post_bias, p.s, p.y, p.s_stride, p.y_stride, p.eps}

template <typename T, int K, bool ADD>
__global__ void __launch_bounds__(256) add_layer_norm_kernel(LN_KARGS) {
    LN_UNPACK
    typedef typename Vec<T>::v8 V8;
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= p.rows) return;
    const int nch = p.C >> 3;
    const T *x = reinterpret_cast<const T *>(p.x) + row * p.x_stride;
    const T *a = ADD ? reinterpret_cast<const T *>(p.a) + row * p.a_stride : nullptr;
    V8 xv[K], av[K], gv[K], bv[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int ch = lane + k * 64;
        const int cc = (ch < nch ? ch : 0) * 8;
        xv[k] = *reinterpret_cast<const V8 *>(x + cc);
        if (ADD) av[k] = *reinterpret_cast<const V8 *>(a + cc);
        gv[k] = p.gamma ? *reinterpret_cast<const V8 *>(reinterpret_cast<const T *>(p.gamma) + cc) : zero8<V8>();
        bv[k] = p.beta ? *reinterpret_cast<const V8 *>(reinterpret_cast<const T *>(p.beta) + cc) : zero8<V8>();
    }
    V8 pv[K];
    if (ADD && p.post_bias) {       // (uniform)
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int ch = lane + k * 64;
            pv[k] = *reinterpret_cast<const V8 *>(reinterpret_cast<const T *>(p.post_bias) + (ch < nch ? ch : 0) * 8);
        }
    }
    float v[K][8];
    float sum = 0.f;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const bool ok = lane + k * 64 < nch;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float h = (float)xv[k][j];
            if (ADD) h = round_to<T>((float)av[k][j] + h);
            v[k][j] = ok ? h : 0.f;
            sum += v[k][j];
        }
    }
    const float mean = wave_sum(sum) / (float)p.C;
    float sq = 0.f;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const bool ok = lane + k * 64 < nch;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float dlt = ok ? v[k][j] - mean : 0.f;
            sq = fmaf(dlt, dlt, sq);
        }
    }
    const float rstd = 1.f / sqrtf(wave_sum(sq) / (float)p.C + p.eps);
    T *y = reinterpret_cast<T *>(p.y) + row * p.y_stride;
    T *s = ADD ? reinterpret_cast<T *>(p.s) + row * p.s_stride : nullptr;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int ch = lane + k * 64;
        if (ch < nch) {
            V8 o, so;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float g = p.gamma ? (float)gv[k][j] : 1.f, b = p.beta ? (float)bv[k][j] : 0.f;
                o[j] = (T)fmaf((v[k][j] - mean) * rstd, g, b);
                so[j] = (ADD && p.post_bias) ? (T)(v[k][j] + (float)pv[k][j]) : (T)v[k][j];
            }
            *reinterpret_cast<V8 *>(y + ch * 8) = o;
            if (ADD) *reinterpret_cast<V8 *>(s + ch * 8) = so;
        }
    }
}

// ---- GEGLU ---------------------------------------------------------------------------------------------------------------------------
struct GegluParams { const void *h; void *y; long rows; int D; long h_stride, y_stride; };

template <typename T>
__global__ void __launch_bounds__(256) geglu_kernel(const void *h_, void *y_, long rows_, long h_stride_, long y_stride_, int D_) {
    const GegluParams p = {h_, y_, rows_, D_, h_stride_, y_stride_};
    typedef typename Vec<T>::v8 V8;
    const int nch = p.D >> 3;
    const long total = p.rows * nch;
    constexpr int UN = 4;
    const long stride = (long)gridDim.x * 256;
    for (long i0 = (long)blockIdx.x * 256 + threadIdx.x; i0 < total; i0 += UN * stride) {
        V8 xv[UN], gv[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const long i = i0 + u * stride < total ? i0 + u * stride : i0;
            const long row = i / nch;
            const int ch = (int)(i - row * nch);
            const T *h = reinterpret_cast<const T *>(p.h) + row * p.h_stride + ch * 8;
            xv[u] = *reinterpret_cast<const V8 *>(h);
            gv[u] = *reinterpret_cast<const V8 *>(h + p.D);
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const long i = i0 + u * stride;
            if (i < total) {
                const long row = i / nch;
                const int ch = (int)(i - row * nch);
                V8 o;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float g = (float)gv[u][j];
                    const float ge = round_to<T>(0.5f * g * (1.f + erff(g * 0.70710678118654752440f)));      // F.gelu (erf form) on a T tensor
                    o[j] = (T)((float)xv[u][j] * ge);
                }
                *reinterpret_cast<V8 *>(reinterpret_cast<T *>(p.y) + row * p.y_stride + ch * 8) = o;
            }
        }
    }
}

// ---- y = T(r + T(v + bias[c])) over a [B, C, H, W] tensor in either memory format --------------------------------------------------------
struct BrParams { const void *r, *v, *bias; void *y; long nchunk; int C, HW, nhwc; };

template <typename T>
__global__ void __launch_bounds__(256) bias_residual_kernel(const void *r_, const void *v_, const void *bias_, void *y_, long nchunk_, int C_, int HW_, int nhwc_) {
    const BrParams p = {r_, v_, bias_, y_, nchunk_, C_, HW_, nhwc_};
    typedef typename Vec<T>::v8 V8;
    constexpr int UN = 4;
    const long stride = (long)gridDim.x * 256;
    const int CH = p.C >> 3, cpr = p.HW >> 3;
    for (long i0 = (long)blockIdx.x * 256 + threadIdx.x; i0 < p.nchunk; i0 += UN * stride) {
        V8 rv[UN], vv[UN], bv[UN];
        float bs[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const long i = i0 + u * stride < p.nchunk ? i0 + u * stride : i0;
            rv[u] = *reinterpret_cast<const V8 *>(reinterpret_cast<const T *>(p.r) + i * 8);
            vv[u] = *reinterpret_cast<const V8 *>(reinterpret_cast<const T *>(p.v) + i * 8);
            if (p.nhwc) bv[u] = *reinterpret_cast<const V8 *>(reinterpret_cast<const T *>(p.bias) + (i % CH) * 8);
            else bs[u] = (float)reinterpret_cast<const T *>(p.bias)[(i / cpr) % p.C];
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const long i = i0 + u * stride;
            if (i < p.nchunk) {
                V8 o;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float b = p.nhwc ? (float)bv[u][j] : bs[u];
                    o[j] = (T)((float)rv[u][j] + round_to<T>((float)vv[u][j] + b));
                }
                *reinterpret_cast<V8 *>(reinterpret_cast<T *>(p.y) + i * 8) = o;
            }
        }
    }
}

int grid_for(long items, int per_thread) {
    long g = (items + 256L * per_thread - 1) / (256L * per_thread);
    return (int)(g < 1 ? 1 : (g > 4096 ? 4096 : g));
}

template <typename T, bool ADD>
int ln_launch(const LnParams &p, hipStream_t stream) {
    const int k = (p.C / 8 + 63) / 64;
    const dim3 grid((unsigned)((p.rows + 3) / 4));
    switch (k) {
    case 1: hipLaunchKernelGGL((add_layer_norm_kernel<T, 1, ADD>), grid, dim3(256), 0, stream, LN_LARGS(p)); break;
    case 2: hipLaunchKernelGGL((add_layer_norm_kernel<T, 2, ADD>), grid, dim3(256), 0, stream, LN_LARGS(p)); break;
    case 3: hipLaunchKernelGGL((add_layer_norm_kernel<T, 3, ADD>), grid, dim3(256), 0, stream, LN_LARGS(p)); break;
    default: hipLaunchKernelGGL((add_layer_norm_kernel<T, 4, ADD>), grid, dim3(256), 0, stream, LN_LARGS(p)); break;
    }
    return check_hip(hipGetLastError(), "add_layer_norm launch");
}

bool aligned16(const void *a, const void *b = nullptr, const void *c = nullptr, const void *d = nullptr, const void *e = nullptr, const void *f = nullptr) {
    return (((uintptr_t)a | (uintptr_t)b | (uintptr_t)c | (uintptr_t)d | (uintptr_t)e | (uintptr_t)f) & 15) == 0;
}

}  // namespace

int add_layer_norm(const void *a, const void *x, const void *gamma, const void *beta, void *s, void *y, const pww_ln_desc_t *d, hipStream_t stream, const void *post_bias) {
    if (!x || !y || !d || (a != nullptr) != (s != nullptr)) { set_error("add_layer_norm: x, y and desc are required; a and s come together"); return PWW_EINVAL; }
    if (post_bias && (!a || (reinterpret_cast<uintptr_t>(post_bias) & 15))) { set_error("add_layer_norm: post_bias needs the add form (a, s) and 16-byte alignment"); return PWW_EINVAL; }
    if (d->rows < 1 || d->C < 8 || d->C % 8 != 0 || d->C > 64 * 8 * LN_MAXK || (d->dtype != PWW_DTYPE_F16 && d->dtype != PWW_DTYPE_BF16)) {
        set_error("add_layer_norm: unsupported description (rows %lld C %d dtype %d): C a multiple of 8, <= %d", (long long)d->rows, d->C, d->dtype, 64 * 8 * LN_MAXK);
        return PWW_ENOTSUP;
    }
    const long xs = d->x_stride ? d->x_stride : d->C, as = d->a_stride ? d->a_stride : d->C, ss = d->s_stride ? d->s_stride : d->C, ys = d->y_stride ? d->y_stride : d->C;
    if (!aligned16(a, x, gamma, beta, s, y) || ((xs | as | ss | ys) & 7)) { set_error("add_layer_norm: pointers must be 16-byte aligned, row strides multiples of 8"); return PWW_EINVAL; }
    if (!arch_ok()) return PWW_ENOTSUP;
    LnParams p;
    p.a = a; p.x = x; p.gamma = gamma; p.beta = beta; p.post_bias = post_bias; p.s = s; p.y = y; p.rows = d->rows; p.C = d->C;
    p.a_stride = as; p.x_stride = xs; p.s_stride = ss; p.y_stride = ys; p.eps = d->eps;
    if (d->dtype == PWW_DTYPE_F16) return a ? ln_launch<f16, true>(p, stream) : ln_launch<f16, false>(p, stream);
    return a ? ln_launch<bf16, true>(p, stream) : ln_launch<bf16, false>(p, stream);
}

int geglu(const void *h, void *y, int64_t rows, int32_t D, int64_t h_stride, int64_t y_stride, int32_t dtype, hipStream_t stream) {
    if (!h || !y || rows < 1 || D < 8 || D % 8 != 0) { set_error("geglu: bad argument (rows %lld D %d)", (long long)rows, D); return PWW_EINVAL; }
    if (dtype != PWW_DTYPE_F16 && dtype != PWW_DTYPE_BF16) { set_error("geglu: dtype %d unsupported", dtype); return PWW_ENOTSUP; }
    const long hs = h_stride ? h_stride : 2L * D, ys = y_stride ? y_stride : D;
    if (!aligned16(h, y) || ((hs | ys) & 7)) { set_error("geglu: pointers must be 16-byte aligned, row strides multiples of 8"); return PWW_EINVAL; }
    if (!arch_ok()) return PWW_ENOTSUP;
    GegluParams p; p.h = h; p.y = y; p.rows = rows; p.D = D; p.h_stride = hs; p.y_stride = ys;
    const int grid = grid_for(rows * (D / 8), 4);
    if (dtype == PWW_DTYPE_F16) hipLaunchKernelGGL(geglu_kernel<f16>, dim3(grid), dim3(256), 0, stream, p.h, p.y, p.rows, p.h_stride, p.y_stride, p.D);
    else hipLaunchKernelGGL(geglu_kernel<bf16>, dim3(grid), dim3(256), 0, stream, p.h, p.y, p.rows, p.h_stride, p.y_stride, p.D);
    return check_hip(hipGetLastError(), "geglu launch");
}

int bias_residual(const void *r, const void *v, const void *bias, void *y, int32_t B, int32_t C, int32_t HW, int32_t layout, int32_t dtype, hipStream_t stream) {
    if (!r || !v || !bias || !y || B < 1 || C < 8 || HW < 1 || C % 8 != 0 || (layout == PWW_LAYOUT_NCHW && HW % 8 != 0)) {
        set_error("bias_residual: bad argument (B %d C %d HW %d layout %d): C a multiple of 8 (and HW in NCHW)", B, C, HW, layout);
        return PWW_EINVAL;
    }
    if ((dtype != PWW_DTYPE_F16 && dtype != PWW_DTYPE_BF16) || (layout != PWW_LAYOUT_NCHW && layout != PWW_LAYOUT_NHWC)) { set_error("bias_residual: dtype %d / layout %d unsupported", dtype, layout); return PWW_ENOTSUP; }
    if (!aligned16(r, v, bias, y)) { set_error("bias_residual: pointers must be 16-byte aligned"); return PWW_EINVAL; }
    if (!arch_ok()) return PWW_ENOTSUP;
    BrParams p; p.r = r; p.v = v; p.bias = bias; p.y = y; p.nchunk = (long)B * C * HW / 8; p.C = C; p.HW = HW; p.nhwc = layout == PWW_LAYOUT_NHWC;
    const int grid = grid_for(p.nchunk, 4);
    if (dtype == PWW_DTYPE_F16) hipLaunchKernelGGL(bias_residual_kernel<f16>, dim3(grid), dim3(256), 0, stream, p.r, p.v, p.bias, p.y, p.nchunk, p.C, p.HW, p.nhwc);
    else hipLaunchKernelGGL(bias_residual_kernel<bf16>, dim3(grid), dim3(256), 0, stream, p.r, p.v, p.bias, p.y, p.nchunk, p.C, p.HW, p.nhwc);
    return check_hip(hipGetLastError(), "bias_residual launch");
}

}  // namespace pww
