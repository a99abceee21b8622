// GroupNorm of the UNet blocks that call the attention path, with the two elementwise neighbours those callers put around it
// (diffusers==0.10.0, pinned by the reference's requirements.txt:1, not under /root/reference):
//   ResnetBlock2D.forward:      h = conv1(silu(norm1(x)));  h = h + time_emb_proj(silu(temb))[:, :, None, None];  h = conv2(silu(norm2(h)))
//   Transformer2DModel.forward: x = norm(hidden_states) -> proj_in -> [BasicTransformerBlock: attn1 / attn2 = the patched CrossAttention]
// i.e. per call   y = act( GroupNorm_G( x + p[c] + t[b, c] ) * gamma[c] + beta[c] ),   act = identity or SiLU;  p (a convolution's bias, handed
// over instead of being added by a launch of its own) and t optional.
//
// Why it is here: at 2 folded rows the stock sequence is 4 - 5 launches per norm (add, row moments with ONE workgroup per (image, group) =
// 64 workgroups on a 256-CU part, fused-parameter kernel, apply, SiLU): 40 - 50 us where the bytes are worth 2 - 3.
// These tensors are 80 KB - 10 MB: the launches are bound by their chain of dependent memory round trips, not by bytes, so the design rule
// is ONE round trip per launch phase, every load of a thread in flight at once, nothing serial behind a single thread:
//   * a group that fits one workgroup's registers (the 16 x 16 and 8 x 8 levels, the 640-channel norms at 32 x 32: about half of the 61
//     norms of a forward): ONE launch, workgroup = (group, image): load -> reduce -> normalise -> store (gn_group_*);
//   * otherwise TWO launches with the kernel boundary as the only device-wide synchronisation: gn_moments_* writes one fp64 partial
//     (sum, sum of squares) per (slab, group); gn_apply_* folds the <= 64 partials of its image's groups in its prologue (fixed order;
//     those loads fly together with the first activations) and streams y = act(a * (x + t) + b). No atomics, no arrival counters, no
//     state that must be zero.
// Rounding points are those of the stock sequence on T tensors (so the fused op can replace it under a parity test): x + t is rounded to T
// before it is normalised, the normalised value is rounded to T before the activation, the activation's result is rounded to T. Statistics
// are BIASED variances of the T-rounded inputs: fp32 per thread over <= 96 values, fp64 from there on (ATen: Welford in fp32),
// rstd = 1 / sqrt(var + eps).
// Both memory formats of a [B, C, H, W] tensor: NCHW (a group is one contiguous run of cg * HW elements) and NHWC = torch.channels_last (a
// pixel's C channels are contiguous: what MIOpen's bf16 convolutions want).
#include "pww_common.h"

namespace pww {

namespace {

constexpr int GN_NT = 256;
constexpr int GN_UN = 8;                      // loads a thread keeps in flight in the streaming loops
constexpr int GN_MAX_SLABS = 64;              // partials per (image, group)
constexpr int GN_GROUP_PIECES = 24;           // single-launch form: pieces a thread keeps in registers

struct GnParams {
    const void *x, *pre, *add, *gamma, *beta;      // pre: [C] per-channel addend applied (and rounded) before `add` -- a convolution's bias
    void *y;
    double *partial;       // [B][nslab][G][2] (NHWC) / [B][G][nslab][2] (NCHW)
    int B, C, HW, G, cg;
    int add_stride;        // elements between two images' rows of `add`
    int nslab, slab_px;    // moments: NHWC pixels per slab / NCHW: slabs per (image, group)
    int apply_px;          // NHWC apply: pixels per workgroup
    int rows_per_wg;       // NCHW apply: (image, channel) rows per workgroup
    float eps;
};

// Kernel arguments in two parts (round 6): the 14 dwords a wave needs to issue its first loads travel as leading SCALAR arguments, which the
// dispatcher preloads into SGPRs (kernarg preload: 16 user SGPRs less the kernarg pointer; a by-value struct is never preloaded), the rest
// in a struct behind them, fetched by an s_load that has the loads' latency to land. Every launch of these kernels starts cold (546 other
// kernels run between two launches of the same code): the first dependent round trip -- to the kernarg segment -- is the one a kernel can drop.
struct GnCold {
    const void *gamma, *beta;
    void *y;
    double *partial;
    int apply_px, rows_per_wg;
    float eps;
};
#define GN_KARGS const void *x_, const void *pre_, const void *add_, int B_, int C_, int HW_, int G_, int cg_, int add_stride_, int nslab_, int slab_px_, const GnCold cold_
#define GN_UNPACK const GnParams p = {x_, pre_, add_, cold_.gamma, cold_.beta, cold_.y, cold_.partial, B_, C_, HW_, G_, cg_, add_stride_, nslab_, slab_px_, cold_.apply_px, cold_.rows_per_wg, cold_.eps};
#define GN_LARGS(p) p.x, p.pre, p.add, p.B, p.C, p.HW, p.G, p.cg, p.add_stride, p.nslab, p.slab_px, GnCold{p.gamma, p.beta, p.y, p.partial, p.apply_px, p.rows_per_wg, p.eps}

typedef double f64x2 __attribute__((ext_vector_type(2)));

template <typename T> __device__ __forceinline__ float round_to(float v) { return (float)(T)v; }

template <typename T, int ACT> __device__ __forceinline__ float finish(float h, float a, float b) {
    float y = round_to<T>(fmaf(a, h, b));
    if (ACT == 1) y = y / (1.f + __expf(-y));        // SiLU on the T-rounded normalised value, like F.silu on a T tensor
    return y;
}

__device__ __forceinline__ void mean_rstd(double S, double Q, double n, float eps, float &mean, float &rstd) {
    const double m = S / n;
    double var = Q / n - m * m;
    var = var > 0.0 ? var : 0.0;
    mean = (float)m;
    rstd = (float)(1.0 / sqrt(var + (double)eps));
}

// workgroup sum of (S, Q) in a fixed order: lanes by shuffle, waves through LDS
__device__ __forceinline__ void wg_sum(double &S, double &Q, double (*red)[2], int tid) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { S += __shfl_down(S, off); Q += __shfl_down(Q, off); }
    if ((tid & 63) == 0) { red[tid >> 6][0] = S; red[tid >> 6][1] = Q; }
    __syncthreads();
    S = 0.0; Q = 0.0;
#pragma unroll
    for (int w = 0; w < GN_NT / 64; ++w) { S += red[w][0]; Q += red[w][1]; }
}

// ==== NHWC ==========================================================================================================================
// Thread t owns the 8-channel chunk t % CH of every PL-th pixel (CH = C / 8 chunks per pixel, PL = NT / CH pixels in flight; threads past
// PL * CH idle: C = 320 -> 240 of 256 busy). NT = 256 threads, 512 for C > 2048 (the 2560-channel inputs of the first up-block).
template <typename T, int NT>
__global__ void __launch_bounds__(NT) gn_moments_nhwc(GN_KARGS) {
    GN_UNPACK
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef typename Vec<T>::v8 V8;
    const int tid = threadIdx.x, b = blockIdx.y, slab = blockIdx.x;
    const int CH = p.C >> 3, PL = NT / CH;
    const int pl = tid / CH, ch = tid - pl * CH;
    float *ls = reinterpret_cast<float *>(smem);                  // [PL][C] per-thread channel sums
    float *lq = ls + PL * p.C;                                    // [PL][C] ... of squares
    double *lk = reinterpret_cast<double *>(lq + PL * p.C);       // [K][G][2]

    if (pl < PL) {
        float s[8], q[8], t[8], pb[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) s[j] = q[j] = t[j] = pb[j] = 0.f;
        const T *x = reinterpret_cast<const T *>(p.x) + (long)b * p.HW * p.C + ch * 8;
        if (p.pre) {
            const V8 pv = *reinterpret_cast<const V8 *>(reinterpret_cast<const T *>(p.pre) + ch * 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) pb[j] = (float)pv[j];
        }
        if (p.add) {
            const V8 tv = *reinterpret_cast<const V8 *>(reinterpret_cast<const T *>(p.add) + (long)b * p.add_stride + ch * 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) t[j] = (float)tv[j];
        }
        const int p0 = slab * p.slab_px, p1 = min(p0 + p.slab_px, p.HW);
        // GN_UN loads in flight per thread, UNCONDITIONAL (a clamped address, the value masked afterwards): a load under an `if`, or one
        // per loop trip, is waited for before the next is issued, and a slab is only a few trips
#pragma unroll 1
        for (int px = p0 + pl; px < p1; px += GN_UN * PL) {
            V8 r[GN_UN];
#pragma unroll
            for (int u = 0; u < GN_UN; ++u) {
                const int pu = px + u * PL;
                r[u] = *reinterpret_cast<const V8 *>(x + (long)(pu < p1 ? pu : px) * p.C);
            }
#pragma unroll
            for (int u = 0; u < GN_UN; ++u) {
                const bool ok = px + u * PL < p1;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    float h = (float)r[u][j];
                    if (p.pre) h = round_to<T>(h + pb[j]);
                    if (p.add) h = round_to<T>(h + t[j]);
                    h = ok ? h : 0.f;
                    s[j] += h;
                    q[j] = fmaf(h, h, q[j]);
                }
            }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) { ls[pl * p.C + ch * 8 + j] = s[j]; lq[pl * p.C + ch * 8 + j] = q[j]; }
    }
    __syncthreads();
    // (group, lane k of K) sums every K-th of the group's PL * cg entries; thread g < G then adds the K lanes in order
    const int K = NT / p.G, g = tid % p.G, k = tid / p.G;
    if (k < K) {
        double S = 0.0, Q = 0.0;
        const int n = PL * p.cg;
        for (int e = k; e < n; e += K) {
            const int r = e / p.cg, c = g * p.cg + (e - r * p.cg);
            S += (double)ls[r * p.C + c]; Q += (double)lq[r * p.C + c];
        }
        lk[(k * p.G + g) * 2] = S; lk[(k * p.G + g) * 2 + 1] = Q;
    }
    __syncthreads();
    if (tid < p.G) {
        double S = 0.0, Q = 0.0;
        for (int r = 0; r < K; ++r) { S += lk[(r * p.G + tid) * 2]; Q += lk[(r * p.G + tid) * 2 + 1]; }
        f64x2 v; v[0] = S; v[1] = Q;
        reinterpret_cast<f64x2 *>(p.partial)[((long)b * p.nslab + slab) * p.G + tid] = v;
    }
}

template <typename T, int ACT, int NT>
__global__ void __launch_bounds__(NT) gn_apply_nhwc(GN_KARGS) {
    GN_UNPACK
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef typename Vec<T>::v8 V8;
    constexpr int NPART = 8;                                         // partials a fold lane takes: K = NT / G >= 8 lanes, <= 64 slabs
    const int tid = threadIdx.x, b = blockIdx.y;
    const int CH = p.C >> 3, PL = NT / CH;
    const int pl = tid / CH, ch = tid - pl * CH;
    const bool active = pl < PL;
    double *lk = reinterpret_cast<double *>(smem);                  // [K][G][2]
    const int K = NT / p.G, g = tid % p.G, k = tid / p.G;
    float *stat = reinterpret_cast<float *>(lk + K * p.G * 2);       // [G][2] mean, rstd

    // everything this thread needs from memory is requested before anything is waited for: its partials, its first pixels, its parameters
    f64x2 part[NPART];
#pragma unroll
    for (int i = 0; i < NPART; ++i) {
        const int sl = k + i * K;
        part[i] = reinterpret_cast<const f64x2 *>(p.partial)[((long)b * p.nslab + (k < K && sl < p.nslab ? sl : 0)) * p.G + g];
    }
    const int c0 = ch * 8;
    V8 gv = zero8<V8>(), bv = zero8<V8>(), tv = zero8<V8>(), pv = zero8<V8>();
    const T *x = reinterpret_cast<const T *>(p.x) + (long)b * p.HW * p.C + c0;
    T *y = reinterpret_cast<T *>(p.y) + (long)b * p.HW * p.C + c0;
    const int p0 = blockIdx.x * p.apply_px, p1 = min(p0 + p.apply_px, p.HW);
    V8 r[GN_UN];
    if (active) {
        if (p.gamma) gv = *reinterpret_cast<const V8 *>(reinterpret_cast<const T *>(p.gamma) + c0);
        if (p.beta) bv = *reinterpret_cast<const V8 *>(reinterpret_cast<const T *>(p.beta) + c0);
        if (p.add) tv = *reinterpret_cast<const V8 *>(reinterpret_cast<const T *>(p.add) + (long)b * p.add_stride + c0);
        if (p.pre) pv = *reinterpret_cast<const V8 *>(reinterpret_cast<const T *>(p.pre) + c0);
#pragma unroll
        for (int u = 0; u < GN_UN; ++u) {
            const int pu = p0 + pl + u * PL;
            r[u] = *reinterpret_cast<const V8 *>(x + (long)(pu < p1 ? pu : p0) * p.C);
        }
    }
    if (k < K) {
        double S = 0.0, Q = 0.0;
#pragma unroll
        for (int i = 0; i < NPART; ++i)
            if (k + i * K < p.nslab) { S += part[i][0]; Q += part[i][1]; }
        lk[(k * p.G + g) * 2] = S; lk[(k * p.G + g) * 2 + 1] = Q;
    }
    __syncthreads();
    if (tid < p.G) {
        double S = 0.0, Q = 0.0;
        for (int rr = 0; rr < K; ++rr) { S += lk[(rr * p.G + tid) * 2]; Q += lk[(rr * p.G + tid) * 2 + 1]; }
        float mean, rstd;
        mean_rstd(S, Q, (double)p.cg * (double)p.HW, p.eps, mean, rstd);
        stat[tid * 2] = mean; stat[tid * 2 + 1] = rstd;
    }
    __syncthreads();
    if (!active) return;
    float a[8], bb[8], t[8], pb[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int gg = (c0 + j) / p.cg;
        a[j] = stat[gg * 2 + 1] * (p.gamma ? (float)gv[j] : 1.f);
        bb[j] = (p.beta ? (float)bv[j] : 0.f) - a[j] * stat[gg * 2];
        t[j] = (float)tv[j];
        pb[j] = (float)pv[j];
    }
    for (int px = p0 + pl; px < p1; px += GN_UN * PL) {
        if (px != p0 + pl) {
#pragma unroll
            for (int u = 0; u < GN_UN; ++u) {
                const int pu = px + u * PL;
                r[u] = *reinterpret_cast<const V8 *>(x + (long)(pu < p1 ? pu : px) * p.C);
            }
        }
#pragma unroll
        for (int u = 0; u < GN_UN; ++u) {
            V8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float h = (float)r[u][j];
                if (p.pre) h = round_to<T>(h + pb[j]);
                if (p.add) h = round_to<T>(h + t[j]);
                o[j] = (T)finish<T, ACT>(h, a[j], bb[j]);
            }
            if (px + u * PL < p1) *reinterpret_cast<V8 *>(y + (long)(px + u * PL) * p.C) = o;
        }
    }
}

// One launch, workgroup = (group, image), NHWC: thread t owns the 4-channel piece t % ppp of every PLs-th pixel (ppp = cg / 4 pieces per
// pixel of the group, PLs = 256 / ppp pixels in flight): its parameters are loaded once, its <= 24 pieces stay in registers.
// (Round 6 measured two variations and kept neither, profiles/r06_gn_single_launch_ab.txt: the groups that do not fit -- the 20 two-launch
// norms of a forward -- as ONE launch of 1024 threads streaming the group twice: -3 % on the headline, 64 workgroups move 80 - 250 KB each
// three times through one CU's path to L2 where the two launches spread the bytes over 512; and this kernel rewritten branch-free with
// buffer offsets and the second phase re-converting the pieces, 2700 instructions instead of 3600: -0.6 %.)
template <typename T, int ACT>
__global__ void __launch_bounds__(GN_NT) gn_group_nhwc(GN_KARGS) {
    GN_UNPACK
    typedef T TV __attribute__((ext_vector_type(4)));
    __shared__ double red[GN_NT / 64][2];
    const int tid = threadIdx.x, g = blockIdx.x, b = blockIdx.y;
    const int ppp = p.cg >> 2, PLs = GN_NT / ppp;
    const int pl = tid / ppp, sub = tid - pl * ppp;
    const bool active = pl < PLs;
    const int c0 = g * p.cg + sub * 4;
    const T *x = reinterpret_cast<const T *>(p.x) + (long)b * p.HW * p.C + c0;
    T *y = reinterpret_cast<T *>(p.y) + (long)b * p.HW * p.C + c0;
    TV raw[GN_GROUP_PIECES];
    const TV z = {(T)0.f, (T)0.f, (T)0.f, (T)0.f};
    TV gv = z, bv = z, tv = z, pv = z;
    if (active) {
#pragma unroll
        for (int j = 0; j < GN_GROUP_PIECES; ++j) {
            const int px = pl + j * PLs;
            raw[j] = *reinterpret_cast<const TV *>(x + (long)(px < p.HW ? px : pl) * p.C);
        }
        if (p.gamma) gv = *reinterpret_cast<const TV *>(reinterpret_cast<const T *>(p.gamma) + c0);
        if (p.beta) bv = *reinterpret_cast<const TV *>(reinterpret_cast<const T *>(p.beta) + c0);
        if (p.add) tv = *reinterpret_cast<const TV *>(reinterpret_cast<const T *>(p.add) + (long)b * p.add_stride + c0);
        if (p.pre) pv = *reinterpret_cast<const TV *>(reinterpret_cast<const T *>(p.pre) + c0);
    }
    float v[GN_GROUP_PIECES][4];
    float s = 0.f, q = 0.f;
    if (active) {
#pragma unroll
        for (int j = 0; j < GN_GROUP_PIECES; ++j) {
            const bool ok = pl + j * PLs < p.HW;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float h = (float)raw[j][e];
                if (p.pre) h = round_to<T>(h + (float)pv[e]);
                if (p.add) h = round_to<T>(h + (float)tv[e]);
                h = ok ? h : 0.f;
                v[j][e] = h;
                s += h;
                q = fmaf(h, h, q);
            }
        }
    }
    double S = (double)s, Q = (double)q;
    wg_sum(S, Q, red, tid);
    float mean, rstd;
    mean_rstd(S, Q, (double)p.cg * (double)p.HW, p.eps, mean, rstd);      // (every thread: no second barrier)
    if (!active) return;
    float a[4], bb[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        a[e] = rstd * (p.gamma ? (float)gv[e] : 1.f);
        bb[e] = (p.beta ? (float)bv[e] : 0.f) - a[e] * mean;
    }
#pragma unroll
    for (int j = 0; j < GN_GROUP_PIECES; ++j) {
        const int px = pl + j * PLs;
        if (px < p.HW) {
            TV o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (T)finish<T, ACT>(v[j][e], a[e], bb[e]);
            *reinterpret_cast<TV *>(y + (long)px * p.C) = o;
        }
    }
}

// ==== NCHW ==========================================================================================================================
// A group is one contiguous run of cg * HW elements (HW a multiple of 8: a 16-byte chunk never straddles two channels).
template <typename T>
__global__ void __launch_bounds__(GN_NT) gn_moments_nchw(GN_KARGS) {
    GN_UNPACK
    typedef typename Vec<T>::v8 V8;
    __shared__ double red[GN_NT / 64][2];
    const int tid = threadIdx.x, seg = blockIdx.x, g = blockIdx.y, b = blockIdx.z;
    const long nchunk = (long)p.cg * p.HW >> 3;
    const long lo = nchunk * seg / p.nslab, hi = nchunk * (seg + 1) / p.nslab;
    const T *x = reinterpret_cast<const T *>(p.x) + ((long)b * p.C + (long)g * p.cg) * p.HW;
    const T *add = p.add ? reinterpret_cast<const T *>(p.add) + (long)b * p.add_stride + g * p.cg : nullptr;
    const T *pre = p.pre ? reinterpret_cast<const T *>(p.pre) + g * p.cg : nullptr;
    float s = 0.f, q = 0.f;
    double S = 0.0, Q = 0.0;
#pragma unroll 1
    for (long k = lo + tid; k < hi; k += GN_UN * GN_NT) {
        V8 r[GN_UN];
        float t[GN_UN], pb[GN_UN];
#pragma unroll
        for (int u = 0; u < GN_UN; ++u) {
            const long ku = k + (long)u * GN_NT < hi ? k + (long)u * GN_NT : k;
            r[u] = *reinterpret_cast<const V8 *>(x + ku * 8);
            t[u] = add ? (float)add[(ku * 8) / p.HW] : 0.f;
            pb[u] = pre ? (float)pre[(ku * 8) / p.HW] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < GN_UN; ++u) {
            const bool ok = k + (long)u * GN_NT < hi;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float h = (float)r[u][j];
                if (pre) h = round_to<T>(h + pb[u]);
                if (add) h = round_to<T>(h + t[u]);
                h = ok ? h : 0.f;
                s += h;
                q = fmaf(h, h, q);
            }
        }
        S += (double)s; Q += (double)q; s = q = 0.f;              // fp32 only over 64 elements at a time
    }
    wg_sum(S, Q, red, tid);
    if (tid == 0) {
        f64x2 v; v[0] = S; v[1] = Q;
        reinterpret_cast<f64x2 *>(p.partial)[((long)b * p.G + g) * p.nslab + seg] = v;
    }
}

// One workgroup = rows_per_wg consecutive (image, channel) rows of HW elements; TPR threads per row, GN_NT / TPR rows in flight. A thread
// folds the (<= 16) partials of ITS row's group itself: the loads of all of them fly together with the row's first activations.
constexpr int GN_NCHW_SLABS = 16;
template <typename T, int ACT>
__global__ void __launch_bounds__(GN_NT) gn_apply_nchw(GN_KARGS) {
    GN_UNPACK
    typedef typename Vec<T>::v8 V8;
    const int tid = threadIdx.x;
    const int cpr = p.HW >> 3;                               // 16-byte chunks per row
    const int TPR = cpr < GN_NT ? cpr : GN_NT;
    const int RP = GN_NT / TPR;
    const int rl = tid / TPR, k0 = tid - rl * TPR;
    if (rl >= RP) return;
    const long nrows = (long)p.B * p.C;
    const long r_end = min((long)(blockIdx.x + 1) * p.rows_per_wg, nrows);
    for (long row = (long)blockIdx.x * p.rows_per_wg + rl; row < r_end; row += RP) {
        const int b = (int)(row / p.C), c = (int)(row - (long)b * p.C), g = c / p.cg;
        const f64x2 *part = reinterpret_cast<const f64x2 *>(p.partial) + ((long)b * p.G + g) * p.nslab;
        const T *x = reinterpret_cast<const T *>(p.x) + row * p.HW;
        T *y = reinterpret_cast<T *>(p.y) + row * p.HW;
        f64x2 pv[GN_NCHW_SLABS];
#pragma unroll
        for (int i = 0; i < GN_NCHW_SLABS; ++i) pv[i] = part[i < p.nslab ? i : 0];
        V8 r[GN_UN];
#pragma unroll
        for (int u = 0; u < GN_UN; ++u) r[u] = *reinterpret_cast<const V8 *>(x + (long)(k0 + u * TPR < cpr ? k0 + u * TPR : k0) * 8);
        const float gm = p.gamma ? (float)reinterpret_cast<const T *>(p.gamma)[c] : 1.f;
        const float bt = p.beta ? (float)reinterpret_cast<const T *>(p.beta)[c] : 0.f;
        const float t = p.add ? (float)reinterpret_cast<const T *>(p.add)[(long)b * p.add_stride + c] : 0.f;
        const float pb = p.pre ? (float)reinterpret_cast<const T *>(p.pre)[c] : 0.f;
        double S = 0.0, Q = 0.0;
#pragma unroll
        for (int i = 0; i < GN_NCHW_SLABS; ++i)
            if (i < p.nslab) { S += pv[i][0]; Q += pv[i][1]; }
        float mean, rstd;
        mean_rstd(S, Q, (double)p.cg * (double)p.HW, p.eps, mean, rstd);
        const float a = rstd * gm, bb = bt - a * mean;
        for (int k = k0; k < cpr; k += GN_UN * TPR) {
            if (k != k0) {
#pragma unroll
                for (int u = 0; u < GN_UN; ++u) r[u] = *reinterpret_cast<const V8 *>(x + (long)(k + u * TPR < cpr ? k + u * TPR : k) * 8);
            }
#pragma unroll
            for (int u = 0; u < GN_UN; ++u) {
                V8 o;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    float h = (float)r[u][j];
                    if (p.pre) h = round_to<T>(h + pb);
                    if (p.add) h = round_to<T>(h + t);
                    o[j] = (T)finish<T, ACT>(h, a, bb);
                }
                if (k + u * TPR < cpr) *reinterpret_cast<V8 *>(y + (long)(k + u * TPR) * 8) = o;
            }
        }
    }
}

// One launch, workgroup = (group, image), NCHW: TPR threads per channel row, RP rows in flight; a thread's <= 12 chunks (rows rl, rl + RP,
// ... x chunks k0, k0 + TPR, ...) stay in registers.
template <typename T, int ACT>
__global__ void __launch_bounds__(GN_NT) gn_group_nchw(GN_KARGS) {
    GN_UNPACK
    typedef typename Vec<T>::v8 V8;
    constexpr int MAXP = GN_GROUP_PIECES / 2;             // 8-value pieces: half as many as the NHWC form for the same registers
    __shared__ double red[GN_NT / 64][2];
    const int tid = threadIdx.x, g = blockIdx.x, b = blockIdx.y;
    const int cpr = p.HW >> 3;
    const int TPR = cpr < GN_NT ? cpr : GN_NT, RP = GN_NT / TPR;
    const int rl = tid / TPR, k0 = tid - rl * TPR;
    const bool active = rl < RP;
    const int kper = (cpr + TPR - 1) / TPR;               // chunks of a row per thread
    const T *x = reinterpret_cast<const T *>(p.x) + ((long)b * p.C + (long)g * p.cg) * p.HW;
    T *y = reinterpret_cast<T *>(p.y) + ((long)b * p.C + (long)g * p.cg) * p.HW;
    V8 raw[MAXP];
    float gm[MAXP], bt[MAXP], tt[MAXP], pp[MAXP];
    // piece j of a thread = (row rl + (j / kper) * RP, chunk k0 + (j % kper) * TPR)
#pragma unroll
    for (int j = 0; j < MAXP; ++j) {
        const int row = rl + (j / kper) * RP, k = k0 + (j % kper) * TPR;
        const bool ok = active && row < p.cg && k < cpr;
        const int rr = ok ? row : 0, kk = ok ? k : 0;
        raw[j] = *reinterpret_cast<const V8 *>(x + (long)rr * p.HW + kk * 8);
        const int c = g * p.cg + rr;
        gm[j] = p.gamma ? (float)reinterpret_cast<const T *>(p.gamma)[c] : 1.f;
        bt[j] = p.beta ? (float)reinterpret_cast<const T *>(p.beta)[c] : 0.f;
        tt[j] = p.add ? (float)reinterpret_cast<const T *>(p.add)[(long)b * p.add_stride + c] : 0.f;
        pp[j] = p.pre ? (float)reinterpret_cast<const T *>(p.pre)[c] : 0.f;
    }
    float v[MAXP][8];
    float s = 0.f, q = 0.f;
#pragma unroll
    for (int j = 0; j < MAXP; ++j) {
        const int row = rl + (j / kper) * RP, k = k0 + (j % kper) * TPR;
        const bool ok = active && row < p.cg && k < cpr;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float h = (float)raw[j][e];
            if (p.pre) h = round_to<T>(h + pp[j]);
            if (p.add) h = round_to<T>(h + tt[j]);
            h = ok ? h : 0.f;
            v[j][e] = h;
            s += h;
            q = fmaf(h, h, q);
        }
    }
    double S = (double)s, Q = (double)q;
    wg_sum(S, Q, red, tid);
    float mean, rstd;
    mean_rstd(S, Q, (double)p.cg * (double)p.HW, p.eps, mean, rstd);
#pragma unroll
    for (int j = 0; j < MAXP; ++j) {
        const int row = rl + (j / kper) * RP, k = k0 + (j % kper) * TPR;
        if (active && row < p.cg && k < cpr) {
            const float a = rstd * gm[j], bb = bt[j] - a * mean;
            V8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (T)finish<T, ACT>(v[j][e], a, bb);
            *reinterpret_cast<V8 *>(y + (long)row * p.HW + k * 8) = o;
        }
    }
}

struct GnPlan { int nslab, slab_px, apply_px, rows_per_wg, nt; bool group; size_t partial_bytes, lds_m, lds_a; };

bool gn_plan(const pww_gn_desc_t *d, GnPlan &pl) {
    if (!d || d->B < 1 || d->C < 8 || d->HW < 8 || d->G < 1 || d->C % d->G != 0 || d->C % 8 != 0 || d->HW % 8 != 0) return false;
    if (d->dtype != PWW_DTYPE_F16 && d->dtype != PWW_DTYPE_BF16) return false;
    if (d->layout != PWW_LAYOUT_NCHW && d->layout != PWW_LAYOUT_NHWC) return false;
    if (d->act != PWW_ACT_NONE && d->act != PWW_ACT_SILU) return false;
    if (d->G > 32 || d->B > 65535) return false;             // (K = threads / G >= 8 fold lanes x 8 partials each cover the 64 slabs)
    pl = GnPlan();
    const int cg = d->C / d->G;
    if (d->layout == PWW_LAYOUT_NHWC) {
        const int CH = d->C / 8;
        if (CH > 512) return false;                          // C <= 4096
        pl.nt = CH <= 256 ? 256 : 512;
        const int PL = pl.nt / CH;
        // single launch: the group's pieces fit the registers of 256 threads
        if (cg % 4 == 0 && cg / 4 <= GN_NT) {
            const int PLs = GN_NT / (cg / 4);
            pl.group = (d->HW + PLs - 1) / PLs <= GN_GROUP_PIECES;
        }
        // moments: <= 64 slabs per image, each at least one full trip of GN_UN loads per thread row
        int px = (d->HW + GN_MAX_SLABS - 1) / GN_MAX_SLABS;
        if (px < GN_UN * PL) px = GN_UN * PL;
        if (px > d->HW) px = d->HW;
        pl.slab_px = px;
        pl.nslab = (d->HW + px - 1) / px;
        // apply: ~512 workgroups per launch, at least one full trip each
        int want = (512 + d->B - 1) / d->B;
        int apx = (d->HW + want - 1) / want;
        if (apx < GN_UN * PL) apx = GN_UN * PL;
        if (apx > d->HW) apx = d->HW;
        pl.apply_px = apx;
        pl.partial_bytes = (size_t)d->B * pl.nslab * d->G * 2 * sizeof(double);
        const size_t lk = (size_t)(pl.nt / d->G) * d->G * 2 * sizeof(double);
        pl.lds_m = (size_t)2 * PL * d->C * sizeof(float) + lk;
        pl.lds_a = lk + (size_t)d->G * 2 * sizeof(float);
    } else {
        const long nchunk = (long)cg * d->HW / 8;
        const int cpr = d->HW / 8, TPR = cpr < GN_NT ? cpr : GN_NT, RP = GN_NT / TPR;
        const int kper = (cpr + TPR - 1) / TPR;
        pl.group = (long)((cg + RP - 1) / RP) * kper <= GN_GROUP_PIECES / 2;
        int want = (1024 + d->B * d->G - 1) / (d->B * d->G);
        long cap = nchunk / ((long)GN_NT * GN_UN);          // at least one full trip per thread
        if (cap < 1) cap = 1;
        pl.nslab = (int)(want < cap ? want : cap);
        if (pl.nslab > GN_NCHW_SLABS) pl.nslab = GN_NCHW_SLABS;
        if (pl.nslab < 1) pl.nslab = 1;
        pl.partial_bytes = (size_t)d->B * d->G * pl.nslab * 2 * sizeof(double);
        int sweeps = (GN_UN * TPR + cpr - 1) / cpr;          // short rows: several rows per thread so that a thread moves >= GN_UN chunks
        if (sweeps < 1) sweeps = 1;
        if (sweeps > 4) sweeps = 4;
        pl.rows_per_wg = RP * sweeps;
    }
    return true;
}

template <typename T>
int gn_launch(const GnParams &p, const pww_gn_desc_t *d, const GnPlan &pl, hipStream_t stream) {
    const bool silu = d->act == PWW_ACT_SILU;
    if (pl.group) {
        const dim3 grid(d->G, d->B);
        if (d->layout == PWW_LAYOUT_NHWC) {
            if (silu) hipLaunchKernelGGL((gn_group_nhwc<T, 1>), grid, dim3(GN_NT), 0, stream, GN_LARGS(p));
            else hipLaunchKernelGGL((gn_group_nhwc<T, 0>), grid, dim3(GN_NT), 0, stream, GN_LARGS(p));
        } else {
            if (silu) hipLaunchKernelGGL((gn_group_nchw<T, 1>), grid, dim3(GN_NT), 0, stream, GN_LARGS(p));
            else hipLaunchKernelGGL((gn_group_nchw<T, 0>), grid, dim3(GN_NT), 0, stream, GN_LARGS(p));
        }
        return check_hip(hipGetLastError(), "group_norm launch");
    }
    if (d->layout == PWW_LAYOUT_NHWC) {
        const dim3 gm(pl.nslab, d->B), ga((d->HW + pl.apply_px - 1) / pl.apply_px, d->B);
        if (pl.nt == 256) {
            hipLaunchKernelGGL((gn_moments_nhwc<T, 256>), gm, dim3(256), pl.lds_m, stream, GN_LARGS(p));
            if (silu) hipLaunchKernelGGL((gn_apply_nhwc<T, 1, 256>), ga, dim3(256), pl.lds_a, stream, GN_LARGS(p));
            else hipLaunchKernelGGL((gn_apply_nhwc<T, 0, 256>), ga, dim3(256), pl.lds_a, stream, GN_LARGS(p));
        } else {
            hipLaunchKernelGGL((gn_moments_nhwc<T, 512>), gm, dim3(512), pl.lds_m, stream, GN_LARGS(p));
            if (silu) hipLaunchKernelGGL((gn_apply_nhwc<T, 1, 512>), ga, dim3(512), pl.lds_a, stream, GN_LARGS(p));
            else hipLaunchKernelGGL((gn_apply_nhwc<T, 0, 512>), ga, dim3(512), pl.lds_a, stream, GN_LARGS(p));
        }
    } else {
        hipLaunchKernelGGL(gn_moments_nchw<T>, dim3(pl.nslab, d->G, d->B), dim3(GN_NT), 0, stream, GN_LARGS(p));
        const long nrows = (long)d->B * d->C;
        const dim3 grid((unsigned)((nrows + pl.rows_per_wg - 1) / pl.rows_per_wg));
        if (silu) hipLaunchKernelGGL((gn_apply_nchw<T, 1>), grid, dim3(GN_NT), 0, stream, GN_LARGS(p));
        else hipLaunchKernelGGL((gn_apply_nchw<T, 0>), grid, dim3(GN_NT), 0, stream, GN_LARGS(p));
    }
    return check_hip(hipGetLastError(), "group_norm launch");
}

}  // namespace

size_t group_norm_workspace_bytes(const pww_gn_desc_t *d) {
    GnPlan pl;
    if (!gn_plan(d, pl)) return 0;
    return pl.group ? 16 : pl.partial_bytes;
}

int group_norm_fwd(const void *x, const void *pre_c, const void *add_bc, const void *gamma, const void *beta, void *y, const pww_gn_desc_t *d,
                   void *workspace, size_t workspace_bytes, hipStream_t stream) {
    GnPlan pl;
    if (!x || !y || !d || !workspace) { set_error("group_norm: null argument"); return PWW_EINVAL; }
    if (!gn_plan(d, pl)) {
        set_error("group_norm: unsupported description (B %d C %d HW %d G %d dtype %d layout %d act %d): C and HW multiples of 8, C %% G == 0, "
                  "G <= 32, C <= 4096 for NHWC", d->B, d->C, d->HW, d->G, d->dtype, d->layout, d->act);
        return PWW_ENOTSUP;
    }
    if (!arch_ok()) return PWW_ENOTSUP;
    const size_t need = group_norm_workspace_bytes(d);
    if (workspace_bytes < need) { set_error("group_norm: workspace of %zu bytes, need %zu", workspace_bytes, need); return PWW_EINVAL; }
    const int add_stride = add_bc ? (d->add_stride > 0 ? d->add_stride : d->C) : 0;
    if ((((uintptr_t)x | (uintptr_t)y | (uintptr_t)workspace | (uintptr_t)add_bc | (uintptr_t)gamma | (uintptr_t)beta | (uintptr_t)pre_c) & 15) || (add_stride & 7)) {
        set_error("group_norm: pointers must be 16-byte aligned (and add_stride a multiple of 8)");
        return PWW_EINVAL;
    }
    GnParams p;
    p.x = x; p.pre = pre_c; p.add = add_bc; p.gamma = gamma; p.beta = beta; p.y = y;
    p.partial = static_cast<double *>(workspace);
    p.B = d->B; p.C = d->C; p.HW = d->HW; p.G = d->G; p.cg = d->C / d->G;
    p.add_stride = add_stride;
    p.nslab = pl.nslab; p.slab_px = pl.slab_px; p.apply_px = pl.apply_px; p.rows_per_wg = pl.rows_per_wg;
    p.eps = d->eps;
    return d->dtype == PWW_DTYPE_F16 ? gn_launch<f16>(p, d, pl, stream) : gn_launch<bf16>(p, d, pl, stream);
}

}  // namespace pww
