// GroupNorm of the UNet blocks that call the attention path, with the two elementwise neighbours those callers put around it
// (diffusers==0.10.0, pinned by the reference's requirements.txt:1, not under /root/reference):
//   ResnetBlock2D.forward:      h = conv1(silu(norm1(x)));  h = h + time_emb_proj(silu(temb))[:, :, None, None];  h = conv2(silu(norm2(h)))
//   Transformer2DModel.forward: x = norm(hidden_states) -> proj_in -> [BasicTransformerBlock: attn1 / attn2 = the patched CrossAttention]
// i.e. per call   y = act( GroupNorm_G( x + t[b, c] ) * gamma[c] + beta[c] ),   act = identity or SiLU,  t optional.
//
// Why it is here: at 2 folded rows the stock sequence is 4 - 5 launches per norm (add, row moments with ONE workgroup per (image, group) =
// 64 workgroups on a 256-CU part, fused-parameter kernel, apply, SiLU): 40 - 50 us where the bytes are worth 2 - 3.
// HBM-bound streaming work: two launches -- moments, apply -- with the kernel boundary as the only device-wide synchronisation:
//   gn_moments_*:  coalesced 16-byte loads, fp32 per-thread sums of a few elements, fp64 from there on; one fp64 partial (sum, sum of squares)
//                  per (workgroup, group); the LAST workgroup of an image (group) to arrive folds the partials in a fixed order and writes the
//                  per-channel affine pair  a = rstd * gamma,  b = beta - a * mean  (what ATen's ComputeFusedParams kernel writes). Arrival
//                  counters are left at zero: no memset node between launches (the same convention as pww_cross.hip's state words).
//   gn_apply_*:    y = act(a * (x + t) + b), 16-byte loads and stores.
// Rounding points are those of the stock sequence on T tensors (so that the fused op can replace it under a parity test): x + t is rounded to
// T before it is normalised, the normalised value is rounded to T before the activation, the activation's result is rounded to T. Statistics
// are BIASED variances in fp64 of the T-rounded inputs (ATen: Welford in fp32), rstd = 1 / sqrt(var + eps).
// Both memory formats of a [B, C, H, W] tensor: NCHW (a group is one contiguous run of cg * HW elements) and NHWC = torch.channels_last (a
// pixel's C channels are contiguous: what MIOpen's bf16 convolutions want).
#include "pww_common.h"

namespace pww {

namespace {

constexpr int GN_NT = 256;
constexpr int GN_COUNTER_BYTES = 4096;        // up to 1024 arrival counters at the front of the workspace

struct GnParams {
    const void *x, *add, *gamma, *beta;
    void *y;
    double *partial;       // NHWC: [B][nslab][G][2]; NCHW: [B][G][nseg][2]
    float *coef;           // [B][C][2]: a, b
    unsigned *count;       // NHWC: [B]; NCHW: [B * G]
    int B, C, HW, G, cg;
    int nslab, slab_px;    // NHWC: pixels per workgroup (moments / apply use the same split)
    int nseg;              // NCHW moments: workgroups per (image, group)
    int rows_per_wg;       // NCHW apply: (image, channel) rows per workgroup
    float eps;
    int act;
};

template <typename T> __device__ __forceinline__ void load8(const T *p, float (&v)[8]) {
    typedef typename Vec<T>::v8 V8;
    const V8 r = *reinterpret_cast<const V8 *>(p);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (float)r[j];
}
template <typename T> __device__ __forceinline__ void store8(T *p, const float (&v)[8]) {
    typedef typename Vec<T>::v8 V8;
    V8 r;
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = (T)v[j];
    *reinterpret_cast<V8 *>(p) = r;
}
template <typename T> __device__ __forceinline__ float round_to(float v) { return (float)(T)v; }

__device__ __forceinline__ double ld_f64_agent(const double *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// mean / rstd of one group from its folded sums, then the affine pairs of the group's channels
template <typename T>
__device__ __forceinline__ void write_coef(const GnParams &p, int b, int g, double S, double Q, int c_lo, int c_hi, int c_step) {
    const double n = (double)p.cg * (double)p.HW;
    const double mean = S / n;
    double var = Q / n - mean * mean;
    var = var > 0.0 ? var : 0.0;
    const float rstd = (float)(1.0 / sqrt(var + (double)p.eps));
    const float meanf = (float)mean;
    const T *gamma = reinterpret_cast<const T *>(p.gamma), *beta = reinterpret_cast<const T *>(p.beta);
    for (int c = g * p.cg + c_lo; c < g * p.cg + c_hi; c += c_step) {
        const float a = rstd * (gamma ? (float)gamma[c] : 1.f);
        const float bb = (beta ? (float)beta[c] : 0.f) - a * meanf;
        p.coef[((long)b * p.C + c) * 2 + 0] = a;
        p.coef[((long)b * p.C + c) * 2 + 1] = bb;
    }
}

template <typename T, int ACT> __device__ __forceinline__ float finish(float h, float a, float b) {
    float y = round_to<T>(fmaf(a, h, b));
    if (ACT == 1) y = y / (1.f + __expf(-y));        // SiLU on the T-rounded normalised value, like F.silu on a T tensor
    return y;
}

// ---- NHWC -------------------------------------------------------------------------------------------------------------------------
// Thread t of a workgroup owns the 8-channel chunk t % CH of every PL-th pixel of the slab (CH = C / 8 chunks per pixel, PL = NT / CH
// pixels in flight; threads past PL * CH idle: C = 320 -> 240 of 256 busy). NT = 256 threads, 512 for C > 2048 (the 2560-channel
// inputs of the first up-block).
template <typename T, int NT>
__global__ void __launch_bounds__(NT) gn_moments_nhwc(const GnParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, b = blockIdx.y, slab = blockIdx.x;
    const int CH = p.C >> 3, PL = NT / CH;
    const int pl = tid / CH, ch = tid - pl * CH;
    const bool active = pl < PL;
    float *ls = reinterpret_cast<float *>(smem);                  // [PL][C] per-thread channel sums
    float *lq = ls + PL * p.C;                                    // [PL][C] ... of squares
    double *lfold = reinterpret_cast<double *>(lq + PL * p.C);    // [NT / G][G][2] (last workgroup only)
    __shared__ int is_last;

    float s[8], q[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) s[j] = q[j] = 0.f;
    if (active) {
        const T *x = reinterpret_cast<const T *>(p.x) + (long)b * p.HW * p.C + ch * 8;
        float t[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) t[j] = 0.f;
        if (p.add) load8(reinterpret_cast<const T *>(p.add) + (long)b * p.C + ch * 8, t);
        const int p0 = slab * p.slab_px, p1 = min(p0 + p.slab_px, p.HW);
        for (int px = p0 + pl; px < p1; px += PL) {
            float v[8];
            load8(x + (long)px * p.C, v);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float h = p.add ? round_to<T>(v[j] + t[j]) : v[j];
                s[j] += h;
                q[j] = fmaf(h, h, q[j]);
            }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) { ls[pl * p.C + ch * 8 + j] = s[j]; lq[pl * p.C + ch * 8 + j] = q[j]; }
    }
    __syncthreads();
    if (tid < p.G) {
        double S = 0.0, Q = 0.0;
        for (int r = 0; r < PL; ++r)
            for (int c = tid * p.cg; c < (tid + 1) * p.cg; ++c) { S += (double)ls[r * p.C + c]; Q += (double)lq[r * p.C + c]; }
        double *dst = p.partial + (((long)b * p.nslab + slab) * p.G + tid) * 2;
        dst[0] = S; dst[1] = Q;
    }
    // last workgroup of the image: plain stores -> agent-scope release -> relaxed ticket (the pattern of pww_reduce.hip)
    __threadfence();
    __syncthreads();
    if (tid == 0) {
        const unsigned ticket = __hip_atomic_fetch_add(p.count + b, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        is_last = ticket == (unsigned)p.nslab - 1u;
    }
    __syncthreads();
    if (!is_last) return;
    __threadfence();
    const int K = NT / p.G;                       // fold lanes per group
    {
        const int g = tid % p.G, k = tid / p.G;
        double S = 0.0, Q = 0.0;
        if (k < K)
            for (int sl = k; sl < p.nslab; sl += K) {
                const double *src = p.partial + (((long)b * p.nslab + sl) * p.G + g) * 2;
                S += ld_f64_agent(src); Q += ld_f64_agent(src + 1);
            }
        if (k < K) { lfold[(k * p.G + g) * 2] = S; lfold[(k * p.G + g) * 2 + 1] = Q; }
    }
    __syncthreads();
    {
        // every thread re-folds its group's K lanes in the same order (cheap) and writes a share of the group's channels
        const int g = tid % p.G, k = tid / p.G;
        if (k < K) {
            double S = 0.0, Q = 0.0;
            for (int r = 0; r < K; ++r) { S += lfold[(r * p.G + g) * 2]; Q += lfold[(r * p.G + g) * 2 + 1]; }
            write_coef<T>(p, b, g, S, Q, k, p.cg, K);
        }
    }
    if (tid == 0) __hip_atomic_store(p.count + b, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <typename T, int ACT, int NT>
__global__ void __launch_bounds__(NT) gn_apply_nhwc(const GnParams p) {
    const int tid = threadIdx.x, b = blockIdx.y, slab = blockIdx.x;
    const int CH = p.C >> 3, PL = NT / CH;
    const int pl = tid / CH, ch = tid - pl * CH;
    if (pl >= PL) return;
    float a[8], bb[8], t[8];
    {
        const float *cf = p.coef + ((long)b * p.C + ch * 8) * 2;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const f32x4 v = *reinterpret_cast<const f32x4 *>(cf + j * 4);
            a[2 * j] = v[0]; bb[2 * j] = v[1]; a[2 * j + 1] = v[2]; bb[2 * j + 1] = v[3];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) t[j] = 0.f;
        if (p.add) load8(reinterpret_cast<const T *>(p.add) + (long)b * p.C + ch * 8, t);
    }
    const T *x = reinterpret_cast<const T *>(p.x) + (long)b * p.HW * p.C + ch * 8;
    T *y = reinterpret_cast<T *>(p.y) + (long)b * p.HW * p.C + ch * 8;
    const int p0 = slab * p.slab_px, p1 = min(p0 + p.slab_px, p.HW);
    for (int px = p0 + pl; px < p1; px += PL) {
        float v[8];
        load8(x + (long)px * p.C, v);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = finish<T, ACT>(p.add ? round_to<T>(v[j] + t[j]) : v[j], a[j], bb[j]);
        store8(y + (long)px * p.C, v);
    }
}

// ---- NCHW -------------------------------------------------------------------------------------------------------------------------
// A group is one contiguous run of cg * HW elements (HW a multiple of 8: a 16-byte chunk never straddles two channels).
template <typename T>
__global__ void __launch_bounds__(GN_NT) gn_moments_nchw(const GnParams p) {
    const int tid = threadIdx.x, seg = blockIdx.x, g = blockIdx.y, b = blockIdx.z;
    __shared__ double red[GN_NT / 64][2];
    __shared__ int is_last;
    const long nchunk = (long)p.cg * p.HW >> 3;
    const long lo = nchunk * seg / p.nseg, hi = nchunk * (seg + 1) / p.nseg;
    const T *x = reinterpret_cast<const T *>(p.x) + ((long)b * p.C + (long)g * p.cg) * p.HW;
    const T *add = p.add ? reinterpret_cast<const T *>(p.add) + (long)b * p.C + g * p.cg : nullptr;
    float s = 0.f, q = 0.f;
    double S = 0.0, Q = 0.0;
    int it = 0;
    for (long k = lo + tid; k < hi; k += GN_NT) {
        float v[8];
        load8(x + k * 8, v);
        const float t = add ? (float)add[(k * 8) / p.HW] : 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float h = add ? round_to<T>(v[j] + t) : v[j];
            s += h;
            q = fmaf(h, h, q);
        }
        if ((++it & 7) == 0) { S += (double)s; Q += (double)q; s = q = 0.f; }     // fp32 only over 64 elements at a time
    }
    S += (double)s; Q += (double)q;
    // wave reduction in a fixed order, then the waves in order
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { S += __shfl_down(S, off); Q += __shfl_down(Q, off); }
    if ((tid & 63) == 0) { red[tid >> 6][0] = S; red[tid >> 6][1] = Q; }
    __syncthreads();
    const long slot = ((long)b * p.G + g) * p.nseg;
    if (tid == 0) {
        double St = 0.0, Qt = 0.0;
        for (int w = 0; w < GN_NT / 64; ++w) { St += red[w][0]; Qt += red[w][1]; }
        p.partial[(slot + seg) * 2] = St; p.partial[(slot + seg) * 2 + 1] = Qt;
        __threadfence();
        const unsigned ticket = __hip_atomic_fetch_add(p.count + b * p.G + g, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        is_last = ticket == (unsigned)p.nseg - 1u;
    }
    __syncthreads();
    if (!is_last) return;
    __threadfence();
    __shared__ double tot[2];
    if (tid == 0) {
        double St = 0.0, Qt = 0.0;
        for (int sg = 0; sg < p.nseg; ++sg) { St += ld_f64_agent(p.partial + (slot + sg) * 2); Qt += ld_f64_agent(p.partial + (slot + sg) * 2 + 1); }
        tot[0] = St; tot[1] = Qt;
        __hip_atomic_store(p.count + b * p.G + g, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    write_coef<T>(p, b, g, tot[0], tot[1], tid, p.cg, GN_NT);
}

// One workgroup = rows_per_wg consecutive (image, channel) rows of HW elements; TPR threads per row, GN_NT / TPR rows in flight.
template <typename T, int ACT>
__global__ void __launch_bounds__(GN_NT) gn_apply_nchw(const GnParams p) {
    const int tid = threadIdx.x;
    const int cpr = p.HW >> 3;                               // 16-byte chunks per row
    const int TPR = cpr < GN_NT ? cpr : GN_NT;
    const int RP = GN_NT / TPR;
    const int rl = tid / TPR, k0 = tid - rl * TPR;
    if (rl >= RP) return;
    const long nrows = (long)p.B * p.C;
    const long r_end = min((long)(blockIdx.x + 1) * p.rows_per_wg, nrows);
    for (long row = (long)blockIdx.x * p.rows_per_wg + rl; row < r_end; row += RP) {
        const float a = p.coef[row * 2], bb = p.coef[row * 2 + 1];
        const float t = p.add ? (float)reinterpret_cast<const T *>(p.add)[row] : 0.f;
        const T *x = reinterpret_cast<const T *>(p.x) + row * p.HW;
        T *y = reinterpret_cast<T *>(p.y) + row * p.HW;
        for (int k = k0; k < cpr; k += TPR) {
            float v[8];
            load8(x + k * 8, v);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = finish<T, ACT>(p.add ? round_to<T>(v[j] + t) : v[j], a, bb);
            store8(y + k * 8, v);
        }
    }
}

struct GnPlan { int nslab, slab_px, nseg, rows_per_wg, nt; size_t partial_bytes, coef_bytes, lds; };

bool gn_plan(const pww_gn_desc_t *d, GnPlan &pl) {
    if (!d || d->B < 1 || d->C < 8 || d->HW < 8 || d->G < 1 || d->C % d->G != 0 || d->C % 8 != 0 || d->HW % 8 != 0) return false;
    if (d->dtype != PWW_DTYPE_F16 && d->dtype != PWW_DTYPE_BF16) return false;
    if (d->layout != PWW_LAYOUT_NCHW && d->layout != PWW_LAYOUT_NHWC) return false;
    if (d->act != PWW_ACT_NONE && d->act != PWW_ACT_SILU) return false;
    if (d->G > GN_NT || (long)d->B * d->G > GN_COUNTER_BYTES / 4) return false;
    pl = GnPlan();
    pl.coef_bytes = (size_t)d->B * d->C * 2 * sizeof(float);
    if (d->layout == PWW_LAYOUT_NHWC) {
        const int CH = d->C / 8;
        if (CH > 512 || d->G > 256) return false;            // C <= 4096
        pl.nt = CH <= 256 ? 256 : 512;
        const int PL = pl.nt / CH;
        // ~1024 workgroups per launch, at least 4 pixels per thread row where the image allows
        int want = (1024 + d->B - 1) / d->B;
        int px = (d->HW + want - 1) / want;
        const int min_px = 4 * PL;
        if (px < min_px) px = min_px;
        if (px > d->HW) px = d->HW;
        pl.slab_px = px;
        pl.nslab = (d->HW + px - 1) / px;
        pl.partial_bytes = (size_t)d->B * pl.nslab * d->G * 2 * sizeof(double);
        pl.lds = (size_t)2 * PL * d->C * sizeof(float) + (size_t)(pl.nt / d->G) * d->G * 2 * sizeof(double);
    } else {
        const long nchunk = (long)(d->C / d->G) * d->HW / 8;
        int want = (1024 + d->B * d->G - 1) / (d->B * d->G);
        long cap = nchunk / (GN_NT * 2);                    // at least two chunks per thread
        if (cap < 1) cap = 1;
        pl.nseg = (int)(want < cap ? want : cap);
        if (pl.nseg < 1) pl.nseg = 1;
        pl.partial_bytes = (size_t)d->B * d->G * pl.nseg * 2 * sizeof(double);
        const int cpr = d->HW / 8, TPR = cpr < GN_NT ? cpr : GN_NT, RP = GN_NT / TPR;
        // a workgroup takes RP rows per sweep; several sweeps when rows are short so that a thread moves >= 4 chunks
        int sweeps = (4 * TPR + cpr - 1) / cpr;
        if (sweeps < 1) sweeps = 1;
        pl.rows_per_wg = RP * sweeps;
    }
    return true;
}

template <typename T>
int gn_launch(const GnParams &p, const pww_gn_desc_t *d, const GnPlan &pl, hipStream_t stream) {
    if (d->layout == PWW_LAYOUT_NHWC) {
        const dim3 grid(pl.nslab, d->B);
        if (pl.nt == 256) {
            hipLaunchKernelGGL((gn_moments_nhwc<T, 256>), grid, dim3(256), pl.lds, stream, p);
            if (d->act == PWW_ACT_SILU) hipLaunchKernelGGL((gn_apply_nhwc<T, 1, 256>), grid, dim3(256), 0, stream, p);
            else hipLaunchKernelGGL((gn_apply_nhwc<T, 0, 256>), grid, dim3(256), 0, stream, p);
        } else {
            hipLaunchKernelGGL((gn_moments_nhwc<T, 512>), grid, dim3(512), pl.lds, stream, p);
            if (d->act == PWW_ACT_SILU) hipLaunchKernelGGL((gn_apply_nhwc<T, 1, 512>), grid, dim3(512), 0, stream, p);
            else hipLaunchKernelGGL((gn_apply_nhwc<T, 0, 512>), grid, dim3(512), 0, stream, p);
        }
    } else {
        hipLaunchKernelGGL(gn_moments_nchw<T>, dim3(pl.nseg, d->G, d->B), dim3(GN_NT), 0, stream, p);
        const long nrows = (long)d->B * d->C;
        const dim3 grid((unsigned)((nrows + pl.rows_per_wg - 1) / pl.rows_per_wg));
        if (d->act == PWW_ACT_SILU) hipLaunchKernelGGL((gn_apply_nchw<T, 1>), grid, dim3(GN_NT), 0, stream, p);
        else hipLaunchKernelGGL((gn_apply_nchw<T, 0>), grid, dim3(GN_NT), 0, stream, p);
    }
    return check_hip(hipGetLastError(), "group_norm launch");
}

}  // namespace

size_t group_norm_workspace_bytes(const pww_gn_desc_t *d) {
    GnPlan pl;
    if (!gn_plan(d, pl)) return 0;
    return GN_COUNTER_BYTES + ((pl.coef_bytes + 255) & ~(size_t)255) + pl.partial_bytes;
}

int group_norm_fwd(const void *x, const void *add_bc, const void *gamma, const void *beta, void *y, const pww_gn_desc_t *d,
                   void *workspace, size_t workspace_bytes, hipStream_t stream) {
    GnPlan pl;
    if (!x || !y || !d || !workspace) { set_error("group_norm: null argument"); return PWW_EINVAL; }
    if (!gn_plan(d, pl)) {
        set_error("group_norm: unsupported description (B %d C %d HW %d G %d dtype %d layout %d act %d): C and HW multiples of 8, C %% G == 0, "
                  "C <= 4096 for NHWC, B * G <= 1024", d->B, d->C, d->HW, d->G, d->dtype, d->layout, d->act);
        return PWW_ENOTSUP;
    }
    if (!arch_ok()) return PWW_ENOTSUP;
    const size_t need = group_norm_workspace_bytes(d);
    if (workspace_bytes < need) { set_error("group_norm: workspace of %zu bytes, need %zu", workspace_bytes, need); return PWW_EINVAL; }
    if (((uintptr_t)x | (uintptr_t)y | (uintptr_t)workspace | (uintptr_t)add_bc) & 15) { set_error("group_norm: pointers must be 16-byte aligned"); return PWW_EINVAL; }
    GnParams p;
    p.x = x; p.add = add_bc; p.gamma = gamma; p.beta = beta; p.y = y;
    char *ws = static_cast<char *>(workspace);
    p.count = reinterpret_cast<unsigned *>(ws);
    p.coef = reinterpret_cast<float *>(ws + GN_COUNTER_BYTES);
    p.partial = reinterpret_cast<double *>(ws + GN_COUNTER_BYTES + ((pl.coef_bytes + 255) & ~(size_t)255));
    p.B = d->B; p.C = d->C; p.HW = d->HW; p.G = d->G; p.cg = d->C / d->G;
    p.nslab = pl.nslab; p.slab_px = pl.slab_px; p.nseg = pl.nseg; p.rows_per_wg = pl.rows_per_wg;
    p.eps = d->eps; p.act = d->act;
    return d->dtype == PWW_DTYPE_F16 ? gn_launch<f16>(p, d, pl, stream) : gn_launch<bf16>(p, d, pl, stream);
}

}  // namespace pww
