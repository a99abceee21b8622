// Score-reduction pre-pass: per-image max / min / sum / sum-of-squares of the RAW score tensor
// S = Q K^T over all heads, query rows and keys, without writing S to HBM.
//
// The reference's weight_function receives the whole score tensor and every shipped variant takes
// a global scalar reduction of it (qk.max(): paint_with_words/paint_with_words.py:402-405,
// runner.py:104; qk.std(): README.md:152). That scalar feeds the bias of the SAME attention call,
// so it has to exist before the fused kernel runs: this kernel recomputes the score tiles with the
// same MFMA path as pww_attn.hip (pww_tile.h) and reduces them wave -> workgroup -> one atomic per
// statistic per workgroup. HBM-bound streaming of Q and K (K is L2-resident after the first
// query block of each head).
#include "pww_tile.h"

namespace pww {

struct ReduceParams {
    const void *q, *k;
    int B, H, N, M, D;
    long q_sb, q_sh, q_sn;
    long k_sb, k_sh, k_sm;
    double *stats;  // [B][4] = max, min, sum, sumsq
    double *partial;        // workspace: [B][blocks_per_image][4]
    unsigned *ticket;       // workspace: [B] arrival counters (zeroed by the init kernel of every call)
    int blocks_per_image;
};

// Re-initialise the arrival counters on the stream ahead of every reduction launch (a polled word must
// be zeroed per call, never left to the previous launch -- guide G16).
__global__ void qk_ticket_init_kernel(unsigned *ticket, int B) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < B) ticket[i] = 0u;
}

template <typename T, int KS, int NW>
__global__ void __launch_bounds__(NW * 64) qk_reduce_kernel(const ReduceParams p) {
    typedef typename Vec<T>::v8 V8;
    typedef KTile<KS> KT;
    constexpr int NT = NW * 64;
    constexpr int KPT = (KT::NCHUNK + NT - 1) / NT;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *Ks = smem;
    float *red = reinterpret_cast<float *>(smem + KT::BYTES);  // [NW][4]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int hi = lane >> 5, l31 = lane & 31;
    const int BH = p.B * p.H;
    const int bh = blockIdx.x % BH, qb = blockIdx.x / BH;
    const int b = bh / p.H, h = bh - b * p.H;
    const T *Qp = reinterpret_cast<const T *>(p.q) + b * p.q_sb + h * p.q_sh;
    const T *Kp = reinterpret_cast<const T *>(p.k) + b * p.k_sb + h * p.k_sh;
    const int qrow = (qb * NW + wave) * 32 + l31;
    const bool qvalid = qrow < p.N;

    V8 qf[KS];
    load_q_frags<T, KS>(qf, Qp + (long)qrow * p.q_sn, qvalid, hi, p.D);

    float vmax = -INFINITY, vmin = INFINITY, vsum = 0.f, vsq = 0.f;
    u32x4 kreg[KPT];
    const int ntiles = (p.M + KVBLK - 1) / KVBLK;
    ktile_load<T, KS, NT, KPT>(kreg, Kp, p.k_sm, 0, p.M, p.D, tid);
    for (int t = 0; t < ntiles; ++t) {
        const int key0 = t * KVBLK;
        __syncthreads();
        ktile_store<KS, NT, KPT>(kreg, Ks, tid, key0, p.M, p.D);
        __syncthreads();
        if (t + 1 < ntiles) ktile_load<T, KS, NT, KPT>(kreg, Kp, p.k_sm, key0 + KVBLK, p.M, p.D, tid);
        f32x16 s[2];
        score_tile<T, KS>(s, qf, Ks, key0, p.M, l31, hi);
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const bool live = qvalid && (key0 + key_of(kb, r, hi) < p.M);
                const float x = s[kb][r];
                vmax = fmaxf(vmax, live ? x : -INFINITY);
                vmin = fminf(vmin, live ? x : INFINITY);
                vsum += live ? x : 0.f;
                vsq += live ? x * x : 0.f;
            }
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        vmax = fmaxf(vmax, __shfl_xor(vmax, off));
        vmin = fminf(vmin, __shfl_xor(vmin, off));
        vsum += __shfl_xor(vsum, off);
        vsq += __shfl_xor(vsq, off);
    }
    __syncthreads();
    if (lane == 0) {
        red[wave * 4 + 0] = vmax; red[wave * 4 + 1] = vmin;
        red[wave * 4 + 2] = vsum; red[wave * 4 + 3] = vsq;
    }
    __syncthreads();
    // Workgroup partial -> workspace; the LAST workgroup of the image to arrive folds all partials into
    // stats[b]. (Four same-address fp64 atomics per workgroup serialise at L2: 512 workgroups x 4 cost
    // ~25 us.) Hand-off = plain stores -> agent-scope release -> relaxed ticket; last arriver: agent-scope
    // acquire -> plain loads: placement-independent (cdna_hip_programming.md, Guideline 16).
    // (the flag lives in the dynamic LDS block: a static __shared__ object would shift the dynamic base
    //  off its 16-byte alignment -- guide G17)
    volatile int *is_last_p = reinterpret_cast<volatile int *>(red + NW * 4);
#define is_last (*is_last_p)
    const int blk = qb * p.H + h;   // index of this workgroup within image b
    if (tid == 0) {
        double dmax = -INFINITY, dmin = INFINITY, dsum = 0.0, dsq = 0.0;
        for (int w = 0; w < NW; ++w) {
            dmax = fmax(dmax, (double)red[w * 4 + 0]);
            dmin = fmin(dmin, (double)red[w * 4 + 1]);
            dsum += (double)red[w * 4 + 2];
            dsq += (double)red[w * 4 + 3];
        }
        double *slot = p.partial + ((long)b * p.blocks_per_image + blk) * 4;
        slot[0] = dmax; slot[1] = dmin; slot[2] = dsum; slot[3] = dsq;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned prev = __hip_atomic_fetch_add(p.ticket + b, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        is_last = prev == (unsigned)(p.blocks_per_image - 1);
        if (is_last) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    if (is_last) {
        double dmax = -INFINITY, dmin = INFINITY, dsum = 0.0, dsq = 0.0;
        const double *base = p.partial + (long)b * p.blocks_per_image * 4;
        for (int i = tid; i < p.blocks_per_image; i += NW * 64) {
            dmax = fmax(dmax, base[i * 4 + 0]);
            dmin = fmin(dmin, base[i * 4 + 1]);
            dsum += base[i * 4 + 2];
            dsq += base[i * 4 + 3];
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            dmax = fmax(dmax, __shfl_xor(dmax, off));
            dmin = fmin(dmin, __shfl_xor(dmin, off));
            dsum += __shfl_xor(dsum, off);
            dsq += __shfl_xor(dsq, off);
        }
        double *fin = reinterpret_cast<double *>(smem);   // K tile no longer needed: [NW][4] doubles
        __syncthreads();
        if (lane == 0) { fin[wave * 4 + 0] = dmax; fin[wave * 4 + 1] = dmin; fin[wave * 4 + 2] = dsum; fin[wave * 4 + 3] = dsq; }
        __syncthreads();
        if (tid == 0) {
            for (int w = 1; w < NW; ++w) {
                dmax = fmax(dmax, fin[w * 4 + 0]); dmin = fmin(dmin, fin[w * 4 + 1]);
                dsum += fin[w * 4 + 2]; dsq += fin[w * 4 + 3];
            }
            double *st = p.stats + b * 4;
            st[0] = dmax; st[1] = dmin; st[2] = dsum; st[3] = dsq;
        }
    }
#undef is_last
}

template <typename T, int KS, int NW>
static int launch_reduce(ReduceParams p, hipStream_t stream) {
    constexpr size_t lds = KTile<KS>::BYTES + NW * 4 * sizeof(float) + 16;
    const int qblocks = (p.N + NW * 32 - 1) / (NW * 32);
    p.blocks_per_image = qblocks * p.H;
    hipLaunchKernelGGL((qk_reduce_kernel<T, KS, NW>), dim3((unsigned)(qblocks * p.B * p.H)), dim3(NW * 64),
                       lds, stream, p);
    return check_hip(hipGetLastError(), "qk_reduce_kernel launch");
}

bool attn_wide_groups_dims(int B, int H, int N, int D);

template <typename T> static int dispatch_reduce(const ReduceParams &p, hipStream_t s) {
    const int D = p.D;
    const bool wide = attn_wide_groups_dims(p.B, p.H, p.N, p.D);   // the attention kernels' rule (pww_attn.hip)
#define PWW_RED(KS_) (wide ? launch_reduce<T, KS_, 4>(p, s) : launch_reduce<T, KS_, 2>(p, s))
    if (D <= 48) return PWW_RED(3);
    if (D <= 64) return PWW_RED(4);
    if (D <= 80) return PWW_RED(5);
    if (D <= 96) return PWW_RED(6);
    if (D <= 128) return PWW_RED(8);
    return PWW_RED(10);
#undef PWW_RED
}

// blocks per image in the worst case (2-wave workgroups) -> workspace layout [B][max_blocks][4] doubles + [B] tickets
static long reduce_max_blocks(const pww_attn_desc_t *d) { return (long)((d->N + 63) / 64) * d->H; }

size_t qk_reduce_workspace_bytes(const pww_attn_desc_t *d) {
    if (!d || d->B <= 0 || d->H <= 0 || d->N <= 0) return 0;
    return (size_t)d->B * reduce_max_blocks(d) * 4 * sizeof(double) + (size_t)((d->B + 1) / 2) * 2 * sizeof(unsigned) + 64;
}

int qk_reduce(const void *q, const void *k, const pww_attn_desc_t *d, double *stats, void *workspace,
              size_t workspace_bytes, hipStream_t stream) {
    if (!d || !q || !k || !stats || !workspace) { set_error("qk_reduce: null argument"); return PWW_EINVAL; }
    if (workspace_bytes < qk_reduce_workspace_bytes(d) || (reinterpret_cast<uintptr_t>(workspace) & 7)) {
        set_error("qk_reduce: workspace too small or misaligned (need %zu bytes, 8-byte aligned)", qk_reduce_workspace_bytes(d));
        return PWW_EINVAL;
    }
    if (d->B <= 0 || d->H <= 0 || d->N <= 0 || d->M <= 0 || d->D <= 0) { set_error("qk_reduce: non-positive dimension"); return PWW_EINVAL; }
    if (d->D % 8 != 0 || d->D > PWW_MAX_HEAD_DIM) { set_error("qk_reduce: head dim %d unsupported", d->D); return PWW_ENOTSUP; }
    if (d->dtype != PWW_DTYPE_F16 && d->dtype != PWW_DTYPE_BF16) { set_error("qk_reduce: dtype %d unsupported", d->dtype); return PWW_ENOTSUP; }
    if ((reinterpret_cast<uintptr_t>(q) & 15) || (reinterpret_cast<uintptr_t>(k) & 15) || (reinterpret_cast<uintptr_t>(stats) & 7)) {
        set_error("qk_reduce: misaligned pointer"); return PWW_EINVAL;
    }
    for (int i = 0; i < 3; ++i)
        if (d->q_stride[i] % 8 || d->k_stride[i] % 8) { set_error("qk_reduce: strides must be multiples of 8 elements"); return PWW_EINVAL; }
    if (!arch_ok()) return PWW_ENOTSUP;
    ReduceParams p;
    p.q = q; p.k = k; p.B = d->B; p.H = d->H; p.N = d->N; p.M = d->M; p.D = d->D;
    p.q_sb = d->q_stride[0]; p.q_sh = d->q_stride[1]; p.q_sn = d->q_stride[2];
    p.k_sb = d->k_stride[0]; p.k_sh = d->k_stride[1]; p.k_sm = d->k_stride[2];
    p.stats = stats;
    p.partial = reinterpret_cast<double *>(workspace);
    p.ticket = reinterpret_cast<unsigned *>(p.partial + (long)d->B * reduce_max_blocks(d) * 4);
    p.blocks_per_image = 0;
    hipLaunchKernelGGL(qk_ticket_init_kernel, dim3((d->B + 63) / 64), dim3(64), 0, stream, p.ticket, d->B);
    if (int rc = check_hip(hipGetLastError(), "qk_ticket_init_kernel launch")) return rc;
    return d->dtype == PWW_DTYPE_F16 ? dispatch_reduce<f16>(p, stream) : dispatch_reduce<bf16>(p, stream);
}

}  // namespace pww
