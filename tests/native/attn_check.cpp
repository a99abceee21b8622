// Native parity harness for libpww_hip.so (test infrastructure; built by tests/native/Makefile,
// run on the GPU box by tests/test_native_gpu.py). Calls the C ABI of include/pww_hip.h exactly
// as a foreign binding would and compares every entry point with an independent fp64 host
// reference written here (plain loops, no shared code with the kernels).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <string>
#include <algorithm>
#include "../../include/pww_hip.h"

#define HIPCHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static inline uint32_t rng_u32() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return (uint32_t)(rng_state >> 32); }
static inline float rng_uniform() { return (rng_u32() >> 8) * (1.0f / 16777216.0f); }
static inline float rng_normal() { float u1 = rng_uniform() + 1e-7f, u2 = rng_uniform(); return sqrtf(-2.f * logf(u1)) * cosf(6.2831853f * u2); }

static uint16_t to_bf16(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); return (uint16_t)(u >> 16); }
static float from_bf16(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }
static uint16_t to_f16(float f) { _Float16 h = (_Float16)f; uint16_t u; memcpy(&u, &h, 2); return u; }
static float from_f16(uint16_t u) { _Float16 h; memcpy(&h, &u, 2); return (float)h; }
static uint16_t to_t(float f, int dt) { return dt == PWW_DTYPE_F16 ? to_f16(f) : to_bf16(f); }
static float from_t(uint16_t u, int dt) { return dt == PWW_DTYPE_F16 ? from_f16(u) : from_bf16(u); }

static int g_fail = 0;
#ifndef PWW_EXPERIMENTS
#define PWW_EXPERIMENTS 0      // 1: built against libpww_hip_experiments.so (attn_check_experiments): + the cases of the moved entry points
#endif
static bool g_product_only = false;  // --product-only: for a cross-attention case, ONLY the launch the product issues (fused_ex with bias_cols and the
                                     // gated-images hint, first half of the gates open), 20 times, no checks: what a PMC pass should see
static bool g_timeline = false;      // --timeline: print the phase time stamps of the case's launches (pww_debug_timeline)

// one launch under pww_debug_timeline: per stamp slot, microseconds since the earliest kernel-entry stamp of the launch
template <typename F> static void timeline_report(const char *name, const char *what, F launch) {
    const size_t wgs = 1 << 16, bytes = wgs * 8 * sizeof(unsigned long long);
    unsigned long long *buf; HIPCHECK(hipMalloc(&buf, bytes));
    for (int rep = 0; rep < 24; ++rep) launch();                      // warm (long enough for the clock to settle under the launch's own load)
    HIPCHECK(hipDeviceSynchronize());
    HIPCHECK(hipMemset(buf, 0, bytes));
    pww_debug_timeline(buf, bytes);
    launch();
    HIPCHECK(hipDeviceSynchronize());
    pww_debug_timeline(nullptr, 0);
    std::vector<unsigned long long> h(wgs * 8);
    HIPCHECK(hipMemcpy(h.data(), buf, bytes, hipMemcpyDeviceToHost));
    unsigned long long t0 = ~0ull; size_t n = 0;
    for (size_t w = 0; w < wgs; ++w) if (h[w * 8]) { t0 = std::min(t0, h[w * 8]); n = w + 1; }
    printf("TIMELINE %-28s %s: %zu workgroups; us since the first workgroup's entry, per stamp [n mean min max]\n", name, what, n);
    {   // slot 7 = shader cycles between entry and exit of the workgroup: cycles / wall time = the clock the launch ran at
        double mhz = 0; long cnt = 0;
        for (size_t w = 0; w < n; ++w) {
            unsigned long long last = 0; for (int sl = 1; sl < 7; ++sl) last = std::max(last, h[w * 8 + sl]);
            if (h[w * 8 + 7] && last > h[w * 8]) { mhz += (double)h[w * 8 + 7] / ((double)(last - h[w * 8]) * 0.01); ++cnt; }
        }
        if (cnt) printf("TIMELINE %-28s   shader clock over the workgroups' lifetimes: %.0f MHz\n", name, mhz / cnt);
    }
    for (int sl = 0; sl < 7; ++sl) {
        double sum = 0, mn = 1e30, mx = 0; long cnt = 0;
        for (size_t w = 0; w < n; ++w) if (h[w * 8 + sl]) { const double us = (double)(h[w * 8 + sl] - t0) * 0.01; sum += us; mn = std::min(mn, us); mx = std::max(mx, us); ++cnt; }
        if (cnt) printf("TIMELINE %-28s   stamp %d: n=%ld mean %.2f min %.2f max %.2f\n", name, sl, cnt, sum / cnt, mn, mx);
    }
    (void)hipFree(buf);
}

struct Case {
    const char *name; int dtype, B, H, N, M, D;
    int bias_mode;   // 0 none, 1 [N,M] shared + coeff[B], 2 full [B,H,N,M]
    bool self;       // K/V come from the same token set as Q (M == N)
    int row_step;    // check every row_step-th query row
    float qk_gain;   // scales Q so logits get a realistic spread
    float late_outlier = 0.f;   // != 0: key row M - 5 of every image is multiplied by this (scores far above everything seen before)
    int bias_cols = 0;          // bias_mode 1: columns >= bias_cols of the map are zero and the *_ex call says so (0 = every column may be non-zero)
    int compact_R = 0;          // bias_mode 1: only compact_R columns (< bias_cols) are non-zero; the *_ex call also gets the compact form
    bool on_request = false;    // profiling shapes (large batches): only run with --only / --match
};

template <typename T> static T *dalloc(size_t n) { T *p; HIPCHECK(hipMalloc(&p, n * sizeof(T) + 64)); return p; }

static void run_case(const Case &c, bool timing) {
    const int B = c.B, H = c.H, N = c.N, M = c.M, D = c.D, C = H * D;
    std::vector<uint16_t> q((size_t)B * N * C), k((size_t)B * M * C), v((size_t)B * M * C);
    std::vector<float> qf(q.size()), kf(k.size()), vf(v.size());
    for (size_t i = 0; i < q.size(); ++i) { q[i] = to_t(rng_normal() * c.qk_gain, c.dtype); qf[i] = from_t(q[i], c.dtype); }
    for (size_t i = 0; i < k.size(); ++i) {
        float kv = rng_normal();
        if (c.late_outlier != 0.f && (int)((i / C) % M) == M - 5) kv *= c.late_outlier;
        k[i] = to_t(kv, c.dtype); kf[i] = from_t(k[i], c.dtype);
    }
    for (size_t i = 0; i < v.size(); ++i) { v[i] = to_t(rng_normal() + 0.1f * (float)(i % 7), c.dtype); vf[i] = from_t(v[i], c.dtype); }
    std::vector<float> bias, coeff;
    pww_attn_desc_t d; memset(&d, 0, sizeof(d));
    d.dtype = c.dtype; d.B = B; d.H = H; d.N = N; d.M = M; d.D = D;
    d.q_stride[0] = (int64_t)N * C; d.q_stride[1] = D; d.q_stride[2] = C;
    d.k_stride[0] = (int64_t)M * C; d.k_stride[1] = D; d.k_stride[2] = C;
    d.v_stride[0] = (int64_t)M * C; d.v_stride[1] = D; d.v_stride[2] = C;
    d.o_stride[0] = (int64_t)N * C; d.o_stride[1] = D; d.o_stride[2] = C;
    d.scale = 1.0f / sqrtf((float)D);
    std::vector<int> ccols;           // compact form: the non-zero columns
    if (c.bias_mode == 1) {
        bias.resize((size_t)N * M); coeff.resize(B);
        for (auto &x : bias) x = (rng_uniform() < 0.3f) ? rng_uniform() * 1.5f : 0.f;
        const int bc = c.bias_cols > 0 ? c.bias_cols : M;
        if (c.compact_R > 0) {        // every (bc / R)-th column below bc, last one = bc - 1
            for (int r = 0; r < c.compact_R; ++r) ccols.push_back(r == c.compact_R - 1 ? bc - 1 : (int)((long)r * bc / c.compact_R));
        }
        for (int n = 0; n < N; ++n) for (int m = 0; m < M; ++m) {
            bool keep = m < bc;
            if (keep && !ccols.empty()) keep = std::find(ccols.begin(), ccols.end(), m) != ccols.end();
            if (!keep) bias[(size_t)n * M + m] = 0.f;
        }
        for (int b = 0; b < B; ++b) coeff[b] = 2.0f + 3.0f * b;
        d.bias_stride[0] = 0; d.bias_stride[1] = 0; d.bias_stride[2] = M; d.bias_stride[3] = 1;
    } else if (c.bias_mode == 2) {
        bias.resize((size_t)B * H * N * M);
        for (auto &x : bias) x = rng_normal() * 2.f;
        d.bias_stride[0] = (int64_t)H * N * M; d.bias_stride[1] = (int64_t)N * M; d.bias_stride[2] = M; d.bias_stride[3] = 1;
    }
    uint16_t *dq = dalloc<uint16_t>(q.size()), *dk = dalloc<uint16_t>(k.size()), *dv = dalloc<uint16_t>(v.size()), *dout = dalloc<uint16_t>(q.size());
    float *dbias = nullptr, *dcoeff = nullptr;
    HIPCHECK(hipMemcpy(dq, q.data(), q.size() * 2, hipMemcpyHostToDevice));
    HIPCHECK(hipMemcpy(dk, k.data(), k.size() * 2, hipMemcpyHostToDevice));
    HIPCHECK(hipMemcpy(dv, v.data(), v.size() * 2, hipMemcpyHostToDevice));
    HIPCHECK(hipMemset(dout, 0xff, q.size() * 2));
    if (!bias.empty()) { dbias = dalloc<float>(bias.size()); HIPCHECK(hipMemcpy(dbias, bias.data(), bias.size() * 4, hipMemcpyHostToDevice)); }
    if (!coeff.empty()) { dcoeff = dalloc<float>(coeff.size()); HIPCHECK(hipMemcpy(dcoeff, coeff.data(), coeff.size() * 4, hipMemcpyHostToDevice)); }
    double *dstats = dalloc<double>(4 * B);
    const size_t ws_bytes = pww_workspace_bytes(&d);
    void *dws = dalloc<char>(ws_bytes + 8);

#if !PWW_EXPERIMENTS
    if (g_product_only) { printf("SKIP %-28s --product-only on this case class launches round 3's fused_ex: attn_check_experiments\n", c.name); return; }
#else
    if (g_product_only) {
        if (c.bias_mode != 1 || M > 128) { printf("SKIP %-28s --product-only needs a cross-attention case with the [N, M] map\n", c.name); return; }
        const size_t fws_bytes = pww_cross_fused_workspace_bytes(&d), sync_bytes = pww_cross_fused_state_bytes(&d);
        void *fws = dalloc<char>(fws_bytes + 8); unsigned *dsync = dalloc<unsigned>(sync_bytes / 4 + 1);
        HIPCHECK(hipMemset(dsync, 0, sync_bytes));
        std::vector<float> gate(coeff); if (B > 1) for (int b = B / 2; b < B; ++b) gate[b] = 0.f;
        float *dgate = dalloc<float>(B); HIPCHECK(hipMemcpy(dgate, gate.data(), B * 4, hipMemcpyHostToDevice));
        pww_cross_opts_t op; memset(&op, 0, sizeof(op)); op.size = sizeof(op); op.bias_cols = c.bias_cols; op.gated_images = B > 1 ? B / 2 : 0;
        int r = 0;
        for (int i = 0; i < 20 && !r; ++i)
            r = pww_cross_attn_fwd_fused_ex(dq, dk, dv, dout, dbias, PWW_STAT_MAX, 0.37f, dgate, &d, nullptr, dsync, sync_bytes, fws, fws_bytes, &op, nullptr);
        HIPCHECK(hipDeviceSynchronize());
        printf("%s %-28s product-only: 20 x fused_ex (bias_cols=%d, gated_images=%d) rc=%d\n", r ? "FAIL" : "PASS", c.name, op.bias_cols, op.gated_images, r);
        if (r) g_fail++;
        return;
    }
#endif
    int rc = c.bias_mode ? pww_cross_attn_fwd(dq, dk, dv, dout, dbias, dcoeff, &d, nullptr)
                         : pww_self_attn_fwd(dq, dk, dv, dout, &d, nullptr);
    if (rc) { printf("FAIL %-28s attn rc=%d err=%s\n", c.name, rc, pww_last_error()); g_fail++; return; }
    rc = pww_qk_reduce(dq, dk, &d, dstats, dws, ws_bytes, nullptr);
    if (rc) { printf("FAIL %-28s reduce rc=%d err=%s\n", c.name, rc, pww_last_error()); g_fail++; return; }
    HIPCHECK(hipDeviceSynchronize());
    std::vector<uint16_t> out(q.size());
    std::vector<double> stats(4 * B);
    HIPCHECK(hipMemcpy(out.data(), dout, out.size() * 2, hipMemcpyDeviceToHost));
    HIPCHECK(hipMemcpy(stats.data(), dstats, stats.size() * 8, hipMemcpyDeviceToHost));

    // pww_cross_attn_fwd_stat: coefficient formed in the kernel from the statistics == explicit coefficient call, bit for bit
    if (c.bias_mode == 1) {
        const double cnt = (double)H * N * M;
        for (int kind : {PWW_STAT_MAX, PWW_STAT_STD, PWW_STAT_ABSMAX, PWW_STAT_MEAN}) {
            const float s0 = 0.37f;
            std::vector<float> cexp(B);
            for (int b = 0; b < B; ++b) {
                const double *st = &stats[4 * b];
                double v;
                if (kind == PWW_STAT_MAX) v = st[0];
                else if (kind == PWW_STAT_ABSMAX) v = std::max(fabs(st[0]), fabs(st[1]));
                else if (kind == PWW_STAT_MEAN) v = st[2] / cnt;
                else { v = (st[3] - st[2] * st[2] / cnt) / (cnt - 1.0); if (v < 0) v = 0; }
                float f = (float)v;
                if (kind == PWW_STAT_STD) f = sqrtf(f);
                cexp[b] = (s0 * f) * coeff[b];
            }
            float *dce = dalloc<float>(B);
            uint16_t *o1 = dalloc<uint16_t>(q.size()), *o2 = dalloc<uint16_t>(q.size());
            HIPCHECK(hipMemcpy(dce, cexp.data(), B * 4, hipMemcpyHostToDevice));
            int r1 = pww_cross_attn_fwd(dq, dk, dv, o1, dbias, dce, &d, nullptr);
            int r2 = pww_cross_attn_fwd_stat(dq, dk, dv, o2, dbias, dstats, kind, cnt, s0, dcoeff, &d, nullptr);
            HIPCHECK(hipDeviceSynchronize());
            std::vector<uint16_t> h1(q.size()), h2(q.size());
            HIPCHECK(hipMemcpy(h1.data(), o1, h1.size() * 2, hipMemcpyDeviceToHost));
            HIPCHECK(hipMemcpy(h2.data(), o2, h2.size() * 2, hipMemcpyDeviceToHost));
            long diff = 0;
            for (size_t i = 0; i < h1.size(); ++i) diff += h1[i] != h2[i];
            const bool ok = r1 == 0 && r2 == 0 && diff == 0;
            printf("%s %-28s stat-coefficient kind=%d: %ld differing elements vs explicit coefficient (rc %d %d)\n", ok ? "PASS" : "FAIL", c.name, kind, diff, r1, r2);
            if (!ok) g_fail++;
            (void)hipFree(dce); (void)hipFree(o1); (void)hipFree(o2);
        }
    }

#if PWW_EXPERIMENTS
    // pww_cross_attn_fwd_fused: statistic + attention in ONE launch == pww_qk_reduce + pww_cross_attn_fwd_stat, bit for bit
    // (output AND statistics), with every gate open and with the last image gated out (the unconditional rows of a
    // CFG-folded batch); repeated launches re-use the same sync words (the kernel leaves them zero).
    if (c.bias_mode == 1 && M <= 128) {
        const size_t fws_bytes = pww_cross_fused_workspace_bytes(&d), sync_bytes = pww_cross_fused_state_bytes(&d);
        void *fws = dalloc<char>(fws_bytes + 8); unsigned *dsync = dalloc<unsigned>(sync_bytes / 4 + 1);
        HIPCHECK(hipMemset(dsync, 0, sync_bytes));
        double *fstats = dalloc<double>(4 * B);
        uint16_t *o1 = dalloc<uint16_t>(q.size()), *o2 = dalloc<uint16_t>(q.size());
        float *dgate = dalloc<float>(B);
        for (int variant = 0; variant < (B > 1 ? 2 : 1); ++variant) {
            std::vector<float> gate(coeff);
            if (variant == 1) gate[B - 1] = 0.f;
            HIPCHECK(hipMemcpy(dgate, gate.data(), B * 4, hipMemcpyHostToDevice));
            for (int kind : {PWW_STAT_MAX, PWW_STAT_STD, PWW_STAT_ABSMAX, PWW_STAT_MEAN, PWW_STAT_MIN, PWW_STAT_NONE}) {
                const float s0 = 0.37f;
                HIPCHECK(hipMemset(fstats, 0, 4 * B * 8));
                HIPCHECK(hipMemset(o1, 0xff, q.size() * 2)); HIPCHECK(hipMemset(o2, 0xee, q.size() * 2));
                int r1 = pww_cross_attn_fwd_stat(dq, dk, dv, o1, dbias, kind == PWW_STAT_NONE ? nullptr : dstats, kind, (double)H * N * M, s0, dgate, &d, nullptr);
                int r2 = 0;
                for (int rep = 0; rep < 3 && !r2; ++rep)
                    r2 = pww_cross_attn_fwd_fused(dq, dk, dv, o2, dbias, kind, s0, dgate, &d, fstats, dsync, sync_bytes, fws, fws_bytes, nullptr);
                HIPCHECK(hipDeviceSynchronize());
                std::vector<uint16_t> h1(q.size()), h2(q.size()); std::vector<double> fs(4 * B); std::vector<unsigned> hs(sync_bytes / 4);
                HIPCHECK(hipMemcpy(h1.data(), o1, h1.size() * 2, hipMemcpyDeviceToHost));
                HIPCHECK(hipMemcpy(h2.data(), o2, h2.size() * 2, hipMemcpyDeviceToHost));
                HIPCHECK(hipMemcpy(fs.data(), fstats, fs.size() * 8, hipMemcpyDeviceToHost));
                HIPCHECK(hipMemcpy(hs.data(), dsync, sync_bytes, hipMemcpyDeviceToHost));
                // statistics: bit for bit. Outputs: the single-stage kernel computes a row with a one-shot softmax (round 4), the general
                // kernel with two online tiles -- same mathematics, other roundings: within two units of the storage type's last place
                long diff = 0, sdiff = 0, dirty = 0; double dmax = 0, omax = 0;
                for (size_t i = 0; i < h1.size(); ++i) {
                    diff += h1[i] != h2[i];
                    const double a = from_t(h1[i], c.dtype), bq = from_t(h2[i], c.dtype);
                    dmax = std::max(dmax, fabs(a - bq)); omax = std::max(omax, fabs(a));
                    if (!(bq == bq)) dmax = 1e30;
                }
                if (kind != PWW_STAT_NONE)
                    for (int b = 0; b < B; ++b) if (gate[b] != 0.f) for (int j = 0; j < 4; ++j) sdiff += memcmp(&fs[4 * b + j], &stats[4 * b + j], 8) != 0;
                for (unsigned w : hs) dirty += w != 0;
                const bool ok = r1 == 0 && r2 == 0 && dmax <= (c.dtype == PWW_DTYPE_F16 ? 1.0 / 512 : 1.0 / 64) * omax && sdiff == 0 && dirty == 0;
                printf("%s %-28s fused kind=%d gate-variant=%d: max output diff %.2e of max|O| %.2f (%ld elements differ), %ld differing statistics, %ld dirty state words (rc %d %d%s%s)\n",
                       ok ? "PASS" : "FAIL", c.name, kind, variant, dmax, omax, diff, sdiff, dirty, r1, r2, r2 ? " " : "", r2 ? pww_last_error() : "");
                if (!ok) g_fail++;
            }
        }
        // pww_cross_attn_fwd_fused_ex: the optional arguments must not change a single bit -- (a) the coefficient scalar read from a
        // device word, (b) the promise that columns >= bias_cols are zero, (c) the compact form of the same map (with and without
        // the dense map beside it), (d) all of them together
        {
            std::vector<float> gate(coeff); if (B > 1) gate[B - 1] = 0.f;
            HIPCHECK(hipMemcpy(dgate, gate.data(), B * 4, hipMemcpyHostToDevice));
            const float s0 = 0.37f;
            float *dscalar = dalloc<float>(4); HIPCHECK(hipMemcpy(dscalar, &s0, 4, hipMemcpyHostToDevice));
            float *dcompact = nullptr; int32_t *dcidx = nullptr; const int R = (int)ccols.size();
            if (R) {
                std::vector<float> comp((size_t)N * R);
                for (int n = 0; n < N; ++n) for (int r = 0; r < R; ++r) comp[(size_t)n * R + r] = bias[(size_t)n * M + ccols[r]];
                std::vector<int32_t> ci(ccols.begin(), ccols.end());
                dcompact = dalloc<float>(comp.size()); dcidx = dalloc<int32_t>(R);
                HIPCHECK(hipMemcpy(dcompact, comp.data(), comp.size() * 4, hipMemcpyHostToDevice));
                HIPCHECK(hipMemcpy(dcidx, ci.data(), R * 4, hipMemcpyHostToDevice));
            }
            HIPCHECK(hipMemset(o1, 0xff, q.size() * 2));
            int r1 = pww_cross_attn_fwd_fused(dq, dk, dv, o1, dbias, PWW_STAT_MAX, s0, dgate, &d, nullptr, dsync, sync_bytes, fws, fws_bytes, nullptr);
            HIPCHECK(hipDeviceSynchronize());
            std::vector<uint16_t> h1(q.size()), h2(q.size());
            HIPCHECK(hipMemcpy(h1.data(), o1, h1.size() * 2, hipMemcpyDeviceToHost));
            // without stats_out the extremes-only statistics (max / min / absmax) take the workgroup-wide running extreme instead of
            // per-block partials: same outputs as the two launches, bit for bit
            for (int kind : {PWW_STAT_MAX, PWW_STAT_MIN, PWW_STAT_ABSMAX}) {
                HIPCHECK(hipMemset(o2, 0xee, q.size() * 2)); HIPCHECK(hipMemset(o1, 0xdd, q.size() * 2));
                int ra = pww_cross_attn_fwd_stat(dq, dk, dv, o2, dbias, dstats, kind, (double)H * N * M, s0, dgate, &d, nullptr);
                int rb = 0;
                for (int rep = 0; rep < 2 && !rb; ++rep)
                    rb = pww_cross_attn_fwd_fused(dq, dk, dv, o1, dbias, kind, s0, dgate, &d, nullptr, dsync, sync_bytes, fws, fws_bytes, nullptr);
                HIPCHECK(hipDeviceSynchronize());
                std::vector<uint16_t> ha(q.size()), hb(q.size()); std::vector<unsigned> hs(sync_bytes / 4);
                HIPCHECK(hipMemcpy(ha.data(), o2, ha.size() * 2, hipMemcpyDeviceToHost)); HIPCHECK(hipMemcpy(hb.data(), o1, hb.size() * 2, hipMemcpyDeviceToHost));
                HIPCHECK(hipMemcpy(hs.data(), dsync, sync_bytes, hipMemcpyDeviceToHost));
                long diff = 0, dirty = 0; double dmax = 0, omax = 0;
                for (size_t i = 0; i < ha.size(); ++i) {
                    diff += ha[i] != hb[i];
                    const double a = from_t(ha[i], c.dtype), bq = from_t(hb[i], c.dtype);
                    dmax = std::max(dmax, fabs(a - bq)); omax = std::max(omax, fabs(a));
                    if (!(bq == bq)) dmax = 1e30;
                }
                for (unsigned w : hs) dirty += w != 0;
                const bool ok = ra == 0 && rb == 0 && dmax <= (c.dtype == PWW_DTYPE_F16 ? 1.0 / 512 : 1.0 / 64) * omax && dirty == 0;
                printf("%s %-28s fused kind=%d without stats_out (running extremes): max output diff %.2e of max|O| %.2f vs two launches (%ld elements differ), %ld dirty state words (rc %d %d)\n", ok ? "PASS" : "FAIL", c.name, kind, dmax, omax, diff, dirty, ra, rb);
                if (!ok) g_fail++;
            }
            HIPCHECK(hipMemset(o1, 0xff, q.size() * 2));
            r1 = pww_cross_attn_fwd_fused(dq, dk, dv, o1, dbias, PWW_STAT_MAX, s0, dgate, &d, nullptr, dsync, sync_bytes, fws, fws_bytes, nullptr);
            HIPCHECK(hipDeviceSynchronize());
            HIPCHECK(hipMemcpy(h1.data(), o1, h1.size() * 2, hipMemcpyDeviceToHost));
            std::vector<uint16_t> htile;
            for (int variant = 0; variant < 8; ++variant) {
                pww_cross_opts_t op; memset(&op, 0, sizeof(op)); op.size = sizeof(op);
                const float *use_bias = dbias;
                const char *what = "";
                if (variant == 0) { op.coeff_scalar_dev = dscalar; what = "coeff_scalar_dev"; }
                else if (variant == 1) { if (!c.bias_cols) continue; op.bias_cols = c.bias_cols; what = "bias_cols"; }
                else if (variant == 2) { if (!R) continue; op.bias_compact = dcompact; op.col_idx = dcidx; op.R = R; op.compact_stride[1] = R; op.bias_cols = c.bias_cols; what = "compact + dense"; }
                else if (variant == 3) { if (!R) continue; op.bias_compact = dcompact; op.col_idx = dcidx; op.R = R; op.compact_stride[1] = R; op.bias_cols = c.bias_cols; use_bias = nullptr; what = "compact only"; }
                else if (variant == 4) { op.coeff_scalar_dev = dscalar; op.bias_cols = c.bias_cols; if (R) { op.bias_compact = dcompact; op.col_idx = dcidx; op.R = R; op.compact_stride[1] = R; } what = "all options"; }
                // the gated-images hint only moves work between workgroups: right (all but the last image), wrong (1 although more are
                // gated in: the others just get the short end), and with the compact form
                else if (variant == 5) { if (B < 2) continue; op.gated_images = B - 1; op.bias_cols = c.bias_cols; what = "gated_images = B - 1 (right)"; }
                else if (variant == 6) { if (B < 3) continue; op.gated_images = 1; op.bias_cols = c.bias_cols; what = "gated_images = 1 (wrong)"; }
                else { if (B < 2 || !R) continue; op.gated_images = B - 1; op.bias_cols = c.bias_cols; op.bias_compact = dcompact; op.col_idx = dcidx; op.R = R; op.compact_stride[1] = R; what = "gated_images + compact"; }
                HIPCHECK(hipMemset(o2, 0xee, q.size() * 2));
                int r2 = pww_cross_attn_fwd_fused_ex(dq, dk, dv, o2, use_bias, PWW_STAT_MAX, op.coeff_scalar_dev ? -1.f : s0, dgate, &d, nullptr, dsync, sync_bytes, fws, fws_bytes, &op, nullptr);
                HIPCHECK(hipDeviceSynchronize());
                if (r2 == PWW_ENOTSUP && !use_bias) { printf("SKIP %-28s fused_ex %s: not resident without the dense map (%s)\n", c.name, what, pww_last_error()); continue; }
                HIPCHECK(hipMemcpy(h2.data(), o2, h2.size() * 2, hipMemcpyDeviceToHost));
                // Forms that stage the bias rows in LDS (bias_cols / compact) run the first / lazy softmax steps, the plain call reads its
                // bias per lane and runs the general online step: bit-identical AMONG the staged forms, equal to the last bit or two of
                // the storage type ACROSS the two families (the same mathematics with another rounding order).
                const bool staged = op.bias_compact || (op.bias_cols && op.bias_cols <= 48);      // (pww_cross.hip: dense rows are staged up to 48 columns)
                long diff = 0, tdiff = 0; double dmax = 0, omax = 0;
                for (size_t i = 0; i < h1.size(); ++i) {
                    diff += h1[i] != h2[i];
                    const double a = from_t(h1[i], c.dtype), bq = from_t(h2[i], c.dtype);
                    dmax = std::max(dmax, fabs(a - bq)); omax = std::max(omax, fabs(a));
                    if (!(bq == bq)) dmax = 1e30;
                }
                if (staged) {
                    if (htile.empty()) htile = h2;
                    else for (size_t i = 0; i < h2.size(); ++i) tdiff += htile[i] != h2[i];
                }
                const bool ok = r1 == 0 && r2 == 0 && (staged ? tdiff == 0 && dmax <= (c.dtype == PWW_DTYPE_F16 ? 1.0 / 512 : 1.0 / 64) * omax : diff == 0);
                printf("%s %-28s fused_ex [%s]: %ld differing outputs vs the plain fused call (max diff %.2e of max|O| %.2f), %ld vs the first staged form (rc %d %d%s%s)\n",
                       ok ? "PASS" : "FAIL", c.name, what, diff, dmax, omax, tdiff, r1, r2, r2 ? " " : "", r2 ? pww_last_error() : "");
                if (!ok) g_fail++;
            }
            // pww_cross_attn_fwd_stat_ex with the device word == pww_cross_attn_fwd_stat with the value
            {
                pww_cross_opts_t op; memset(&op, 0, sizeof(op)); op.size = sizeof(op); op.coeff_scalar_dev = dscalar;
                int r3 = pww_cross_attn_fwd_stat(dq, dk, dv, o1, dbias, dstats, PWW_STAT_STD, (double)H * N * M, s0, dgate, &d, nullptr);
                int r4 = pww_cross_attn_fwd_stat_ex(dq, dk, dv, o2, dbias, dstats, PWW_STAT_STD, (double)H * N * M, -1.f, dgate, &d, &op, nullptr);
                HIPCHECK(hipDeviceSynchronize());
                HIPCHECK(hipMemcpy(h1.data(), o1, h1.size() * 2, hipMemcpyDeviceToHost));
                HIPCHECK(hipMemcpy(h2.data(), o2, h2.size() * 2, hipMemcpyDeviceToHost));
                long diff = 0; for (size_t i = 0; i < h1.size(); ++i) diff += h1[i] != h2[i];
                const bool ok = r3 == 0 && r4 == 0 && diff == 0;
                printf("%s %-28s stat_ex [coeff_scalar_dev]: %ld differing outputs (rc %d %d)\n", ok ? "PASS" : "FAIL", c.name, diff, r3, r4);
                if (!ok) g_fail++;
            }
            for (void *ptr : {(void *)dscalar, (void *)dcompact, (void *)dcidx}) if (ptr) (void)hipFree(ptr);
        }
        if (timing) {
            hipEvent_t e0, e1; HIPCHECK(hipEventCreate(&e0)); HIPCHECK(hipEventCreate(&e1));
            const int iters = 50;
            std::vector<float> gate(coeff); if (B > 1) for (int b = B / 2; b < B; ++b) gate[b] = 0.f;     // folded CFG batch: second half unconditional
            HIPCHECK(hipMemcpy(dgate, gate.data(), B * 4, hipMemcpyHostToDevice));
            if (g_timeline) {
                pww_cross_opts_t op; memset(&op, 0, sizeof(op)); op.size = sizeof(op); op.bias_cols = c.bias_cols; op.gated_images = B > 1 ? B / 2 : 0;
                timeline_report(c.name, "fused cross-attention with bias_cols and the gated-images hint (stamps: 0 entry, 1 K/V staged, 2 partials published, 3 statistic folded, 4 outputs stored, 5 exit)", [&]() {
                    pww_cross_attn_fwd_fused_ex(dq, dk, dv, o2, dbias, PWW_STAT_MAX, 0.37f, dgate, &d, nullptr, dsync, sync_bytes, fws, fws_bytes, &op, nullptr); });
            }
            float ms_f = 0, ms_s = 0;
            for (int pass = 0; pass < 2; ++pass) {
                for (int i = 0; i < 5 + iters; ++i) {
                    if (i == 5) HIPCHECK(hipEventRecord(e0, nullptr));
                    if (pass == 0) pww_cross_attn_fwd_fused(dq, dk, dv, o2, dbias, PWW_STAT_MAX, 0.37f, dgate, &d, nullptr, dsync, sync_bytes, fws, fws_bytes, nullptr);
                    else { pww_qk_reduce(dq, dk, &d, dstats, dws, ws_bytes, nullptr); pww_cross_attn_fwd_stat(dq, dk, dv, o1, dbias, dstats, PWW_STAT_MAX, (double)H * N * M, 0.37f, dgate, &d, nullptr); }
                }
                HIPCHECK(hipEventRecord(e1, nullptr)); HIPCHECK(hipEventSynchronize(e1));
                HIPCHECK(hipEventElapsedTime(pass == 0 ? &ms_f : &ms_s, e0, e1));
            }
            printf("TIME %-28s fused statistic+attention %.2f us/call | pww_qk_reduce + pww_cross_attn_fwd_stat %.2f us/call (gates: first half 1, second half 0)\n",
                   c.name, ms_f * 1e3 / iters, ms_s * 1e3 / iters);
            if (c.bias_cols) {
                pww_cross_opts_t op; memset(&op, 0, sizeof(op)); op.size = sizeof(op); op.bias_cols = c.bias_cols;
                float ms_e = 0;
                for (int i = 0; i < 5 + iters; ++i) {
                    if (i == 5) HIPCHECK(hipEventRecord(e0, nullptr));
                    pww_cross_attn_fwd_fused_ex(dq, dk, dv, o2, dbias, PWW_STAT_MAX, 0.37f, dgate, &d, nullptr, dsync, sync_bytes, fws, fws_bytes, &op, nullptr);
                }
                HIPCHECK(hipEventRecord(e1, nullptr)); HIPCHECK(hipEventSynchronize(e1));
                HIPCHECK(hipEventElapsedTime(&ms_e, e0, e1));
                printf("TIME %-28s fused_ex with bias_cols=%d: %.2f us/call\n", c.name, c.bias_cols, ms_e * 1e3 / iters);
                if (B > 1) {               // what the product passes for a CFG-folded batch: + the hint that the first half is gated in
                    op.gated_images = B / 2;
                    for (int i = 0; i < 5 + iters; ++i) {
                        if (i == 5) HIPCHECK(hipEventRecord(e0, nullptr));
                        pww_cross_attn_fwd_fused_ex(dq, dk, dv, o2, dbias, PWW_STAT_MAX, 0.37f, dgate, &d, nullptr, dsync, sync_bytes, fws, fws_bytes, &op, nullptr);
                    }
                    HIPCHECK(hipEventRecord(e1, nullptr)); HIPCHECK(hipEventSynchronize(e1));
                    HIPCHECK(hipEventElapsedTime(&ms_e, e0, e1));
                    printf("TIME %-28s fused_ex with bias_cols=%d and gated_images=%d: %.2f us/call\n", c.name, c.bias_cols, B / 2, ms_e * 1e3 / iters);
                    op.gated_images = 0;
                }
                if (!ccols.empty()) {      // the compact form of the same map
                    const int R = (int)ccols.size();
                    std::vector<float> comp((size_t)N * R);
                    for (int n = 0; n < N; ++n) for (int r = 0; r < R; ++r) comp[(size_t)n * R + r] = bias[(size_t)n * M + ccols[r]];
                    std::vector<int32_t> ci(ccols.begin(), ccols.end());
                    float *dcompact = dalloc<float>(comp.size()); int32_t *dcidx = dalloc<int32_t>(R);
                    HIPCHECK(hipMemcpy(dcompact, comp.data(), comp.size() * 4, hipMemcpyHostToDevice));
                    HIPCHECK(hipMemcpy(dcidx, ci.data(), R * 4, hipMemcpyHostToDevice));
                    op.bias_compact = dcompact; op.col_idx = dcidx; op.R = R; op.compact_stride[1] = R;
                    for (int i = 0; i < 5 + iters; ++i) {
                        if (i == 5) HIPCHECK(hipEventRecord(e0, nullptr));
                        pww_cross_attn_fwd_fused_ex(dq, dk, dv, o2, dbias, PWW_STAT_MAX, 0.37f, dgate, &d, nullptr, dsync, sync_bytes, fws, fws_bytes, &op, nullptr);
                    }
                    HIPCHECK(hipEventRecord(e1, nullptr)); HIPCHECK(hipEventSynchronize(e1));
                    HIPCHECK(hipEventElapsedTime(&ms_e, e0, e1));
                    printf("TIME %-28s fused_ex with the compact form (R=%d, bias_cols=%d): %.2f us/call\n", c.name, R, c.bias_cols, ms_e * 1e3 / iters);
                    (void)hipFree(dcompact); (void)hipFree(dcidx);
                }
            }
        }
        for (void *ptr : {(void *)fws, (void *)dsync, (void *)fstats, (void *)o1, (void *)o2, (void *)dgate}) (void)hipFree(ptr);
    }

#endif  // PWW_EXPERIMENTS

    // fp64 reference on sampled rows: O = softmax((QK^T + c*bias) * scale) V
    double max_err = 0, max_ref = 0; long nchk = 0; int nan_count = 0;
    std::vector<double> logit(M), ref(D);
    for (int b = 0; b < B; ++b) for (int h = 0; h < H; ++h) for (int n = 0; n < N; ++n) {
        const bool sample = (n % c.row_step == 0) || n == N - 1 || n == (N > 33 ? 33 : 0);
        if (!sample) continue;
        const float *qr = &qf[((size_t)b * N + n) * C + h * D];
        double mx = -1e300;
        for (int m = 0; m < M; ++m) {
            const float *kr = &kf[((size_t)b * M + m) * C + h * D];
            double s = 0; for (int x = 0; x < D; ++x) s += (double)qr[x] * kr[x];
            double bv = 0;
            if (c.bias_mode == 1) bv = (double)coeff[b] * bias[(size_t)n * M + m];
            else if (c.bias_mode == 2) bv = bias[(((size_t)b * H + h) * N + n) * M + m];
            logit[m] = (s + bv) * (double)d.scale; mx = std::max(mx, logit[m]);
        }
        double den = 0; std::fill(ref.begin(), ref.end(), 0.0);
        for (int m = 0; m < M; ++m) {
            const double pr = exp(logit[m] - mx); den += pr;
            const float *vr = &vf[((size_t)b * M + m) * C + h * D];
            for (int x = 0; x < D; ++x) ref[x] += pr * vr[x];
        }
        for (int x = 0; x < D; ++x) {
            const double r = ref[x] / den;
            const float g = from_t(out[((size_t)b * N + n) * C + h * D + x], c.dtype);
            if (!(g == g)) nan_count++;
            max_err = std::max(max_err, fabs((double)g - r)); max_ref = std::max(max_ref, fabs(r)); nchk++;
        }
    }
    const double tol = (c.dtype == PWW_DTYPE_F16 ? 2e-3 : 1.6e-2) * max_ref;
    const bool ok_attn = nan_count == 0 && max_err <= tol;

    // statistics reference over ALL rows (float accumulate of products is what the MFMA does; compare loosely)
    bool ok_stats = true; double stat_err = 0;
    {
        for (int b = 0; b < B; ++b) {
            double smax = -1e300, smin = 1e300, ssum = 0, ssq = 0;
            const long total = (long)H * N * M;
            const int rstep = total > 40000000L ? 16 : 1;   // subsample rows for the big shapes (max/min then compared one-sided)
            for (int h = 0; h < H; ++h) for (int n = 0; n < N; n += rstep) {
                const float *qr = &qf[((size_t)b * N + n) * C + h * D];
                for (int m = 0; m < M; ++m) {
                    const float *kr = &kf[((size_t)b * M + m) * C + h * D];
                    double s = 0; for (int x = 0; x < D; ++x) s += (double)qr[x] * kr[x];
                    smax = std::max(smax, s); smin = std::min(smin, s); ssum += s; ssq += s * s;
                }
            }
            const double *g = &stats[4 * b];
            if (rstep == 1) {
                const double mag = std::max(fabs(smax), fabs(smin)) + 1e-6;
                double e = std::max(fabs(g[0] - smax), fabs(g[1] - smin)) / mag;
                // mean and (unbiased) std, the quantities weight_function derives from sum / sumsq
                const double mean_r = ssum / total, var_r = total > 1 ? (ssq - ssum * ssum / total) / (total - 1) : 0.0;
                const double mean_g = g[2] / total, var_g = total > 1 ? (g[3] - g[2] * g[2] / total) / (total - 1) : 0.0;
                const double sd = sqrt(std::max(var_r, 0.0)) + 1e-6;
                e = std::max(e, fabs(mean_g - mean_r) / (sd + fabs(mean_r)));
                e = std::max(e, fabs(sqrt(std::max(var_g, 0.0)) - sqrt(std::max(var_r, 0.0))) / sd);
                stat_err = std::max(stat_err, e);
                if (e > 2e-3) ok_stats = false;
            } else {
                if (g[0] < smax - 1e-2 * fabs(smax) || g[1] > smin + 1e-2 * fabs(smin)) ok_stats = false;
            }
        }
    }
    printf("%s %-28s dtype=%s B=%d H=%d N=%d M=%d D=%d bias=%d | attn max_err=%.3e (tol %.3e, max|O|=%.3f, nan=%d, n=%ld) | stats rel_err=%.2e %s\n",
           (ok_attn && ok_stats) ? "PASS" : "FAIL", c.name, c.dtype == PWW_DTYPE_F16 ? "f16" : "bf16", B, H, N, M, D, c.bias_mode,
           max_err, tol, max_ref, nan_count, nchk, stat_err, ok_stats ? "ok" : "BAD");
    if (!(ok_attn && ok_stats)) g_fail++;

    if (timing && g_timeline)
        timeline_report(c.name, "attention kernel (stamps: 0 entry, 1 first stage staged, 2 key loop done, 3 outputs stored)", [&]() {
            c.bias_mode ? pww_cross_attn_fwd(dq, dk, dv, dout, dbias, dcoeff, &d, nullptr) : pww_self_attn_fwd(dq, dk, dv, dout, &d, nullptr); });
    if (timing) {
        hipEvent_t e0, e1; HIPCHECK(hipEventCreate(&e0)); HIPCHECK(hipEventCreate(&e1));
        const int iters = 50;
        for (int i = 0; i < 5; ++i) c.bias_mode ? pww_cross_attn_fwd(dq, dk, dv, dout, dbias, dcoeff, &d, nullptr) : pww_self_attn_fwd(dq, dk, dv, dout, &d, nullptr);
        HIPCHECK(hipEventRecord(e0, nullptr));
        for (int i = 0; i < iters; ++i) c.bias_mode ? pww_cross_attn_fwd(dq, dk, dv, dout, dbias, dcoeff, &d, nullptr) : pww_self_attn_fwd(dq, dk, dv, dout, &d, nullptr);
        HIPCHECK(hipEventRecord(e1, nullptr)); HIPCHECK(hipEventSynchronize(e1));
        float ms = 0; HIPCHECK(hipEventElapsedTime(&ms, e0, e1));
        const double us = ms * 1e3 / iters, flops = 4.0 * B * H * (double)N * M * D;
        HIPCHECK(hipEventRecord(e0, nullptr));
        for (int i = 0; i < iters; ++i) pww_qk_reduce(dq, dk, &d, dstats, dws, ws_bytes, nullptr);
        HIPCHECK(hipEventRecord(e1, nullptr)); HIPCHECK(hipEventSynchronize(e1));
        float ms2 = 0; HIPCHECK(hipEventElapsedTime(&ms2, e0, e1));
        // kernel-only duration of one more launch (pww_profile_*: the dispatch's own start / end timestamps)
        float kus = -1.f;
        const int slot = pww_profile_arm();
        c.bias_mode ? pww_cross_attn_fwd(dq, dk, dv, dout, dbias, dcoeff, &d, nullptr) : pww_self_attn_fwd(dq, dk, dv, dout, &d, nullptr);
        if (slot < 0 || pww_profile_elapsed_us(slot, &kus) != PWW_OK || !(kus > 0.f && kus < 2.f * us + 20.f)) {
            printf("FAIL %-28s pww_profile_*: slot %d, %.2f us (event-timed loop %.2f us): %s\n", c.name, slot, kus, us, pww_last_error());
            g_fail++;
        }
        pww_profile_reset();
        printf("TIME %-28s attn %.2f us/call (kernel-only, single launch: %.2f us)  %.1f TFLOP/s (algorithmic 4BHNMD) | qk_reduce %.2f us/call\n", c.name, us, kus,
               flops / us * 1e-6, ms2 * 1e3 / iters);
    }
    for (void *ptr : {(void *)dws, (void *)dq, (void *)dk, (void *)dv, (void *)dout, (void *)dstats, (void *)dbias, (void *)dcoeff})
        if (ptr) (void)hipFree(ptr);
}

// ---- mask build ------------------------------------------------------------------------------
static int round_half_up_div(int a, int r) { return (int)floor((double)a / r + 0.5); }

static void check_mask() {
    const int H = 512, W = 384, T = 77, R = 5;
    std::vector<uint8_t> rgb((size_t)H * W * 3);
    pww_region_t regs[R] = {{0, 0, 0, 0, 1.0f}, {255, 255, 255, 0, 1.0f}, {13, 255, 0, 0, 1.5f}, {90, 206, 255, 0, 0.2f}, {74, 18, 1, 0, 0.2f}};
    for (int y = 0; y < H; ++y) for (int x = 0; x < W; ++x) {
        int r = ((x / 37) + (y / 53) * 3) % (R + 1);   // region R = unmatched colour
        uint8_t *p = &rgb[((size_t)y * W + x) * 3];
        if (r < R) { p[0] = regs[r].r; p[1] = regs[r].g; p[2] = regs[r].b; } else { p[0] = 1; p[1] = 2; p[2] = 3; }
    }
    // columns: region 0 -> {5}, region 1 -> {7,8}, region 2 -> {9, 7 (overlap)}, region 3 -> {13}, region 4 -> {13 again twice}
    std::vector<std::vector<int>> cols(T);
    cols[5] = {0}; cols[7] = {1, 2}; cols[8] = {1}; cols[9] = {2}; cols[13] = {3, 4, 4};
    std::vector<int32_t> col_ptr(T + 1, 0), col_reg;
    for (int t = 0; t < T; ++t) { col_ptr[t] = (int)col_reg.size(); for (int r : cols[t]) col_reg.push_back(r); }
    col_ptr[T] = (int)col_reg.size();
    uint8_t *drgb = dalloc<uint8_t>(rgb.size()); pww_region_t *dregs = dalloc<pww_region_t>(R);
    int32_t *dptr = dalloc<int32_t>(T + 1), *dreg = dalloc<int32_t>(col_reg.size() + 1);
    HIPCHECK(hipMemcpy(drgb, rgb.data(), rgb.size(), hipMemcpyHostToDevice));
    HIPCHECK(hipMemcpy(dregs, regs, sizeof(regs), hipMemcpyHostToDevice));
    HIPCHECK(hipMemcpy(dptr, col_ptr.data(), (T + 1) * 4, hipMemcpyHostToDevice));
    HIPCHECK(hipMemcpy(dreg, col_reg.data(), col_reg.size() * 4, hipMemcpyHostToDevice));
    const int ratios[4] = {8, 16, 32, 64};
    float *douts[4]; size_t npix[4]; int Hr[4], Wr[4];
    for (int i = 0; i < 4; ++i) { Hr[i] = round_half_up_div(H, ratios[i]); Wr[i] = round_half_up_div(W, ratios[i]); npix[i] = (size_t)Hr[i] * Wr[i]; douts[i] = dalloc<float>(npix[i] * T); HIPCHECK(hipMemset(douts[i], 0xff, npix[i] * T * 4)); }
    int rc = pww_mask_build(drgb, H, W, dregs, R, dptr, dreg, T, douts[0], douts[1], douts[2], douts[3], nullptr);
    if (rc) { printf("FAIL mask_build rc=%d err=%s\n", rc, pww_last_error()); g_fail++; return; }
    HIPCHECK(hipDeviceSynchronize());
    for (int i = 0; i < 4; ++i) {
        std::vector<float> got(npix[i] * T); HIPCHECK(hipMemcpy(got.data(), douts[i], got.size() * 4, hipMemcpyDeviceToHost));
        const float sy = Hr[i] > 1 ? (float)(H - 1) / (float)(Hr[i] - 1) : 0.f, sx = Wr[i] > 1 ? (float)(W - 1) / (float)(Wr[i] - 1) : 0.f;
        double max_err = 0; long mism = 0; double checksum = 0;
        for (int oy = 0; oy < Hr[i]; ++oy) for (int ox = 0; ox < Wr[i]; ++ox) {
            volatile float fy = sy * (float)oy, fx = sx * (float)ox;
            const int y0 = (int)fy, x0 = (int)fx, y1 = y0 + (y0 < H - 1), x1 = x0 + (x0 < W - 1);
            volatile float ly = fy - (float)y0, lx = fx - (float)x0; volatile float hy = 1.f - ly, hx = 1.f - lx;
            float val[R];
            for (int r = 0; r < R; ++r) {
                auto tap = [&](int y, int x) { const uint8_t *p = &rgb[((size_t)y * W + x) * 3]; return (p[0] == regs[r].r && p[1] == regs[r].g && p[2] == regs[r].b) ? regs[r].strength : 0.f; };
                volatile float a = tap(y0, x0) * hx, b2 = tap(y0, x1) * lx, c2 = tap(y1, x0) * hx, d2 = tap(y1, x1) * lx;
                volatile float top = a + b2, bot = c2 + d2; volatile float t1 = top * hy, t2 = bot * ly; val[r] = t1 + t2;
            }
            for (int t = 0; t < T; ++t) {
                volatile float acc = 0.f; for (int r : cols[t]) acc = acc + val[r];
                const float g = got[((size_t)oy * Wr[i] + ox) * T + t];
                if (g != acc) mism++; max_err = std::max(max_err, (double)fabsf(g - acc)); checksum += g;
            }
        }
        const bool ok = mism == 0;
        printf("%s mask_build ratio=%d out=%dx%d bit-mismatches=%ld max_err=%.3e checksum=%.4f\n", ok ? "PASS" : "FAIL", ratios[i], Hr[i], Wr[i], mism, max_err, checksum);
        if (!ok) g_fail++;
    }
}

static void check_cfg() {
    const long n = 4 * 64 * 64 + 3;
    std::vector<uint16_t> c(n), u(n); std::vector<float> ref(n);
    for (long i = 0; i < n; ++i) { c[i] = to_bf16(rng_normal()); u[i] = to_bf16(rng_normal()); volatile float d = from_bf16(c[i]) - from_bf16(u[i]); volatile float m = 7.5f * d; ref[i] = from_bf16(u[i]) + m; }
    uint16_t *dc = dalloc<uint16_t>(n), *du = dalloc<uint16_t>(n); float *dout = dalloc<float>(n);
    HIPCHECK(hipMemcpy(dc, c.data(), n * 2, hipMemcpyHostToDevice)); HIPCHECK(hipMemcpy(du, u.data(), n * 2, hipMemcpyHostToDevice));
    int rc = pww_cfg_combine(dc, du, 7.5f, dout, n, PWW_DTYPE_BF16, nullptr);
    HIPCHECK(hipDeviceSynchronize());
    std::vector<float> got(n); HIPCHECK(hipMemcpy(got.data(), dout, n * 4, hipMemcpyDeviceToHost));
    long mism = 0; for (long i = 0; i < n; ++i) if (got[i] != ref[i]) mism++;
    printf("%s cfg_combine rc=%d mismatches=%ld\n", (rc == 0 && mism == 0) ? "PASS" : "FAIL", rc, mism);
    if (rc || mism) g_fail++;
}


// ---- pww_qproj_stat + pww_cross_attn_fwd_parts: Q = X W^T with the score statistic's partials out of the GEMM epilogue; the attention
// launch folds them. Checked against (1) an fp64 GEMM on sampled rows, (2) pww_qk_reduce on the Q the kernel wrote, (3) the two-step path
// pww_qk_reduce + pww_cross_attn_fwd_stat on the same Q; timed against pww_cross_attn_fwd_fused_ex (the round-3 launch).
struct QCase { const char *name; int dtype, B, N, Cin, H, D, M; bool shared_k; bool on_request; };

static void run_qproj(const QCase &c, bool timing) {
    const int B = c.B, N = c.N, Cin = c.Cin, H = c.H, D = c.D, M = c.M, C = H * D, Bk = c.shared_k ? 1 : B;
    std::vector<uint16_t> x((size_t)B * N * Cin), w((size_t)C * Cin), k((size_t)Bk * M * C), v((size_t)Bk * M * C);
    std::vector<float> xf(x.size()), wf(w.size()), kf(k.size());
    const float ws = 1.0f / sqrtf((float)Cin);
    for (size_t i = 0; i < x.size(); ++i) { x[i] = to_t(rng_normal(), c.dtype); xf[i] = from_t(x[i], c.dtype); }
    for (size_t i = 0; i < w.size(); ++i) { w[i] = to_t(rng_normal() * ws, c.dtype); wf[i] = from_t(w[i], c.dtype); }
    for (size_t i = 0; i < k.size(); ++i) { k[i] = to_t(rng_normal(), c.dtype); kf[i] = from_t(k[i], c.dtype); }
    for (size_t i = 0; i < v.size(); ++i) v[i] = to_t(rng_normal() + 0.1f * (float)(i % 7), c.dtype);
    pww_qproj_desc_t qd; memset(&qd, 0, sizeof(qd));
    qd.dtype = c.dtype; qd.B = B; qd.N = N; qd.Cin = Cin; qd.H = H; qd.D = D; qd.M = M;
    qd.x_stride[0] = (int64_t)N * Cin; qd.x_stride[1] = Cin; qd.q_stride[0] = (int64_t)N * C; qd.q_stride[1] = C;
    qd.k_stride[0] = c.shared_k ? 0 : (int64_t)M * C; qd.k_stride[1] = C;
    const int nparts = pww_qproj_parts(&qd);
    if (nparts <= 0) { printf("FAIL %-30s pww_qproj_parts = %d (%s)\n", c.name, nparts, pww_last_error()); g_fail++; return; }
    uint16_t *dx = dalloc<uint16_t>(x.size()), *dw = dalloc<uint16_t>(w.size()), *dk = dalloc<uint16_t>(k.size()), *dv = dalloc<uint16_t>(v.size());
    uint16_t *dq = dalloc<uint16_t>((size_t)B * N * C), *o1 = dalloc<uint16_t>((size_t)B * N * C), *o2 = dalloc<uint16_t>((size_t)B * N * C);
    double *dparts = dalloc<double>((size_t)B * nparts * 4), *dstats = dalloc<double>(4 * B), *dfold = dalloc<double>(4 * B);
    HIPCHECK(hipMemcpy(dx, x.data(), x.size() * 2, hipMemcpyHostToDevice)); HIPCHECK(hipMemcpy(dw, w.data(), w.size() * 2, hipMemcpyHostToDevice));
    HIPCHECK(hipMemcpy(dk, k.data(), k.size() * 2, hipMemcpyHostToDevice)); HIPCHECK(hipMemcpy(dv, v.data(), v.size() * 2, hipMemcpyHostToDevice));
    HIPCHECK(hipMemset(dq, 0xff, (size_t)B * N * C * 2)); HIPCHECK(hipMemset(dparts, 0xff, (size_t)B * nparts * 32));
    std::vector<float> gate(B, 1.f); if (B > 1) gate[B - 1] = 0.f;        // last image gated out: Q still written, no partials
    float *dgate = dalloc<float>(B); HIPCHECK(hipMemcpy(dgate, gate.data(), B * 4, hipMemcpyHostToDevice));
    if (g_product_only) {        // --product-only: ONLY the two launches the product issues per cross-attention layer (for PMC passes)
        std::vector<float> g2(B, 1.f); if (B > 1) for (int b = B / 2; b < B; ++b) g2[b] = 0.f;
        HIPCHECK(hipMemcpy(dgate, g2.data(), B * 4, hipMemcpyHostToDevice));
        std::vector<float> bias((size_t)N * M, 0.f);
        for (int n = 0; n < N; ++n) for (int m = 0; m < 32 && m < M; ++m) bias[(size_t)n * M + m] = (rng_uniform() < 0.3f) ? rng_uniform() * 1.5f : 0.f;
        float *dbias = dalloc<float>(bias.size()); HIPCHECK(hipMemcpy(dbias, bias.data(), bias.size() * 4, hipMemcpyHostToDevice));
        pww_attn_desc_t d; memset(&d, 0, sizeof(d));
        d.dtype = c.dtype; d.B = B; d.H = H; d.N = N; d.M = M; d.D = D;
        d.q_stride[0] = (int64_t)N * C; d.q_stride[1] = D; d.q_stride[2] = C;
        d.k_stride[0] = c.shared_k ? 0 : (int64_t)M * C; d.k_stride[1] = D; d.k_stride[2] = C;
        d.v_stride[0] = d.k_stride[0]; d.v_stride[1] = D; d.v_stride[2] = C;
        d.o_stride[0] = (int64_t)N * C; d.o_stride[1] = D; d.o_stride[2] = C;
        d.scale = 1.0f / sqrtf((float)D); d.bias_stride[2] = M; d.bias_stride[3] = 1;
        pww_cross_opts_t op; memset(&op, 0, sizeof(op)); op.size = sizeof(op); op.bias_cols = 32; op.gated_images = B > 1 ? B / 2 : 0;
        int r = 0;
        // the product's route for this layer (pww_hip/attention.py qproj_route): to_q with the statistic in its epilogue where that wins
        // (C = 320, and C = 640 when 160 % D == 0); elsewhere the stock to_q GEMM + pww_qk_parts over the finished Q
        const bool k1_route = C <= 320 || (C <= 640 && 160 % D == 0);
        const int np2 = pww_qk_parts_count(&d);
        double *dp2 = dalloc<double>((size_t)B * std::max(np2, 1) * 4);
        if (!k1_route) r = pww_qproj_stat(dx, dw, dq, dk, dgate, &qd, PWW_STAT_MAX, dparts, (size_t)B * nparts * 32, nullptr);      // (stands in for the stock GEMM: Q once)
        for (int i = 0; i < 20 && !r; ++i) {
            if (k1_route) {
                r = pww_qproj_stat(dx, dw, dq, dk, dgate, &qd, PWW_STAT_MAX, dparts, (size_t)B * nparts * 32, nullptr);
                if (!r) r = pww_cross_attn_fwd_parts(dq, dk, dv, o2, dbias, PWW_STAT_MAX, 0.37f, dgate, &d, dparts, nparts, nullptr, &op, nullptr);
            } else {
                r = pww_qk_parts(dq, dk, dgate, &d, PWW_STAT_MAX, op.gated_images, dp2, (size_t)B * np2 * 32, nullptr);
                if (!r) r = pww_cross_attn_fwd_parts(dq, dk, dv, o2, dbias, PWW_STAT_MAX, 0.37f, dgate, &d, dp2, np2, nullptr, &op, nullptr);
            }
        }
        HIPCHECK(hipDeviceSynchronize());
        (void)hipFree(dp2);
        printf("%s %-30s product-only: 20 x (%s, cross_attn_fwd_parts) rc=%d %s\n", r ? "FAIL" : "PASS", c.name, k1_route ? "qproj_stat" : "qk_parts", r, r ? pww_last_error() : "");
        if (r) g_fail++;
        return;
    }
    int rc = pww_qproj_stat(dx, dw, dq, dk, dgate, &qd, PWW_STAT_ALL, dparts, (size_t)B * nparts * 32, nullptr);
    HIPCHECK(hipDeviceSynchronize());
    if (rc) { printf("FAIL %-30s pww_qproj_stat rc=%d err=%s\n", c.name, rc, pww_last_error()); g_fail++; return; }
    std::vector<uint16_t> q((size_t)B * N * C); std::vector<double> parts((size_t)B * nparts * 4);
    HIPCHECK(hipMemcpy(q.data(), dq, q.size() * 2, hipMemcpyDeviceToHost)); HIPCHECK(hipMemcpy(parts.data(), dparts, parts.size() * 8, hipMemcpyDeviceToHost));
    // (1) Q vs fp64 GEMM: sampled rows, every channel; one rounding of the result to T (2^-9 bf16 / 2^-12 f16 relative) + fp32 accumulation
    double qerr = 0, qmax = 0; long bad = 0, nq = 0;
    const double ulp = c.dtype == PWW_DTYPE_F16 ? 1.0 / 2048 : 1.0 / 256;
    const int rstep = std::max(1, N / 61);
    for (int b = 0; b < B; ++b) for (int n = 0; n < N; ++n) {
        if (n % rstep && n != N - 1) continue;
        const float *xr = &xf[((size_t)b * N + n) * Cin];
        for (int ch = 0; ch < C; ++ch) {
            const float *wr = &wf[(size_t)ch * Cin];
            double s = 0; for (int i = 0; i < Cin; ++i) s += (double)xr[i] * wr[i];
            const double g = from_t(q[((size_t)b * N + n) * C + ch], c.dtype);
            const double e = fabs(g - s);
            qerr = std::max(qerr, e); qmax = std::max(qmax, fabs(s)); ++nq;
            if (!(e <= ulp * fabs(s) + 2e-5)) ++bad;
        }
    }
    // (2) folded partials vs pww_qk_reduce on the Q the kernel wrote
    pww_attn_desc_t d; memset(&d, 0, sizeof(d));
    d.dtype = c.dtype; d.B = B; d.H = H; d.N = N; d.M = M; d.D = D;
    d.q_stride[0] = (int64_t)N * C; d.q_stride[1] = D; d.q_stride[2] = C;
    d.k_stride[0] = c.shared_k ? 0 : (int64_t)M * C; d.k_stride[1] = D; d.k_stride[2] = C;
    d.v_stride[0] = d.k_stride[0]; d.v_stride[1] = D; d.v_stride[2] = C;
    d.o_stride[0] = (int64_t)N * C; d.o_stride[1] = D; d.o_stride[2] = C;
    d.scale = 1.0f / sqrtf((float)D);
    d.bias_stride[0] = 0; d.bias_stride[1] = 0; d.bias_stride[2] = M; d.bias_stride[3] = 1;
    const size_t ws_bytes = pww_workspace_bytes(&d); void *dws = dalloc<char>(ws_bytes + 8);
    rc = pww_qk_reduce(dq, dk, &d, dstats, dws, ws_bytes, nullptr);
    HIPCHECK(hipDeviceSynchronize());
    std::vector<double> stats(4 * B);
    HIPCHECK(hipMemcpy(stats.data(), dstats, stats.size() * 8, hipMemcpyDeviceToHost));
    double serr = 0; bool untouched = true;
    const double cnt = (double)H * N * M;
    for (int b = 0; b < B; ++b) {
        const double *pp = &parts[(size_t)b * nparts * 4];
        if (gate[b] == 0.f) { for (int i = 0; i < nparts * 4; ++i) { uint64_t u; memcpy(&u, &pp[i], 8); untouched = untouched && u == ~0ull; } continue; }
        double f[4] = {-1e300, 1e300, 0, 0};
        for (int i = 0; i < nparts; ++i) { f[0] = std::max(f[0], pp[i * 4]); f[1] = std::min(f[1], pp[i * 4 + 1]); f[2] += pp[i * 4 + 2]; f[3] += pp[i * 4 + 3]; }
        const double *g = &stats[4 * b];
        const double mag = std::max(fabs(g[0]), fabs(g[1])) + 1e-9;
        serr = std::max(serr, std::max(fabs(f[0] - g[0]), fabs(f[1] - g[1])) / mag);
        const double sd = sqrt(std::max((g[3] - g[2] * g[2] / cnt) / (cnt - 1), 0.0)) + 1e-9;
        serr = std::max(serr, fabs(f[2] - g[2]) / cnt / sd);                        // the means, in units of the std
        serr = std::max(serr, fabs(f[3] - g[3]) / fabs(g[3]));
    }
    const bool ok_q = rc == 0 && bad == 0, ok_s = serr <= 1e-6 && untouched;
    printf("%s %-30s dtype=%s B=%d N=%d Cin=%d H=%d D=%d M=%d parts/image=%d | Q max_err=%.3e (max|Q|=%.2f, %ld of %ld beyond 1 rounding) | folded partials vs pww_qk_reduce rel_err=%.2e %s%s\n",
           ok_q && ok_s ? "PASS" : "FAIL", c.name, c.dtype == PWW_DTYPE_F16 ? "f16" : "bf16", B, N, Cin, H, D, M, nparts, qerr, qmax, bad, nq, serr, ok_s ? "ok" : "BAD",
           untouched ? "" : " (partials of a gated-out image were written)");
    if (!(ok_q && ok_s)) g_fail++;
    // (2b) pww_qk_parts over the finished Q (the route of the layers whose to_q stays the stock GEMM): folded partials vs pww_qk_reduce
    const int nparts2 = pww_qk_parts_count(&d);
    double *dparts2 = dalloc<double>((size_t)B * std::max(nparts2, 1) * 4);
    {
        HIPCHECK(hipMemset(dparts2, 0xff, (size_t)B * std::max(nparts2, 1) * 32));
        const int r = nparts2 > 0 ? pww_qk_parts(dq, dk, dgate, &d, PWW_STAT_ALL, B > 1 ? B - 1 : 0, dparts2, (size_t)B * nparts2 * 32, nullptr) : -1;
        HIPCHECK(hipDeviceSynchronize());
        std::vector<double> p2((size_t)B * std::max(nparts2, 1) * 4);
        HIPCHECK(hipMemcpy(p2.data(), dparts2, p2.size() * 8, hipMemcpyDeviceToHost));
        double e2 = 0; bool untouched2 = true;
        for (int b = 0; b < B && r == 0; ++b) {
            const double *pp = &p2[(size_t)b * nparts2 * 4];
            if (gate[b] == 0.f) { for (int i = 0; i < nparts2 * 4; ++i) { uint64_t u; memcpy(&u, &pp[i], 8); untouched2 = untouched2 && u == ~0ull; } continue; }
            double f[4] = {-1e300, 1e300, 0, 0};
            for (int i = 0; i < nparts2; ++i) { f[0] = std::max(f[0], pp[i * 4]); f[1] = std::min(f[1], pp[i * 4 + 1]); f[2] += pp[i * 4 + 2]; f[3] += pp[i * 4 + 3]; }
            const double *g = &stats[4 * b];
            const double mag = std::max(fabs(g[0]), fabs(g[1])) + 1e-9;
            e2 = std::max(e2, std::max(fabs(f[0] - g[0]), fabs(f[1] - g[1])) / mag);
            const double sd = sqrt(std::max((g[3] - g[2] * g[2] / cnt) / (cnt - 1), 0.0)) + 1e-9;
            e2 = std::max(e2, fabs(f[2] - g[2]) / cnt / sd);
            e2 = std::max(e2, fabs(f[3] - g[3]) / fabs(g[3]));
        }
        const bool ok2 = r == 0 && e2 <= 1e-6 && untouched2;
        printf("%s %-30s pww_qk_parts: %d partials / image, folded vs pww_qk_reduce rel_err=%.2e%s (rc %d %s)\n", ok2 ? "PASS" : "FAIL", c.name, nparts2, e2,
               untouched2 ? "" : " (partials of a gated-out image were written)", r, r ? pww_last_error() : "");
        if (!ok2) g_fail++;
    }
    // (3) attention from the partials vs pww_cross_attn_fwd_stat from pww_qk_reduce's statistics, same Q
    std::vector<float> bias((size_t)N * M);
    for (auto &bv : bias) bv = (rng_uniform() < 0.3f) ? rng_uniform() * 1.5f : 0.f;
    for (int n = 0; n < N; ++n) for (int m = 32; m < M; ++m) bias[(size_t)n * M + m] = 0.f;
    float *dbias = dalloc<float>(bias.size()); HIPCHECK(hipMemcpy(dbias, bias.data(), bias.size() * 4, hipMemcpyHostToDevice));
    pww_cross_opts_t op; memset(&op, 0, sizeof(op)); op.size = sizeof(op); op.bias_cols = 32; op.gated_images = B > 1 ? B - 1 : 0;
    for (int kind : {PWW_STAT_MAX, PWW_STAT_STD, PWW_STAT_ABSMAX, PWW_STAT_MEAN, PWW_STAT_MIN}) {
        HIPCHECK(hipMemset(o1, 0xff, (size_t)B * N * C * 2)); HIPCHECK(hipMemset(o2, 0xee, (size_t)B * N * C * 2)); HIPCHECK(hipMemset(dfold, 0, 4 * B * 8));
        int r1 = pww_cross_attn_fwd_stat(dq, dk, dv, o1, dbias, dstats, kind, cnt, 0.37f, dgate, &d, nullptr);
        int r2 = pww_cross_attn_fwd_parts(dq, dk, dv, o2, dbias, kind, 0.37f, dgate, &d, dparts, nparts, dfold, &op, nullptr);
        HIPCHECK(hipDeviceSynchronize());
        std::vector<uint16_t> h1((size_t)B * N * C), h2(h1.size());
        HIPCHECK(hipMemcpy(h1.data(), o1, h1.size() * 2, hipMemcpyDeviceToHost)); HIPCHECK(hipMemcpy(h2.data(), o2, h2.size() * 2, hipMemcpyDeviceToHost));
        double dmax = 0, omax = 0; long ndiff = 0, nan = 0;
        for (size_t i = 0; i < h1.size(); ++i) {
            const double a = from_t(h1[i], c.dtype), bq = from_t(h2[i], c.dtype);
            if (!(bq == bq)) ++nan;
            dmax = std::max(dmax, fabs(a - bq)); omax = std::max(omax, fabs(a)); ndiff += h1[i] != h2[i];
        }
        const bool ok = r1 == 0 && r2 == 0 && nan == 0 && dmax <= 4 * ulp * omax;
        printf("%s %-30s parts-attention kind=%d: max diff %.3e vs two-step path (max|O| %.2f, %ld of %zu elements differ, nan=%ld, rc %d %d%s%s)\n", ok ? "PASS" : "FAIL",
               c.name, kind, dmax, omax, ndiff, h1.size(), nan, r1, r2, r2 ? " " : "", r2 ? pww_last_error() : "");
        if (!ok) g_fail++;
        // the same from pww_qk_parts' partials (formed for THIS statistic only), and the folded statistics it hands back
        HIPCHECK(hipMemset(o2, 0xee, (size_t)B * N * C * 2)); HIPCHECK(hipMemset(dfold, 0, 4 * B * 8));
        int r3 = pww_qk_parts(dq, dk, dgate, &d, kind, B > 1 ? B - 1 : 0, dparts2, (size_t)B * nparts2 * 32, nullptr);
        if (!r3) r3 = pww_cross_attn_fwd_parts(dq, dk, dv, o2, dbias, kind, 0.37f, dgate, &d, dparts2, nparts2, nullptr, &op, nullptr);
        HIPCHECK(hipDeviceSynchronize());
        HIPCHECK(hipMemcpy(h2.data(), o2, h2.size() * 2, hipMemcpyDeviceToHost));
        double dmax3 = 0; long nan3 = 0;
        for (size_t i = 0; i < h1.size(); ++i) {
            const double a = from_t(h1[i], c.dtype), bq = from_t(h2[i], c.dtype);
            if (!(bq == bq)) ++nan3;
            dmax3 = std::max(dmax3, fabs(a - bq));
        }
        const bool ok3 = r3 == 0 && nan3 == 0 && dmax3 <= 4 * ulp * omax;
        printf("%s %-30s qk_parts + parts-attention kind=%d: max diff %.3e vs two-step path (nan=%ld, rc %d %s)\n", ok3 ? "PASS" : "FAIL", c.name, kind, dmax3, nan3, r3, r3 ? pww_last_error() : "");
        if (!ok3) g_fail++;
    }
#if PWW_EXPERIMENTS
    // (4) the same attention WITH the layer's output projection in the launch (pww_cross_attn_fwd_parts_out, the C = 320 layers): against the
    // fp64 projection of the O the two-launch route just stored (sampled rows, every output channel) -- one rounding of the storage type
    if (pww_cross_attn_out_supported(&d, C, op.bias_cols)) {
        std::vector<uint16_t> wo((size_t)C * C), wob(C);
        std::vector<float> wof(wo.size()), wobf(C);
        for (size_t i = 0; i < wo.size(); ++i) { wo[i] = to_t(rng_normal() / sqrtf((float)C), c.dtype); wof[i] = from_t(wo[i], c.dtype); }
        for (int i = 0; i < C; ++i) { wob[i] = to_t(rng_normal() * 0.1f, c.dtype); wobf[i] = from_t(wob[i], c.dtype); }
        uint16_t *dwo = dalloc<uint16_t>(wo.size()), *dwob = dalloc<uint16_t>(C);
        HIPCHECK(hipMemcpy(dwo, wo.data(), wo.size() * 2, hipMemcpyHostToDevice)); HIPCHECK(hipMemcpy(dwob, wob.data(), C * 2, hipMemcpyHostToDevice));
        HIPCHECK(hipMemset(o1, 0xff, (size_t)B * N * C * 2)); HIPCHECK(hipMemset(o2, 0xee, (size_t)B * N * C * 2));
        int r1 = pww_cross_attn_fwd_parts(dq, dk, dv, o2, dbias, PWW_STAT_MAX, 0.37f, dgate, &d, dparts, nparts, nullptr, &op, nullptr);
        int r2 = pww_cross_attn_fwd_parts_out(dq, dk, dv, o1, dbias, PWW_STAT_MAX, 0.37f, dgate, &d, dparts, nparts, nullptr, &op, dwo, dwob, nullptr, nullptr, nullptr);
        HIPCHECK(hipDeviceSynchronize());
        std::vector<uint16_t> ho((size_t)B * N * C), hout(ho.size());
        HIPCHECK(hipMemcpy(ho.data(), o2, ho.size() * 2, hipMemcpyDeviceToHost)); HIPCHECK(hipMemcpy(hout.data(), o1, hout.size() * 2, hipMemcpyDeviceToHost));
        double pmax = 0; std::vector<double> pref; std::vector<size_t> pidx;
        for (int b = 0; b < B; ++b) for (int n = 0; n < N; ++n) {
            if (n % rstep && n != N - 1) continue;
            for (int ch = 0; ch < C; ++ch) {
                double sacc = wobf[ch];
                for (int i = 0; i < C; ++i) sacc += (double)from_t(ho[((size_t)b * N + n) * C + i], c.dtype) * wof[(size_t)ch * C + i];
                pref.push_back(sacc); pidx.push_back(((size_t)b * N + n) * C + ch); pmax = std::max(pmax, fabs(sacc));
            }
        }
        double perr = 0; long pnan = 0;
        for (size_t i = 0; i < pref.size(); ++i) {
            const double g = from_t(hout[pidx[i]], c.dtype);
            if (!(g == g)) ++pnan;
            perr = std::max(perr, fabs(g - pref[i]) / (fabs(pref[i]) + pmax / 64));
        }
        const bool okp = r1 == 0 && r2 == 0 && pnan == 0 && perr <= 1.5 * ulp;
        printf("%s %-30s attention + to_out in one launch: max err %.2f half-spacings vs the fp64 projection of the two-launch O (%zu samples, nan=%ld, rc %d %d%s%s)\n",
               okp ? "PASS" : "FAIL", c.name, perr / ulp, pref.size(), pnan, r1, r2, r2 ? " " : "", r2 ? pww_last_error() : "");
        if (!okp) g_fail++;
        (void)hipFree(dwo); (void)hipFree(dwob);
    }
#endif  // PWW_EXPERIMENTS
    if (timing && g_timeline) {
        std::vector<float> g2(B, 1.f); if (B > 1) for (int b = B / 2; b < B; ++b) g2[b] = 0.f;
        HIPCHECK(hipMemcpy(dgate, g2.data(), B * 4, hipMemcpyHostToDevice));
        timeline_report(c.name, "qproj_stat (stamps: 0 entry, 1 K tile + chunk 0 staged, 2 contraction done, 3 Q tile in LDS, 4 statistic partials written, 5 tile stores issued)", [&]() {
            pww_qproj_stat(dx, dw, dq, dk, dgate, &qd, PWW_STAT_MAX, dparts, (size_t)B * nparts * 32, nullptr); });
        op.gated_images = B > 1 ? B / 2 : 0;
        timeline_report(c.name, "cross_attn_fwd_parts (stamps: 0 entry, 1 K/V staged, 3 partials folded, 4 outputs stored, 5 exit)", [&]() {
            pww_cross_attn_fwd_parts(dq, dk, dv, o2, dbias, PWW_STAT_MAX, 0.37f, dgate, &d, dparts, nparts, nullptr, &op, nullptr); });
    }
    if (timing) {
        hipEvent_t e0, e1; HIPCHECK(hipEventCreate(&e0)); HIPCHECK(hipEventCreate(&e1));
        const int iters = 50;
        std::vector<float> g2(B, 1.f); if (B > 1) for (int b = B / 2; b < B; ++b) g2[b] = 0.f;       // a CFG-folded batch
        HIPCHECK(hipMemcpy(dgate, g2.data(), B * 4, hipMemcpyHostToDevice));
        op.gated_images = B > 1 ? B / 2 : 0;
#if PWW_EXPERIMENTS
        const size_t fws_bytes = pww_cross_fused_workspace_bytes(&d), sync_bytes = pww_cross_fused_state_bytes(&d);
        void *fws = dalloc<char>(fws_bytes + 8); unsigned *dsync = dalloc<unsigned>(sync_bytes / 4 + 1);
        HIPCHECK(hipMemset(dsync, 0, sync_bytes));
#else
        void *fws = nullptr; unsigned *dsync = nullptr;
#endif
        float ms[6] = {0, 0, 0, 0, 0, 0};
        for (int pass = 0; pass < 6; ++pass) {
            for (int i = 0; i < 5 + iters; ++i) {
                if (i == 5) HIPCHECK(hipEventRecord(e0, nullptr));
                if (pass == 0) pww_qproj_stat(dx, dw, dq, dk, dgate, &qd, PWW_STAT_MAX, dparts, (size_t)B * nparts * 32, nullptr);
                else if (pass == 1) pww_cross_attn_fwd_parts(dq, dk, dv, o2, dbias, PWW_STAT_MAX, 0.37f, dgate, &d, dparts, nparts, nullptr, &op, nullptr);
#if PWW_EXPERIMENTS
                else if (pass == 2) pww_cross_attn_fwd_fused_ex(dq, dk, dv, o1, dbias, PWW_STAT_MAX, 0.37f, dgate, &d, nullptr, dsync, sync_bytes, fws, fws_bytes, &op, nullptr);
#else
                else if (pass == 2) { }       // (round 3's fused launch: attn_check_experiments)
#endif
                else if (pass == 3) { pww_qproj_stat(dx, dw, dq, dk, dgate, &qd, PWW_STAT_MAX, dparts, (size_t)B * nparts * 32, nullptr);
                       pww_cross_attn_fwd_parts(dq, dk, dv, o2, dbias, PWW_STAT_MAX, 0.37f, dgate, &d, dparts, nparts, nullptr, &op, nullptr); }
                else if (pass == 4) pww_qk_parts(dq, dk, dgate, &d, PWW_STAT_MAX, op.gated_images, dparts2, (size_t)B * nparts2 * 32, nullptr);
                else { pww_qk_parts(dq, dk, dgate, &d, PWW_STAT_MAX, op.gated_images, dparts2, (size_t)B * nparts2 * 32, nullptr);
                       pww_cross_attn_fwd_parts(dq, dk, dv, o2, dbias, PWW_STAT_MAX, 0.37f, dgate, &d, dparts2, nparts2, nullptr, &op, nullptr); }
            }
            HIPCHECK(hipEventRecord(e1, nullptr)); HIPCHECK(hipEventSynchronize(e1));
            HIPCHECK(hipEventElapsedTime(&ms[pass], e0, e1));
        }
        float kq = -1.f, ka = -1.f;
        int slot = pww_profile_arm();
        pww_qproj_stat(dx, dw, dq, dk, dgate, &qd, PWW_STAT_MAX, dparts, (size_t)B * nparts * 32, nullptr);
        if (slot >= 0) pww_profile_elapsed_us(slot, &kq);
        slot = pww_profile_arm();
        pww_cross_attn_fwd_parts(dq, dk, dv, o2, dbias, PWW_STAT_MAX, 0.37f, dgate, &d, dparts, nparts, nullptr, &op, nullptr);
        if (slot >= 0) pww_profile_elapsed_us(slot, &ka);
        pww_profile_reset();
        const double gemm_flops = 2.0 * B * N * (double)C * Cin, bytes = 2.0 * ((double)B * N * Cin + (double)B * N * C + (double)C * Cin);
        printf("TIME %-30s qproj_stat %.2f us (kernel-only %.2f; %.1f TFLOP/s, %.0f GB/s algorithmic) | parts-attention %.2f us (kernel-only %.2f) | both back to back %.2f us | "
               "round-3 fused launch alone (needs its own to_q GEMM before it) %.2f us | qk_parts %.2f us, qk_parts + parts-attention %.2f us (needs its own to_q GEMM before it)\n", c.name, ms[0] * 1e3 / iters, kq, gemm_flops / (ms[0] * 1e3 / iters) * 1e-6,
               bytes / (ms[0] * 1e3 / iters) * 1e-3, ms[1] * 1e3 / iters, ka, ms[3] * 1e3 / iters, ms[2] * 1e3 / iters, ms[4] * 1e3 / iters, ms[5] * 1e3 / iters);
        (void)hipFree(fws); (void)hipFree(dsync);
    }
    for (void *ptr : {(void *)dx, (void *)dw, (void *)dk, (void *)dv, (void *)dq, (void *)o1, (void *)o2, (void *)dparts, (void *)dstats, (void *)dfold, (void *)dgate, (void *)dws, (void *)dbias, (void *)dparts2})
        (void)hipFree(ptr);
}

static void check_qproj(const char *only, const char *match, bool quick) {
    const QCase cases[] = {
        // the 16 cross-attention layers of SD1.5 at 512 x 512: 2 folded rows (config 2) and 16 (configs 3 / 4)
        {"qproj_sd15_n4096_b2", PWW_DTYPE_BF16, 2, 4096, 320, 8, 40, 77, false, false},
        {"qproj_sd15_n1024_b2", PWW_DTYPE_BF16, 2, 1024, 640, 8, 80, 77, false, false},
        {"qproj_sd15_n256_b2", PWW_DTYPE_BF16, 2, 256, 1280, 8, 160, 77, false, false},
        {"qproj_sd15_n64_b2", PWW_DTYPE_BF16, 2, 64, 1280, 8, 160, 77, false, false},
        {"qproj_sd15_n4096_b16_f16", PWW_DTYPE_F16, 16, 4096, 320, 8, 40, 77, false, false},
        {"qproj_sd15_n4096_b16", PWW_DTYPE_BF16, 16, 4096, 320, 8, 40, 77, false, false},
        {"qproj_sd15_n1024_b16", PWW_DTYPE_BF16, 16, 1024, 640, 8, 80, 77, false, false},
        {"qproj_sd15_n256_b16", PWW_DTYPE_F16, 16, 256, 1280, 8, 160, 77, false, false},
        {"qproj_sd15_n64_b16", PWW_DTYPE_BF16, 16, 64, 1280, 8, 160, 77, false, false},
        // SD2.1 at 768 x 768 (head dim 64: TN = 320 = 5 heads), 8 folded rows
        {"qproj_sd21_n9216_b8", PWW_DTYPE_BF16, 8, 9216, 320, 5, 64, 77, false, false},
        {"qproj_sd21_n2304_b8", PWW_DTYPE_BF16, 8, 2304, 640, 10, 64, 77, false, false},
        {"qproj_sd21_n576_b8", PWW_DTYPE_BF16, 8, 576, 1280, 20, 64, 77, false, false},
        {"qproj_sd21_n144_b8", PWW_DTYPE_F16, 8, 144, 1280, 20, 64, 77, false, false},
        // ragged and odd shapes: rows past N, one shared prompt, other key counts, every tile shape
        {"qproj_n3990_b3_shared_k", PWW_DTYPE_F16, 3, 3990, 320, 8, 40, 77, true, false},
        {"qproj_n100_m40_d80", PWW_DTYPE_BF16, 2, 100, 640, 8, 80, 40, false, false},
        {"qproj_n333_m128_d160", PWW_DTYPE_BF16, 1, 333, 1280, 8, 160, 128, false, false},
        {"qproj_n50_m1_d32", PWW_DTYPE_F16, 5, 50, 192, 5, 32, 1, true, false},
        {"qproj_n700_m33_d16_c160", PWW_DTYPE_BF16, 7, 700, 128, 10, 16, 33, false, false},
        {"qproj_n2000_b40_d40", PWW_DTYPE_BF16, 40, 2000, 320, 8, 40, 77, false, false},
    };
    for (const QCase &c : cases) {
        const bool big = (long)c.B * c.N >= 32768;
        if (quick && big) continue;
        if (only && strcmp(only, c.name)) continue;
        if (match && !strstr(c.name, match)) continue;
        run_qproj(c, true);
    }
}

static void check_errors() {
    pww_attn_desc_t d; memset(&d, 0, sizeof(d));
    d.dtype = PWW_DTYPE_F16; d.B = 1; d.H = 1; d.N = 32; d.M = 32; d.D = 36; d.scale = 1.f;
    for (int i = 0; i < 3; ++i) { d.q_stride[i] = d.k_stride[i] = d.v_stride[i] = d.o_stride[i] = 40; }
    uint16_t *p = dalloc<uint16_t>(4096);
    int rc1 = pww_self_attn_fwd(p, p, p, p, &d, nullptr);           // D not a multiple of 8
    d.D = 200; int rc2 = pww_self_attn_fwd(p, p, p, p, &d, nullptr);  // D too large
    d.D = 40; int rc3 = pww_self_attn_fwd(nullptr, p, p, p, &d, nullptr);
    d.dtype = 7; int rc4 = pww_self_attn_fwd(p, p, p, p, &d, nullptr);
    const bool ok = rc1 == PWW_ENOTSUP && rc2 == PWW_ENOTSUP && rc3 == PWW_EINVAL && rc4 == PWW_ENOTSUP && strlen(pww_last_error()) > 0;
    printf("%s error codes: %d %d %d %d last='%s'\n", ok ? "PASS" : "FAIL", rc1, rc2, rc3, rc4, pww_last_error());
    if (!ok) g_fail++;
    (void)hipFree(p);
}

int main(int argc, char **argv) {
    if (argc > 1 && !strcmp(argv[1], "--timeline")) { g_timeline = true; --argc; ++argv; }
    if (argc > 1 && !strcmp(argv[1], "--product-only")) { g_product_only = true; --argc; ++argv; }
    const bool quick = argc > 1 && !strcmp(argv[1], "--quick");
    const char *only = (argc > 2 && !strcmp(argv[1], "--only")) ? argv[2] : nullptr;
    const char *match = (argc > 2 && !strcmp(argv[1], "--match")) ? argv[2] : nullptr;     // substring of the case name
    char arch[64] = ""; int rc = pww_device_arch(arch, sizeof(arch));
    printf("libpww_hip version %d, device arch '%s' (rc=%d)\n", pww_version(), arch, rc);
    std::vector<Case> cases = {
        {"tiny_self_d40", PWW_DTYPE_F16, 1, 1, 32, 32, 40, 0, true, 1, 1.0f},
        {"tiny_self_d40_bf16", PWW_DTYPE_BF16, 1, 1, 32, 32, 40, 0, true, 1, 1.0f},
        {"one_row_one_key", PWW_DTYPE_F16, 1, 2, 1, 1, 64, 0, false, 1, 1.0f},
        {"m64_exact", PWW_DTYPE_F16, 1, 2, 70, 64, 64, 0, false, 1, 1.0f},
        {"m65_ragged", PWW_DTYPE_BF16, 2, 3, 100, 65, 64, 2, false, 1, 1.0f},
        {"sd15_mid_self_n64_d160", PWW_DTYPE_F16, 2, 8, 64, 64, 160, 0, true, 1, 0.5f},
        {"sd15_mid_cross_n64_d160", PWW_DTYPE_BF16, 2, 8, 64, 77, 160, 1, false, 1, 0.5f},
        {"sd15_self_n256_d160", PWW_DTYPE_BF16, 2, 8, 256, 256, 160, 0, true, 3, 0.5f},
        {"sd15_cross_n256_d160", PWW_DTYPE_F16, 1, 8, 256, 77, 160, 1, false, 3, 0.5f},
        {"sd15_self_n1024_d80", PWW_DTYPE_F16, 1, 8, 1024, 1024, 80, 0, true, 17, 0.7f},
        {"sd15_cross_n1024_d80", PWW_DTYPE_BF16, 2, 8, 1024, 77, 80, 1, false, 17, 0.7f},
        {"sd15_self_n4096_d40", PWW_DTYPE_BF16, 1, 8, 4096, 4096, 40, 0, true, 97, 1.0f},
        {"sd15_self_n4096_d40_f16_b2", PWW_DTYPE_F16, 2, 8, 4096, 4096, 40, 0, true, 193, 1.0f},
        {"sd15_self_n4096_d40_bf16_b2", PWW_DTYPE_BF16, 2, 8, 4096, 4096, 40, 0, true, 193, 1.0f},   // the bench's dominant launch (bf16, 2 folded rows)
        {"sd15_cross_n4096_d40", PWW_DTYPE_BF16, 2, 8, 4096, 77, 40, 1, false, 97, 1.0f},
        {"sd15_cross_n4096_d40_b16", PWW_DTYPE_BF16, 16, 8, 4096, 77, 40, 1, false, 397, 1.0f},     // 8 images folded: several query blocks per workgroup
        // bias column bound + compact bias (pww_cross_attn_fwd_fused_ex): the map of a prompt is zero past its last region phrase
        {"sd15_cross_n4096_d40_cols32", PWW_DTYPE_BF16, 2, 8, 4096, 77, 40, 1, false, 97, 1.0f, 0.f, 32, 9},
        {"sd15_cross_n4096_d40_b16_cols32", PWW_DTYPE_BF16, 16, 8, 4096, 77, 40, 1, false, 397, 1.0f, 0.f, 32, 9},
        {"sd15_cross_n4096_f16_b16_cols48", PWW_DTYPE_F16, 16, 8, 4096, 77, 40, 1, false, 397, 1.0f, 0.f, 48, 17},
        {"sd15_cross_n4096_d40_b16_cols16", PWW_DTYPE_BF16, 16, 8, 4096, 77, 40, 1, false, 397, 1.0f, 0.f, 16, 4},     // narrowest tile: 2 LDS-direct copies per thread
        {"cross_n3990_d40_b16_cols32_ragged", PWW_DTYPE_F16, 16, 8, 3990, 77, 40, 1, false, 397, 1.0f, 0.f, 32, 9},      // several blocks per workgroup, last block ragged (rows past N arrive as zeros)
        {"cross_n2000_d64_b20_cols48_ragged", PWW_DTYPE_BF16, 20, 5, 2000, 77, 64, 1, false, 211, 0.8f, 0.f, 48, 17},    // odd block count per workgroup, 64-float tile rows, padding chunks
        {"sd15_cross_n1024_d80_cols32", PWW_DTYPE_BF16, 2, 8, 1024, 77, 80, 1, false, 17, 0.7f, 0.f, 32, 5},
        {"sd15_cross_n1024_d80_b16_cols48", PWW_DTYPE_F16, 16, 8, 1024, 77, 80, 1, false, 67, 0.7f, 0.f, 48, 17},
        {"sd15_cross_n256_d160_cols32", PWW_DTYPE_F16, 2, 8, 256, 77, 160, 1, false, 3, 0.5f, 0.f, 32, 9},
        {"sd15_cross_n256_d160_b16_cols32", PWW_DTYPE_BF16, 16, 8, 256, 77, 160, 1, false, 5, 0.5f, 0.f, 32, 9},
        {"sd15_cross_n64_d160_cols16", PWW_DTYPE_BF16, 2, 8, 64, 77, 160, 1, false, 1, 0.5f, 0.f, 16, 3},
        {"sd21_cross_n9216_d64_b8_cols32", PWW_DTYPE_BF16, 8, 5, 9216, 77, 64, 1, false, 1531, 0.8f, 0.f, 32, 12},
        {"cross_n200_m128_d64_cols112", PWW_DTYPE_BF16, 3, 5, 200, 128, 64, 1, false, 1, 0.8f, 0.f, 112, 32},
        {"cross_n100_m40_d80_cols16", PWW_DTYPE_F16, 2, 4, 100, 40, 80, 1, false, 1, 0.8f, 0.f, 16, 1},
        {"cross_n333_m77_d96_cols80", PWW_DTYPE_BF16, 2, 3, 333, 77, 96, 1, false, 1, 0.6f, 0.f, 77, 20},
        {"cross_n64_d160_b72_split", PWW_DTYPE_F16, 72, 8, 64, 77, 160, 1, false, 7, 0.5f},        // more (image, head) pairs than resident workgroups: two-launch path
        {"cross_n200_m128_d64", PWW_DTYPE_BF16, 3, 5, 200, 128, 64, 1, false, 1, 0.8f},
        {"cross_n100_m40_d80", PWW_DTYPE_F16, 2, 4, 100, 40, 80, 1, false, 1, 0.8f},
        {"sd15_cross_fullbias_n4096", PWW_DTYPE_F16, 1, 8, 4096, 77, 40, 2, false, 97, 1.0f},
        {"sd21_self_n2304_d64", PWW_DTYPE_BF16, 1, 10, 2304, 2304, 64, 0, true, 61, 0.8f},
        {"sd21_self_n9216_d64_b4", PWW_DTYPE_BF16, 4, 5, 9216, 9216, 64, 0, true, 1531, 0.8f},   // BASELINE config 5: 768x768, 2 images folded
        // the dominant launches of BASELINE configs 3 / 4 / 5 at the batch sizes bench.py runs them (profiling: tools/gpu_profile.sh)
        {"sd15_self_n4096_d40_f16_b16", PWW_DTYPE_F16, 16, 8, 4096, 4096, 40, 0, true, 2039, 1.0f, 0.f, 0, 0, true},
        {"sd15_self_n4096_d40_bf16_b16", PWW_DTYPE_BF16, 16, 8, 4096, 4096, 40, 0, true, 2039, 1.0f, 0.f, 0, 0, true},
        {"sd21_self_n9216_d64_b8", PWW_DTYPE_BF16, 8, 5, 9216, 9216, 64, 0, true, 4603, 0.8f, 0.f, 0, 0, true},
        {"d96_n200_m130", PWW_DTYPE_F16, 1, 2, 200, 130, 96, 2, false, 1, 0.6f},
        {"d128_n130_m200", PWW_DTYPE_BF16, 1, 2, 130, 200, 128, 0, false, 1, 0.6f},
        {"d48_n33_m1", PWW_DTYPE_F16, 1, 1, 33, 1, 48, 0, false, 1, 1.0f},
        {"d8_self_n300", PWW_DTYPE_F16, 2, 4, 300, 300, 8, 0, true, 1, 1.5f},
        {"d16_self_n200", PWW_DTYPE_BF16, 1, 4, 200, 200, 16, 0, true, 1, 1.5f},
        {"d24_cross_n130", PWW_DTYPE_F16, 1, 3, 130, 77, 24, 1, false, 1, 1.0f},
        {"d32_self_n129", PWW_DTYPE_F16, 1, 4, 129, 129, 32, 0, true, 1, 1.0f},
        {"d56_self_n257", PWW_DTYPE_BF16, 1, 2, 257, 257, 56, 0, true, 1, 1.0f},
        {"d72_self_n192", PWW_DTYPE_F16, 1, 2, 192, 192, 72, 0, true, 1, 1.0f},
        {"d88_self_n128", PWW_DTYPE_F16, 1, 2, 128, 128, 88, 0, true, 1, 1.0f},
        {"d104_self_n128", PWW_DTYPE_BF16, 1, 2, 128, 128, 104, 0, true, 1, 1.0f},
        {"d152_self_n128", PWW_DTYPE_F16, 1, 2, 128, 128, 152, 0, true, 1, 0.7f},
        // folded-reference variant (D % 16 == 8): ragged tails, every workgroup width, and logit spreads wide enough
        // that the lazy reference has to be raised many times per row (gain 4: logits ~ N(0, 4^2) * log2 e)
        {"d40_self_n300_f16", PWW_DTYPE_F16, 2, 4, 300, 300, 40, 0, true, 1, 1.0f},
        {"d40_self_n300_bf16_hot", PWW_DTYPE_BF16, 2, 4, 300, 300, 40, 0, true, 1, 4.0f},
        {"d40_n1000_m1000_hot", PWW_DTYPE_F16, 2, 8, 1000, 1000, 40, 0, true, 7, 4.0f},
        {"d40_n520_m129_cold", PWW_DTYPE_F16, 1, 8, 520, 129, 40, 0, false, 3, 0.05f},
        {"d40_n2048_b4_bf16", PWW_DTYPE_BF16, 4, 8, 2048, 2048, 40, 0, true, 61, 2.0f},
        {"d24_self_n200", PWW_DTYPE_BF16, 1, 4, 200, 200, 24, 0, true, 1, 2.0f},
        // range-free bf16 mode of the folded kernel: one key near the END of the sequence scores ~100 binary orders above the first
        // stage's maximum for many rows -> exp2 overflows in the fast path -> the workgroup must fall back to the exact online softmax
        {"d40_late_outlier_bf16", PWW_DTYPE_BF16, 2, 4, 700, 700, 40, 0, true, 1, 1.0f, 120.f},
        {"d40_late_outlier_n4096_bf16", PWW_DTYPE_BF16, 1, 8, 4096, 4096, 40, 0, true, 31, 1.0f, 120.f},
        {"d64_late_outlier_bf16", PWW_DTYPE_BF16, 2, 5, 1100, 1100, 64, 0, true, 3, 1.0f, 90.f},        // range-free general kernel (SD2.x head dim): fallback taken
        {"d80_late_outlier_bf16", PWW_DTYPE_BF16, 4, 8, 1024, 1024, 80, 0, true, 7, 0.7f, 90.f},
        {"d160_late_outlier_bf16", PWW_DTYPE_BF16, 2, 8, 300, 300, 160, 0, true, 3, 0.5f, 90.f},
        // f16 range-free mode (round 3: no headroom, 16 binary orders of room above the first stage's maximum): a late key far above it
        // overflows P to inf -> the workgroup must notice and take the exact path; a mild one (+8 natural units) must stay on the fast path
        {"d40_late_outlier_f16", PWW_DTYPE_F16, 2, 4, 700, 700, 40, 0, true, 1, 1.0f, 30.f},
        {"d40_late_outlier_n4096_f16", PWW_DTYPE_F16, 1, 8, 4096, 4096, 40, 0, true, 31, 1.0f, 30.f},
        {"d64_late_outlier_f16", PWW_DTYPE_F16, 2, 5, 1100, 1100, 64, 0, true, 3, 1.0f, 40.f},
        {"d80_late_outlier_f16", PWW_DTYPE_F16, 4, 8, 1024, 1024, 80, 0, true, 7, 0.7f, 40.f},
        {"d160_late_outlier_f16", PWW_DTYPE_F16, 2, 8, 300, 300, 160, 0, true, 3, 0.5f, 40.f},
        {"d40_mild_outlier_f16", PWW_DTYPE_F16, 2, 4, 700, 700, 40, 0, true, 1, 1.0f, 3.f},
        {"d64_self_n2304_f16_b8", PWW_DTYPE_F16, 8, 10, 2304, 2304, 64, 0, true, 193, 0.8f},
        {"d64_self_n2304_bf16_b8", PWW_DTYPE_BF16, 8, 10, 2304, 2304, 64, 0, true, 193, 0.8f},
        {"d40_mild_outlier_bf16", PWW_DTYPE_BF16, 2, 4, 700, 700, 40, 0, true, 1, 1.0f, 10.f},      // scaled logits to +-40: inside the range, fast path only
        // magnitude guard of the folded-reference kernel: row maxima of ~80 and ~120 natural units (gain g gives scaled logits ~ N(0, g^2)):
        // far beyond where the extra rounding of Q * scale * log2(e) keeps the bar -- the kernel has to take its exact path by itself
        {"d40_logit80_bf16", PWW_DTYPE_BF16, 2, 4, 700, 700, 40, 0, true, 1, 24.f},
        {"d40_logit80_f16", PWW_DTYPE_F16, 2, 4, 700, 700, 40, 0, true, 1, 24.f},
        {"d40_logit120_bf16", PWW_DTYPE_BF16, 2, 4, 1100, 1100, 40, 0, true, 3, 36.f},
        {"d40_logit120_f16", PWW_DTYPE_F16, 2, 4, 1100, 1100, 40, 0, true, 3, 36.f},
        {"d40_logit30_f16", PWW_DTYPE_F16, 2, 8, 1024, 1024, 40, 0, true, 7, 8.f},                   // f16: above the f16 limit (20), below the bf16 one
        {"d40_logit12_f16", PWW_DTYPE_F16, 2, 8, 1024, 1024, 40, 0, true, 7, 3.0f},                  // f16 fast path (folded), realistic logit spread
        {"d40_n4096_hot_f16_b2", PWW_DTYPE_F16, 2, 8, 4096, 4096, 40, 0, true, 193, 3.0f},
    };
    for (auto &c : cases) {
        const bool big = (long)c.N * c.M >= 1024L * 1024L || c.N >= 4096;
        if (quick && big) continue;
        if (c.on_request && !only && !match) continue;
        if (only && strcmp(only, c.name)) continue;
        if (match && !strstr(c.name, match)) continue;
        run_case(c, big || c.bias_mode == 1);
    }
    check_qproj(only, match, quick);
    if (!only && !match) {
        check_mask();
        check_cfg();
        check_errors();
    }
    printf("%s: %d failure(s)\n", g_fail ? "NATIVE CHECK FAILED" : "NATIVE CHECK OK", g_fail);
    return g_fail ? 1 : 0;
}
