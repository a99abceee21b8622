// extern "C" surface of libpww_hip.so (declared in include/pww_hip.h) + error plumbing.
#include <stdlib.h>
#include <string.h>
#include <mutex>
#include <vector>
#include "pww_common.h"

namespace pww {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int check_hip(hipError_t e, const char *what) {
    if (e == hipSuccess) return PWW_OK;
    set_error("%s: %s (%s)", what, hipGetErrorString(e), hipGetErrorName(e));
    return PWW_EHIP;
}

const DebugKnobs &debug_knobs() {
    static const DebugKnobs knobs = [] {
        DebugKnobs k;
        const char *e = getenv("PWW_DEBUG");
        if (!e) return k;
        const struct { const char *name; int *value; } table[] = {
            {"attn_rf", &k.attn_rf}, {"attn_ksplit", &k.attn_ksplit}, {"attn_fold", &k.attn_fold}, {"attn_nw8", &k.attn_nw8},
            {"attn_pair_major", &k.attn_pair_major}, {"attn_head_pairs", &k.attn_head_pairs}, {"cross_head_major", &k.cross_head_major}, {"attn_wide_store", &k.attn_wide_store}, {"cross_wg_per_cu", &k.cross_wg_per_cu},
            {"cross_assume_resident", &k.cross_assume_resident}, {"cross_gate_weight", &k.cross_gate_weight},
            {"cross_tile_nbuf", &k.cross_tile_nbuf}, {"cross_bias_lds", &k.cross_bias_lds}, {"cross_lean", &k.cross_lean}, {"cross_lean_multi", &k.cross_lean_multi},
            {"cross_lean_nw", &k.cross_lean_nw}, {"attn_ksplit_nw", &k.attn_ksplit_nw}, {"attn_ksplit1", &k.attn_ksplit1}, {"attn_ksplit_half", &k.attn_ksplit_half},
            {"attn_hot_sum", &k.attn_hot_sum}, {"attn_fold_limit_f16", &k.attn_fold_limit_f16}};
        const char *p = e;
        while (*p) {
            const char *eq = strchr(p, '='), *end = strchr(p, ',');
            if (!end) end = p + strlen(p);
            if (eq && eq < end) {
                bool known = false;
                for (const auto &t : table)
                    if (strlen(t.name) == (size_t)(eq - p) && !strncmp(t.name, p, eq - p)) { *t.value = atoi(eq + 1); known = true; }
                if (!known) fprintf(stderr, "libpww_hip: PWW_DEBUG: unknown knob '%.*s'\n", (int)(eq - p), p);
            }
            p = *end ? end + 1 : end;
        }
        return k;
    }();
    return knobs;
}

static int query_arch(char *buf, size_t n) {
    int dev = 0;
    if (int rc = check_hip(hipGetDevice(&dev), "hipGetDevice")) return rc;
    hipDeviceProp_t prop;
    if (int rc = check_hip(hipGetDeviceProperties(&prop, dev), "hipGetDeviceProperties")) return rc;
    // gcnArchName looks like "gfx950:sramecc+:xnack-"
    size_t len = strcspn(prop.gcnArchName, ":");
    if (len >= n) len = n ? n - 1 : 0;
    if (n) { memcpy(buf, prop.gcnArchName, len); buf[len] = 0; }
    return PWW_OK;
}

// The code objects in this library are built for gfx950 only; refuse anything else loudly instead
// of failing inside a launch.
bool arch_ok() {
    static thread_local int cached = -1;  // -1 unknown, 0 bad, 1 ok
    if (cached < 0) {
        char arch[64] = "";
        if (query_arch(arch, sizeof(arch)) != PWW_OK) return false;
        cached = strcmp(arch, "gfx950") == 0 ? 1 : 0;
        if (!cached) set_error("libpww_hip is built for gfx950 (MI355X) only; device reports '%s'", arch);
    } else if (cached == 0) {
        set_error("libpww_hip is built for gfx950 (MI355X) only");
    }
    return cached == 1;
}

// ---- kernel-only timing slots (pww_profile_*): event pairs owned by the library, handed to the next attention launch of the
// arming thread (launch_attn_kernel in pww_common.h)
// The pool is bounded: slot numbers keep counting, the event pair behind slot n is pair n % PWW_PROFILE_SLOTS (created on first
// use, re-used afterwards), so a long-running process that arms a slot per launch holds at most 2 * PWW_PROFILE_SLOTS events.
struct ProfileSlot { hipEvent_t start, stop; bool used; int owner; };
static std::mutex g_prof_mutex;
static std::vector<ProfileSlot> g_prof_slots;
static int g_prof_next = 0;
static thread_local int g_prof_armed = -1;

bool profile_take(hipEvent_t *start, hipEvent_t *stop, hipStream_t stream) {
    if (g_prof_armed < 0) return false;
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &st) != hipSuccess || st != hipStreamCaptureStatusNone) return false;   // event-stamped launches cannot be captured: stay armed
    std::lock_guard<std::mutex> lock(g_prof_mutex);
    const size_t idx = (size_t)(g_prof_armed % PWW_PROFILE_SLOTS);
    if (idx >= g_prof_slots.size()) { g_prof_armed = -1; return false; }      // another thread reset the pool since this thread armed
    ProfileSlot &sl = g_prof_slots[idx];
    const bool mine = sl.owner == g_prof_armed;
    g_prof_armed = -1;
    if (!mine) return false;          // the pair was recycled by a later arm before this launch happened
    sl.used = true;
    *start = sl.start;
    *stop = sl.stop;
    return true;
}

static int profile_arm() {
    std::lock_guard<std::mutex> lock(g_prof_mutex);
    const int slot = g_prof_next;
    const size_t idx = (size_t)(slot % PWW_PROFILE_SLOTS);
    if (idx >= g_prof_slots.size()) {
        ProfileSlot sl{nullptr, nullptr, false, -1};
        if (check_hip(hipEventCreate(&sl.start), "hipEventCreate") || check_hip(hipEventCreate(&sl.stop), "hipEventCreate")) return PWW_EHIP;
        g_prof_slots.push_back(sl);
    }
    g_prof_slots[idx].used = false;
    g_prof_slots[idx].owner = slot;
    g_prof_next = slot == 0x7ffffffe ? 0 : slot + 1;
    g_prof_armed = slot;
    return slot;
}

static int profile_elapsed_us(int slot, float *us) {
    hipEvent_t e0, e1;
    {
        std::lock_guard<std::mutex> lock(g_prof_mutex);
        const size_t idx = (size_t)(slot % PWW_PROFILE_SLOTS);
        if (!us || slot < 0 || idx >= g_prof_slots.size() || g_prof_slots[idx].owner != slot) { set_error("pww_profile_elapsed_us: no such slot (or recycled: the pool holds %d)", PWW_PROFILE_SLOTS); return PWW_EINVAL; }
        if (!g_prof_slots[idx].used) { set_error("pww_profile_elapsed_us: slot %d was armed but no attention kernel was launched", slot); return PWW_EINVAL; }
        e0 = g_prof_slots[idx].start;
        e1 = g_prof_slots[idx].stop;
    }
    if (int rc = check_hip(hipEventSynchronize(e1), "hipEventSynchronize")) return rc;
    float ms = 0.f;
    if (int rc = check_hip(hipEventElapsedTime(&ms, e0, e1), "hipEventElapsedTime")) return rc;
    *us = ms * 1e3f;
    return PWW_OK;
}

static void profile_reset() {
    std::lock_guard<std::mutex> lock(g_prof_mutex);
    for (ProfileSlot &sl : g_prof_slots) { (void)hipEventDestroy(sl.start); (void)hipEventDestroy(sl.stop); }
    g_prof_slots.clear();
    g_prof_next = 0;
    g_prof_armed = -1;
}

// ---- phase time stamps (pww_debug_timeline): one process-wide debug pointer handed to every attention launch
// (pointer and size are read by concurrent launches: both live behind the same mutex as the timing slots)
static unsigned long long *g_timeline = nullptr;
static size_t g_timeline_bytes = 0;
unsigned long long *debug_timeline() { std::lock_guard<std::mutex> lock(g_prof_mutex); return g_timeline_bytes ? g_timeline : nullptr; }
size_t debug_timeline_bytes() { std::lock_guard<std::mutex> lock(g_prof_mutex); return g_timeline ? g_timeline_bytes : 0; }
// ---- path counters of the folded-reference self-attention kernel (pww_debug_path_counts): { range-free, lazy, exact } workgroups
static unsigned *g_path_counts = nullptr;
unsigned *debug_path_counts() { std::lock_guard<std::mutex> lock(g_prof_mutex); return g_path_counts; }

int attn_fwd(const void *q, const void *k, const void *v, void *o, const float *bias, const float *bias_coeff,
             const pww_attn_desc_t *d, hipStream_t stream, const double *stats = nullptr, int stat_kind = PWW_STAT_NONE,
             double stat_count = 1.0, float coeff_scalar = 1.f, const float *coeff_scalar_dev = nullptr);
int qk_reduce(const void *q, const void *k, const pww_attn_desc_t *d, double *stats, void *workspace,
              size_t workspace_bytes, hipStream_t stream);
size_t qk_reduce_workspace_bytes(const pww_attn_desc_t *d);
int cross_attn_fused(const void *q, const void *k, const void *v, void *o, const float *bias, int stat_kind, float coeff_scalar,
                     const float *gate, const pww_attn_desc_t *d, double *stats_out, void *state, size_t state_bytes,
                     void *workspace, size_t workspace_bytes, const pww_cross_opts_t *opts, hipStream_t stream,
                     const double *ext_part = nullptr, int ext_nparts = 0, bool pass2_only = false);
int qproj_stat(const void *x, const void *w, void *q, const void *k, const float *gate, const pww_qproj_desc_t *d, int stat_kind,
               double *partials, size_t partials_bytes, hipStream_t stream);
int qproj_parts(const pww_qproj_desc_t *d);
int qk_parts(const void *q, const void *k, const float *gate, const pww_attn_desc_t *d, int stat_kind, int gated_images, double *partials,
             size_t partials_bytes, hipStream_t stream);
int qk_parts_count(const pww_attn_desc_t *d);
int cross_attn_out(const void *q, const void *k, const void *v, void *out, const float *bias, int stat_kind, float coeff_scalar, const float *gate,
                   const pww_attn_desc_t *d, const double *parts, int nparts, double *stats_out, const pww_cross_opts_t *opts, const void *w,
                   const void *w_bias, const void *residual, const int64_t *residual_stride, hipStream_t stream);
int cross_attn_out_supported(const pww_attn_desc_t *d, int C, int bias_cols);
int mask_build_f32_levels(const float *masks, int H, int W, int R, const int32_t *col_ptr, const int32_t *col_reg, int T,
                          float *out8, float *out16, float *out32, float *out64, hipStream_t stream);
size_t cross_fused_workspace_bytes(const pww_attn_desc_t *d);
size_t cross_fused_state_bytes(const pww_attn_desc_t *d);
int mask_build(const uint8_t *rgb, int H, int W, const pww_region_t *regions, int R, const int32_t *col_ptr,
               const int32_t *col_reg, int T, float *out8, float *out16, float *out32, float *out64,
               hipStream_t stream);
int mask_build_rgb(const uint8_t *rgb, int H, int W, const pww_region_t *regions, int R, const int32_t *col_ptr,
                   const int32_t *col_reg, int T, int ratio, float *out, hipStream_t stream);
int mask_build_f32(const float *masks, int H, int W, int R, const int32_t *col_ptr, const int32_t *col_reg, int T,
                   int ratio, float *out, hipStream_t stream);
int resize_tokens(const float *orig, int H, int W, int T, int oh, int ow, int n_tokens, float *out, hipStream_t stream);
int gauss_blur(const float *in, float *out, int H, int W, const float *weights, int ksize, double *tmp, hipStream_t stream);
int inpaint_prep(const uint8_t *rgb, const uint8_t *mask, int H, int W, int h, int w, float *mask_out, float *masked,
                 float *mask_lat, hipStream_t stream);
int cfg_combine(const void *cond, const void *uncond, float g, float *out, long n, int dtype, hipStream_t stream);
int store_f32(float *dst, const float *values, int n, hipStream_t stream);
size_t group_norm_workspace_bytes(const pww_gn_desc_t *d);
int group_norm_fwd(const void *x, const void *pre_c, const void *add_bc, const void *gamma, const void *beta, void *y, const pww_gn_desc_t *d,
                   void *workspace, size_t workspace_bytes, hipStream_t stream);
int add_layer_norm(const void *a, const void *x, const void *gamma, const void *beta, void *s, void *y, const pww_ln_desc_t *d, hipStream_t stream, const void *post_bias = nullptr);
int geglu(const void *h, void *y, int64_t rows, int32_t D, int64_t h_stride, int64_t y_stride, int32_t dtype, hipStream_t stream);
int bias_residual(const void *r, const void *v, const void *bias, void *y, int32_t B, int32_t C, int32_t HW, int32_t layout, int32_t dtype, hipStream_t stream);

}  // namespace pww

extern "C" {

int pww_version(void) { return PWW_VERSION; }
int pww_has_experiments(void) { return PWW_EXPERIMENTS; }

int pww_profile_arm(void) { return pww::profile_arm(); }
int pww_profile_elapsed_us(int slot, float *us) { return pww::profile_elapsed_us(slot, us); }
void pww_profile_reset(void) { pww::profile_reset(); }

const char *pww_last_error(void) { return pww::g_err; }

int pww_device_arch(char *buf, size_t n) {
    if (!buf || !n) { pww::set_error("pww_device_arch: null buffer"); return PWW_EINVAL; }
    return pww::query_arch(buf, n);
}

int pww_self_attn_fwd(const void *q, const void *k, const void *v, void *o, const pww_attn_desc_t *desc,
                      void *stream) {
    return pww::attn_fwd(q, k, v, o, nullptr, nullptr, desc, static_cast<hipStream_t>(stream));
}

int pww_cross_attn_fwd(const void *q, const void *k, const void *v, void *o, const float *bias,
                       const float *bias_coeff, const pww_attn_desc_t *desc, void *stream) {
    return pww::attn_fwd(q, k, v, o, bias, bias_coeff, desc, static_cast<hipStream_t>(stream));
}

int pww_cross_attn_fwd_stat(const void *q, const void *k, const void *v, void *o, const float *bias,
                            const double *stats, int32_t stat_kind, double stat_count, float coeff_scalar,
                            const float *gate, const pww_attn_desc_t *desc, void *stream) {
    return pww::attn_fwd(q, k, v, o, bias, gate, desc, static_cast<hipStream_t>(stream), stats, stat_kind, stat_count,
                         coeff_scalar);
}

#if PWW_EXPERIMENTS      // libpww_hip_experiments.so only (include/pww_hip.h, "experiments")
int pww_cross_attn_fwd_fused(const void *q, const void *k, const void *v, void *o, const float *bias, int32_t stat_kind,
                             float coeff_scalar, const float *gate, const pww_attn_desc_t *desc, double *stats_out, void *state,
                             size_t state_bytes, void *workspace, size_t workspace_bytes, void *stream) {
    return pww::cross_attn_fused(q, k, v, o, bias, stat_kind, coeff_scalar, gate, desc, stats_out, state, state_bytes, workspace,
                                 workspace_bytes, nullptr, static_cast<hipStream_t>(stream));
}

int pww_cross_attn_fwd_fused_ex(const void *q, const void *k, const void *v, void *o, const float *bias, int32_t stat_kind,
                                float coeff_scalar, const float *gate, const pww_attn_desc_t *desc, double *stats_out, void *state,
                                size_t state_bytes, void *workspace, size_t workspace_bytes, const pww_cross_opts_t *opts, void *stream) {
    return pww::cross_attn_fused(q, k, v, o, bias, stat_kind, coeff_scalar, gate, desc, stats_out, state, state_bytes, workspace,
                                 workspace_bytes, opts, static_cast<hipStream_t>(stream));
}
#endif

int pww_cross_attn_fwd_stat_ex(const void *q, const void *k, const void *v, void *o, const float *bias,
                               const double *stats, int32_t stat_kind, double stat_count, float coeff_scalar,
                               const float *gate, const pww_attn_desc_t *desc, const pww_cross_opts_t *opts, void *stream) {
    const float *dev = nullptr;
    if (opts) {
        if (opts->size < 16) { pww::set_error("pww_cross_attn_fwd_stat_ex: pww_cross_opts_t.size = %u is not a known layout", opts->size); return PWW_EINVAL; }
        dev = opts->coeff_scalar_dev;
    }
    return pww::attn_fwd(q, k, v, o, bias, gate, desc, static_cast<hipStream_t>(stream), stats, stat_kind, stat_count,
                         coeff_scalar, dev);
}

int pww_qproj_stat(const void *x, const void *w, void *q, const void *k, const float *gate, const pww_qproj_desc_t *desc,
                   int32_t stat_kind, double *partials, size_t partials_bytes, void *stream) {
    return pww::qproj_stat(x, w, q, k, gate, desc, stat_kind, partials, partials_bytes, static_cast<hipStream_t>(stream));
}

int32_t pww_qproj_parts(const pww_qproj_desc_t *desc) { return pww::qproj_parts(desc); }

int pww_qk_parts(const void *q, const void *k, const float *gate, const pww_attn_desc_t *desc, int32_t stat_kind, int32_t gated_images,
                 double *partials, size_t partials_bytes, void *stream) {
    return pww::qk_parts(q, k, gate, desc, stat_kind, gated_images, partials, partials_bytes, static_cast<hipStream_t>(stream));
}

int32_t pww_qk_parts_count(const pww_attn_desc_t *desc) { return pww::qk_parts_count(desc); }

#if PWW_EXPERIMENTS
int pww_cross_attn_fwd_parts_out(const void *q, const void *k, const void *v, void *out, const float *bias, int32_t stat_kind, float coeff_scalar,
                                 const float *gate, const pww_attn_desc_t *desc, const double *partials, int32_t nparts, double *stats_out,
                                 const pww_cross_opts_t *opts, const void *w, const void *w_bias, const void *residual,
                                 const int64_t *residual_stride, void *stream) {
    return pww::cross_attn_out(q, k, v, out, bias, stat_kind, coeff_scalar, gate, desc, partials, partials ? nparts : 0, stats_out, opts, w, w_bias,
                               residual, residual_stride, static_cast<hipStream_t>(stream));
}

int32_t pww_cross_attn_out_supported(const pww_attn_desc_t *desc, int32_t c_out, int32_t bias_cols) {
    return pww::cross_attn_out_supported(desc, c_out, bias_cols);
}
#endif

int pww_cross_attn_fwd_parts(const void *q, const void *k, const void *v, void *o, const float *bias, int32_t stat_kind,
                             float coeff_scalar, const float *gate, const pww_attn_desc_t *desc, const double *partials,
                             int32_t nparts, double *stats_out, const pww_cross_opts_t *opts, void *stream) {
    if (!partials && stat_kind != PWW_STAT_NONE) { pww::set_error("pww_cross_attn_fwd_parts: null partials"); return PWW_EINVAL; }
    return pww::cross_attn_fused(q, k, v, o, bias, stat_kind, coeff_scalar, gate, desc, stats_out, nullptr, 0, nullptr, 0, opts,
                                 static_cast<hipStream_t>(stream), partials, partials ? nparts : 0, true);
}

int pww_mask_build_f32_levels(const float *masks, int32_t H, int32_t W, int32_t R, const int32_t *col_ptr, const int32_t *col_reg,
                              int32_t T, float *out8, float *out16, float *out32, float *out64, void *stream) {
    return pww::mask_build_f32_levels(masks, H, W, R, col_ptr, col_reg, T, out8, out16, out32, out64, static_cast<hipStream_t>(stream));
}

void pww_debug_path_counts(void *device_buffer) {
    std::lock_guard<std::mutex> lock(pww::g_prof_mutex);
    pww::g_path_counts = static_cast<unsigned *>(device_buffer);
}

void pww_debug_timeline(void *device_buffer, size_t bytes) {
    std::lock_guard<std::mutex> lock(pww::g_prof_mutex);
    pww::g_timeline = static_cast<unsigned long long *>(device_buffer);
    pww::g_timeline_bytes = device_buffer ? bytes : 0;
}

#if PWW_EXPERIMENTS
size_t pww_cross_fused_workspace_bytes(const pww_attn_desc_t *desc) { return pww::cross_fused_workspace_bytes(desc); }

size_t pww_cross_fused_state_bytes(const pww_attn_desc_t *desc) { return pww::cross_fused_state_bytes(desc); }
#endif

int pww_qk_reduce(const void *q, const void *k, const pww_attn_desc_t *desc, double *stats, void *workspace,
                  size_t workspace_bytes, void *stream) {
    return pww::qk_reduce(q, k, desc, stats, workspace, workspace_bytes, static_cast<hipStream_t>(stream));
}

int pww_mask_build(const uint8_t *rgb, int32_t H, int32_t W, const pww_region_t *regions, int32_t R,
                   const int32_t *col_ptr, const int32_t *col_reg, int32_t T, float *out8, float *out16,
                   float *out32, float *out64, void *stream) {
    return pww::mask_build(rgb, H, W, regions, R, col_ptr, col_reg, T, out8, out16, out32, out64,
                           static_cast<hipStream_t>(stream));
}

int pww_mask_build_rgb(const uint8_t *rgb, int32_t H, int32_t W, const pww_region_t *regions, int32_t R,
                       const int32_t *col_ptr, const int32_t *col_reg, int32_t T, int32_t ratio, float *out,
                       void *stream) {
    return pww::mask_build_rgb(rgb, H, W, regions, R, col_ptr, col_reg, T, ratio, out, static_cast<hipStream_t>(stream));
}

int pww_mask_build_f32(const float *masks, int32_t H, int32_t W, int32_t R, const int32_t *col_ptr,
                       const int32_t *col_reg, int32_t T, int32_t ratio, float *out, void *stream) {
    return pww::mask_build_f32(masks, H, W, R, col_ptr, col_reg, T, ratio, out, static_cast<hipStream_t>(stream));
}

int pww_resize_tokens(const float *orig, int32_t H, int32_t W, int32_t T, int32_t oh, int32_t ow, int32_t n_tokens, float *out,
                      void *stream) {
    return pww::resize_tokens(orig, H, W, T, oh, ow, n_tokens, out, static_cast<hipStream_t>(stream));
}

int pww_gauss_blur(const float *in, float *out, int32_t H, int32_t W, const float *weights, int32_t ksize, double *tmp,
                   void *stream) {
    return pww::gauss_blur(in, out, H, W, weights, ksize, tmp, static_cast<hipStream_t>(stream));
}

int pww_inpaint_prep(const uint8_t *rgb, const uint8_t *mask, int32_t H, int32_t W, int32_t h, int32_t w, float *mask_out,
                     float *masked_image, float *mask_lat, void *stream) {
    return pww::inpaint_prep(rgb, mask, H, W, h, w, mask_out, masked_image, mask_lat, static_cast<hipStream_t>(stream));
}

int pww_cfg_combine(const void *cond, const void *uncond, float guidance, float *out, int64_t n, int32_t dtype,
                    void *stream) {
    return pww::cfg_combine(cond, uncond, guidance, out, (long)n, dtype, static_cast<hipStream_t>(stream));
}

int pww_store_f32(float *dst, const float *values, int32_t n, void *stream) { return pww::store_f32(dst, values, n, static_cast<hipStream_t>(stream)); }

size_t pww_workspace_bytes(const pww_attn_desc_t *desc) { return pww::qk_reduce_workspace_bytes(desc); }

size_t pww_group_norm_workspace_bytes(const pww_gn_desc_t *desc) { return pww::group_norm_workspace_bytes(desc); }
int pww_group_norm_fwd(const void *x, const void *pre_c, const void *add_bc, const void *gamma, const void *beta, void *y, const pww_gn_desc_t *desc,
                       void *workspace, size_t workspace_bytes, void *stream) {
    return pww::group_norm_fwd(x, pre_c, add_bc, gamma, beta, y, desc, workspace, workspace_bytes, static_cast<hipStream_t>(stream));
}
int pww_add_layer_norm(const void *a, const void *x, const void *gamma, const void *beta, void *s, void *y, const pww_ln_desc_t *desc, void *stream) {
    return pww::add_layer_norm(a, x, gamma, beta, s, y, desc, static_cast<hipStream_t>(stream));
}
int pww_add_layer_norm_bias(const void *a, const void *x, const void *gamma, const void *beta, const void *post_bias, void *s, void *y, const pww_ln_desc_t *desc, void *stream) {
    return pww::add_layer_norm(a, x, gamma, beta, s, y, desc, static_cast<hipStream_t>(stream), post_bias);
}
int pww_geglu(const void *h, void *y, int64_t rows, int32_t D, int64_t h_stride, int64_t y_stride, int32_t dtype, void *stream) {
    return pww::geglu(h, y, rows, D, h_stride, y_stride, dtype, static_cast<hipStream_t>(stream));
}
int pww_bias_residual(const void *r, const void *v, const void *bias, void *y, int32_t B, int32_t C, int32_t HW, int32_t layout, int32_t dtype, void *stream) {
    return pww::bias_residual(r, v, bias, y, B, C, HW, layout, dtype, static_cast<hipStream_t>(stream));
}

}  // extern "C"
