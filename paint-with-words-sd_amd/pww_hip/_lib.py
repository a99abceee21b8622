"""ctypes binding of libpww_hip.so (the C ABI declared in include/pww_hip.h).

The library is plain HIP behind `extern "C"`; nothing here depends on torch. Loading fails LOUDLY:
there is no CPU or PyTorch fallback for the kernels (the oracle under oracle/ is test-only).
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PWW_HIP_LIB", os.path.join(_HERE, "libpww_hip.so"))
# the product library compiled with -DPWW_EXPERIMENTS=1: + the forms that were measured and not made a default (include/pww_hip.h, section
# "experiments"). Tests and A/B tools load it (`load_experiments()`, or a whole process through PWW_HIP_LIB); the product never does.
EXPERIMENTS_LIB_PATH = os.environ.get("PWW_HIP_EXPERIMENTS_LIB", os.path.join(_HERE, "libpww_hip_experiments.so"))

PWW_OK, PWW_EINVAL, PWW_ENOTSUP, PWW_EHIP = 0, -22, -95, -5
MIN_VERSION = 126        # oldest libpww_hip ABI (pww_version(): major * 100 + minor) this package drives
DTYPE_F16, DTYPE_BF16 = 0, 1
LAYOUT_NCHW, LAYOUT_NHWC = 0, 1
ACT_NONE, ACT_SILU = 0, 1
MAX_HEAD_DIM = 160

# every symbol include/pww_hip.h declares for libpww_hip.so (tests check the library exports all of them) ...
EXPORTS = ("pww_version", "pww_has_experiments", "pww_last_error", "pww_device_arch", "pww_self_attn_fwd", "pww_cross_attn_fwd",
           "pww_cross_attn_fwd_stat", "pww_cross_attn_fwd_stat_ex",
           "pww_qk_reduce", "pww_mask_build", "pww_mask_build_rgb", "pww_mask_build_f32", "pww_resize_tokens", "pww_gauss_blur", "pww_inpaint_prep", "pww_cfg_combine", "pww_store_f32",
           "pww_workspace_bytes", "pww_profile_arm", "pww_profile_elapsed_us", "pww_profile_reset", "pww_debug_timeline", "pww_debug_path_counts",
           "pww_qproj_stat", "pww_qproj_parts", "pww_cross_attn_fwd_parts", "pww_mask_build_f32_levels", "pww_qk_parts", "pww_qk_parts_count",
           "pww_group_norm_fwd", "pww_group_norm_workspace_bytes", "pww_add_layer_norm", "pww_add_layer_norm_bias", "pww_geglu", "pww_bias_residual")
# ... and what only libpww_hip_experiments.so has on top of them (the header's "experiments" section)
EXPERIMENT_EXPORTS = ("pww_cross_attn_fwd_fused", "pww_cross_attn_fwd_fused_ex", "pww_cross_fused_workspace_bytes", "pww_cross_fused_state_bytes",
                      "pww_cross_attn_fwd_parts_out", "pww_cross_attn_out_supported")


class AttnDesc(ctypes.Structure):
    """struct pww_attn_desc (include/pww_hip.h)."""
    _fields_ = [("dtype", ctypes.c_int32), ("B", ctypes.c_int32), ("H", ctypes.c_int32), ("N", ctypes.c_int32),
                ("M", ctypes.c_int32), ("D", ctypes.c_int32),
                ("q_stride", ctypes.c_int64 * 3), ("k_stride", ctypes.c_int64 * 3),
                ("v_stride", ctypes.c_int64 * 3), ("o_stride", ctypes.c_int64 * 3),
                ("scale", ctypes.c_float), ("bias_stride", ctypes.c_int64 * 4)]


class CrossOpts(ctypes.Structure):
    """struct pww_cross_opts (optional arguments of the *_ex cross-attention entry points)."""
    _fields_ = [("size", ctypes.c_uint32), ("bias_cols", ctypes.c_int32), ("coeff_scalar_dev", ctypes.c_void_p),
                ("bias_compact", ctypes.c_void_p), ("col_idx", ctypes.c_void_p), ("R", ctypes.c_int32), ("gated_images", ctypes.c_int32),
                ("compact_stride", ctypes.c_int64 * 2), ("col_idx_stride", ctypes.c_int64)]


class QprojDesc(ctypes.Structure):
    """struct pww_qproj_desc (query projection + score-statistic partials)."""
    _fields_ = [("dtype", ctypes.c_int32), ("B", ctypes.c_int32), ("N", ctypes.c_int32), ("Cin", ctypes.c_int32), ("H", ctypes.c_int32),
                ("D", ctypes.c_int32), ("M", ctypes.c_int32), ("x_stride", ctypes.c_int64 * 2), ("q_stride", ctypes.c_int64 * 2),
                ("k_stride", ctypes.c_int64 * 2)]


class GnDesc(ctypes.Structure):
    """struct pww_gn_desc (GroupNorm + addend + activation)."""
    _fields_ = [("dtype", ctypes.c_int32), ("layout", ctypes.c_int32), ("B", ctypes.c_int32), ("C", ctypes.c_int32), ("HW", ctypes.c_int32),
                ("G", ctypes.c_int32), ("eps", ctypes.c_float), ("act", ctypes.c_int32), ("add_stride", ctypes.c_int32), ("_pad", ctypes.c_int32)]


class LnDesc(ctypes.Structure):
    """struct pww_ln_desc (add + LayerNorm)."""
    _fields_ = [("dtype", ctypes.c_int32), ("C", ctypes.c_int32), ("rows", ctypes.c_int64), ("a_stride", ctypes.c_int64), ("x_stride", ctypes.c_int64),
                ("s_stride", ctypes.c_int64), ("y_stride", ctypes.c_int64), ("eps", ctypes.c_float), ("_pad", ctypes.c_int32)]


class Region(ctypes.Structure):
    """struct pww_region."""
    _fields_ = [("r", ctypes.c_uint8), ("g", ctypes.c_uint8), ("b", ctypes.c_uint8), ("_pad", ctypes.c_uint8),
                ("strength", ctypes.c_float)]


class PwwHipError(RuntimeError):
    pass


_lib = None
_exp = None


def _bind(lib, experiments):
    vp, i32, i64, f32 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_float
    lib.pww_version.restype = ctypes.c_int
    lib.pww_last_error.restype = ctypes.c_char_p
    lib.pww_device_arch.argtypes = [ctypes.c_char_p, ctypes.c_size_t]
    lib.pww_self_attn_fwd.argtypes = [vp, vp, vp, vp, ctypes.POINTER(AttnDesc), vp]
    lib.pww_cross_attn_fwd.argtypes = [vp, vp, vp, vp, vp, vp, ctypes.POINTER(AttnDesc), vp]
    lib.pww_cross_attn_fwd_stat.argtypes = [vp, vp, vp, vp, vp, vp, i32, ctypes.c_double, f32, vp, ctypes.POINTER(AttnDesc), vp]
    lib.pww_cross_attn_fwd_stat_ex.argtypes = [vp, vp, vp, vp, vp, vp, i32, ctypes.c_double, f32, vp, ctypes.POINTER(AttnDesc),
                                               ctypes.POINTER(CrossOpts), vp]
    lib.pww_qproj_stat.argtypes = [vp, vp, vp, vp, vp, ctypes.POINTER(QprojDesc), i32, vp, ctypes.c_size_t, vp]
    lib.pww_qproj_stat.restype = ctypes.c_int
    lib.pww_qproj_parts.argtypes = [ctypes.POINTER(QprojDesc)]
    lib.pww_qproj_parts.restype = ctypes.c_int32
    lib.pww_cross_attn_fwd_parts.argtypes = [vp, vp, vp, vp, vp, i32, f32, vp, ctypes.POINTER(AttnDesc), vp, i32, vp, ctypes.POINTER(CrossOpts), vp]
    lib.pww_cross_attn_fwd_parts.restype = ctypes.c_int
    lib.pww_qk_parts.argtypes = [vp, vp, vp, ctypes.POINTER(AttnDesc), i32, i32, vp, ctypes.c_size_t, vp]
    lib.pww_qk_parts.restype = ctypes.c_int
    lib.pww_qk_parts_count.argtypes = [ctypes.POINTER(AttnDesc)]
    lib.pww_qk_parts_count.restype = ctypes.c_int32
    lib.pww_mask_build_f32_levels.argtypes = [vp, i32, i32, i32, vp, vp, i32, vp, vp, vp, vp, vp]
    lib.pww_mask_build_f32_levels.restype = ctypes.c_int
    lib.pww_group_norm_workspace_bytes.argtypes = [ctypes.POINTER(GnDesc)]
    lib.pww_group_norm_workspace_bytes.restype = ctypes.c_size_t
    lib.pww_group_norm_fwd.argtypes = [vp, vp, vp, vp, vp, vp, ctypes.POINTER(GnDesc), vp, ctypes.c_size_t, vp]
    lib.pww_group_norm_fwd.restype = ctypes.c_int
    lib.pww_add_layer_norm.argtypes = [vp, vp, vp, vp, vp, vp, ctypes.POINTER(LnDesc), vp]
    lib.pww_add_layer_norm.restype = ctypes.c_int
    lib.pww_add_layer_norm_bias.argtypes = [vp, vp, vp, vp, vp, vp, vp, ctypes.POINTER(LnDesc), vp]
    lib.pww_add_layer_norm_bias.restype = ctypes.c_int
    lib.pww_geglu.argtypes = [vp, vp, ctypes.c_int64, i32, ctypes.c_int64, ctypes.c_int64, i32, vp]
    lib.pww_geglu.restype = ctypes.c_int
    lib.pww_bias_residual.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, i32, vp]
    lib.pww_bias_residual.restype = ctypes.c_int
    lib.pww_debug_timeline.argtypes = [vp, ctypes.c_size_t]
    lib.pww_debug_timeline.restype = None
    lib.pww_debug_path_counts.argtypes = [vp]
    lib.pww_debug_path_counts.restype = None
    lib.pww_qk_reduce.argtypes = [vp, vp, ctypes.POINTER(AttnDesc), vp, vp, ctypes.c_size_t, vp]
    lib.pww_mask_build.argtypes = [vp, i32, i32, vp, i32, vp, vp, i32, vp, vp, vp, vp, vp]
    lib.pww_mask_build_rgb.argtypes = [vp, i32, i32, vp, i32, vp, vp, i32, i32, vp, vp]
    lib.pww_mask_build_f32.argtypes = [vp, i32, i32, i32, vp, vp, i32, i32, vp, vp]
    lib.pww_resize_tokens.argtypes = [vp, i32, i32, i32, i32, i32, i32, vp, vp]
    lib.pww_gauss_blur.argtypes = [vp, vp, i32, i32, vp, i32, vp, vp]
    lib.pww_inpaint_prep.argtypes = [vp, vp, i32, i32, i32, i32, vp, vp, vp, vp]
    lib.pww_cfg_combine.argtypes = [vp, vp, f32, vp, i64, i32, vp]
    lib.pww_store_f32.argtypes = [vp, ctypes.POINTER(ctypes.c_float), i32, vp]
    lib.pww_store_f32.restype = ctypes.c_int
    lib.pww_workspace_bytes.argtypes = [ctypes.POINTER(AttnDesc)]
    lib.pww_workspace_bytes.restype = ctypes.c_size_t
    lib.pww_profile_arm.argtypes = []
    lib.pww_profile_arm.restype = ctypes.c_int
    lib.pww_profile_elapsed_us.argtypes = [i32, ctypes.POINTER(ctypes.c_float)]
    lib.pww_profile_elapsed_us.restype = ctypes.c_int
    lib.pww_profile_reset.argtypes = []
    lib.pww_profile_reset.restype = None
    for name in ("pww_device_arch", "pww_self_attn_fwd", "pww_cross_attn_fwd", "pww_cross_attn_fwd_stat", "pww_cross_attn_fwd_stat_ex", "pww_qk_reduce",
                 "pww_mask_build", "pww_mask_build_rgb", "pww_mask_build_f32", "pww_resize_tokens", "pww_gauss_blur", "pww_inpaint_prep", "pww_cfg_combine"):
        getattr(lib, name).restype = ctypes.c_int
    if lib.pww_version() // 100 != 1 or lib.pww_version() < MIN_VERSION:
        raise PwwHipError("libpww_hip ABI version %d.%02d is not 1.x >= 1.%02d (rebuild: python paint-with-words-sd_amd/build.py)"
                          % (lib.pww_version() // 100, lib.pww_version() % 100, MIN_VERSION % 100))
    lib.pww_has_experiments.restype = ctypes.c_int
    if experiments:
        lib.pww_cross_attn_fwd_fused.argtypes = [vp, vp, vp, vp, vp, i32, f32, vp, ctypes.POINTER(AttnDesc), vp, vp, ctypes.c_size_t, vp, ctypes.c_size_t, vp]
        lib.pww_cross_attn_fwd_fused_ex.argtypes = [vp, vp, vp, vp, vp, i32, f32, vp, ctypes.POINTER(AttnDesc), vp, vp, ctypes.c_size_t, vp,
                                                    ctypes.c_size_t, ctypes.POINTER(CrossOpts), vp]
        lib.pww_cross_attn_fwd_fused.restype = lib.pww_cross_attn_fwd_fused_ex.restype = ctypes.c_int
        lib.pww_cross_fused_workspace_bytes.argtypes = [ctypes.POINTER(AttnDesc)]
        lib.pww_cross_fused_workspace_bytes.restype = ctypes.c_size_t
        lib.pww_cross_fused_state_bytes.argtypes = [ctypes.POINTER(AttnDesc)]
        lib.pww_cross_fused_state_bytes.restype = ctypes.c_size_t
        lib.pww_cross_attn_fwd_parts_out.argtypes = [vp, vp, vp, vp, vp, i32, f32, vp, ctypes.POINTER(AttnDesc), vp, i32, vp, ctypes.POINTER(CrossOpts),
                                                     vp, vp, vp, ctypes.POINTER(ctypes.c_int64), vp]
        lib.pww_cross_attn_fwd_parts_out.restype = ctypes.c_int
        lib.pww_cross_attn_out_supported.argtypes = [ctypes.POINTER(AttnDesc), ctypes.c_int32, ctypes.c_int32]
        lib.pww_cross_attn_out_supported.restype = ctypes.c_int32
    return lib


def load():
    """Load libpww_hip.so once; raise PwwHipError with build instructions if it is missing. (PWW_HIP_LIB may point a whole process at
    libpww_hip_experiments.so: the same ABI plus the experiments section.)"""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        raise PwwHipError(
            "libpww_hip.so not found at %s. Build it with `python paint-with-words-sd_amd/build.py` "
            "(or __graft_entry__.build()). There is no CPU/PyTorch fallback for the PwW kernels." % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    lib.pww_has_experiments.restype = ctypes.c_int
    _lib = _bind(lib, bool(lib.pww_has_experiments()) if hasattr(lib, "pww_has_experiments") else False)
    return _lib


def has_experiments():
    """True if the library `load()` returns carries the experiments section (a process started with PWW_HIP_LIB=...experiments.so)."""
    return bool(load().pww_has_experiments())


def load_experiments():
    """libpww_hip_experiments.so (test / tool infrastructure: round 3's in-launch statistic, the attention + to_out launch, the A/B kernels
    behind PWW_DEBUG). The product never calls this; a missing library raises with the build command."""
    global _exp
    if _exp is not None:
        return _exp
    if has_experiments():
        _exp = load()
        return _exp
    if not os.path.isfile(EXPERIMENTS_LIB_PATH):
        raise PwwHipError("libpww_hip_experiments.so not found at %s: this entry point is not part of the product library. Build it with "
                          "`python paint-with-words-sd_amd/build.py --experiments` (tests: the `experiments_lib` fixture does)." % EXPERIMENTS_LIB_PATH)
    lib = ctypes.CDLL(EXPERIMENTS_LIB_PATH)
    lib.pww_has_experiments.restype = ctypes.c_int
    if not lib.pww_has_experiments():
        raise PwwHipError("%s was not built with -DPWW_EXPERIMENTS=1" % EXPERIMENTS_LIB_PATH)
    _exp = _bind(lib, True)
    return _exp


class experiments:
    """TEST / TOOL scope: inside `with _lib.experiments():` every op of this process runs on libpww_hip_experiments.so (`load()` returns
    it), e.g. a whole request with the compact bias form; the product library is back afterwards."""

    def __enter__(self):
        global _lib
        load()
        self._prev = _lib
        _lib = load_experiments()
        return _lib

    def __exit__(self, *exc):
        global _lib
        _lib = self._prev
        return False


def check(rc, what, lib=None):
    if rc != PWW_OK:
        msg = (lib or load()).pww_last_error().decode("utf-8", "replace")
        raise PwwHipError("%s failed (rc=%d): %s" % (what, rc, msg))


def device_arch():
    buf = ctypes.create_string_buffer(64)
    check(load().pww_device_arch(buf, 64), "pww_device_arch")
    return buf.value.decode()
