"""Torch-facing wrappers of the C ABI: tensors in, tensors out, launched on torch's current stream.

PyTorch here is plumbing (device memory from the caching allocator, the current HIP stream, dtype /
stride bookkeeping); all arithmetic happens in libpww_hip.so. Every function requires tensors on a
HIP device and raises if the library is missing -- there is no fallback path.
"""
import ctypes

import torch

from . import _lib
from ._lib import AttnDesc, CrossOpts, QprojDesc, Region, PwwHipError

_DT = {torch.float16: _lib.DTYPE_F16, torch.bfloat16: _lib.DTYPE_BF16}


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def _require_gpu(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise PwwHipError("pww_hip ops need tensors on a HIP device (got %s); there is no CPU path" % t.device)


def _rows_ok(t):
    """last dim contiguous, rows 16-byte aligned"""
    return t.stride(-1) == 1 and t.stride(0) % 8 == 0 and t.stride(1) % 8 == 0 and t.data_ptr() % 16 == 0


def _desc(q, k, v, o, heads, scale):
    if q.dim() != 3 or k.dim() != 3 or (v is not None and v.dim() != 3):
        raise PwwHipError("q/k/v must be [batch, tokens, heads*dim] tensors")
    B, N, C = q.shape
    M = k.shape[1]
    if C % heads:
        raise PwwHipError("channels %d not divisible by heads %d" % (C, heads))
    if k.shape[2] != C or k.shape[0] not in (1, B):
        raise PwwHipError("key shape %s does not match query shape %s" % (tuple(k.shape), tuple(q.shape)))
    if v is not None and tuple(v.shape) != tuple(k.shape):
        raise PwwHipError("value shape %s differs from key shape %s" % (tuple(v.shape), tuple(k.shape)))
    if M < 1 or N < 1 or B < 1:
        raise PwwHipError("empty attention problem (B=%d N=%d M=%d)" % (B, N, M))
    D = C // heads
    d = AttnDesc()
    d.dtype = _DT[q.dtype]
    d.B, d.H, d.N, d.M, d.D = B, heads, N, M, D
    d.q_stride[:] = [q.stride(0), D, q.stride(1)]
    kb = k.stride(0) if k.shape[0] == B and B > 1 else 0     # a batch-1 context is shared by every image
    d.k_stride[:] = [kb, D, k.stride(1)]
    if v is not None:
        d.v_stride[:] = [v.stride(0) if v.shape[0] == B and B > 1 else 0, D, v.stride(1)]
    if o is not None:
        d.o_stride[:] = [o.stride(0), D, o.stride(1)]
    d.scale = float(scale)
    return d


def _prep(t):
    if t.dtype not in _DT:
        raise PwwHipError("dtype %s unsupported (float16 / bfloat16)" % t.dtype)
    return t if _rows_ok(t) else t.contiguous()


STAT_NONE, STAT_MAX, STAT_MIN, STAT_MEAN, STAT_STD, STAT_ABSMAX, STAT_ALL = 0, 1, 2, 3, 4, 5, 6   # PWW_STAT_* of include/pww_hip.h


FUSED_MAX_KEYS = 128   # pww_cross_attn_fwd_fused: one K/V stage


class FusedScratch:
    """EXPERIMENTS LIBRARY (libpww_hip_experiments.so; tests and A/B tools): device buffers of pww_cross_attn_fwd_fused for one call site (one attention layer): the persistent state words
    (zeroed ONCE here -- the kernel leaves them zero after every launch, so captured hipGraphs replay without a memset
    node) and the scratch of the two-launch path. Allocated outside stream capture (the first, eager call of a layer
    creates them) and kept alive by their owner, because captured graphs hold their addresses: a buffer that has to grow
    is RETIRED, never freed (graphs captured for the smaller geometry keep replaying into it)."""

    def __init__(self):
        self.state = None
        self.ws = None
        self._err_index = 0
        self._retired = []
        self._err_sites = {}       # id(state buffer) -> (buffer, {error-word indices of every geometry launched into it})

    def _alloc(self, current, nbytes, device, zero):
        if current is not None and current.device == device and current.numel() * 8 >= nbytes:
            return current
        if torch.cuda.is_current_stream_capturing():
            raise PwwHipError("fused cross-attention buffers must exist before hipGraph capture (run one eager call first)")
        if current is not None:
            self._retired.append(current)
        n = (nbytes + 7) // 8
        return (torch.zeros if zero else torch.empty)((n,), dtype=torch.int64, device=device)

    def ensure(self, lib, d, device):
        lib = _lib.load_experiments()      # (the hand-off form's size queries live where the launch lives)
        self.state = self._alloc(self.state, int(lib.pww_cross_fused_state_bytes(ctypes.byref(d))), device, True)
        self.ws = self._alloc(self.ws, int(lib.pww_cross_fused_workspace_bytes(ctypes.byref(d))), device, False)
        self._err_index = d.B * d.H + d.B
        self._err_sites.setdefault(id(self.state), (self.state, set()))[1].add(self._err_index)
        return self.state, self.ws

    def error_word(self):
        """The error word of the last call's geometry as a 0-dim device tensor (non-zero: a hand-off inside the fused kernel
        timed out and the affected outputs are NaN), or None before the first call. No synchronisation."""
        return None if self.state is None else self.state.view(torch.int32)[self._err_index]

    def error_words(self):
        """The error words of EVERY geometry this call site was ever launched with, retired buffers included (graphs captured for
        a smaller geometry keep replaying into them; a layer that alternates between batch geometries has its word at another
        index each time): a list of 0-dim device tensors. No synchronisation."""
        return [buf.view(torch.int32)[i] for buf, idxs in self._err_sites.values() for i in sorted(idxs)]

    def error(self):
        """True if a hand-off timed out (synchronises). The state words must then be re-zeroed: `reset()`."""
        words = self.error_words()
        return bool(words) and bool(torch.stack(words).ne(0).any().item())

    def reset(self):
        for t in [self.state] + self._retired:
            if t is not None:
                t.zero_()


class FusedErrorWatch:
    """check_fused_errors without stopping the host: after a request's last launch the error words of all layers are reduced on the
    device and copied to a pinned host flag behind an event (`post`); `poll` looks at the flags whose event has completed -- or
    waits for all of them -- and raises like check_fused_errors. The sampler posts at the end of every request and polls at the
    start of the next one, the PIL-returning entry points poll with wait=True after the decode they synchronise on anyway: the
    host keeps running ahead of the GPU between requests (a synchronous check after every image costs ~3 % images/s at batch 1:
    the next request's conditioning would start only when the GPU is idle)."""

    def __init__(self):
        self.pending = []

    def post(self, modules):
        scratches = [m.__dict__["_pww_fused_scratch"] for m in modules if "_pww_fused_scratch" in getattr(m, "__dict__", {})]
        words = [w for s in scratches for w in s.error_words()]
        if not words:
            return False
        flag = torch.stack(words).ne(0).any().to(torch.int32)
        host = torch.empty((), dtype=torch.int32).pin_memory()
        host.copy_(flag, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self.pending.append((ev, host, scratches))
        return True

    def poll(self, wait=False):
        while self.pending:
            ev, host, scratches = self.pending[0]
            if wait:
                ev.synchronize()
            elif not ev.query():
                return
            self.pending.pop(0)
            if int(host.item()) != 0:
                torch.cuda.synchronize()
                for s in scratches:
                    s.reset()
                self.pending.clear()
                raise PwwHipError(_TIMEOUT_MESSAGE)


_TIMEOUT_MESSAGE = ("a fused cross-attention launch timed out waiting for its own workgroups (the GPU is shared with another process or "
                    "stream?); its outputs are NaN and were discarded. State re-zeroed. Set PWW_FUSED_CROSS=0 to use the two-launch "
                    "path on a shared device.")


def check_fused_errors(modules):
    """One device -> host read for a whole request: raise PwwHipError if any fused cross-attention launch of the given
    modules' scratch buffers timed out in its hand-off (the launch needs every workgroup resident at once: another process or
    stream holding compute units can starve it; PWW_FUSED_CROSS=0 selects the two-launch path). Dirty state is re-zeroed so
    that the next request starts clean."""
    scratches = [m.__dict__["_pww_fused_scratch"] for m in modules if "_pww_fused_scratch" in getattr(m, "__dict__", {})]
    words = [w for s in scratches for w in s.error_words()]
    if not words:
        return
    if bool(torch.stack(words).ne(0).any().item()):
        torch.cuda.synchronize()
        for s in scratches:
            s.reset()
        raise PwwHipError(_TIMEOUT_MESSAGE)


def _cross_opts(B, N, coeff_dev, bias_cols, compact, keep, gated=0):
    """pww_cross_opts_t for the *_ex entry points (None if nothing optional is asked for)."""
    if coeff_dev is None and not bias_cols and compact is None and not gated:
        return None
    op = CrossOpts()
    op.size = ctypes.sizeof(CrossOpts)
    op.bias_cols = int(bias_cols or 0)
    op.gated_images = int(gated or 0)
    if coeff_dev is not None:
        if coeff_dev.dtype != torch.float32 or coeff_dev.numel() != 1 or not coeff_dev.is_cuda:
            raise PwwHipError("coeff_dev must be a one-element float32 device tensor")
        op.coeff_scalar_dev = coeff_dev.data_ptr()
    if compact is not None:
        wc, idx = compact
        if wc.dtype != torch.float32 or idx.dtype != torch.int32 or wc.stride(-1) != 1 or idx.stride(-1) != 1:
            raise PwwHipError("compact bias: float32 values and int32 column indices, innermost stride 1")
        if wc.dim() == 2:
            wc = wc.unsqueeze(0)
        if idx.dim() == 1:
            idx = idx.unsqueeze(0)
        R = wc.shape[-1]
        if wc.shape[1] != N or wc.shape[0] not in (1, B) or idx.shape[-1] != R or idx.shape[0] not in (1, B):
            raise PwwHipError("compact bias shapes %s / %s do not fit B=%d N=%d" % (tuple(wc.shape), tuple(idx.shape), B, N))
        op.bias_compact, op.col_idx, op.R = wc.data_ptr(), idx.data_ptr(), R
        op.compact_stride[:] = [wc.stride(0) if wc.shape[0] == B and B > 1 else 0, wc.stride(1)]
        op.col_idx_stride = idx.stride(0) if idx.shape[0] == B and B > 1 else 0
        keep.extend([wc, idx])
    return op


COMPACT_MAX_R = 32     # pww_cross.hip: the compact form holds at most 32 non-zero columns


def attention(q, k, v, heads, scale, bias=None, bias_coeff=None, stat=None, scratch=None, stats_out=None, coeff_dev=None,
              bias_cols=0, compact=None, gated=0, parts=None):
    """softmax((Q K^T + c * bias) * scale) V on [B, tokens, heads*D] tensors (diffusers layout, no
    head-split copies). bias: fp32 tensor broadcastable to [B, heads, N, M] or None;
    bias_coeff: optional fp32 [B] device tensor of per-image coefficients (with `stat`: the row gate);
    stat: optional (stats, STAT_* kind, python scalar): the kernel forms c[b] = scalar * stat(stats[b]) * bias_coeff[b]
    itself. `stats` is either the float64 [B,4] tensor of qk_stats (pww_cross_attn_fwd_stat) or None together with a
    FusedScratch in `scratch`: the statistic is then computed in the same launch (pww_cross_attn_fwd_fused, M <= 128).
    With `stat`: coeff_dev = one-element fp32 device tensor that replaces the python scalar when the kernel runs (hipGraph
    replays across denoise steps); bias_cols = columns >= bias_cols of the map are zero; compact = (values [B?, N, R] fp32,
    col_idx [B?, R] int32) the compact form of the same map (fused launch only); gated = the caller's hint that bias_coeff is
    non-zero exactly for the first `gated` images (a CFG-folded batch), 0 = unknown (fused launch only: work distribution).
    parts: float64 [B, nparts, 4] partials of the statistic formed by qproj_stat with q (stat = (None, kind, scalar)): the launch
    folds them at entry (pww_cross_attn_fwd_parts) -- no scratch, no hand-off."""
    _require_gpu(q, k, v, bias, bias_coeff)
    if not (q.dtype == k.dtype == v.dtype):
        raise PwwHipError("q/k/v dtypes differ: %s %s %s" % (q.dtype, k.dtype, v.dtype))
    q, k, v = _prep(q), _prep(k), _prep(v)
    B, N, C = q.shape
    M = k.shape[1]
    out = torch.empty((B, N, C), dtype=q.dtype, device=q.device)
    d = _desc(q, k, v, out, heads, scale)
    lib = _lib.load()
    with torch.cuda.device(q.device):
        if bias is None:
            rc = lib.pww_self_attn_fwd(_ptr(q), _ptr(k), _ptr(v), _ptr(out), ctypes.byref(d), _stream())
            _lib.check(rc, "pww_self_attn_fwd")
        else:
            if bias.dtype != torch.float32:
                bias = bias.float()
            if bias.dim() == 3 and bias.shape[0] == B * heads and B * heads != 1:
                bias = bias.reshape(B, heads, bias.shape[1], bias.shape[2])
            bias = torch.broadcast_to(bias, (B, heads, N, M))  # view: broadcast axes get stride 0
            d.bias_stride[:] = list(bias.stride())
            if bias_coeff is not None:
                bias_coeff = bias_coeff.to(torch.float32).reshape(-1).contiguous()
                if bias_coeff.numel() == 1 and B > 1:
                    bias_coeff = bias_coeff.expand(B).contiguous()
                if bias_coeff.numel() != B:
                    raise PwwHipError("bias_coeff must have B=%d elements" % B)
            if stat is not None and stat[0] is None and stat[1] == STAT_NONE and parts is None and M > FUSED_MAX_KEYS:
                # a statistic-free weight function over a context longer than one K/V stage: nothing to fold, nothing to form -- the
                # general launch (no key limit) with its scalar (or the hipGraph mode's device word) as the coefficient
                _, kind, scalar = stat
                op = _cross_opts(B, N, coeff_dev, 0, None, [])
                rc = lib.pww_cross_attn_fwd_stat_ex(_ptr(q), _ptr(k), _ptr(v), _ptr(out), _ptr(bias), None, int(STAT_NONE),
                                                    float(heads * N * M), float(scalar), _ptr(bias_coeff), ctypes.byref(d),
                                                    ctypes.byref(op) if op is not None else None, _stream())
                _lib.check(rc, "pww_cross_attn_fwd_stat_ex")
            elif stat is not None and stat[0] is None and scratch is None:
                # pass-2-only launch: the statistic's partials came from qproj_stat / qk_parts (none needed for STAT_NONE); nothing in it
                # waits for another workgroup
                _, kind, scalar = stat
                if kind != STAT_NONE and parts is None:
                    raise PwwHipError("a statistic formed outside the attention launch needs its partials (qproj_stat / qk_parts), or a FusedScratch for the in-launch form")
                if parts is not None and (parts.dtype != torch.float64 or parts.dim() != 3 or parts.shape[0] != B or parts.shape[2] != 4 or not parts.is_contiguous()):
                    raise PwwHipError("parts must be a contiguous float64 [B, nparts, 4] tensor")
                if M > FUSED_MAX_KEYS:
                    raise PwwHipError("the pass-2-only cross-attention launch takes at most %d keys (partials over more keys: fold them with "
                                      "fold_parts and pass stat=(stats, kind, scalar))" % FUSED_MAX_KEYS)
                if stats_out is not None and (stats_out.dtype != torch.float64 or tuple(stats_out.shape) != (B, 4) or not stats_out.is_contiguous()):
                    raise PwwHipError("stats_out must be a contiguous float64 [B, 4] tensor")
                keep = []
                if compact is not None and (compact[0].shape[-1] > COMPACT_MAX_R or not _lib.has_experiments()):
                    compact = None          # (the compact form of the map is an experiments-library form: the dense map serves)
                op = _cross_opts(B, N, coeff_dev, bias_cols, compact, keep, gated)
                rc = lib.pww_cross_attn_fwd_parts(_ptr(q), _ptr(k), _ptr(v), _ptr(out), _ptr(bias), int(kind), float(scalar), _ptr(bias_coeff),
                                                  ctypes.byref(d), _ptr(parts), int(parts.shape[1]) if parts is not None else 0, _ptr(stats_out),
                                                  ctypes.byref(op) if op is not None else None, _stream())
                _lib.check(rc, "pww_cross_attn_fwd_parts")
            elif stat is not None and stat[0] is None:
                _, kind, scalar = stat
                if M > FUSED_MAX_KEYS:
                    raise PwwHipError("fused statistic needs a FusedScratch and at most %d keys" % FUSED_MAX_KEYS)
                xlib = _lib.load_experiments()      # round 3's in-launch statistic is not part of the product library
                state, ws = scratch.ensure(xlib, d, q.device)
                if stats_out is not None and (stats_out.dtype != torch.float64 or tuple(stats_out.shape) != (B, 4) or not stats_out.is_contiguous()):
                    raise PwwHipError("stats_out must be a contiguous float64 [B, 4] tensor")
                keep = []
                if compact is not None and compact[0].shape[-1] > COMPACT_MAX_R:
                    compact = None
                op = _cross_opts(B, N, coeff_dev, bias_cols, compact, keep, gated)
                rc = xlib.pww_cross_attn_fwd_fused_ex(_ptr(q), _ptr(k), _ptr(v), _ptr(out), _ptr(bias), int(kind), float(scalar),
                                                      _ptr(bias_coeff), ctypes.byref(d), _ptr(stats_out), _ptr(state), state.numel() * 8,
                                                      _ptr(ws), ws.numel() * 8, ctypes.byref(op) if op is not None else None, _stream())
                _lib.check(rc, "pww_cross_attn_fwd_fused_ex", xlib)
            elif stat is not None:
                stats, kind, scalar = stat
                if stats is not None and (stats.dtype != torch.float64 or tuple(stats.shape) != (B, 4) or not stats.is_contiguous()):
                    raise PwwHipError("stat: stats must be a contiguous float64 [B, 4] tensor")
                op = _cross_opts(B, N, coeff_dev, 0, None, [])
                rc = lib.pww_cross_attn_fwd_stat_ex(_ptr(q), _ptr(k), _ptr(v), _ptr(out), _ptr(bias), _ptr(stats), int(kind),
                                                    float(heads * N * M), float(scalar), _ptr(bias_coeff), ctypes.byref(d),
                                                    ctypes.byref(op) if op is not None else None, _stream())
                _lib.check(rc, "pww_cross_attn_fwd_stat_ex")
            else:
                rc = lib.pww_cross_attn_fwd(_ptr(q), _ptr(k), _ptr(v), _ptr(out), _ptr(bias), _ptr(bias_coeff),
                                            ctypes.byref(d), _stream())
                _lib.check(rc, "pww_cross_attn_fwd")
    return out


def _bias_view(bias, B, heads, N, M):
    if bias.dtype != torch.float32:
        bias = bias.float()
    if bias.dim() == 3 and bias.shape[0] == B * heads and B * heads != 1:
        bias = bias.reshape(B, heads, bias.shape[1], bias.shape[2])
    return torch.broadcast_to(bias, (B, heads, N, M))  # view: broadcast axes get stride 0


def attention_out_supported(q, k, heads, bias, weight, bias_cols=0):
    """EXPERIMENTS LIBRARY. True if attention_out (cross-attention + to_out in one launch) takes this problem: C = heads * D = 320 with D <= 64, 64 <= M <= 128,
    a dense fp32 map shared by the heads with at most 64 non-zero columns, weight [C, C] of q's dtype."""
    if q.dtype not in _DT or q.dim() != 3 or k.dim() != 3 or bias is None or not torch.is_tensor(weight):
        return False
    B, N, C = q.shape
    M = k.shape[1]
    if tuple(weight.shape) != (C, C) or weight.dtype != q.dtype or not weight.is_contiguous() or C % heads:
        return False
    try:
        bv = _bias_view(bias, B, heads, N, M)
    except RuntimeError:
        return False
    d = AttnDesc()
    d.dtype = _DT[q.dtype]
    d.B, d.H, d.N, d.M, d.D = B, heads, N, M, C // heads
    d.q_stride[:] = [q.stride(0), C // heads, q.stride(1)]
    d.bias_stride[:] = list(bv.stride())
    return bool(_lib.load_experiments().pww_cross_attn_out_supported(ctypes.byref(d), C, int(bias_cols)))


def attention_out(q, k, v, heads, scale, bias, weight, weight_bias=None, residual=None, bias_coeff=None, stat=None, parts=None, stats_out=None,
                  coeff_dev=None, bias_cols=0, gated=0):
    """EXPERIMENTS LIBRARY (libpww_hip_experiments.so: measured slower than the two-launch route at batch 1, a tie at 16 rows --
    profiles/r05_to_out_epilogue.md; not a product route since round 6).
    linear(attention(q, k, v, ...), weight, weight_bias) [+ residual] in ONE launch (pww_cross_attn_fwd_parts_out: reference
    paint_with_words.py:106-123): the pass-2-only cross-attention of `attention(..., stat=(None, kind, scalar), parts=parts)` with the
    layer's to_out projection applied to the heads' outputs while they are still in registers. stat = (None, kind, scalar) or None
    (= a plain `bias_coeff * bias`). Ask attention_out_supported first; anything else raises."""
    _require_gpu(q, k, v, bias, bias_coeff, weight, weight_bias, residual)
    if not (q.dtype == k.dtype == v.dtype == weight.dtype):
        raise PwwHipError("q/k/v/weight dtypes differ: %s %s %s %s" % (q.dtype, k.dtype, v.dtype, weight.dtype))
    q, k, v = _prep(q), _prep(k), _prep(v)
    B, N, C = q.shape
    M = k.shape[1]
    if tuple(weight.shape) != (C, C) or not weight.is_contiguous():
        raise PwwHipError("attention_out: weight must be a contiguous [%d, %d] tensor" % (C, C))
    if weight_bias is not None and (weight_bias.dtype != q.dtype or tuple(weight_bias.shape) != (C,) or not weight_bias.is_contiguous()):
        raise PwwHipError("attention_out: weight_bias must be a contiguous [%d] tensor of q's dtype" % C)
    if residual is not None and (residual.dtype != q.dtype or tuple(residual.shape) != (B, N, C) or residual.stride(2) != 1):
        raise PwwHipError("attention_out: residual must be a [B, N, C] tensor of q's dtype with unit channel stride")
    out = torch.empty((B, N, C), dtype=q.dtype, device=q.device)
    d = _desc(q, k, v, out, heads, scale)
    bias = _bias_view(bias, B, heads, N, M)
    d.bias_stride[:] = list(bias.stride())
    if bias_coeff is not None:
        bias_coeff = bias_coeff.to(torch.float32).reshape(-1).contiguous()
        if bias_coeff.numel() == 1 and B > 1:
            bias_coeff = bias_coeff.expand(B).contiguous()
        if bias_coeff.numel() != B:
            raise PwwHipError("bias_coeff must have B=%d elements" % B)
    kind, scalar = (STAT_NONE, 1.0) if stat is None else (stat[1], stat[2])
    if stat is not None and stat[0] is not None:
        raise PwwHipError("attention_out folds partials (stat = (None, kind, scalar), parts = qproj_stat / qk_parts output)")
    if kind != STAT_NONE and parts is None:
        raise PwwHipError("attention_out: a statistic needs its partials (qproj_stat / qk_parts)")
    if parts is not None and (parts.dtype != torch.float64 or parts.dim() != 3 or parts.shape[0] != B or parts.shape[2] != 4 or not parts.is_contiguous()):
        raise PwwHipError("parts must be a contiguous float64 [B, nparts, 4] tensor")
    if stats_out is not None and (stats_out.dtype != torch.float64 or tuple(stats_out.shape) != (B, 4) or not stats_out.is_contiguous()):
        raise PwwHipError("stats_out must be a contiguous float64 [B, 4] tensor")
    keep = []
    op = _cross_opts(B, N, coeff_dev, bias_cols, None, keep, gated)
    rs = (ctypes.c_int64 * 2)(residual.stride(0), residual.stride(1)) if residual is not None else None
    lib = _lib.load_experiments()
    with torch.cuda.device(q.device):
        rc = lib.pww_cross_attn_fwd_parts_out(_ptr(q), _ptr(k), _ptr(v), _ptr(out), _ptr(bias), int(kind), float(scalar), _ptr(bias_coeff), ctypes.byref(d),
                                              _ptr(parts), int(parts.shape[1]) if parts is not None else 0, _ptr(stats_out),
                                              ctypes.byref(op) if op is not None else None, _ptr(weight), _ptr(weight_bias), _ptr(residual), rs, _stream())
    _lib.check(rc, "pww_cross_attn_fwd_parts_out", lib)
    return out


def _qproj_desc(x, weight, k, heads):
    if x.dim() != 3 or k.dim() != 3 or weight.dim() != 2:
        raise PwwHipError("qproj: x [B, N, Cin], weight [C, Cin], k [B or 1, M, C]")
    B, N, Cin = x.shape
    C = weight.shape[0]
    if weight.shape[1] != Cin or k.shape[2] != C or k.shape[0] not in (1, B) or C % heads:
        raise PwwHipError("qproj: shapes do not fit (x %s, weight %s, k %s, heads %d)" % (tuple(x.shape), tuple(weight.shape), tuple(k.shape), heads))
    d = QprojDesc()
    d.dtype = _DT[x.dtype]
    d.B, d.N, d.Cin, d.H, d.D, d.M = B, N, Cin, heads, C // heads, k.shape[1]
    d.x_stride[:] = [x.stride(0), x.stride(1)]
    d.k_stride[:] = [k.stride(0) if k.shape[0] == B and B > 1 else 0, k.stride(1)]
    return d


def qproj_parts(x, weight, k, heads):
    """Partials per image pww_qproj_stat forms for this problem, 0 if the shape is not supported (the caller then keeps its own
    projection and the in-launch statistic)."""
    if x.dtype not in _DT or not (x.dtype == weight.dtype == k.dtype) or k.shape[1] > FUSED_MAX_KEYS:
        return 0
    d = _qproj_desc(x, weight, k, heads)
    d.q_stride[:] = [x.shape[1] * weight.shape[0], weight.shape[0]]
    return int(_lib.load().pww_qproj_parts(ctypes.byref(d)))


def qproj_stat(x, weight, k, heads, kind, gate=None):
    """q = x @ weight.T (a bias-free to_q, paint_with_words.py:76) together with the partials of the per-image score statistic
    of q k^T (:87; kind = STAT_*: only the fields that statistic needs are formed) -- one launch, the statistic comes out of the GEMM's
    epilogue. Returns (q [B, N, C], parts float64 [B, nparts, 4]). gate: optional fp32 [B]; images with gate 0 get no partials."""
    _require_gpu(x, weight, k, gate)
    if not (x.dtype == weight.dtype == k.dtype):
        raise PwwHipError("qproj: dtypes differ: %s %s %s" % (x.dtype, weight.dtype, k.dtype))
    x, k = _prep(x), _prep(k)
    weight = weight if weight.is_contiguous() else weight.contiguous()
    B, N, Cin = x.shape
    C = weight.shape[0]
    q = torch.empty((B, N, C), dtype=x.dtype, device=x.device)
    d = _qproj_desc(x, weight, k, heads)
    d.q_stride[:] = [q.stride(0), q.stride(1)]
    lib = _lib.load()
    nparts = int(lib.pww_qproj_parts(ctypes.byref(d)))
    if nparts <= 0:
        raise PwwHipError("qproj_stat: unsupported shape (C = %d, head dim %d, Cin = %d): ask qproj_parts() first" % (C, C // heads, Cin))
    parts = torch.empty((B, nparts, 4), dtype=torch.float64, device=x.device)
    if gate is not None:
        gate = gate.to(torch.float32).reshape(-1).contiguous()
        if gate.numel() != B:
            raise PwwHipError("gate must have B=%d elements" % B)
    with torch.cuda.device(x.device):
        _lib.check(lib.pww_qproj_stat(_ptr(x), _ptr(weight), _ptr(q), _ptr(k), _ptr(gate), ctypes.byref(d), int(kind), _ptr(parts),
                                      parts.numel() * 8, _stream()), "pww_qproj_stat")
    return q, parts


def qk_parts(q, k, heads, kind, gate=None, gated=0):
    """Partials of the per-image score statistic of q k^T over a FINISHED q (paint_with_words.py:87 + the reduction weight_function
    applies): float64 [B, nparts, 4], the input of attention(..., parts=...). One small launch (pww_qk_parts): the layers whose to_q
    stays the stock GEMM. gate: optional fp32 [B] (images with gate 0 get no partials); gated: hint that exactly the first `gated`
    images have a non-zero gate."""
    _require_gpu(q, k, gate)
    if q.dtype != k.dtype:
        raise PwwHipError("qk_parts: dtypes differ: %s %s" % (q.dtype, k.dtype))
    q, k = _prep(q), _prep(k)
    d = _desc(q, k, None, None, heads, 1.0)
    lib = _lib.load()
    nparts = int(lib.pww_qk_parts_count(ctypes.byref(d)))
    if nparts <= 0:
        raise PwwHipError("qk_parts: unsupported problem %s x %s, %d heads" % (tuple(q.shape), tuple(k.shape), heads))
    B = q.shape[0]
    parts = torch.empty((B, nparts, 4), dtype=torch.float64, device=q.device)
    if gate is not None:
        gate = gate.to(torch.float32).reshape(-1).contiguous()
        if gate.numel() != B:
            raise PwwHipError("gate must have B=%d elements" % B)
    with torch.cuda.device(q.device):
        _lib.check(lib.pww_qk_parts(_ptr(q), _ptr(k), _ptr(gate), ctypes.byref(d), int(kind), int(gated or 0), _ptr(parts), parts.numel() * 8, _stream()),
                   "pww_qk_parts")
    return parts


def fold_parts(parts):
    """[B, nparts, 4] partials -> [B, 4] statistics (max, min, sum, sum of squares), as the attention kernel folds them."""
    return torch.stack([parts[:, :, 0].amax(1), parts[:, :, 1].amin(1), parts[:, :, 2].sum(1), parts[:, :, 3].sum(1)], dim=1)


def qk_stats(q, k, heads):
    """Per-image statistics of the raw scores Q K^T over heads x rows x keys: float64 [B, 4] =
    (max, min, sum, sum of squares). q: [B, N, heads*D], k: [B, M, heads*D]."""
    _require_gpu(q, k)
    q, k = _prep(q), _prep(k)
    stats = torch.empty((q.shape[0], 4), dtype=torch.float64, device=q.device)
    d = _desc(q, k, None, None, heads, 1.0)
    lib = _lib.load()
    nbytes = int(lib.pww_workspace_bytes(ctypes.byref(d)))
    ws = torch.empty(((nbytes + 7) // 8,), dtype=torch.float64, device=q.device)   # caching allocator: stream-ordered reuse
    with torch.cuda.device(q.device):
        _lib.check(lib.pww_qk_reduce(_ptr(q), _ptr(k), ctypes.byref(d), _ptr(stats), _ptr(ws), ws.numel() * 8, _stream()),
                   "pww_qk_reduce")
    return stats


def _regions_tensor(regions, device):
    """[(r, g, b, strength)] -> device byte tensor laid out as struct pww_region[R]."""
    arr = (Region * len(regions))()
    for i, (r, g, b, s) in enumerate(regions):
        arr[i] = Region(int(r), int(g), int(b), 0, float(s))
    raw = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8)
    return raw.to(device)


def _csr(cols, device):
    ptr, reg = [0], []
    for lst in cols:
        reg.extend(int(x) for x in lst)
        ptr.append(len(reg))
    col_ptr = torch.tensor(ptr, dtype=torch.int32).to(device)
    col_reg = torch.tensor(reg if reg else [0], dtype=torch.int32).to(device)
    return col_ptr, col_reg


def round_half_up_div(a, ratio):
    return (2 * a + ratio) // (2 * ratio)


def mask_build(rgb, regions, cols, ratios=(8, 16, 32, 64)):
    """RGB color map (uint8 [H, W, 3] device tensor) -> {ratio: fp32 [Hr*Wr, T]} token weight maps.
    regions: [(r, g, b, strength)]; cols: per prompt position, the region ordinals added to it."""
    _require_gpu(rgb)
    if rgb.dtype != torch.uint8 or rgb.dim() != 3 or rgb.shape[2] != 3:
        raise PwwHipError("rgb must be uint8 [H, W, 3]")
    rgb = rgb.contiguous()
    H, W = rgb.shape[:2]
    T = len(cols)
    dev = rgb.device
    regs = _regions_tensor(regions, dev)
    col_ptr, col_reg = _csr(cols, dev)
    outs = {r: torch.empty((round_half_up_div(H, r) * round_half_up_div(W, r), T), dtype=torch.float32, device=dev)
            for r in ratios}
    lib = _lib.load()
    with torch.cuda.device(dev):
        if tuple(ratios) == (8, 16, 32, 64):
            rc = lib.pww_mask_build(_ptr(rgb), H, W, _ptr(regs), len(regions), _ptr(col_ptr), _ptr(col_reg), T,
                                    _ptr(outs[8]), _ptr(outs[16]), _ptr(outs[32]), _ptr(outs[64]), _stream())
            _lib.check(rc, "pww_mask_build")
        else:
            for r in ratios:
                rc = lib.pww_mask_build_rgb(_ptr(rgb), H, W, _ptr(regs), len(regions), _ptr(col_ptr), _ptr(col_reg),
                                            T, r, _ptr(outs[r]), _stream())
                _lib.check(rc, "pww_mask_build_rgb")
    return outs


def mask_build_f32(masks, cols, ratios=(8, 16, 32, 64)):
    """Same accumulation from R float32 region masks [R, H, W] (strength-scaled, possibly blurred)."""
    _require_gpu(masks)
    masks = masks.to(torch.float32).contiguous()
    R, H, W = masks.shape
    T = len(cols)
    dev = masks.device
    col_ptr, col_reg = _csr(cols, dev)
    outs = {}
    lib = _lib.load()
    with torch.cuda.device(dev):
        for r in ratios:
            outs[r] = torch.empty((round_half_up_div(H, r) * round_half_up_div(W, r), T), dtype=torch.float32, device=dev)
        rest = list(ratios)
        if all(r in outs for r in (8, 16, 32, 64)):       # the four maps of a request: ONE launch
            rc = lib.pww_mask_build_f32_levels(_ptr(masks), H, W, R, _ptr(col_ptr), _ptr(col_reg), T, _ptr(outs[8]), _ptr(outs[16]),
                                               _ptr(outs[32]), _ptr(outs[64]), _stream())
            _lib.check(rc, "pww_mask_build_f32_levels")
            rest = [r for r in ratios if r not in (8, 16, 32, 64)]
        for r in rest:
            rc = lib.pww_mask_build_f32(_ptr(masks), H, W, R, _ptr(col_ptr), _ptr(col_reg), T, r, _ptr(outs[r]), _stream())
            _lib.check(rc, "pww_mask_build_f32")
    return outs


def resize_tokens(w_orig, n_tokens):
    """CROSS_ATTENTION_WEIGHT_ORIG [H, W, T] (fp32, device) -> [n_tokens, T]: the reference's fallback resize
    (paint_with_words.py:96-101). The intermediate size is computed like torch does for `scale_factor`."""
    import math
    _require_gpu(w_orig)
    if w_orig.dim() != 3 or w_orig.dtype != torch.float32:
        raise PwwHipError("resize_tokens needs an fp32 [H, W, T] map")
    w_orig = w_orig.contiguous()
    H, W, T = w_orig.shape
    ratio = math.sqrt(H * W / n_tokens)
    oh, ow = int(math.floor(H * (1 / ratio))), int(math.floor(W * (1 / ratio)))
    out = torch.empty((n_tokens, T), dtype=torch.float32, device=w_orig.device)
    with torch.cuda.device(w_orig.device):
        _lib.check(_lib.load().pww_resize_tokens(_ptr(w_orig), H, W, T, oh, ow, int(n_tokens), _ptr(out), _stream()), "pww_resize_tokens")
    return out


def gaussian_kernel1d(sigma, ksize=39):
    """torchvision's _get_gaussian_kernel1d in fp32: exp(-0.5 (x / sigma)^2) on linspace(-h, h, ksize), normalised."""
    half = (ksize - 1) * 0.5
    x = torch.linspace(-half, half, steps=ksize, dtype=torch.float32)
    pdf = torch.exp(-0.5 * (x / sigma).pow(2))
    return pdf / pdf.sum()


def gauss_blur(mask, sigma, ksize=39):
    """GaussianBlur(ksize x ksize, sigma) with reflect padding of one fp32 [H, W] device mask (paint_with_words.py:307-312)."""
    _require_gpu(mask)
    if mask.dim() != 2:
        raise PwwHipError("gauss_blur needs an [H, W] mask")
    mask = mask.to(torch.float32).contiguous()
    H, W = mask.shape
    k = gaussian_kernel1d(float(sigma), ksize).to(mask.device)
    tmp = torch.empty((H, W), dtype=torch.float64, device=mask.device)
    out = torch.empty_like(mask)
    with torch.cuda.device(mask.device):
        _lib.check(_lib.load().pww_gauss_blur(_ptr(mask), _ptr(out), H, W, _ptr(k), int(ksize), _ptr(tmp), _stream()), "pww_gauss_blur")
    return out


def inpaint_prep(rgb, mask, lat_h, lat_w):
    """uint8 init image [H, W, 3] + uint8 mask [H, W] (device) -> (mask [1,1,H,W], masked_image [1,3,H,W], latent-size
    mask [1,1,h,w]) fp32, as paint_with_words_inpaint.py:92-106 / :115 compute them."""
    _require_gpu(rgb, mask)
    if rgb.dtype != torch.uint8 or mask.dtype != torch.uint8 or rgb.dim() != 3 or rgb.shape[2] != 3 or tuple(mask.shape) != tuple(rgb.shape[:2]):
        raise PwwHipError("inpaint_prep needs uint8 rgb [H, W, 3] and uint8 mask [H, W]")
    rgb, mask = rgb.contiguous(), mask.contiguous()
    H, W = mask.shape
    dev = rgb.device
    m = torch.empty((1, 1, H, W), dtype=torch.float32, device=dev)
    masked = torch.empty((1, 3, H, W), dtype=torch.float32, device=dev)
    ml = torch.empty((1, 1, lat_h, lat_w), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        _lib.check(_lib.load().pww_inpaint_prep(_ptr(rgb), _ptr(mask), H, W, int(lat_h), int(lat_w), _ptr(m), _ptr(masked), _ptr(ml), _stream()),
                   "pww_inpaint_prep")
    return m, masked, ml


def store_f32(dst, values):
    """dst[:len(values)] = values (Python floats, at most 64) as ONE asynchronous launch on the current stream: the values ride
    in the kernel arguments -- no host staging buffer that a later call could overwrite, no synchronous copy."""
    _require_gpu(dst)
    n = len(values)
    if dst.dtype != torch.float32 or not dst.is_contiguous() or dst.numel() < n:
        raise PwwHipError("store_f32 needs a contiguous float32 device tensor with room for %d values" % n)
    arr = (ctypes.c_float * n)(*[float(v) for v in values])
    with torch.cuda.device(dst.device):
        _lib.check(_lib.load().pww_store_f32(_ptr(dst), arr, n, _stream()), "pww_store_f32")


def cfg_combine(cond, uncond, guidance_scale):
    """uncond + g * (cond - uncond) in fp32 (returns fp32)."""
    _require_gpu(cond, uncond)
    if cond.dtype != uncond.dtype or cond.dtype not in _DT or cond.shape != uncond.shape:
        raise PwwHipError("cfg_combine needs two same-shape float16/bfloat16 tensors")
    cond, uncond = cond.contiguous(), uncond.contiguous()
    out = torch.empty(cond.shape, dtype=torch.float32, device=cond.device)
    with torch.cuda.device(cond.device):
        _lib.check(_lib.load().pww_cfg_combine(_ptr(cond), _ptr(uncond), float(guidance_scale), _ptr(out),
                                               cond.numel(), _DT[cond.dtype], _stream()), "pww_cfg_combine")
    return out


# ---- GroupNorm (+ per-(image, channel) addend, + SiLU) of the blocks that call the attention path ------------------------------------
def group_norm(x, num_groups, weight=None, bias=None, eps=1e-5, add=None, act=None, workspace=None, out=None, pre_bias=None):
    """act(GroupNorm(x + add[:, :, None, None])) for a [B, C, H, W] float16 / bfloat16 tensor in NCHW-contiguous or channels_last memory
    format (the output has x's format); `add` [B, C] (unit stride along C; rows may be views into a wider tensor) or None; `pre_bias` [C] or
    None: a per-channel addend applied, and rounded, before `add` (the bias of the convolution that produced x); act None | "silu". SURVEY.md section 8 row a17 (diffusers 0.10.0 ResnetBlock2D / Transformer2DModel, the callers of the patched CrossAttention).
    The scratch for the partial sums (<= 1 MB; none for the single-launch form) comes from the caching allocator per call -- stream-ordered,
    so concurrent streams never share it and a hipGraph capture takes it from the graph's pool -- unless the caller passes `workspace`."""
    _require_gpu(x)
    if x.dim() != 4 or x.dtype not in _DT:
        raise PwwHipError("group_norm needs a 4-d float16/bfloat16 tensor (got %s %s)" % (tuple(x.shape), x.dtype))
    B, C, H, W = x.shape
    if x.is_contiguous():
        layout = _lib.LAYOUT_NCHW
    elif x.is_contiguous(memory_format=torch.channels_last):
        layout = _lib.LAYOUT_NHWC
    else:
        x = x.contiguous()
        layout = _lib.LAYOUT_NCHW
    if out is None:
        out = torch.empty_like(x)     # (preserves the memory format)
    elif out.shape != x.shape or out.dtype != x.dtype or out.stride() != x.stride():
        raise PwwHipError("group_norm: `out` must have x's shape, dtype and strides")
    for name, t, shape in (("weight", weight, (C,)), ("bias", bias, (C,)), ("add", add, (B, C)), ("pre_bias", pre_bias, (C,))):
        if t is not None and (t.dtype != x.dtype or tuple(t.shape) != shape or t.device != x.device):
            raise PwwHipError("group_norm: `%s` must be %s %s on %s (got %s %s)" % (name, x.dtype, shape, x.device, t.dtype, tuple(t.shape)))
    weight = weight.contiguous() if weight is not None else None
    bias = bias.contiguous() if bias is not None else None
    pre_bias = pre_bias.contiguous() if pre_bias is not None else None
    add_stride = 0
    if add is not None:
        if add.stride(1) != 1 or add.stride(0) % 8 != 0 or add.data_ptr() % 16 != 0:
            add = add.contiguous()
        add_stride = add.stride(0) if B > 1 else C
    if act not in (None, "silu"):
        raise PwwHipError("group_norm: act must be None or 'silu'")
    d = _lib.GnDesc(_DT[x.dtype], layout, B, C, H * W, int(num_groups), float(eps), _lib.ACT_SILU if act == "silu" else _lib.ACT_NONE, add_stride, 0)
    lib = _lib.load()
    need = lib.pww_group_norm_workspace_bytes(ctypes.byref(d))
    if need == 0:
        raise PwwHipError("group_norm: unsupported shape B %d C %d HW %d groups %d (C and H*W multiples of 8, C %% groups == 0, "
                          "groups <= 32, C <= 4096 in channels_last)" % (B, C, H * W, num_groups))
    ws = workspace if workspace is not None else torch.empty(int(need), dtype=torch.uint8, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(lib.pww_group_norm_fwd(_ptr(x), _ptr(pre_bias) if pre_bias is not None else None, _ptr(add) if add is not None else None,
                                          _ptr(weight) if weight is not None else None,
                                          _ptr(bias) if bias is not None else None, _ptr(out), ctypes.byref(d), _ptr(ws), ws.numel(), _stream()),
                   "pww_group_norm_fwd")
    return out


def _rows(t, name):
    """[..., C] tensor as rows: (rows, C, row stride) -- unit stride along C, ONE stride between consecutive rows."""
    if t.dtype not in _DT or t.dim() < 2:
        raise PwwHipError("%s: a float16/bfloat16 tensor of at least two dims is needed (got %s %s)" % (name, t.dtype, tuple(t.shape)))
    if t.stride(-1) != 1 or t.stride(-2) % 8 != 0 or t.data_ptr() % 16 != 0:
        return None                                 # (the caller makes it contiguous: e.g. the [B, HW, C] view of an NCHW tensor)
    C = t.shape[-1]
    rows = t.numel() // C
    stride = t.stride(-2)
    exp = stride
    for d in range(t.dim() - 2, -1, -1):          # every leading dim must continue the same row pitch
        if t.shape[d] != 1 and t.stride(d) != exp:
            return None
        exp *= t.shape[d]
    return rows, C, stride


def add_layer_norm(x, weight, bias, eps, a=None, post_bias=None):
    """LayerNorm over the last dim of x ([..., C]); with `a` (same shape): s = a + x, returns (s, LayerNorm(s)) from ONE launch -- the
    `attn(norm(h)) + h` of diffusers' BasicTransformerBlock together with the block's next norm. Without `a`: returns LayerNorm(x).
    post_bias ([C], with `a`): the returned sum is s + post_bias (the LayerNorm still is that of s): the bias of the GEMM whose output the
    caller adds to it next, through that GEMM's C operand."""
    _require_gpu(x)
    rx = _rows(x, "add_layer_norm")
    if rx is None:
        x = x.contiguous()
        rx = _rows(x, "add_layer_norm")
    rows, C, xs = rx
    ra = None
    if a is not None:
        if a.shape != x.shape or a.dtype != x.dtype:
            raise PwwHipError("add_layer_norm: `a` must have x's shape and dtype")
        ra = _rows(a, "add_layer_norm")
        if ra is None:
            a = a.contiguous()
            ra = _rows(a, "add_layer_norm")
    for name, t in (("weight", weight), ("bias", bias), ("post_bias", post_bias)):
        if t is not None and (t.dtype != x.dtype or tuple(t.shape) != (C,) or not t.is_contiguous()):
            raise PwwHipError("add_layer_norm: `%s` must be a contiguous %s [%d]" % (name, x.dtype, C))
    if post_bias is not None and a is None:
        raise PwwHipError("add_layer_norm: post_bias needs the add form (a)")
    y = torch.empty(x.shape, dtype=x.dtype, device=x.device)
    s = torch.empty(x.shape, dtype=x.dtype, device=x.device) if a is not None else None
    d = _lib.LnDesc(_DT[x.dtype], C, rows, ra[2] if ra else 0, xs, C, C, float(eps), 0)
    with torch.cuda.device(x.device):
        if post_bias is not None:
            _lib.check(_lib.load().pww_add_layer_norm_bias(_ptr(a), _ptr(x), _ptr(weight) if weight is not None else None, _ptr(bias) if bias is not None else None,
                                                           _ptr(post_bias), _ptr(s), _ptr(y), ctypes.byref(d), _stream()), "pww_add_layer_norm_bias")
        else:
            _lib.check(_lib.load().pww_add_layer_norm(_ptr(a) if a is not None else None, _ptr(x), _ptr(weight) if weight is not None else None,
                                                      _ptr(bias) if bias is not None else None, _ptr(s) if s is not None else None, _ptr(y),
                                                      ctypes.byref(d), _stream()), "pww_add_layer_norm")
    return (s, y) if a is not None else y


def geglu(h):
    """h[..., :D] * gelu(h[..., D:]) for the output of GEGLU's projection (diffusers FeedForward.net[0]); erf form of gelu."""
    _require_gpu(h)
    r = _rows(h, "geglu")
    if r is None:
        h = h.contiguous()
        r = _rows(h, "geglu")
    rows, C2, hs = r
    if C2 % 16 != 0:
        raise PwwHipError("geglu: the last dim must be 2 * D with D a multiple of 8 (got %d)" % C2)
    D = C2 // 2
    y = torch.empty(h.shape[:-1] + (D,), dtype=h.dtype, device=h.device)
    with torch.cuda.device(h.device):
        _lib.check(_lib.load().pww_geglu(_ptr(h), _ptr(y), rows, D, hs, D, _DT[h.dtype], _stream()), "pww_geglu")
    return y


def bias_residual(residual, value, bias):
    """residual + (value + bias[None, :, None, None]) for [B, C, H, W] tensors of ONE memory format (NCHW-contiguous or channels_last), the two
    roundings of the stock sequence kept; one launch."""
    _require_gpu(residual, value, bias)
    if residual.shape != value.shape or residual.dtype != value.dtype or residual.dtype not in _DT or residual.dim() != 4 or bias.dtype != residual.dtype:
        raise PwwHipError("bias_residual: two same-shape 4-d float16/bfloat16 tensors and a bias of that type are needed")
    B, C, H, W = residual.shape
    if tuple(bias.shape) != (C,):
        raise PwwHipError("bias_residual: bias must have shape (%d,), got %s" % (C, tuple(bias.shape)))
    nhwc = value.is_contiguous(memory_format=torch.channels_last) and not value.is_contiguous()
    fmt = torch.channels_last if nhwc else torch.contiguous_format
    value, residual = value.contiguous(memory_format=fmt), residual.contiguous(memory_format=fmt)       # (no-ops when both already are)
    layout = _lib.LAYOUT_NHWC if nhwc else _lib.LAYOUT_NCHW
    y = torch.empty_like(value)
    with torch.cuda.device(value.device):
        _lib.check(_lib.load().pww_bias_residual(_ptr(residual), _ptr(value), _ptr(bias.contiguous()), _ptr(y), B, C, H * W, layout, _DT[value.dtype], _stream()),
                   "pww_bias_residual")
    return y
