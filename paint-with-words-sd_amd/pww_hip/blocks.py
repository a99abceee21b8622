"""The blocks that CALL the attention path (diffusers==0.10.0 ``ResnetBlock2D`` / ``Transformer2DModel`` /
``UNet2DConditionModel.conv_norm_out``), with their GroupNorm (+ time-embedding addend, + SiLU) on the HIP kernels of ``csrc/pww_norm.hip``.

Installed per INSTANCE (``module.forward = ...``; ``nn.Module.__call__`` and its hooks stay), by class NAME like the reference's own plug
(paint_with_words.py:193-195 selects ``__class__.__name__ == "CrossAttention"``):

* ``ResnetBlock2D`` -- the 0.10.0 forward restated with the two ``norm -> nonlinearity`` pairs and the ``+ temb`` in front of ``norm2`` as ONE op
  each::

      h = conv1(silu(norm1(x)));  h = h + time_emb_proj(silu(temb))[:, :, None, None];  h = conv2(dropout(silu(norm2(h))));
      return (conv_shortcut(x) + h) / output_scale_factor

  and ``time_emb_proj(silu(temb))`` of all blocks as one GEMM per forward (``TembProjections``).
  Blocks this restatement does not cover (``upsample`` / ``downsample`` inside the block, ``time_embedding_norm != "default"``, a nonlinearity
  other than SiLU) keep their own forward.
  The biases of ``conv1`` / ``conv2`` ride along as operands of the next fused op (``pre_bias`` of norm2, ``ops.bias_residual`` for the block's
  output) instead of costing an add launch each.
* every other ``nn.GroupNorm`` of the model (``Transformer2DModel.norm``, ``conv_norm_out``) -- the same kernels without addend / activation.
* ``BasicTransformerBlock`` -- ``attn(norm(h)) + h`` and the block's NEXT LayerNorm as one launch (``ops.add_layer_norm``); ``GEGLU`` --
  ``x * gelu(gate)`` as one launch behind its projection; 1 x 1 ``nn.Conv2d`` (``proj_in`` / ``proj_out`` / ``conv_shortcut``) on channels_last
  tensors -- a GEMM over the ``[B H W, C]`` view with the bias in its epilogue.

Inputs the kernels do not take (CPU tensors, fp32, a channel count that is not a multiple of 8 ...) go through the module's ORIGINAL forward:
that is the stock op of the dependency, not a second implementation of this repository. On a GPU without the HIP library ``ops.group_norm``
raises like every other op here.
"""
import os
import types
import warnings

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops

FUSED_NORM = os.environ.get("PWW_FUSED_NORM", "1") != "0"      # A/B switch (bench.py --no-fused-norm sets it)
BATCHED_TEMB = os.environ.get("PWW_BATCHED_TEMB", "1") != "0"  # A/B switch: one time-embedding projection GEMM per forward for all blocks
FOLD_CONV_BIAS = os.environ.get("PWW_FOLD_CONV_BIAS", "1") != "0"      # A/B: conv1 / conv2 biases of a ResnetBlock2D ride in the next fused op
CONV1X1_AS_LINEAR = os.environ.get("PWW_CONV1X1_AS_LINEAR", "1") != "0"  # A/B: 1 x 1 convolutions on channels_last tensors as GEMMs
# `ff(norm3(h)) + h` through the output GEMM's C operand (no add launch): built in round 5, measured NEUTRAL on the headline (4.04 / 4.06 vs 4.08 /
# 4.06 images/s, profiles/r05_blocks_ab.md: hipBLASLt's beta = 1 kernels cost what the add launch saves) -> opt-in, off by default
FUSE_FF_RESIDUAL = os.environ.get("PWW_FUSE_FF_RESIDUAL", "0") == "1"


# ---- what the plug did with the calls it saw (a call the kernels decline must not go unnoticed) ----------------
# per op: [calls taken by the HIP kernels, calls handed to the module's own forward / the stock op]. Counted on the host, i.e. in eager
# passes and while a hipGraph is captured (a replay runs no Python). `stats()` returns a copy, `reset_stats()` zeroes; the first decline of
# a HIP half-precision tensor per (op, reason) also warns once -- under torch.autocast, the reference's own run mode, the whole plug steps
# aside (autocast runs the norms in fp32) and that is worth knowing when a benchmark claims the fused blocks.
_STATS = {k: [0, 0] for k in ("group_norm", "resnet_block", "transformer_block", "geglu", "conv1x1")}
_warned = set()


def stats():
    """{op: {"fused": n, "declined": n}} since the last reset_stats(), and the hit rate over the norm / elementwise ops (the 1 x 1
    convolutions are listed but not rated: an NCHW tensor keeps MIOpen's convolution by the caller's choice of memory format)."""
    out = {k: {"fused": v[0], "declined": v[1]} for k, v in _STATS.items()}
    rated = [v for k, v in _STATS.items() if k != "conv1x1"]
    tot = sum(v[0] + v[1] for v in rated)
    out["hit_rate"] = (sum(v[0] for v in rated) / tot) if tot else None
    return out


def reset_stats():
    for v in _STATS.values():
        v[0] = v[1] = 0


def _count(op, fused, x=None, reason=None):
    _STATS[op][0 if fused else 1] += 1
    if not fused and FUSED_NORM and torch.is_tensor(x) and x.is_cuda and x.dtype in (torch.float16, torch.bfloat16):
        key = (op, reason)
        if key not in _warned:
            _warned.add(key)
            warnings.warn("pww_hip.blocks: %s call on a HIP %s tensor handed to the stock op (%s); pww_hip.blocks.stats() counts them"
                          % (op, str(x.dtype).split(".")[-1], reason or "shape / dtype the kernels do not take"), stacklevel=3)


def _why_not(x):
    if torch.is_autocast_enabled():
        return "torch.autocast is on: the norms run in fp32 there"
    if torch.is_grad_enabled() and torch.is_tensor(x) and x.requires_grad:
        return "autograd is recording: the kernels are forward-only"
    return None


def _takes(x, norm):
    if not (FUSED_NORM and torch.is_tensor(x) and x.is_cuda and x.dim() == 4 and x.dtype in (torch.float16, torch.bfloat16)):
        return False
    B, C, H, W = x.shape
    w = norm.weight
    return (C % 8 == 0 and (H * W) % 8 == 0 and C % norm.num_groups == 0 and norm.num_groups <= 32 and C <= 4096
            and (w is None or w.dtype == x.dtype) and not torch.is_autocast_enabled()         # (autocast runs group_norm in fp32)
            and not (torch.is_grad_enabled() and (x.requires_grad or (w is not None and w.requires_grad))))


def fused_group_norm(norm, x, add=None, act=None):
    """act(norm(x + add[:, :, None, None])) through the HIP kernels when they take the input, else the same thing from stock ops."""
    if _takes(x, norm) and (add is None or add.dtype == x.dtype):
        return ops.group_norm(x, norm.num_groups, norm.weight, norm.bias, norm.eps, add=add, act=act)
    if add is not None:
        x = x + add[:, :, None, None]
    y = F.group_norm(x, norm.num_groups, norm.weight, norm.bias, norm.eps)
    return F.silu(y) if act == "silu" else y


def _group_norm_forward(self, x, *args, **kwargs):
    if args or kwargs or not _takes(x, self):
        _count("group_norm", False, x, "unexpected arguments" if (args or kwargs) else _why_not(x))
        return self._pww_orig_forward(x, *args, **kwargs)
    _count("group_norm", True)
    return fused_group_norm(self, x)


def _is_silu(m):
    return m is None or isinstance(m, nn.SiLU) or m is F.silu


def _resnet_covered(m):
    return (isinstance(getattr(m, "norm1", None), nn.GroupNorm) and isinstance(getattr(m, "norm2", None), nn.GroupNorm)
            and hasattr(m, "conv1") and hasattr(m, "conv2") and getattr(m, "upsample", None) is None and getattr(m, "downsample", None) is None
            and getattr(m, "time_embedding_norm", "default") == "default" and _is_silu(getattr(m, "nonlinearity", None)))


class TembProjections:
    """``time_emb_proj(silu(temb))`` of ALL the plugged ResnetBlock2D of a model as ONE GEMM per forward (22 blocks in the SD1.5 UNet: 22
    SiLU launches on the same [B, 1280] tensor and 22 GEMMs of 2 rows become one each). The concatenated weight is a COPY: it is rebuilt
    when any source parameter was replaced, modified in place (tensor version counter), moved or cast. The result is cached ON the temb
    tensor object (a fresh one every forward; the hit test also compares the tensor's version counter and data pointer), block i takes its
    columns as a view -- `ops.group_norm` reads strided rows."""

    def __init__(self):
        self.linears = []          # (block, offset, width) in registration order
        self.width = 0
        self._sig = None
        self._w = self._b = None

    def register(self, block):
        lin = block.time_emb_proj
        slot = (self, self.width, lin.out_features)
        self.linears.append((lin, self.width, lin.out_features))
        self.width += (lin.out_features + 7) // 8 * 8             # (every block's columns start 16-byte aligned)
        return slot

    def _signature(self, dtype, device):
        sig = [dtype, device]
        for lin, _, _ in self.linears:
            sig.append((lin.weight.data_ptr(), lin.weight._version, lin.weight.dtype))
            sig.append(None if lin.bias is None else (lin.bias.data_ptr(), lin.bias._version))
        return tuple(sig)

    def project(self, temb):
        # the cache entry is valid for THIS tensor in THIS state: a caller that keeps one temb tensor and updates it in place between
        # forwards (static buffers, hipGraph-style input copies) bumps its version counter
        hit = temb.__dict__.get("_pww_temb_proj")
        if hit is not None and hit[0] is self and hit[2] == (temb._version, temb.data_ptr()):
            return hit[1]
        sig = self._signature(temb.dtype, temb.device)
        if sig != self._sig:
            w = torch.zeros(self.width, self.linears[0][0].in_features, dtype=temb.dtype, device=temb.device)
            b = torch.zeros(self.width, dtype=temb.dtype, device=temb.device)
            for lin, off, n in self.linears:
                w[off:off + n] = lin.weight.detach().to(temb.dtype)
                if lin.bias is not None:
                    b[off:off + n] = lin.bias.detach().to(temb.dtype)
            self._w, self._b, self._sig = w, b, sig
        out = F.linear(F.silu(temb), self._w, self._b)
        temb.__dict__["_pww_temb_proj"] = (self, out, (temb._version, temb.data_ptr()))
        return out


def _conv_no_bias(conv, x):
    """`conv(x)` without its bias (the caller hands the bias to the next fused op instead of paying an add launch for it)."""
    return F.conv2d(x, conv.weight, None, conv.stride, conv.padding, conv.dilation, conv.groups)


def _plain_conv(conv):
    return isinstance(conv, nn.Conv2d) and conv.bias is not None and conv.padding_mode == "zeros" and type(conv).forward is nn.Conv2d.forward \
        and "forward" not in conv.__dict__ and not conv._forward_hooks and not conv._forward_pre_hooks


def _resnet_forward(self, *args, **kwargs):
    """diffusers 0.10.0 ``ResnetBlock2D.forward(input_tensor, temb)`` (``up`` / ``down`` = False, ``time_embedding_norm == "default"``). Any other
    call -- later diffusers pass ``scale=`` (0.21 - 0.26: the LoRA scale) -- reaches the module's own forward with its arguments as given."""
    covered = 1 <= len(args) <= 2 and set(kwargs) <= {"temb"} and not (len(args) == 2 and kwargs)
    input_tensor = args[0] if args else None
    if not covered or not _takes(input_tensor, self.norm1):
        _count("resnet_block", False, input_tensor, ("a call beyond (input_tensor, temb): %d positional, keywords %s" % (len(args), sorted(kwargs))) if not covered else _why_not(input_tensor))
        return self._pww_orig_forward(*args, **kwargs)
    temb = args[1] if len(args) == 2 else kwargs.get("temb")
    _count("resnet_block", True)
    fold = FOLD_CONV_BIAS and _plain_conv(self.conv1) and _plain_conv(self.conv2) and self.conv1.bias.dtype == input_tensor.dtype
    h = fused_group_norm(self.norm1, input_tensor, act="silu")
    h = _conv_no_bias(self.conv1, h) if fold else self.conv1(h)
    add = None
    if temb is not None and getattr(self, "time_emb_proj", None) is not None:
        slot = self.__dict__.get("_pww_temb_slot")
        if slot is not None and temb.dtype == h.dtype and temb.dim() == 2 and not torch.is_grad_enabled():
            add = slot[0].project(temb)[:, slot[1]:slot[1] + slot[2]]
        else:
            add = self.time_emb_proj(F.silu(temb))
        if add.dtype != h.dtype:
            add = add.to(h.dtype)
    if fold and _takes(h, self.norm2):
        h = ops.group_norm(h, self.norm2.num_groups, self.norm2.weight, self.norm2.bias, self.norm2.eps, add=add, act="silu", pre_bias=self.conv1.bias)
    else:
        if fold:
            h = h + self.conv1.bias[None, :, None, None]
        h = fused_group_norm(self.norm2, h, add=add, act="silu")
    dropout = getattr(self, "dropout", None)
    if dropout is not None:
        h = dropout(h)
    if getattr(self, "conv_shortcut", None) is not None:
        input_tensor = self.conv_shortcut(input_tensor)
    scale = getattr(self, "output_scale_factor", 1.0)
    if fold and input_tensor.dtype == h.dtype and input_tensor.shape[1] % 8 == 0:
        out = ops.bias_residual(input_tensor, _conv_no_bias(self.conv2, h), self.conv2.bias)          # input + (conv2(h) + bias): one launch
    else:
        out = input_tensor + self.conv2(h)
    return out if scale == 1.0 else out / scale


# ---- 1 x 1 convolutions on channels_last tensors are GEMMs over the [B H W, C] view: bias in the GEMM's epilogue, no layout kernels ----
def _conv1x1_forward(self, x, *args, **kwargs):
    if (not args and not kwargs and FUSED_NORM and CONV1X1_AS_LINEAR and torch.is_tensor(x) and x.is_cuda and x.dim() == 4 and x.dtype == self.weight.dtype
            and x.shape[2] * x.shape[3] > 1 and x.is_contiguous(memory_format=torch.channels_last) and not torch.is_autocast_enabled()):
        B, C, H, W = x.shape
        _STATS["conv1x1"][0] += 1
        y = F.linear(x.permute(0, 2, 3, 1).reshape(B * H * W, C), self.weight.reshape(self.out_channels, C), self.bias)
        return y.view(B, H, W, self.out_channels).permute(0, 3, 1, 2)
    _STATS["conv1x1"][1] += 1          # (an NCHW tensor keeps MIOpen's 1 x 1 convolution: a layout choice of the caller, not worth a warning)
    return self._pww_orig_forward(x, *args, **kwargs)


def _is_conv1x1(m):
    return (isinstance(m, nn.Conv2d) and type(m).forward is nn.Conv2d.forward and m.kernel_size == (1, 1) and m.stride == (1, 1) and m.padding == (0, 0)
            and m.dilation == (1, 1) and m.groups == 1 and m.padding_mode == "zeros")


# ---- BasicTransformerBlock / GEGLU -----------------------------------------------------------------------------------------------------
def _ln_takes(x, norm):
    return (FUSED_NORM and torch.is_tensor(x) and x.is_cuda and x.dtype in (torch.float16, torch.bfloat16) and x.dim() >= 2
            and len(norm.normalized_shape) == 1 and x.shape[-1] == norm.normalized_shape[0] and x.shape[-1] % 8 == 0 and x.shape[-1] <= 2048
            and (norm.weight is None or norm.weight.dtype == x.dtype) and not torch.is_autocast_enabled()
            and not (torch.is_grad_enabled() and (x.requires_grad or (norm.weight is not None and norm.weight.requires_grad))))


def _transformer_block_covered(m):
    return (all(isinstance(getattr(m, n, None), nn.LayerNorm) for n in ("norm1", "norm2", "norm3")) and hasattr(m, "attn1") and hasattr(m, "attn2")
            and hasattr(m, "ff") and not getattr(m, "only_cross_attention", False))


def _transformer_block_forward(self, *args, **kwargs):
    """diffusers 0.10.0 ``BasicTransformerBlock.forward(hidden_states, context=None, timestep=None)``: ``h = attn1(norm1(h)) + h;
    h = attn2(norm2(h), context) + h; h = ff(norm3(h)) + h`` -- with each residual add and the NEXT norm as one launch. 0.10.0's
    ``Transformer2DModel`` always passes ``timestep=`` (None unless the norms are AdaLayerNorm, which `_transformer_block_covered` excludes):
    a None timestep is accepted and ignored. Every other call -- a real timestep, more positionals, keywords this restatement does not know
    (later diffusers: ``encoder_hidden_states=``, ``attention_mask=``, ``cross_attention_kwargs=`` ...) -- reaches the module's own forward
    with its arguments exactly as they were given."""
    covered = 1 <= len(args) <= 2 and set(kwargs) <= {"context", "timestep"} and kwargs.get("timestep") is None and not (len(args) == 2 and "context" in kwargs)
    hidden_states = args[0] if args else None
    if not covered or not _ln_takes(hidden_states, self.norm1):
        _count("transformer_block", False, hidden_states, ("a call beyond (hidden_states, context, timestep=None): %d positional, keywords %s" % (len(args), sorted(kwargs)))
               if not covered else _why_not(hidden_states))
        return self._pww_orig_forward(*args, **kwargs)
    context = args[1] if len(args) == 2 else kwargs.get("context")
    _count("transformer_block", True)
    h = hidden_states
    n = ops.add_layer_norm(h, self.norm1.weight, self.norm1.bias, self.norm1.eps)
    h, n = ops.add_layer_norm(h, self.norm2.weight, self.norm2.bias, self.norm2.eps, a=self.attn1(n))
    lin = _ff_output_linear(self.ff) if FUSE_FF_RESIDUAL else None
    if lin is not None and lin.weight.dtype == h.dtype:
        # `ff(norm3(h)) + h` without an add launch (round 5; 16 of the 32 residual adds of a forward): the sum that feeds norm3 leaves its
        # launch already carrying the output GEMM's bias, and that GEMM adds it as its C operand (beta = 1)
        h, n = ops.add_layer_norm(h, self.norm3.weight, self.norm3.bias, self.norm3.eps, a=self.attn2(n, context=context), post_bias=lin.bias)
        g = self.ff.net[0](n)
        return torch.addmm(h.reshape(-1, h.shape[-1]), g.reshape(-1, g.shape[-1]), lin.weight.t()).view(h.shape)
    h, n = ops.add_layer_norm(h, self.norm3.weight, self.norm3.bias, self.norm3.eps, a=self.attn2(n, context=context))
    return self.ff(n) + h


def _ff_output_linear(ff):
    """The output projection of a diffusers FeedForward `net = [GEGLU, Dropout, Linear]` when the block's tail can run as GEGLU -> one GEMM
    with the residual as its C operand (a dropout that does nothing, a plain biased nn.Linear without hooks), else None."""
    net = getattr(ff, "net", None)
    # the stock class only: a subclass (or a patched class) that brings its own `forward` -- LoRA-scaled, chunked -- keeps it
    if net is None or len(net) != 3 or type(ff).__name__ != "FeedForward" or "forward" in ff.__dict__:
        return None
    owner = next((c for c in type(ff).__mro__ if "forward" in c.__dict__), None)
    if owner is None or owner.__name__ != "FeedForward":
        return None
    act, drop, lin = net[0], net[1], net[2]
    if act.__class__.__name__ != "GEGLU" or not isinstance(drop, nn.Dropout) or (drop.training and drop.p > 0):
        return None
    if type(lin) is not nn.Linear or lin.bias is None or lin._forward_hooks or lin._forward_pre_hooks or "forward" in lin.__dict__ or ff._forward_hooks or ff._forward_pre_hooks:
        return None
    return lin


def _geglu_forward(self, x, *args, **kwargs):
    """diffusers ``GEGLU.forward(hidden_states)``; later versions add ``scale`` (0.21 - 0.26): anything extra -> the module's own forward."""
    if args or kwargs:
        _count("geglu", False, x, "arguments beyond (hidden_states): %s" % (sorted(kwargs) or len(args)))
        return self._pww_orig_forward(x, *args, **kwargs)
    h = self.proj(x)
    if FUSED_NORM and h.is_cuda and h.dtype in (torch.float16, torch.bfloat16) and h.shape[-1] % 16 == 0 and not torch.is_autocast_enabled() \
            and not (torch.is_grad_enabled() and h.requires_grad):
        _count("geglu", True)
        return ops.geglu(h)
    _count("geglu", False, h, _why_not(h))
    a, gate = h.chunk(2, dim=-1)
    return a * F.gelu(gate)


def install_blocks(unet):
    """Put the fused norms under every ``ResnetBlock2D`` and every other ``nn.GroupNorm`` of `unet`. Returns (resnets, norms) patched."""
    n_res = n_gn = 0
    owned = set()
    plans = {}                 # time-embedding width -> one batched projection for all blocks that share it
    for m in unet.modules():
        if m.__class__.__name__ == "ResnetBlock2D" and _resnet_covered(m):
            if "_pww_orig_forward" not in m.__dict__:
                m.__dict__["_pww_orig_forward"] = m.forward
                m.forward = types.MethodType(_resnet_forward, m)
                lin = getattr(m, "time_emb_proj", None)
                if BATCHED_TEMB and isinstance(lin, nn.Linear):
                    m.__dict__["_pww_temb_slot"] = plans.setdefault(lin.in_features, TembProjections()).register(m)
            owned.update((id(m.norm1), id(m.norm2)))
            n_res += 1
    for m in unet.modules():
        name, fwd = m.__class__.__name__, None
        if isinstance(m, nn.GroupNorm) and id(m) not in owned:
            fwd = _group_norm_forward
            n_gn += 1
        elif name == "BasicTransformerBlock" and _transformer_block_covered(m):
            fwd = _transformer_block_forward
        elif name == "GEGLU" and isinstance(getattr(m, "proj", None), nn.Linear):
            fwd = _geglu_forward
        elif _is_conv1x1(m):
            fwd = _conv1x1_forward
        if fwd is not None and "_pww_orig_forward" not in m.__dict__:
            m.__dict__["_pww_orig_forward"] = m.forward
            m.forward = types.MethodType(fwd, m)
    return n_res, n_gn


def uninstall_blocks(unet):
    for m in unet.modules():
        if "_pww_orig_forward" in m.__dict__:
            del m.__dict__["_pww_orig_forward"]
            m.__dict__.pop("forward", None)
            m.__dict__.pop("_pww_temb_slot", None)
