"""Helpers for the GPU parity tests (test infrastructure)."""
import numpy as np
import torch

from oracle import pww_oracle as O

TOL = {torch.float16: 2e-3, torch.bfloat16: 1.6e-2}   # per-call bar of BASELINE.md section 4 (x max|O|)


def unfused_inj_forward(module, hidden_states, context=None, mask=None):
    """The reference's op sequence as plain torch ops on the GPU with autocast-like precision
    (half matmuls, fp32 softmax): the 'reference on the same GPU' calibration path."""
    is_dict = isinstance(context, dict)
    ctx = hidden_states if context is None else (context["CONTEXT_TENSOR"] if is_dict else context)
    wdt = module.to_q.weight.dtype
    q = O.split_heads(module.to_q(hidden_states.to(wdt)), module.heads)
    k = O.split_heads(module.to_k(ctx.to(wdt)), module.heads)
    v = O.split_heads(module.to_v(ctx.to(wdt)), module.heads)
    scores = torch.matmul(q, k.transpose(-1, -2))
    bias = 0.0
    if is_dict:
        n = scores.shape[-2]
        try:
            w = context[f"CROSS_ATTENTION_WEIGHT_{n}"]
        except KeyError:        # the reference's fallback (:95-101): resize CROSS_ATTENTION_WEIGHT_ORIG to n tokens, plain torch ops
            w = context["CROSS_ATTENTION_WEIGHT_ORIG"]
            if torch.is_tensor(w):
                ratio = (w.shape[0] * w.shape[1] / n) ** 0.5
                nc = w.shape[2]
                small = torch.nn.functional.interpolate(w.permute(2, 0, 1)[None].float(), scale_factor=1 / ratio, mode="bilinear", align_corners=True)
                w = torch.nn.functional.interpolate(small.reshape(1, nc, -1), size=n, mode="nearest")[0].t()       # [n, nc]
            else:
                w = 0
        bias = context["WEIGHT_FUNCTION"](w, context["SIGMA"], scores)
    probs = ((scores.float() + bias) * module.scale).softmax(dim=-1)
    out = O.merge_heads(torch.matmul(probs.to(v.dtype), v), module.heads)
    return module.to_out[0](out)


def install_unfused(unet):
    for m in unet.modules():
        if m.__class__.__name__ == "CrossAttention":
            m.__class__.__call__ = unfused_inj_forward


def uninstall_all():
    from sd_standin import CrossAttention
    if "__call__" in CrossAttention.__dict__:
        del CrossAttention.__call__


def rel_l2(a, b):
    a = a.detach().double().cpu().numpy() if torch.is_tensor(a) else np.asarray(a, np.float64)
    b = b.detach().double().cpu().numpy() if torch.is_tensor(b) else np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / np.linalg.norm(b))
