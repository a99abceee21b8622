"""Round-5 GPU tests:
  * pww_qk_parts: the statistic's partials over a finished Q vs pww_qk_reduce (item 1a: the C = 1280 layers leave the in-kernel hand-off);
  * the pass-2-only launch (pww_cross_attn_fwd_parts -> the small one-block-per-workgroup kernel) vs an fp64 restatement of
    paint_with_words.py:87-116 on the same rounded inputs, every head dim / column bound / statistic, ragged shapes, gated rows;
  * the default product path creates no hand-off state on ANY layer (no `_pww_fused_scratch`), also for the 1/8-width model;
  * self-attention: the K-fragment prefetch of score_tile leaves every output bit where it was (kernel vs fp64 on the reference's shapes
    is test_attention_gpu.py; here: run-to-run bit identity and the key-split variants against each other);
  * function-API img2img pinned to the reference's own loop (item 3).
"""
import math
import os
import subprocess
import sys

import numpy as np
import pytest
import torch
from PIL import Image

import pww_cases as cases
from gpu_util import TOL, uninstall_all, rel_l2

pytestmark = pytest.mark.gpu
G = cases.GOLDEN

QKP_SHAPES = [
    # name, B, N, heads, D, M, shared prompt
    ("sd15_n256", 2, 256, 8, 160, 77, False),
    ("sd15_n64", 2, 64, 8, 160, 77, True),
    ("sd15_n1024", 2, 1024, 8, 80, 77, False),
    ("sd15_n4096", 2, 4096, 8, 40, 77, False),
    ("sd21_n576_b8", 8, 576, 20, 64, 77, False),
    ("sd21_n144", 2, 144, 20, 64, 77, False),
    ("ragged_n100_m40", 3, 100, 8, 80, 40, True),
    ("ragged_n333_m128", 1, 333, 8, 160, 128, False),
    ("tiny_n50_m1_d32", 5, 50, 5, 32, 1, True),
    ("d16_n700_m33", 7, 700, 10, 16, 33, False),
    ("d96_n90_m100", 2, 90, 4, 96, 100, False),
    ("d128_n200_m77", 2, 200, 3, 128, 77, False),
    ("d24_n200_m77", 2, 200, 4, 24, 77, False),      # head dims whose ones column (the softmax denominator from the PV MFMA) sits in the last padding chunk
    ("d8_n96_m77", 2, 96, 8, 8, 77, True),
    ("d56_n130_m90", 2, 130, 4, 56, 90, False),
    ("d72_n260_m77", 2, 260, 4, 72, 77, False),
    ("d104_n70_m128", 1, 70, 2, 104, 128, False),
]


def _qkv(name, B, N, H, D, M, shared, dtype, dev, gain=0.6):
    g = torch.Generator().manual_seed(sum(map(ord, name)) % 997 + 3)
    C = H * D
    q = (torch.randn(B, N, C, generator=g) * gain).to(dtype).to(dev)
    k = torch.randn(1 if shared else B, M, C, generator=g).to(dtype).to(dev)
    v = (torch.randn(1 if shared else B, M, C, generator=g) + 0.1).to(dtype).to(dev)
    return q, k, v


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("name,B,N,H,D,M,shared", QKP_SHAPES)
def test_qk_parts_matches_qk_reduce(gpu_device, dtype, name, B, N, H, D, M, shared):
    """folded partials of pww_qk_parts vs pww_qk_reduce on the same Q: extremes exactly (both are maxima of the same fp32 MFMA scores),
    mean to 1e-6 of a standard deviation, sum of squares to 1e-6 relative; only the fields a statistic is made of are formed."""
    from pww_hip import ops
    q, k, v = _qkv(name, B, N, H, D, M, shared, dtype, gpu_device)
    gate = torch.ones(B, device=gpu_device)
    if B > 1:
        gate[B - 1] = 0.0
    nb = B - 1 if B > 1 else B
    parts = ops.qk_parts(q, k, H, ops.STAT_ALL, gate=gate, gated=nb if B > 1 else 0)
    nrb, nkb = (N + 31) // 32, (M + 31) // 32
    assert parts.shape == (B, H * nrb * nkb if H * nrb * nkb <= 256 else H * ((nrb + 3) // 4), 4)      # a partial per wave, or per workgroup of 4 row blocks
    stats = ops.qk_stats(q, k, H).cpu()
    folded = ops.fold_parts(parts[:nb]).cpu()
    cnt = H * N * M
    sd = ((stats[:nb, 3] - stats[:nb, 2] ** 2 / cnt) / max(cnt - 1, 1)).clamp_min(0).sqrt() + 1e-12
    assert torch.equal(folded[:, :2], stats[:nb, :2]), (name, folded[:, :2], stats[:nb, :2])
    e_mean = ((folded[:, 2] - stats[:nb, 2]).abs() / cnt / sd).max().item()
    e_sq = ((folded[:, 3] - stats[:nb, 3]).abs() / stats[:nb, 3]).max().item()
    print(f"qk_parts {name} {dtype}: {parts.shape[1]} partials / image, mean {e_mean:.1e}, sumsq {e_sq:.1e}")
    assert e_mean <= 1e-6 and e_sq <= 1e-6
    p_max = ops.qk_parts(q, k, H, ops.STAT_MAX, gate=gate, gated=nb if B > 1 else 0)
    assert torch.equal(p_max[:nb, :, 0], parts[:nb, :, 0]) and bool((p_max[:nb, :, 2] == 0).all()) and bool(torch.isinf(p_max[:nb, :, 1]).all())
    # the hint only shrinks the grid: without it the same partials
    p_nohint = ops.qk_parts(q, k, H, ops.STAT_ALL, gate=gate)
    assert torch.equal(p_nohint[:nb], parts[:nb])


def _reference(q, k, v, H, scale, bias, coeff):
    """paint_with_words.py:87-116 in fp64 on the given (already rounded) q / k / v: per image coefficient coeff[b] on the bias map"""
    B, N, C = q.shape
    D = C // H
    qh = q.double().reshape(B, N, H, D).permute(0, 2, 1, 3)
    kh = k.double().expand(B, -1, -1).reshape(B, -1, H, D).permute(0, 2, 1, 3)
    vh = v.double().expand(B, -1, -1).reshape(B, -1, H, D).permute(0, 2, 1, 3)
    s = qh @ kh.transpose(-1, -2)
    s = s + coeff.double().reshape(B, 1, 1, 1) * bias.double()[None, None]
    p = (s * scale).softmax(-1)
    return (p @ vh).permute(0, 2, 1, 3).reshape(B, N, C), qh @ kh.transpose(-1, -2)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("name,B,N,H,D,M,shared", [s for s in QKP_SHAPES if s[5] >= 64] + [("sd15_n4096_b8", 8, 4096, 8, 40, 77, True)])
@pytest.mark.parametrize("cols", [16, 32, 48, 64])
def test_pass2_only_launch_vs_fp64(gpu_device, dtype, name, B, N, H, D, M, shared, cols):
    """qk_parts + pww_cross_attn_fwd_parts (the small kernel for M >= 64, at most 64 bias columns, <= 1024 blocks of 128 rows) against
    fp64: the BASELINE.md bar per attention call on the same rounded inputs. Statistic kinds rotate over the cases; the last image is
    gated out (CFG's unconditional row) where there are several."""
    from pww_hip import ops
    q, k, v = _qkv(name, B, N, H, D, M, shared, dtype, gpu_device)
    g = torch.Generator().manual_seed(cols)
    bias = ((torch.rand(N, M, generator=g) < 0.3).float() * torch.rand(N, M, generator=g) * 1.5)
    bias[:, cols:] = 0
    bias = bias.to(gpu_device)
    gate = torch.ones(B, device=gpu_device)
    if B > 1:
        gate[B - 1] = 0.0
    gated = B - 1 if B > 1 else 0
    kind = [ops.STAT_MAX, ops.STAT_STD, ops.STAT_ABSMAX, ops.STAT_MEAN, ops.STAT_MIN][(cols // 16 + N) % 5]
    scale, c0 = D ** -0.5, 0.37
    parts = ops.qk_parts(q, k, H, kind, gate=gate, gated=gated)
    stats_out = torch.zeros(B, 4, dtype=torch.float64, device=gpu_device)
    out = ops.attention(q, k, v, H, scale, bias=bias, bias_coeff=gate, stat=(None, kind, c0), parts=parts, bias_cols=cols, gated=gated, stats_out=stats_out)
    # coefficient from the raw fp64 scores
    _, s = _reference(q, k, v, H, scale, bias, torch.zeros(B, device=q.device))
    sb = s.reshape(B, -1)
    stat = {ops.STAT_MAX: sb.max(1).values, ops.STAT_MIN: sb.min(1).values, ops.STAT_MEAN: sb.mean(1), ops.STAT_STD: sb.std(1), ops.STAT_ABSMAX: sb.abs().max(1).values}[kind]
    coeff = c0 * stat * gate.double()
    ref, _ = _reference(q, k, v, H, scale, bias, coeff)
    err = (out.double() - ref).abs().max().item() / ref.abs().max().item()
    print(f"pass-2-only {name} {dtype} cols {cols} kind {kind}: max err / max|O| = {err:.2e}")
    assert torch.isfinite(out).all() and err <= TOL[dtype]
    # the statistics handed back are the folded partials
    f = ops.fold_parts(parts[: B - 1 if B > 1 else B])
    so = stats_out[: B - 1 if B > 1 else B]
    assert torch.equal(so[:, 0], f[:, 0]) or kind not in (ops.STAT_MAX, ops.STAT_ABSMAX)
    # bitwise repeatable
    out2 = ops.attention(q, k, v, H, scale, bias=bias, bias_coeff=gate, stat=(None, kind, c0), parts=parts, bias_cols=cols, gated=gated)
    assert torch.equal(out, out2)
    # a statistic-free coefficient needs no partials: c = c0 * gate
    out3 = ops.attention(q, k, v, H, scale, bias=bias, bias_coeff=gate, stat=(None, ops.STAT_NONE, c0), bias_cols=cols, gated=gated)
    ref3, _ = _reference(q, k, v, H, scale, bias, c0 * gate.double())
    err3 = (out3.double() - ref3).abs().max().item() / ref3.abs().max().item()
    assert err3 <= TOL[dtype], err3


OUT_SHAPES = [
    # name, B, N, heads, D, M, shared prompt          (C = heads * D = 320: the layers pww_cross_attn_fwd_parts_out takes)
    ("sd15_n4096", 2, 4096, 8, 40, 77, False),
    ("sd15_n4096_b5", 5, 4096, 8, 40, 77, True),
    ("sd21_n9216", 2, 9216, 5, 64, 77, False),
    ("ragged_n333_m128", 3, 333, 8, 40, 128, False),
    ("d32_n200_m64", 2, 200, 10, 32, 64, True),
    ("d48_n100_m90", 1, 100, 5, 64, 90, False),
]


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("name,B,N,H,D,M,shared", OUT_SHAPES)
def test_cross_attention_with_to_out_in_the_launch(gpu_device, experiments_lib, dtype, name, B, N, H, D, M, shared):
    """pww_cross_attn_fwd_parts_out (row f-1: attention + to_out + bias [+ residual] in one launch) against the two-launch route it
    replaces -- `linear(pww_cross_attn_fwd_parts(...), W, b) [+ residual]` in fp64 ON THE KERNEL'S OWN rounded O (the projection must see
    exactly the O the two-launch path stores) -- and against the fp64 reference of the whole expression (paint_with_words.py:106-123).
    Bitwise repeatable; the statistic kinds rotate; the last image is gated out where there are several."""
    from pww_hip import ops
    q, k, v = _qkv(name, B, N, H, D, M, shared, dtype, gpu_device)
    C = H * D
    g = torch.Generator().manual_seed(len(name) + N)
    cols = [16, 32, 48, 64][N % 4]
    bias = ((torch.rand(N, M, generator=g) < 0.3).float() * torch.rand(N, M, generator=g) * 1.5)
    bias[:, cols:] = 0
    bias = bias.to(gpu_device)
    w = (torch.randn(C, C, generator=g) / math.sqrt(C)).to(dtype).to(gpu_device)
    wb = (torch.randn(C, generator=g) * 0.1).to(dtype).to(gpu_device)
    res = torch.randn(B, N, C, generator=g).to(dtype).to(gpu_device)
    gate = torch.ones(B, device=gpu_device)
    if B > 1:
        gate[B - 1] = 0.0
    gated = B - 1 if B > 1 else 0
    kind = [ops.STAT_MAX, ops.STAT_STD, ops.STAT_NONE, ops.STAT_MEAN][(N + H) % 4]
    scale, c0 = D ** -0.5, 0.37
    assert ops.attention_out_supported(q, k, H, bias, w, cols)
    parts = ops.qk_parts(q, k, H, kind, gate=gate, gated=gated) if kind != ops.STAT_NONE else None
    kw = dict(bias_coeff=gate, stat=(None, kind, c0), parts=parts, bias_cols=cols, gated=gated)
    o = ops.attention(q, k, v, H, scale, bias=bias, **kw)                       # the two-launch route's O (rounded to the storage type)
    out = ops.attention_out(q, k, v, H, scale, bias, w, wb, **kw)
    two = o.double() @ w.double().t() + wb.double()
    ulp = 2.0 ** (-8 if dtype == torch.bfloat16 else -11)
    # one rounding of an fp32 sum that differs from the fp64 one by summation order only: within an ulp of the result's magnitude
    err = ((out.double() - two).abs() / (two.abs() + two.abs().max() * 2 ** -6)).max().item()
    print(f"attention_out {name} {dtype} cols {cols} kind {kind}: max err vs the two-launch route {err / ulp:.2f} ulp")
    assert torch.isfinite(out).all() and err <= 1.5 * ulp, err / ulp      # (half a spacing of the storage type + the fp32 summation order; the O of a large batch comes from the general kernel: last-bit differences)
    # the whole expression in fp64 (BASELINE.md's bar per attention call, through the projection)
    _, s = _reference(q, k, v, H, scale, bias, torch.zeros(B, device=q.device))
    sb = s.reshape(B, -1)
    stat = {ops.STAT_MAX: sb.max(1).values, ops.STAT_STD: sb.std(1), ops.STAT_MEAN: sb.mean(1), ops.STAT_NONE: torch.ones(B, dtype=torch.float64, device=q.device)}[kind]
    ref_o, _ = _reference(q, k, v, H, scale, bias, c0 * stat * gate.double())
    ref = ref_o @ w.double().t() + wb.double()
    e64 = (out.double() - ref).abs().max().item() / ref.abs().max().item()
    assert e64 <= TOL[dtype], e64
    # with the residual: a second rounding, like `linear(...) + residual`
    out_r = ops.attention_out(q, k, v, H, scale, bias, w, wb, residual=res, **kw)
    assert torch.equal(out_r, out + res)
    # without a bias vector; bitwise repeatable
    out_nb = ops.attention_out(q, k, v, H, scale, bias, w, None, **kw)
    two_nb = o.double() @ w.double().t()
    assert ((out_nb.double() - two_nb).abs() / (two_nb.abs() + two_nb.abs().max() * 2 ** -6)).max().item() <= 1.5 * ulp
    assert torch.equal(out, ops.attention_out(q, k, v, H, scale, bias, w, wb, **kw))


def test_cross_attention_with_to_out_declines_what_it_does_not_take(gpu_device, experiments_lib):
    from pww_hip import ops
    dev, dt = gpu_device, torch.bfloat16
    q, k, v = _qkv("decl", 2, 256, 8, 80, 77, False, dt, dev)          # C = 640
    bias = torch.rand(256, 77, device=dev)
    w = torch.randn(640, 640, device=dev).to(dt)
    assert not ops.attention_out_supported(q, k, 8, bias, w, 32)
    with pytest.raises(ops.PwwHipError):
        ops.attention_out(q, k, v, 8, 80 ** -0.5, bias, w, None, stat=(None, ops.STAT_NONE, 1.0))
    q, k, v = _qkv("decl2", 2, 256, 8, 40, 40, False, dt, dev)         # M < 64
    assert not ops.attention_out_supported(q, k, 8, torch.rand(256, 40, device=dev), torch.randn(320, 320, device=dev).to(dt), 32)
    q, k, v = _qkv("decl3", 2, 256, 8, 40, 77, False, dt, dev)         # a map per head
    assert not ops.attention_out_supported(q, k, 8, torch.rand(2, 8, 256, 77, device=dev), torch.randn(320, 320, device=dev).to(dt), 32)


def test_default_path_has_no_handoff_state_on_any_layer(gpu_device):
    """VERDICT round 4 item 1a "done" criterion: pww_cross_attn_fwd_fused is not reachable with default settings -- after full requests
    through the SD1.5 topology (1/8 width: no pww_qproj_stat tile on any layer -> every layer takes pww_qk_parts) and through the
    full-size attention modules of every BASELINE shape class, no module owns a FusedScratch."""
    import paint_with_words as pw
    import pww_hip
    import pww_hip.attention as A
    assert not A.FUSED_CROSS
    tools = cases.build_tools("tiny", dtype=torch.bfloat16, device=gpu_device)
    try:
        for mode in ("eager", "folded", "graph"):
            mod = __import__("importlib").import_module("paint_with_words.paint_with_words")
            old, mod.DEFAULT_MODE = mod.DEFAULT_MODE, mode
            try:
                lat = pw.paint_with_words(color_context=dict(cases.RUNNER_CONTEXT), color_map_image=Image.fromarray(cases.load_example_rgb()),
                                          input_prompt=cases.RUNNER_PROMPT, num_inference_steps=3, guidance_scale=7.5, seed=0, device=str(gpu_device),
                                          weight_function=cases.weight_fn_runner, preloaded_utils=tools, return_latents=True)
            finally:
                mod.DEFAULT_MODE = old
            assert torch.isfinite(lat).all()
        assert not [m for m in tools[1].modules() if "_pww_fused_scratch" in getattr(m, "__dict__", {})]
    finally:
        uninstall_all()
    for shape in ("sd15_n4096", "sd15_n1024", "sd15_n256", "sd15_n64", "sd21_n576"):
        case = cases.make_attention_case(shape)
        mod = case["attn_cross"].to(gpu_device, torch.bfloat16)
        ctx = {"CONTEXT_TENSOR": case["ctx"].to(gpu_device, torch.bfloat16), f"CROSS_ATTENTION_WEIGHT_{case['N']}": case["w"].to(gpu_device),
               "SIGMA": torch.tensor(7.84), "WEIGHT_FUNCTION": cases.weight_fn_runner}
        y = pww_hip.inj_forward(mod, case["hidden"].to(gpu_device, torch.bfloat16), ctx)
        assert torch.isfinite(y).all() and "_pww_fused_scratch" not in mod.__dict__, shape


@pytest.mark.parametrize("shape", ["sd15_n4096", "sd21_n2304"])
def test_plug_with_to_out_inside_the_launch(gpu_device, experiments_lib, monkeypatch, shape):
    """PWW_FUSE_TO_OUT (row f-1, opt-in): inj_forward of a C = 320 cross-attention layer with the projection inside the attention launch
    against the default two-launch route -- the same module, the same PwW dict; eager and with a CFG-folded row gate. A layer the kernel
    does not cover (C = 640) silently keeps the two-launch route."""
    import pww_hip
    from pww_hip import attention
    monkeypatch.setitem(cases.ATTN_SHAPES, "sd21_n2304", (2304, 320, 5, 1024))      # SD2.1's finest level at 384 x 384: 5 heads x 64
    case = cases.make_attention_case(shape)
    mod = case["attn_cross"].to(gpu_device, torch.bfloat16)
    hidden = case["hidden"].to(gpu_device, torch.bfloat16)
    ctx = {"CONTEXT_TENSOR": case["ctx"].to(gpu_device, torch.bfloat16), f"CROSS_ATTENTION_WEIGHT_{case['N']}": case["w"].to(gpu_device),
           "SIGMA": torch.tensor(7.84), "WEIGHT_FUNCTION": cases.weight_fn_runner,
           attention.BIAS_COLS: 32}      # (the pipeline's hint: columns >= 32 of the map are zero -- the kernel takes maps of at most 64 columns)
    seen = []
    orig = attention.ops.attention_out
    monkeypatch.setattr(attention.ops, "attention_out", lambda *a, **kw: (seen.append(1), orig(*a, **kw))[1])
    two = pww_hip.inj_forward(mod, hidden, dict(ctx))
    assert not seen
    monkeypatch.setattr(attention, "FUSE_TO_OUT", True)
    one = pww_hip.inj_forward(mod, hidden, dict(ctx))
    assert seen, "the fused-projection launch was not taken"
    d = (one.double() - two.double()).abs()
    print(f"to_out inside the launch, {shape}: max |diff| {d.max().item():.2e} of max |y| {two.abs().max().item():.2e}, {int((d > 0).sum())} elements differ")
    assert torch.isfinite(one).all() and d.max().item() <= 2.0 ** -7 * two.abs().max().item()
    # uncovered layer: same route as before, same bits
    case2 = cases.make_attention_case("sd15_n1024")
    mod2 = case2["attn_cross"].to(gpu_device, torch.bfloat16)
    ctx2 = {"CONTEXT_TENSOR": case2["ctx"].to(gpu_device, torch.bfloat16), f"CROSS_ATTENTION_WEIGHT_{case2['N']}": case2["w"].to(gpu_device),
            "SIGMA": torch.tensor(7.84), "WEIGHT_FUNCTION": cases.weight_fn_runner}
    n = len(seen)
    y_on = pww_hip.inj_forward(mod2, case2["hidden"].to(gpu_device, torch.bfloat16), dict(ctx2))
    monkeypatch.setattr(attention, "FUSE_TO_OUT", False)
    y_off = pww_hip.inj_forward(mod2, case2["hidden"].to(gpu_device, torch.bfloat16), dict(ctx2))
    assert len(seen) == n and torch.equal(y_on, y_off)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("N,C,H", [(1024, 640, 8), (256, 1280, 8), (64, 1280, 8), (576, 1280, 20), (2304, 640, 10), (300, 768, 8),
                                   (600, 640, 8), (200, 1280, 8), (256, 1024, 8), (512, 768, 8), (130, 1280, 8), (515, 640, 4)])
def test_small_self_attention_vs_fp64_and_repeatable(gpu_device, dtype, N, C, H):
    """Self-attention at the coarse UNet levels (round 5: K fragments prefetched; N >= 512 at d = 80 / 96 and N >= 128 at d = 128 / 160 on
    the single-buffered key-split kernel -- ragged last stages, key groups that see no key at all): vs fp64 on the same rounded q / k / v
    inside the per-call bar, and bit-identical from run to run."""
    from pww_hip import ops
    D = C // H
    g = torch.Generator().manual_seed(N + C)
    qkv = (torch.randn(2, N, 3 * C, generator=g) * 0.8).to(dtype).to(gpu_device)
    q, k, v = qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:]
    out = ops.attention(q, k, v, H, D ** -0.5)
    qh = q.double().reshape(2, N, H, D).permute(0, 2, 1, 3)
    kh = k.double().reshape(2, N, H, D).permute(0, 2, 1, 3)
    vh = v.double().reshape(2, N, H, D).permute(0, 2, 1, 3)
    ref = (((qh @ kh.transpose(-1, -2)) * D ** -0.5).softmax(-1) @ vh).permute(0, 2, 1, 3).reshape(2, N, C)
    err = (out.double() - ref).abs().max().item() / ref.abs().max().item()
    print(f"self N={N} d={D} {dtype}: max err / max|O| = {err:.2e}")
    assert err <= TOL[dtype]
    assert torch.equal(out, ops.attention(q, k, v, H, D ** -0.5))


def test_single_buffer_key_split_kernel_behind_its_knob(gpu_device, experiments_lib):
    """The single-buffered key-split self-attention kernel of round 5 (attn_ksplit1_kernel: VERDICT round 4 item 1b asked for more waves in
    flight at N <= 1024) measured slower than the double-buffered form it was to replace (profiles/r05_small_attn.md) and is off by default;
    it stays correct behind PWW_DEBUG=attn_ksplit1=1: every head dim x token count of the sweep inside the per-call bar (own process: the
    library reads its knobs once)."""
    env = dict(os.environ, PWW_DEBUG="attn_ksplit1=1", PWW_HIP_LIB=experiments_lib)      # (an experiments-library kernel since round 6)
    out = subprocess.run(["timeout", "300", sys.executable, os.path.join(cases.REPO, "tools", "diag_selfattn_dims.py")], capture_output=True, text=True, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if " err " in l]
    assert len(lines) >= 90 and not [l for l in lines if "FAIL" in l], [l for l in lines if "FAIL" in l][:5]


@pytest.mark.parametrize("mode", ["eager", "graph"])
@pytest.mark.parametrize("strength", [0.5, 0.8])
def test_function_api_img2img_vs_the_references_own_loop(gpu_device, mode, strength):
    """paint_with_words(..., init_image=, strength=) -- the function API's img2img (paint_with_words.py:434-441 the
    shortened schedule, :459-468 vae.encode -> x 0.18215 -> add_noise with noise from the GLOBAL generator) -- against the final latents of
    the REFERENCE's own function (tests/golden/loop_tiny_img2img_lms10.npz, oracle/make_golden.py gen_img2img: tiny UNet, 10 LMS steps),
    the global generator seeded the same way, in the reference's call pattern (eager) and through the captured graph."""
    import paint_with_words as pw
    g = np.load(os.path.join(G, "loop_tiny_img2img_lms10.npz"))
    ref = g["latents_s%02d" % int(strength * 10)]
    mod = __import__("importlib").import_module("paint_with_words.paint_with_words")
    for dtype, bar in ((torch.float16, 1e-2), (torch.bfloat16, 5e-2)):
        tools = cases.build_tools("tiny", dtype=dtype, device=gpu_device)
        old, mod.DEFAULT_MODE = mod.DEFAULT_MODE, mode
        try:
            torch.manual_seed(int(g["global_seed"]))
            lat = pw.paint_with_words(color_context=dict(cases.RUNNER_CONTEXT), color_map_image=Image.fromarray(cases.load_example_rgb()),
                                      input_prompt=cases.RUNNER_PROMPT, num_inference_steps=int(g["steps"]), guidance_scale=7.5, seed=0, device=str(gpu_device),
                                      weight_function=cases.weight_fn_runner, preloaded_utils=tools, init_image=Image.fromarray(cases.synthetic_init_image(512, 3)),
                                      strength=strength, return_latents=True)
        finally:
            mod.DEFAULT_MODE = old
            uninstall_all()
        err = rel_l2(lat.float().cpu().numpy(), ref)
        print(f"function-API img2img strength {strength} {mode} {dtype}: rel L2 vs the reference's loop = {err:.3e}")
        assert lat.shape == ref.shape and err <= bar, err
    # the two strengths really are different runs (the other fixture is far away)
    other = g["latents_s%02d" % (8 if strength == 0.5 else 5)]
    assert rel_l2(lat.float().cpu().numpy(), other) > 0.2
