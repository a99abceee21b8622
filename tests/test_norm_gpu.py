"""GPU tests of pww_group_norm_fwd (csrc/pww_norm.hip) and of the block-level plug that uses it (pww_hip/blocks.py) -- SURVEY.md section 8
row a17: the GroupNorm (+ time-embedding addend, + SiLU) of the diffusers 0.10.0 blocks that call the attention path.

Reference of a floating-point kernel = plain PyTorch in fp32 with the SAME rounding points (x + t rounded to the storage type, the normalised
value rounded before the activation), statistics in fp64. Bars: an output may differ from that reference by one rounding step of the storage
type (the statistics' summation order moves a value across a rounding boundary now and then): |dy| <= 2^-7 |y| + 2^-7 * 1e-2 * max|y| for
bfloat16, 2^-10 for float16 -- and against the STOCK half-precision op sequence on the same GPU by two steps.
"""
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DEV = "cuda:0"
ULP = {torch.bfloat16: 2.0 ** -7, torch.float16: 2.0 ** -10}


def _reference(x, add, weight, bias, groups, eps, act, dtype):
    """fp32 / fp64 torch with the kernel's rounding points (NCHW logical layout)."""
    h = x.float()
    if add is not None:
        h = (h + add.float()[:, :, None, None]).to(dtype).float()
    B, C, H, W = h.shape
    hd = h.double().reshape(B, groups, -1)
    mean = hd.mean(-1)
    var = hd.var(-1, unbiased=False)
    rstd = (1.0 / torch.sqrt(var + eps)).float()
    a = rstd[:, :, None] * (weight.float() if weight is not None else torch.ones(C, device=x.device)).reshape(1, groups, -1)
    b = (bias.float() if bias is not None else torch.zeros(C, device=x.device)).reshape(1, groups, -1) - a * mean.float()[:, :, None]
    y = (h * a.reshape(B, C, 1, 1) + b.reshape(B, C, 1, 1)).to(dtype).float()
    if act == "silu":
        y = y / (1.0 + torch.exp(-y))
    return y.to(dtype)


def _close(y, ref, dtype, steps=1):
    yf, rf = y.float(), ref.float()
    tol = steps * ULP[dtype] * (rf.abs() + 1e-2 * rf.abs().max())
    bad = ((yf - rf).abs() > tol)
    return int(bad.sum()), float((yf - rf).abs().max() / rf.abs().max())


# the norms of the SD1.5 / SD2.1 UNets at 2 folded rows (and one 16-row case): (B, C, H, W)
SHAPES = [(2, 320, 64, 64), (2, 640, 32, 32), (2, 1280, 16, 16), (2, 1280, 8, 8), (2, 2560, 8, 8), (2, 1920, 16, 16), (2, 960, 32, 32),
          (16, 640, 32, 32), (2, 320, 96, 96), (3, 64, 8, 8), (1, 32, 4, 2),
          # (round 6) more of the shapes that take the two-launch form at 2 rows (10- and 30-channel groups, the 64 x 64 level), one odd batch and one
          # non-square level; [2, 320, 96, 96] above is SD2.1's
          (2, 640, 64, 64), (2, 960, 64, 64), (2, 1280, 32, 32), (2, 1920, 32, 32), (2, 320, 32, 32), (1, 320, 64, 64), (3, 320, 40, 24)]


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("channels_last", [False, True])
@pytest.mark.parametrize("shape", SHAPES)
def test_group_norm_matches_the_fp32_reference(shape, channels_last, dtype):
    from pww_hip import ops
    B, C, H, W = shape
    g = torch.Generator(device="cpu").manual_seed(B * 1000 + C + H)
    x = (torch.randn(shape, generator=g) * 1.7 + 0.4 * torch.randn(1, C, 1, 1, generator=g)).to(DEV, dtype)
    if channels_last:
        x = x.contiguous(memory_format=torch.channels_last)
    w = (1.0 + 0.3 * torch.randn(C, generator=g)).to(DEV, dtype)
    b = (0.2 * torch.randn(C, generator=g)).to(DEV, dtype)
    add = (0.8 * torch.randn(B, C, generator=g)).to(DEV, dtype)
    for use_add, act in ((False, None), (False, "silu"), (True, "silu"), (True, None)):
        y = ops.group_norm(x, 32, w, b, 1e-5, add=add if use_add else None, act=act)
        assert y.shape == x.shape and y.stride() == x.stride() and y.dtype == dtype
        ref = _reference(x, add if use_add else None, w, b, 32, 1e-5, act, dtype)
        # (SiLU amplifies one rounding step of a normalised value of -4 .. -6 fourfold relative to its small result: two steps there)
        nbad, rel = _close(y, ref, dtype, steps=2 if act else 1)
        assert nbad == 0, (shape, channels_last, dtype, use_add, act, nbad, rel)
        # the stock op sequence on tensors of the same type: ATen's fp32 Welford statistics put 0.5 % of ITS outputs one step away from the
        # fp64-statistics reference (measured: 14 670 of 2.6 M elements at [2, 320, 64, 64]), so this bar is relative to the tensor's maximum
        h = x if not use_add else x + add[:, :, None, None]
        stock = F.group_norm(h, 32, w, b, 1e-5)
        stock = F.silu(stock) if act == "silu" else stock
        rel2 = _close(y, stock, dtype)[1]
        assert rel2 <= 2 * ULP[dtype], ("vs stock", shape, channels_last, dtype, use_add, act, rel2)
    # the addend as rows of a wider tensor (how the block plug hands over one batched time-embedding projection): no copy, same bits
    wide = torch.zeros(B, C + 64, device=DEV, dtype=dtype)
    wide[:, 32:32 + C] = add
    assert torch.equal(ops.group_norm(x, 32, w, b, 1e-5, add=wide[:, 32:32 + C], act="silu"), ops.group_norm(x, 32, w, b, 1e-5, add=add, act="silu"))


def test_group_norm_statistics_survive_a_large_mean():
    """mean 300, spread 0.5 -- sum-of-squares minus mean^2 in fp64 of bf16-rounded inputs; no affine."""
    from pww_hip import ops
    g = torch.Generator(device="cpu").manual_seed(5)
    for cl in (False, True):
        x = (300.0 + 0.5 * torch.randn(2, 320, 64, 64, generator=g)).to(DEV, torch.bfloat16)
        x = x.contiguous(memory_format=torch.channels_last) if cl else x
        y = ops.group_norm(x, 32, None, None, 1e-6)
        ref = _reference(x, None, None, None, 32, 1e-6, None, torch.bfloat16)
        nbad, rel = _close(y, ref, torch.bfloat16)
        assert nbad == 0, (cl, nbad, rel)
        assert abs(float(y.float().mean())) < 2e-2 and abs(float(y.float().std()) - 1.0) < 2e-2


def test_group_norm_in_place_graph_replay_and_errors():
    from pww_hip import ops, PwwHipError
    g = torch.Generator(device="cpu").manual_seed(9)
    x = torch.randn(2, 640, 32, 32, generator=g).to(DEV, torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = torch.randn(640, generator=g).to(DEV, torch.bfloat16)
    b = torch.randn(640, generator=g).to(DEV, torch.bfloat16)
    y = ops.group_norm(x, 32, w, b, 1e-5, act="silu")
    x2 = x.clone(memory_format=torch.preserve_format)
    assert ops.group_norm(x2, 32, w, b, 1e-5, act="silu", out=x2) is x2 and torch.equal(x2, y)
    # replayed from a hipGraph: bit-identical every time (no state between launches)
    static = x.clone(memory_format=torch.preserve_format)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        ops.group_norm(static, 32, w, b, 1e-5, act="silu")
    torch.cuda.current_stream().wait_stream(s)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        out = ops.group_norm(static, 32, w, b, 1e-5, act="silu")
    for _ in range(3):
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, y)
    # run to run: fixed summation order
    assert torch.equal(ops.group_norm(x, 32, w, b, 1e-5, act="silu"), y)
    with pytest.raises(PwwHipError):
        ops.group_norm(x.float(), 32)
    with pytest.raises(PwwHipError):
        ops.group_norm(torch.zeros(2, 36, 8, 8, device=DEV, dtype=torch.bfloat16), 4)          # C % 8
    with pytest.raises(PwwHipError):
        ops.group_norm(torch.zeros(2, 64, 3, 3, device=DEV, dtype=torch.bfloat16), 32)         # HW % 8
    with pytest.raises(PwwHipError):
        ops.group_norm(x, 32, w.float(), b)


@pytest.mark.parametrize("channels_last", [False, True])
def test_resnet_block_and_transformer_norm_through_the_plug(channels_last):
    """install() puts the fused norms under the stand-in's ResnetBlock2D / GroupNorm instances; the patched block equals the original
    forward to two rounding steps of bf16 per norm (compared on the block's output, relative to its spread); uninstall() restores."""
    import pww_hip
    from pww_hip import blocks
    from sd_standin import unet as U
    torch.manual_seed(3)
    res = U.ResnetBlock2D(320, 640, 1280).to(DEV, torch.bfloat16).eval().requires_grad_(False)
    norm = nn.GroupNorm(32, 640, eps=1e-6).to(DEV, torch.bfloat16)
    holder = nn.ModuleList([res, norm])
    x = torch.randn(2, 320, 32, 32, device=DEV).to(torch.bfloat16)
    temb = torch.randn(2, 1280, device=DEV).to(torch.bfloat16)
    if channels_last:
        holder.to(memory_format=torch.channels_last)
        x = x.contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        want = res(x, temb)
        want_n = norm(want)
        assert blocks.install_blocks(holder) == (1, 1)
        calls = []
        orig = blocks.ops.group_norm
        blocks.ops.group_norm = lambda *a, **k: (calls.append(k.get("act")), orig(*a, **k))[1]
        try:
            got = res(x, temb)
            got_n = norm(want)
            with torch.autocast("cuda", dtype=torch.float16):           # autocast runs group_norm in fp32: the stock op keeps it
                res(x, temb)
        finally:
            blocks.ops.group_norm = orig
        assert calls == ["silu", "silu", None]
        spread = want.float().std().item()
        assert (got.float() - want.float()).abs().max().item() <= 4 * ULP[torch.bfloat16] * 4 * spread
        assert _close(got_n, want_n, torch.bfloat16, steps=2)[0] == 0
        blocks.uninstall_blocks(holder)
        assert "forward" not in res.__dict__ and "forward" not in norm.__dict__
        # (the stock convolutions are not bit-repeatable run to run: DESIGN.md section 2)
        assert (res(x, temb).float() - want.float()).abs().max().item() <= 4 * ULP[torch.bfloat16] * 4 * spread
    assert pww_hip.blocks is blocks


# ---- elementwise glue of the same blocks (csrc/pww_blocks.hip) ------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("shape", [(2, 4096, 320), (2, 1024, 640), (2, 256, 1280), (16, 64, 1280), (3, 50, 1288), (1, 7, 2048), (5, 8)])
def test_add_layer_norm_matches_the_stock_sequence(shape, dtype):
    """s = a + x and LayerNorm(s) from one launch: s bit-equal to the stock add; the norm within one rounding step of an fp32 LayerNorm of
    that s (and of the stock half-precision LayerNorm); strided rows (a column slice of a wider tensor) give the same bits."""
    from pww_hip import ops
    g = torch.Generator(device="cpu").manual_seed(sum(shape))
    C = shape[-1]
    x = (torch.randn(shape, generator=g) * 2.0 + 0.3).to(DEV, dtype)
    a = torch.randn(shape, generator=g).to(DEV, dtype)
    w = (1.0 + 0.2 * torch.randn(C, generator=g)).to(DEV, dtype)
    b = (0.1 * torch.randn(C, generator=g)).to(DEV, dtype)
    s, y = ops.add_layer_norm(x, w, b, 1e-5, a=a)
    assert torch.equal(s, a + x)
    ref = F.layer_norm(s.float(), (C,), w.float(), b.float(), 1e-5).to(dtype)
    assert _close(y, ref, dtype)[0] == 0
    assert _close(y, F.layer_norm(s, (C,), w, b, 1e-5), dtype, steps=2)[0] == 0
    y0 = ops.add_layer_norm(s, w, b, 1e-5)
    assert torch.equal(y0, y)                                   # plain form on the same rows: same arithmetic
    wide = torch.zeros(shape[:-1] + (C + 24,), device=DEV, dtype=dtype)
    wide[..., 8:8 + C] = x
    s2, y2 = ops.add_layer_norm(wide[..., 8:8 + C], w, b, 1e-5, a=a)
    assert torch.equal(s2, s) and torch.equal(y2, y)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("shape", [(2, 4096, 2560), (2, 256, 10240), (16, 64, 10240), (3, 5, 48)])
def test_geglu_matches_the_stock_sequence(shape, dtype):
    from pww_hip import ops
    g = torch.Generator(device="cpu").manual_seed(shape[1])
    h = (torch.randn(shape, generator=g) * 1.5).to(DEV, dtype)
    y = ops.geglu(h)
    xa, gate = h.chunk(2, dim=-1)
    stock = xa * F.gelu(gate)
    ref = (xa.float() * F.gelu(gate.float()).to(dtype).float()).to(dtype)
    assert y.shape == stock.shape and _close(y, ref, dtype)[0] == 0 and _close(y, stock, dtype)[0] == 0


@pytest.mark.parametrize("channels_last", [False, True])
def test_bias_residual_is_the_stock_sequence_bit_for_bit(channels_last):
    from pww_hip import ops
    g = torch.Generator(device="cpu").manual_seed(4)
    for dtype in (torch.bfloat16, torch.float16):
        for shape in ((2, 320, 64, 64), (2, 1280, 8, 8), (3, 64, 4, 2)):
            r = torch.randn(shape, generator=g).to(DEV, dtype)
            v = torch.randn(shape, generator=g).to(DEV, dtype)
            bias = torch.randn(shape[1], generator=g).to(DEV, dtype)
            if channels_last:
                r, v = r.contiguous(memory_format=torch.channels_last), v.contiguous(memory_format=torch.channels_last)
            y = ops.bias_residual(r, v, bias)
            assert y.stride() == v.stride() and torch.equal(y, r + (v + bias[None, :, None, None]))


@pytest.mark.parametrize("channels_last", [False, True])
def test_transformer_model_through_the_plug(channels_last):
    """Transformer2DModel of the stand-in (GroupNorm -> 1x1 proj_in -> BasicTransformerBlock{LayerNorms, attn1, attn2, GEGLU feed-forward} ->
    1x1 proj_out -> residual) with install(): every fused op is taken (and the attention runs on the HIP path), the output stays within the
    half-precision noise of the unplugged module."""
    import pww_hip
    from pww_hip import blocks
    from sd_standin import unet as U
    torch.manual_seed(5)
    tm = U.Transformer2DModel(8, 40, 320, 768).to(DEV, torch.bfloat16).eval().requires_grad_(False)
    x = torch.randn(2, 320, 32, 32, device=DEV).to(torch.bfloat16)
    ctx = torch.randn(2, 77, 768, device=DEV).to(torch.bfloat16)
    if channels_last:
        tm.to(memory_format=torch.channels_last)
        x = x.contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        pww_hip.uninstall(tm)
        want = tm(x, ctx)
        pww_hip.install(tm)
        seen = []
        names = ("group_norm", "add_layer_norm", "geglu")
        orig = {n: getattr(blocks.ops, n) for n in names}
        for n in names:
            setattr(blocks.ops, n, (lambda n: lambda *a, **k: (seen.append(n), orig[n](*a, **k))[1])(n))
        try:
            got = tm(x, ctx)
        finally:
            for n in names:
                setattr(blocks.ops, n, orig[n])
            pww_hip.uninstall(tm)
        assert seen == ["group_norm", "add_layer_norm", "add_layer_norm", "add_layer_norm", "geglu"]
        assert got.shape == want.shape and got.stride() == want.stride()
        assert (got.float() - want.float()).abs().max().item() <= 3e-2 * want.float().abs().max().item()
        assert "forward" not in tm.proj_in.__dict__ and "forward" not in tm.transformer_blocks[0].__dict__


def test_conv1x1_as_gemm_on_channels_last():
    from pww_hip import blocks
    torch.manual_seed(6)
    conv = nn.Conv2d(320, 640, 1).to(DEV, torch.bfloat16)
    holder = nn.ModuleList([conv])
    x = torch.randn(2, 320, 16, 16, device=DEV).to(torch.bfloat16)
    xcl = x.contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        want = conv(xcl)
        blocks.install_blocks(holder)
        got = conv(xcl)
        assert got.shape == want.shape and got.is_contiguous(memory_format=torch.channels_last)
        assert (got.float() - want.float()).abs().max().item() <= 2 * ULP[torch.bfloat16] * want.float().abs().max().item()
        assert torch.equal(conv(x), conv._pww_orig_forward(x))              # NCHW input: the convolution itself
        blocks.uninstall_blocks(holder)
