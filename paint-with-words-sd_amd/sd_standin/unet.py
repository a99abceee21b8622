"""Stand-in for the diffusers==0.10.0 ``UNet2DConditionModel`` the reference drives.

``diffusers`` is pinned by the reference (requirements.txt:1) but is neither vendored in it nor
installable here (no network), so the SD1.x / SD2.x UNet topology is restated from the published
architecture (config mapping visible in the reference's change_model_path.py:231-275): the modules
the Paint-with-Words hot path touches keep the exact class and attribute names its code relies on
(paint_with_words/paint_with_words.py:76-85, :112, :118-123, :193-195):

  * a class literally named ``CrossAttention`` with ``to_q/to_k/to_v`` (no bias),
    ``to_out = [Linear, Dropout]``, ``scale``, ``heads``, ``reshape_heads_to_batch_dim`` and
    ``reshape_batch_dim_to_heads``;
  * ``BasicTransformerBlock`` calling ``attn1(norm_x)`` and ``attn2(norm_x, context=...)`` and
    handing ``encoder_hidden_states`` through UNTOUCHED (it is a dict in the PwW protocol,
    :370-386);
  * ``UNet2DConditionModel(sample, timestep, encoder_hidden_states).sample`` and ``.in_channels``.

Weights are random-init from a seed (there are no checkpoints offline); this is the "synthetic
random-init UNet" of BASELINE.json. Everything that is not attention stays stock PyTorch ops
(MIOpen convolutions, hipBLASLt GEMMs on ROCm).
"""
import math
from types import SimpleNamespace

import torch
import torch.nn as nn
import torch.nn.functional as F


class CrossAttention(nn.Module):
    """Attention module whose ``__call__`` the reference replaces at class level (:193-195)."""

    def __init__(self, query_dim, cross_attention_dim=None, heads=8, dim_head=64, dropout=0.0):
        super().__init__()
        inner_dim = dim_head * heads
        cross_attention_dim = cross_attention_dim if cross_attention_dim is not None else query_dim
        self.scale = dim_head ** -0.5
        self.heads = heads
        self.to_q = nn.Linear(query_dim, inner_dim, bias=False)
        self.to_k = nn.Linear(cross_attention_dim, inner_dim, bias=False)
        self.to_v = nn.Linear(cross_attention_dim, inner_dim, bias=False)
        self.to_out = nn.ModuleList([nn.Linear(inner_dim, query_dim), nn.Dropout(dropout)])

    def reshape_heads_to_batch_dim(self, tensor):
        b, s, c = tensor.shape
        h = self.heads
        return tensor.reshape(b, s, h, c // h).permute(0, 2, 1, 3).reshape(b * h, s, c // h)

    def reshape_batch_dim_to_heads(self, tensor):
        bh, s, d = tensor.shape
        h = self.heads
        return tensor.reshape(bh // h, h, s, d).permute(0, 2, 1, 3).reshape(bh // h, s, d * h)

    def forward(self, hidden_states, context=None, mask=None):
        # vanilla (un-patched) attention; never used once the PwW plug is installed
        context = hidden_states if context is None else context
        q = self.reshape_heads_to_batch_dim(self.to_q(hidden_states))
        k = self.reshape_heads_to_batch_dim(self.to_k(context))
        v = self.reshape_heads_to_batch_dim(self.to_v(context))
        probs = (torch.matmul(q, k.transpose(-1, -2)) * self.scale).softmax(dim=-1)
        out = self.reshape_batch_dim_to_heads(torch.matmul(probs, v))
        return self.to_out[1](self.to_out[0](out))


class GEGLU(nn.Module):
    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out * 2)

    def forward(self, x):
        x, gate = self.proj(x).chunk(2, dim=-1)
        return x * F.gelu(gate)


class FeedForward(nn.Module):
    def __init__(self, dim, mult=4, dropout=0.0):
        super().__init__()
        self.net = nn.ModuleList([GEGLU(dim, dim * mult), nn.Dropout(dropout), nn.Linear(dim * mult, dim)])

    def forward(self, x):
        for m in self.net:
            x = m(x)
        return x


class BasicTransformerBlock(nn.Module):
    def __init__(self, dim, heads, dim_head, cross_attention_dim):
        super().__init__()
        self.attn1 = CrossAttention(dim, None, heads, dim_head)
        self.ff = FeedForward(dim)
        self.attn2 = CrossAttention(dim, cross_attention_dim, heads, dim_head)
        self.norm1 = nn.LayerNorm(dim)
        self.norm2 = nn.LayerNorm(dim)
        self.norm3 = nn.LayerNorm(dim)

    def forward(self, hidden_states, context=None):
        hidden_states = self.attn1(self.norm1(hidden_states)) + hidden_states
        hidden_states = self.attn2(self.norm2(hidden_states), context=context) + hidden_states
        hidden_states = self.ff(self.norm3(hidden_states)) + hidden_states
        return hidden_states


class Transformer2DModel(nn.Module):
    def __init__(self, heads, dim_head, in_channels, cross_attention_dim, norm_num_groups=32,
                 use_linear_projection=False, depth=1):
        super().__init__()
        inner = heads * dim_head
        self.use_linear_projection = use_linear_projection
        self.norm = nn.GroupNorm(norm_num_groups, in_channels, eps=1e-6, affine=True)
        if use_linear_projection:
            self.proj_in = nn.Linear(in_channels, inner)
            self.proj_out = nn.Linear(inner, in_channels)
        else:
            self.proj_in = nn.Conv2d(in_channels, inner, 1)
            self.proj_out = nn.Conv2d(inner, in_channels, 1)
        self.transformer_blocks = nn.ModuleList(
            [BasicTransformerBlock(inner, heads, dim_head, cross_attention_dim) for _ in range(depth)])

    def forward(self, hidden_states, encoder_hidden_states=None):
        b, c, h, w = hidden_states.shape
        residual = hidden_states
        x = self.norm(hidden_states)
        if self.use_linear_projection:
            x = self.proj_in(x.permute(0, 2, 3, 1).reshape(b, h * w, c))
        else:
            x = self.proj_in(x).permute(0, 2, 3, 1).reshape(b, h * w, -1)
        for blk in self.transformer_blocks:
            x = blk(x, context=encoder_hidden_states)
        if self.use_linear_projection:
            x = self.proj_out(x).reshape(b, h, w, c).permute(0, 3, 1, 2)
        else:
            x = self.proj_out(x.reshape(b, h, w, -1).permute(0, 3, 1, 2))
        return x + residual


class ResnetBlock2D(nn.Module):
    def __init__(self, in_channels, out_channels, temb_channels, groups=32, eps=1e-5):
        super().__init__()
        self.norm1 = nn.GroupNorm(groups, in_channels, eps=eps)
        self.conv1 = nn.Conv2d(in_channels, out_channels, 3, padding=1)
        self.time_emb_proj = nn.Linear(temb_channels, out_channels)
        self.norm2 = nn.GroupNorm(groups, out_channels, eps=eps)
        self.conv2 = nn.Conv2d(out_channels, out_channels, 3, padding=1)
        self.conv_shortcut = nn.Conv2d(in_channels, out_channels, 1) if in_channels != out_channels else None

    def forward(self, x, temb):
        h = self.conv1(F.silu(self.norm1(x)))
        h = h + self.time_emb_proj(F.silu(temb))[:, :, None, None]
        h = self.conv2(F.silu(self.norm2(h)))
        if self.conv_shortcut is not None:
            x = self.conv_shortcut(x)
        return x + h


class Downsample2D(nn.Module):
    def __init__(self, channels):
        super().__init__()
        self.conv = nn.Conv2d(channels, channels, 3, stride=2, padding=1)

    def forward(self, x):
        return self.conv(x)


class Upsample2D(nn.Module):
    def __init__(self, channels):
        super().__init__()
        self.conv = nn.Conv2d(channels, channels, 3, padding=1)

    def forward(self, x):
        return self.conv(F.interpolate(x, scale_factor=2.0, mode="nearest"))


class DownBlock(nn.Module):
    def __init__(self, in_ch, out_ch, temb_ch, layers, heads, cross_dim, has_attn, add_down, groups, linear_proj):
        super().__init__()
        self.resnets = nn.ModuleList(
            [ResnetBlock2D(in_ch if i == 0 else out_ch, out_ch, temb_ch, groups) for i in range(layers)])
        self.attentions = nn.ModuleList(
            [Transformer2DModel(heads, out_ch // heads, out_ch, cross_dim, groups, linear_proj)
             for _ in range(layers)]) if has_attn else None
        self.downsamplers = nn.ModuleList([Downsample2D(out_ch)]) if add_down else None

    def forward(self, x, temb, ctx):
        outs = ()
        for i, res in enumerate(self.resnets):
            x = res(x, temb)
            if self.attentions is not None:
                x = self.attentions[i](x, ctx)
            outs += (x,)
        if self.downsamplers is not None:
            x = self.downsamplers[0](x)
            outs += (x,)
        return x, outs


class MidBlock(nn.Module):
    def __init__(self, ch, temb_ch, heads, cross_dim, groups, linear_proj):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(ch, ch, temb_ch, groups), ResnetBlock2D(ch, ch, temb_ch, groups)])
        self.attentions = nn.ModuleList([Transformer2DModel(heads, ch // heads, ch, cross_dim, groups, linear_proj)])

    def forward(self, x, temb, ctx):
        x = self.resnets[0](x, temb)
        x = self.attentions[0](x, ctx)
        return self.resnets[1](x, temb)


class UpBlock(nn.Module):
    def __init__(self, in_ch, out_ch, prev_ch, temb_ch, layers, heads, cross_dim, has_attn, add_up, groups, linear_proj):
        super().__init__()
        resnets = []
        for i in range(layers):
            skip = in_ch if i == layers - 1 else out_ch
            res_in = prev_ch if i == 0 else out_ch
            resnets.append(ResnetBlock2D(res_in + skip, out_ch, temb_ch, groups))
        self.resnets = nn.ModuleList(resnets)
        self.attentions = nn.ModuleList(
            [Transformer2DModel(heads, out_ch // heads, out_ch, cross_dim, groups, linear_proj)
             for _ in range(layers)]) if has_attn else None
        self.upsamplers = nn.ModuleList([Upsample2D(out_ch)]) if add_up else None

    def forward(self, x, skips, temb, ctx):
        for i, res in enumerate(self.resnets):
            x = torch.cat([x, skips[-1]], dim=1)
            skips = skips[:-1]
            x = res(x, temb)
            if self.attentions is not None:
                x = self.attentions[i](x, ctx)
        if self.upsamplers is not None:
            x = self.upsamplers[0](x)
        return x, skips


# SD2.x-flavoured reduced config: head dim 64 everywhere (heads 1/2/4/4), linear projections, 768x768 -> sample 96
TINY_SD2_CONFIG = dict(sample_size=96, in_channels=4, out_channels=4, block_out_channels=(64, 128, 256, 256),
                       layers_per_block=1, attention_head_dim=(1, 2, 4, 4), cross_attention_dim=96,
                       norm_num_groups=16, use_linear_projection=True)


def timestep_embedding(timesteps, dim):
    """Sinusoidal embedding, flip_sin_to_cos=True, freq_shift=0 (the SD configuration)."""
    half = dim // 2
    exponent = -math.log(10000.0) * torch.arange(half, dtype=torch.float32, device=timesteps.device) / half
    emb = timesteps[:, None].float() * torch.exp(exponent)[None, :]
    return torch.cat([torch.cos(emb), torch.sin(emb)], dim=-1)


SD15_CONFIG = dict(sample_size=64, in_channels=4, out_channels=4, block_out_channels=(320, 640, 1280, 1280),
                   layers_per_block=2, attention_head_dim=8, cross_attention_dim=768,
                   norm_num_groups=32, use_linear_projection=False)
SD15_INPAINT_CONFIG = dict(SD15_CONFIG, in_channels=9)
SD21_CONFIG = dict(sample_size=96, in_channels=4, out_channels=4, block_out_channels=(320, 640, 1280, 1280),
                   layers_per_block=2, attention_head_dim=(5, 10, 20, 20), cross_attention_dim=1024,
                   norm_num_groups=32, use_linear_projection=True)
# Same block structure at 1/8 width: used by CPU tests so the oracle loop finishes in seconds.
TINY_CONFIG = dict(sample_size=64, in_channels=4, out_channels=4, block_out_channels=(32, 64, 128, 128),
                   layers_per_block=1, attention_head_dim=4, cross_attention_dim=64,
                   norm_num_groups=8, use_linear_projection=False)


class UNet2DConditionModel(nn.Module):
    def __init__(self, sample_size=64, in_channels=4, out_channels=4, block_out_channels=(320, 640, 1280, 1280),
                 layers_per_block=2, attention_head_dim=8, cross_attention_dim=768, norm_num_groups=32,
                 use_linear_projection=False):
        super().__init__()
        self.in_channels = in_channels
        self.config = SimpleNamespace(sample_size=sample_size, in_channels=in_channels, out_channels=out_channels,
                                      block_out_channels=tuple(block_out_channels), layers_per_block=layers_per_block,
                                      attention_head_dim=attention_head_dim, cross_attention_dim=cross_attention_dim)
        nblocks = len(block_out_channels)
        heads = attention_head_dim if isinstance(attention_head_dim, (tuple, list)) else (attention_head_dim,) * nblocks
        temb_ch = block_out_channels[0] * 4
        g = norm_num_groups
        self.time_proj_dim = block_out_channels[0]
        self.time_embedding = nn.ModuleList([nn.Linear(block_out_channels[0], temb_ch), nn.Linear(temb_ch, temb_ch)])
        self.conv_in = nn.Conv2d(in_channels, block_out_channels[0], 3, padding=1)

        self.down_blocks = nn.ModuleList()
        out_ch = block_out_channels[0]
        for i in range(nblocks):
            in_ch, out_ch = out_ch, block_out_channels[i]
            last = i == nblocks - 1
            self.down_blocks.append(DownBlock(in_ch, out_ch, temb_ch, layers_per_block, heads[i], cross_attention_dim,
                                              has_attn=not last, add_down=not last, groups=g,
                                              linear_proj=use_linear_projection))
        self.mid_block = MidBlock(block_out_channels[-1], temb_ch, heads[-1], cross_attention_dim, g,
                                  use_linear_projection)
        self.up_blocks = nn.ModuleList()
        rev = list(reversed(block_out_channels))
        rev_heads = list(reversed(heads))
        out_ch = rev[0]
        for i in range(nblocks):
            prev_ch, out_ch = out_ch, rev[i]
            in_ch = rev[min(i + 1, nblocks - 1)]
            last = i == nblocks - 1
            self.up_blocks.append(UpBlock(in_ch, out_ch, prev_ch, temb_ch, layers_per_block + 1, rev_heads[i],
                                          cross_attention_dim, has_attn=i > 0, add_up=not last, groups=g,
                                          linear_proj=use_linear_projection))
        self.conv_norm_out = nn.GroupNorm(g, block_out_channels[0], eps=1e-5)
        self.conv_out = nn.Conv2d(block_out_channels[0], out_channels, 3, padding=1)

    @property
    def dtype(self):
        return self.conv_in.weight.dtype

    @property
    def device(self):
        return self.conv_in.weight.device

    def forward(self, sample, timestep, encoder_hidden_states=None):
        if not torch.is_tensor(timestep):
            timestep = torch.tensor([timestep], dtype=torch.float32, device=sample.device)
        timestep = timestep.to(sample.device).reshape(-1).expand(sample.shape[0])
        temb = timestep_embedding(timestep, self.time_proj_dim).to(self.dtype)
        temb = self.time_embedding[1](F.silu(self.time_embedding[0](temb)))

        x = self.conv_in(sample.to(self.dtype))
        skips = (x,)
        for blk in self.down_blocks:
            x, outs = blk(x, temb, encoder_hidden_states)
            skips += outs
        x = self.mid_block(x, temb, encoder_hidden_states)
        for blk in self.up_blocks:
            x, skips = blk(x, skips, temb, encoder_hidden_states)
        x = self.conv_out(F.silu(self.conv_norm_out(x)))
        return SimpleNamespace(sample=x)


def build_unet(config=None, seed=1234, dtype=torch.float32, device="cpu", qk_gain=1.0):
    """Random-init UNet with a fixed seed (seed 1234, default nn init).

    ``qk_gain`` optionally scales the to_q weights so random-init attention logits have a
    non-trivial spread (default init gives near-uniform softmax); 1.0 keeps the default init.
    """
    cfg = dict(SD15_CONFIG if config is None else config)
    gen_state = torch.random.get_rng_state()
    torch.manual_seed(seed)
    try:
        unet = UNet2DConditionModel(**cfg)
    finally:
        torch.random.set_rng_state(gen_state)
    if qk_gain != 1.0:
        with torch.no_grad():
            for m in unet.modules():
                if isinstance(m, CrossAttention):
                    m.to_q.weight.mul_(qk_gain)
    return unet.to(device=device, dtype=dtype).eval().requires_grad_(False)
