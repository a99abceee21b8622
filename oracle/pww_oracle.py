"""CPU oracle for the Paint-with-Words hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may import this module, and
only as the CHECKER. The product (paint-with-words-sd_amd/) never imports it and fails loudly when
libpww_hip.so is missing.

Every function restates one piece of the reference algorithm (cloneofsimo/paint-with-words-sd,
file:line given per function, paths relative to the reference root) in plain numpy / fp32 torch-CPU
arithmetic. It is a restatement, not a copy: integer / byte / mask work is explicit numpy loops and
index arithmetic (including the bilinear formulas, which the reference delegates to ATen), the
floating-point attention is fp32 torch on CPU.

PARITY PIN: the reference ships no tests or golden vectors, so this oracle is
pinned against OUTPUTS OF THE REFERENCE ITSELF: oracle/make_golden.py executes the reference's own
functions (AST-loaded, unmodified -- oracle/ref_loader.py) in the build container and commits the
results under tests/golden/; tests/test_oracle_golden.py checks this module against them.
"""
import math
from types import SimpleNamespace

import numpy as np
import torch

# --------------------------------------------------------------------------------------------
# small helpers


def always_round(x):
    """Round half up for x >= 0 (paint_with_words/paint_with_words.py:18-26: even integer part ->
    explicit half-up; odd integer part -> Python round(), whose half-to-even also rounds up)."""
    return int(math.floor(x + 0.5))


def extract_seed_and_sigma(color_context, ignore_seed=-1):
    """:279-297. "text,strength[,seed[,sigma]]" -> strips seed/sigma, MUTATING the dict like the
    reference (:296); returns (color_context, {ordinal: seed}, {ordinal: sigma})."""
    seeds, sigmas = {}, {}
    for i, (key, value) in enumerate(list(color_context.items())):
        parts = value.split(",")
        if len(parts) > 2:
            try:
                seed = int(parts[-2])
                sigma = float(parts[-1])
                parts = parts[:-2]
                sigmas[i] = sigma
            except ValueError:
                seed = int(parts[-1])
                parts = parts[:-1]
            if seed != ignore_seed:
                seeds[i] = seed
        color_context[key] = ",".join(parts)
    return color_context, seeds, sigmas


def _parse_color(color):
    """:228-230: "#rrggbb" or an (r, g, b) tuple."""
    if isinstance(color, str):
        return (int(color[1:3], 16), int(color[3:5], 16), int(color[5:7], 16))
    return tuple(int(c) for c in color)


def separate_regions(img_rgb, color_context, tokenizer, verbose=False):
    """:207-244. img_rgb: uint8 [H, W, 3] (or None). Returns ([(token_ids, float32 mask[H, W])], W, H)
    with mask = strength * (pixel == color), exact RGB equality (:231-236)."""
    regions = []
    if img_rgb is not None:
        H, W = img_rgb.shape[:2]
        for color, value in color_context.items():
            fields = value.split(",")
            strength = float(fields[-1])
            text = ",".join(fields[:-1])
            ids = tokenizer(text, max_length=tokenizer.model_max_length, truncation=True)["input_ids"][1:-1]
            rgb = np.array(_parse_color(color), dtype=img_rgb.dtype)
            hit = (img_rgb == rgb[None, None, :]).all(axis=-1)
            if verbose and not hit.any():
                print(f"Warning : not a single color {tuple(rgb)} not found in image")
            regions.append((list(ids), hit.astype(np.float32) * np.float32(strength)))
    else:
        W, H = 512, 512
    if not regions:
        regions.append(([-1], np.zeros((W, H), dtype=np.float32)))
    return regions, W, H


def _source_index_weights(in_size, out_size, align_corners):
    """ATen's linear-interpolation index/weight computation (what F.interpolate(mode='bilinear') runs,
    reference :39-45 align_corners=True and :302-303 align_corners=False), all in fp32."""
    f = np.float32
    i0 = np.zeros(out_size, dtype=np.int64)
    lam = np.zeros(out_size, dtype=np.float32)
    if align_corners:
        scale = f(in_size - 1) / f(out_size - 1) if out_size > 1 else f(0)
    else:
        scale = f(in_size) / f(out_size)
    for o in range(out_size):
        if align_corners:
            src = scale * f(o)
        else:
            src = scale * (f(o) + f(0.5)) - f(0.5)
            if src < 0:
                src = f(0)
        idx = int(src)
        if idx > in_size - 1:
            idx = in_size - 1
        i0[o] = idx
        lam[o] = min(max(f(src) - f(idx), f(0)), f(1))
    i1 = np.minimum(i0 + 1, in_size - 1)
    return i0, i1, lam


def bilinear_resize(img, out_h, out_w, align_corners=True):
    """Bilinear resize of a float32 [H, W] map: out = hy*(hx*p00 + lx*p01) + ly*(hx*p10 + lx*p11)."""
    img = np.asarray(img, dtype=np.float32)
    H, W = img.shape
    y0, y1, ly = _source_index_weights(H, out_h, align_corners)
    x0, x1, lx = _source_index_weights(W, out_w, align_corners)
    hy, hx = (np.float32(1) - ly)[:, None], (np.float32(1) - lx)[None, :]
    ly, lx = ly[:, None], lx[None, :]
    top = img[y0][:, x0] * hx + img[y0][:, x1] * lx
    bot = img[y1][:, x0] * hx + img[y1][:, x1] * lx
    return (top * hy + bot * ly).astype(np.float32)


def gaussian_blur(mask, sigma, ksize=39):
    """:307-312 (torchvision GaussianBlur(39x39, sigma)): separable Gaussian with reflect padding."""
    half = (ksize - 1) * 0.5
    x = np.linspace(-half, half, ksize, dtype=np.float32)
    k = np.exp(np.float32(-0.5) * (x / np.float32(sigma)) ** 2).astype(np.float32)
    k = k / k.sum(dtype=np.float32)
    k2 = np.outer(k, k).astype(np.float32)
    p = ksize // 2
    padded = np.pad(np.asarray(mask, np.float32), p, mode="reflect")
    t = torch.nn.functional.conv2d(torch.from_numpy(padded)[None, None], torch.from_numpy(k2)[None, None])
    return t[0, 0].numpy()


def tokens_img_attention_weight(regions, token_ids, ratio=8, original_shape=False, verbose=False):
    """:247-276. Per-token weight map [Hr*Wr, len(token_ids)] (float32): for every prompt position
    where a region's phrase matches, the bilinear(align_corners=True) downsample of the region mask is
    ADDED to the phrase's columns (overlaps and repeats accumulate, in region-then-position order)."""
    token_ids = list(token_ids)
    H, W = regions[0][1].shape
    Hr, Wr = always_round(H / ratio), always_round(W / ratio)
    out = np.zeros((Hr * Wr, len(token_ids)), dtype=np.float32)
    for ids, mask in regions:
        found = False
        L = len(ids)
        for idx in range(len(token_ids)):
            if token_ids[idx: idx + L] == list(ids):
                found = True
                down = bilinear_resize(mask, Hr, Wr, align_corners=True).reshape(-1, 1)
                out[:, idx: idx + L] += down
        if verbose and not found:
            print(f"Warning ratio {ratio} : tokens {ids} not found in text")
    if original_shape:
        out = out.reshape(Hr, Wr, len(token_ids))
    return out


def column_region_lists(regions, token_ids):
    """Host-side index the HIP mask kernel consumes: for each prompt position, the region ordinals
    added to it, in the reference's accumulation order (:257-268)."""
    token_ids = list(token_ids)
    cols = [[] for _ in token_ids]
    for r, (ids, _) in enumerate(regions):
        L = len(ids)
        for idx in range(len(token_ids)):
            if token_ids[idx: idx + L] == list(ids):
                for c in range(idx, min(idx + L, len(token_ids))):
                    cols[c].append(r)
    return cols


def orig_weight_fallback(w_orig, n_tokens):
    """:96-101: resize CROSS_ATTENTION_WEIGHT_ORIG [H, W, T] to [n_tokens, T] when no per-resolution
    key exists: bilinear(align_corners=True) by scale 1/sqrt(H*W/n), then 1-D nearest to n_tokens."""
    w_orig = np.asarray(w_orig, np.float32)
    H, W, T = w_orig.shape
    ratio = math.sqrt(H * W / n_tokens)
    oh, ow = int(math.floor(H * (1 / ratio))), int(math.floor(W * (1 / ratio)))
    small = np.stack([bilinear_resize(w_orig[:, :, t], oh, ow, True) for t in range(T)], 0).reshape(T, -1)
    n_in = small.shape[1]
    scale = np.float32(n_in) / np.float32(n_tokens)
    src = np.minimum(np.floor(np.arange(n_tokens, dtype=np.float32) * scale).astype(np.int64), n_in - 1)
    return small[:, src].T.copy()


def encode_text_color_inputs(text_encoder, tokenizer, color_map_rgb, color_context, input_prompt,
                             unconditional_input_prompt="", verbose=False, use_sigma=True):
    """:315-388. Returns (extra_seeds, regions, cond_dict, uncond_dict) with the reference's key names.
    use_sigma=False: the pipeline classes' own copy of this function (:561-627) parses the blur sigmas and drops them (:574)."""
    text_input = tokenizer([input_prompt], padding="max_length", max_length=tokenizer.model_max_length,
                           truncation=True, return_tensors="pt")
    color_context, extra_seeds, extra_sigmas = extract_seed_and_sigma(color_context)
    regions, width, height = separate_regions(color_map_rgb, color_context, tokenizer, verbose)
    for k, sigma in (extra_sigmas.items() if use_sigma else ()):
        regions[k] = (regions[k][0], gaussian_blur(regions[k][1], sigma))
    ids = text_input["input_ids"][0].tolist()
    cond = {"CONTEXT_TENSOR": text_encoder(text_input.input_ids)[0],
            "CROSS_ATTENTION_WEIGHT_ORIG": torch.from_numpy(tokens_img_attention_weight(regions, ids, 1, True))}
    uncond = {"CROSS_ATTENTION_WEIGHT_ORIG": 0}
    for r in (8, 16, 32, 64):
        key = f"CROSS_ATTENTION_WEIGHT_{always_round(height / r) * always_round(width / r)}"
        cond[key] = torch.from_numpy(tokens_img_attention_weight(regions, ids, r, verbose=verbose))
        uncond[key] = 0
    uncond_input = tokenizer([unconditional_input_prompt], padding="max_length",
                             max_length=text_input.input_ids.shape[-1], return_tensors="pt")
    uncond["CONTEXT_TENSOR"] = text_encoder(uncond_input.input_ids)[0]
    return extra_seeds, regions, cond, uncond


# --------------------------------------------------------------------------------------------
# attention (fp32 torch-CPU reference of the floating-point kernel)


def attention_core(q, k, v, bias, scale):
    """:87, :112-116 on head-split fp32 tensors q [BH, N, d], k/v [BH, M, d]:
    softmax((q k^T + bias) * scale) v -- the bias joins the RAW scores, before the scaling."""
    scores = torch.matmul(q, k.transpose(-1, -2))
    probs = ((scores + bias) * scale).softmax(dim=-1)
    return torch.matmul(probs, v), scores


def split_heads(t, heads):
    b, s, c = t.shape
    return t.reshape(b, s, heads, c // heads).permute(0, 2, 1, 3).reshape(b * heads, s, c // heads)


def merge_heads(t, heads):
    bh, s, d = t.shape
    return t.reshape(bh // heads, heads, s, d).permute(0, 2, 1, 3).reshape(bh // heads, s, d * heads)


def inj_forward(module, hidden_states, context=None, mask=None):
    """:60-125 restated for a diffusers-0.10-style CrossAttention module, fp32 on CPU.

    context: None (self-attention), a tensor (vanilla cross-attention) or the PwW dict
    {CONTEXT_TENSOR, CROSS_ATTENTION_WEIGHT_<N>, CROSS_ATTENTION_WEIGHT_ORIG, SIGMA, WEIGHT_FUNCTION}."""
    is_dict = isinstance(context, dict)
    if context is None:
        ctx = hidden_states
    else:
        ctx = context["CONTEXT_TENSOR"] if is_dict else context
    h = module.heads
    q = split_heads(torch.nn.functional.linear(hidden_states, module.to_q.weight), h)
    k = split_heads(torch.nn.functional.linear(ctx, module.to_k.weight), h)
    v = split_heads(torch.nn.functional.linear(ctx, module.to_v.weight), h)
    scores = torch.matmul(q, k.transpose(-1, -2))
    n_img = scores.shape[-2]
    bias = 0.0
    if context is not None and is_dict:
        f = context["WEIGHT_FUNCTION"]
        key = f"CROSS_ATTENTION_WEIGHT_{n_img}"
        if key in context:
            w = context[key]
        else:
            w = context["CROSS_ATTENTION_WEIGHT_ORIG"]
            w = 0 if isinstance(w, int) else torch.from_numpy(orig_weight_fallback(w.cpu().numpy(), n_img))
        bias = f(w, context["SIGMA"], scores)
    probs = ((scores + bias) * module.scale).softmax(dim=-1)
    out = merge_heads(torch.matmul(probs, v), h)
    out = torch.nn.functional.linear(out, module.to_out[0].weight, module.to_out[0].bias)
    return out


def install_oracle_attention(unet):
    """Class-level plug of the oracle (same mechanism as :193-195) -- for oracle runs only."""
    for m in unet.modules():
        if m.__class__.__name__ == "CrossAttention":
            m.__class__.__call__ = inj_forward


# --------------------------------------------------------------------------------------------
# latents, guidance, sampling loop


def region_binary_masks(regions, extra_seeds, size):
    """:300-304: (mask > 0) as float, bilinear (align_corners=False) to the latent size."""
    return [bilinear_resize((regions[k][1] > 0).astype(np.float32), size[0], size[1], align_corners=False)
            for k in extra_seeds.keys()]


def initial_latents(seed, in_channels, height, width, regions=None, extra_seeds=None):
    """:445-455: CPU-generator randn; optional region-based seeding."""
    size = (1, in_channels, height // 8, width // 8)
    latents = torch.randn(size, generator=torch.manual_seed(seed))
    if extra_seeds:
        multi = [torch.randn(size, generator=torch.manual_seed(s)) for s in extra_seeds.values()]
        masks = [torch.from_numpy(m)[None, None] for m in region_binary_masks(regions, extra_seeds, size[-2:])]
        foreground = (sum(masks) > 0).squeeze()
        summed = sum(l * m for l, m in zip(multi, masks))
        latents[:, :, foreground] = summed[:, :, foreground]
    return latents


def cfg_combine(cond, uncond, guidance_scale):
    """:501-503."""
    return uncond + guidance_scale * (cond - uncond)


def sample_latents(unet, scheduler, cond, uncond, latents, num_inference_steps, guidance_scale, weight_function,
                   extra_channels=None, on_step=None, t_start=None):
    """:431, :457, :471-506 (inpaint: paint_with_words_inpaint.py:230-266 with `extra_channels` =
    cat([mask, masked_image_latents]) appended to the latent input). Two batch-1 UNet calls per step
    (cond dict, then uncond dict with a zero weight function), CFG, scheduler.step. `t_start` (img2img, :434-441): the loop walks
    scheduler.timesteps[t_start:] over latents that already carry their noise (no init_noise_sigma scaling, :459-468)."""
    scheduler.set_timesteps(num_inference_steps)
    if t_start is None:
        latents = latents * scheduler.init_noise_sigma
    first = 0 if t_start is None else t_start
    for i, t in enumerate(scheduler.timesteps[first:], first):
        sigma = scheduler.sigmas[i]      # (:473-474: the step index of t in the FULL schedule)
        x = scheduler.scale_model_input(latents, t)
        if extra_channels is not None:
            x = torch.cat([x, extra_channels], dim=1)
        cond.update({"SIGMA": sigma, "WEIGHT_FUNCTION": weight_function})
        eps_c = unet(x, t, encoder_hidden_states=cond).sample
        uncond.update({"SIGMA": sigma, "WEIGHT_FUNCTION": lambda w, sigma, qk: 0.0})
        eps_u = unet(x, t, encoder_hidden_states=uncond).sample
        eps = cfg_combine(eps_c, eps_u, guidance_scale)
        latents = scheduler.step(eps, t, latents).prev_sample
        if on_step is not None:
            on_step(i, latents)
    return latents


def paint_with_words_latents(color_context, color_map_rgb, input_prompt, unet, text_encoder, tokenizer, scheduler,
                             num_inference_steps=30, guidance_scale=7.5, seed=0,
                             weight_function=lambda w, sigma, qk: 0.1 * w * math.log(sigma + 1) * qk.max(),
                             unconditional_input_prompt=""):
    """:391-506 (txt2img branch) up to the final latent (the VAE decode of :508 is outside the path)."""
    H, W = color_map_rgb.shape[:2]
    extra_seeds, regions, cond, uncond = encode_text_color_inputs(
        text_encoder, tokenizer, color_map_rgb, color_context, input_prompt, unconditional_input_prompt)
    latents = initial_latents(seed, unet.in_channels, H, W, regions, extra_seeds)
    return sample_latents(unet, scheduler, cond, uncond, latents, num_inference_steps, guidance_scale,
                          weight_function)


def paint_with_words_img2img_latents(color_context, color_map_rgb, input_prompt, init_rgb, vae, unet, text_encoder, tokenizer, scheduler,
                                     num_inference_steps=30, guidance_scale=7.5, strength=0.5,
                                     weight_function=lambda w, sigma, qk: 0.1 * w * math.log(sigma + 1) * qk.max(),
                                     unconditional_input_prompt=""):
    """:391-506, img2img branch, up to the final latent: the shortened schedule of :434-441 (strength -> t_start), the init image through
    `preprocess` (:28-35; here for sides that are multiples of 32: no resampling) and the VAE encoder, x 0.18215, noise from the GLOBAL
    torch generator (:464: `torch.randn(shape)` without a generator -- the caller seeds it), scheduler.add_noise at the first kept
    timestep (:467), then the loop of :471-506 from t_start."""
    H, W = init_rgb.shape[:2]
    assert H % 32 == 0 and W % 32 == 0
    _, _, cond, uncond = encode_text_color_inputs(text_encoder, tokenizer, color_map_rgb, color_context, input_prompt, unconditional_input_prompt)
    scheduler.set_timesteps(num_inference_steps)
    offset = scheduler.config.get("steps_offset", 0)
    init_timestep = min(int(num_inference_steps * strength) + offset, num_inference_steps)
    t_start = max(num_inference_steps - init_timestep + offset, 0)
    image = 2.0 * torch.from_numpy((np.asarray(init_rgb, dtype=np.float32) / 255.0)[None].transpose(0, 3, 1, 2)) - 1.0
    init_latents = 0.18215 * vae.encode(image).latent_dist.sample()
    noise = torch.randn(init_latents.shape)
    latents = scheduler.add_noise(init_latents, noise, scheduler.timesteps[t_start:t_start + 1])
    return sample_latents(unet, scheduler, cond, uncond, latents, num_inference_steps, guidance_scale, weight_function, t_start=t_start)


# --------------------------------------------------------------------------------------------
# inpaint pre-processing (paint_with_words/paint_with_words_inpaint.py)


def prepare_mask_and_masked_image(image_rgb, mask_l):
    """paint_with_words_inpaint.py:92-106 (PIL / numpy branch): image uint8 [H, W, 3] -> [-1, 1] NCHW,
    mask uint8 [H, W] ("L") -> binarised at 0.5, masked_image = image * (mask < 0.5)."""
    image = torch.from_numpy(np.asarray(image_rgb)[None].transpose(0, 3, 1, 2)).to(torch.float32) / 127.5 - 1.0
    mask = np.asarray(mask_l).astype(np.float32) / 255.0
    mask = mask[None, None]
    mask = np.where(mask >= 0.5, np.float32(1), np.float32(0))
    mask = torch.from_numpy(mask)
    return mask, image * (mask < 0.5)


def nearest_resize(mask, out_h, out_w):
    """F.interpolate default mode='nearest' (paint_with_words_inpaint.py:115): src = floor(dst * in/out)."""
    H, W = mask.shape[-2:]
    ys = np.minimum(np.floor(np.arange(out_h, dtype=np.float32) * (np.float32(H) / np.float32(out_h))).astype(np.int64), H - 1)
    xs = np.minimum(np.floor(np.arange(out_w, dtype=np.float32) * (np.float32(W) / np.float32(out_w))).astype(np.int64), W - 1)
    return mask[..., ys, :][..., xs]


def paint_with_words_inpaint_latents(color_context, color_map_rgb, mask_l, init_rgb, input_prompt, vae, unet, text_encoder,
                                     tokenizer, scheduler, num_inference_steps=150, guidance_scale=7.5, seed=0,
                                     weight_function=lambda w, sigma, qk: 0.1 * w * math.log(sigma + 1) * qk.max(),
                                     strength=1.0):
    """paint_with_words_inpaint.py:137-266 up to the final latent, for inputs already at the init image's size
    (the reference's NEAREST resize of color map / mask, :172-173, is then the identity) and a side that is a
    multiple of 32 (so `preprocess`, paint_with_words.py:28-35, does not resample)."""
    H, W = init_rgb.shape[:2]
    assert color_map_rgb.shape[:2] == (H, W) and mask_l.shape == (H, W) and H % 32 == 0 and W % 32 == 0
    _, _, cond, uncond = encode_text_color_inputs(text_encoder, tokenizer, color_map_rgb, color_context, input_prompt, "")
    mask, masked_image = prepare_mask_and_masked_image(init_rgb, mask_l)
    scheduler.set_timesteps(num_inference_steps)
    offset = scheduler.config.get("steps_offset", 0)
    init_timestep = min(int(num_inference_steps * strength) + offset, num_inference_steps)
    t_start = max(num_inference_steps - init_timestep + offset, 0)
    timesteps = scheduler.timesteps[t_start:]
    generator = torch.manual_seed(seed)
    image = 2.0 * torch.from_numpy(init_rgb.astype(np.float32) / 255.0)[None].permute(0, 3, 1, 2) - 1.0
    init_latents = 0.18215 * vae.encode(image).latent_dist.sample()
    noise = torch.randn(init_latents.shape, generator=generator)
    latents = scheduler.add_noise(init_latents, noise, timesteps[:1])
    # :201-214: mask to latent resolution (nearest), masked image through the VAE encoder
    mask_lat = torch.from_numpy(nearest_resize(mask.numpy(), H // 8, W // 8))
    masked_latents = 0.18215 * vae.encode(masked_image).latent_dist.sample()
    extra = torch.cat([mask_lat, masked_latents], dim=1)
    assert latents.shape[1] + extra.shape[1] == unet.in_channels
    for t in timesteps:
        i = int((scheduler.timesteps == t).nonzero().item())
        sigma = scheduler.sigmas[i]
        x = torch.cat([scheduler.scale_model_input(latents, t), extra], dim=1)
        cond.update({"SIGMA": sigma, "WEIGHT_FUNCTION": weight_function})
        eps_c = unet(x, t, encoder_hidden_states=cond).sample
        uncond.update({"SIGMA": sigma, "WEIGHT_FUNCTION": lambda w, sigma, qk: 0.0})
        eps_u = unet(x, t, encoder_hidden_states=uncond).sample
        latents = scheduler.step(cfg_combine(eps_c, eps_u, guidance_scale), t, latents).prev_sample
    return latents


# --------------------------------------------------------------------------------------------
# pipeline classes (paint_with_words.py:513-842, paint_with_words_inpaint.py:273-575)


def _denoise(unet, scheduler, cond, uncond, latents, timesteps, guidance_scale, weight_function, extra=None, callback=None,
             callback_steps=1):
    """The classes' denoising loop (:787-816 / paint_with_words_inpaint.py:520-559): sigma looked up by timestep equality,
    two batch-1 UNet calls, CFG, scheduler.step, `callback(i, t, latents)` when i % callback_steps == 0."""
    for i, t in enumerate(timesteps):
        sigma = scheduler.sigmas[int((scheduler.timesteps == t).nonzero().item())]
        x = scheduler.scale_model_input(latents, t)
        if extra is not None:
            x = torch.cat([x, extra], dim=1)
        cond.update({"SIGMA": sigma, "WEIGHT_FUNCTION": weight_function})
        eps_c = unet(x, t, encoder_hidden_states=cond).sample
        uncond.update({"SIGMA": sigma, "WEIGHT_FUNCTION": lambda w, sigma, qk: 0.0})
        eps_u = unet(x, t, encoder_hidden_states=uncond).sample
        latents = scheduler.step(cfg_combine(eps_c, eps_u, guidance_scale), t, latents).prev_sample
        if callback is not None and i % callback_steps == 0:
            callback(i, t, latents)
    return latents


def _strength_timesteps(scheduler, num_inference_steps, strength):
    """:733-741 (and :434-441): the tail of the schedule an img2img / inpaint request runs."""
    offset = scheduler.config.get("steps_offset", 0)
    init_timestep = min(int(num_inference_steps * strength) + offset, num_inference_steps)
    return scheduler.timesteps[max(num_inference_steps - init_timestep + offset, 0):]


def _preprocess_rgb(rgb):
    """paint_with_words.py:28-35 for an image whose sides are multiples of 32 (no resampling): uint8 [H, W, 3] -> [-1, 1] NCHW."""
    assert rgb.shape[0] % 32 == 0 and rgb.shape[1] % 32 == 0
    return 2.0 * torch.from_numpy(rgb.astype(np.float32) / 255.0)[None].permute(0, 3, 1, 2) - 1.0


def pipeline_call_latents(vae, unet, text_encoder, tokenizer, scheduler, prompt, color_map_rgb=None, color_context=None,
                          weight_function=lambda w, sigma, qk: 0.1 * w * math.log(sigma + 1) * qk.max(), height=None, width=None,
                          num_inference_steps=30, guidance_scale=7.5, negative_prompt="", eta=0.5, seed=0, image_rgb=None,
                          callback=None, callback_steps=1):
    """PaintWithWord_StableDiffusionPipeline.__call__ (:629-842) up to the final latent. What differs from the function API:
    height / width default to sample_size * 8 and size the LATENT (:700-701, :756) while the color map sizes the weight maps;
    blur sigmas are dropped (:574); `negative_prompt` is the unconditional prompt; with `image` the schedule is cut by `eta`
    (:735) and the noise comes from the GLOBAL generator (:771)."""
    height = height or unet.config.sample_size * 8
    width = width or unet.config.sample_size * 8
    extra_seeds, regions, cond, uncond = encode_text_color_inputs(text_encoder, tokenizer, color_map_rgb, dict(color_context or {}),
                                                                  prompt, negative_prompt, use_sigma=False)
    scheduler.set_timesteps(num_inference_steps)
    if image_rgb is None:
        timesteps = scheduler.timesteps
        latents = initial_latents(seed, unet.in_channels, height, width, regions, extra_seeds) * scheduler.init_noise_sigma
    else:
        timesteps = _strength_timesteps(scheduler, num_inference_steps, eta)
        init_latents = 0.18215 * vae.encode(_preprocess_rgb(image_rgb)).latent_dist.sample()
        latents = scheduler.add_noise(init_latents, torch.randn(init_latents.shape), timesteps[:1])
    return _denoise(unet, scheduler, cond, uncond, latents, timesteps, guidance_scale, weight_function, None, callback, callback_steps)


def inpaint_pipeline_call_latents(vae, unet, text_encoder, tokenizer, scheduler, prompt, image_rgb, mask_l, color_map_rgb=None,
                                  color_context=None, weight_function=lambda w, sigma, qk: 0.1 * w * math.log(sigma + 1) * qk.max(),
                                  height=None, width=None, num_inference_steps=30, guidance_scale=7.5, negative_prompt="", eta=1.0,
                                  seed=0, callback=None, callback_steps=1):
    """PaintWithWord_StableDiffusionInpaintPipeline.__call__ (paint_with_words_inpaint.py:340-575) up to the final latent: color map
    and mask are taken as given (no resize to the init image), the latent mask is sized by height / width (:427-432, :498-508),
    `eta` is the strength (:441), the noise comes from `seed` (:467-473)."""
    height = height or unet.config.sample_size * 8
    width = width or unet.config.sample_size * 8
    _, _, cond, uncond = encode_text_color_inputs(text_encoder, tokenizer, color_map_rgb, dict(color_context or {}), prompt,
                                                  negative_prompt, use_sigma=False)
    mask, masked_image = prepare_mask_and_masked_image(image_rgb, mask_l)
    scheduler.set_timesteps(num_inference_steps)
    timesteps = _strength_timesteps(scheduler, num_inference_steps, eta)
    generator = torch.manual_seed(seed)
    init_latents = 0.18215 * vae.encode(_preprocess_rgb(image_rgb)).latent_dist.sample()
    latents = scheduler.add_noise(init_latents, torch.randn(init_latents.shape, generator=generator), timesteps[:1])
    mask_lat = torch.from_numpy(nearest_resize(mask.numpy(), height // 8, width // 8))
    extra = torch.cat([mask_lat, 0.18215 * vae.encode(masked_image).latent_dist.sample()], dim=1)
    assert latents.shape[1] + extra.shape[1] == unet.in_channels
    return _denoise(unet, scheduler, cond, uncond, latents, timesteps, guidance_scale, weight_function, extra, callback, callback_steps)
