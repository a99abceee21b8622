"""Drop-in for the reference module paint_with_words/paint_with_words_inpaint.py (function API):
the inpainting variant feeds a 9-channel UNet input cat([latents, mask, masked_image_latents])
(reference :237, :250); the attention path is the same fused HIP op, and the mask / masked-image
preparation (reference :20-134) is one HIP kernel (pww_inpaint_prep) over the uint8 pixels."""
import math
from typing import Callable, Dict, Optional, Sequence, Tuple, Union

import numpy as np
import torch
import torch.nn.functional as F
from PIL import Image

import pww_hip
from pww_hip import ops
_pw_module_name = __name__.rsplit(".", 1)[0] + ".paint_with_words"      # the function-API module: its DEFAULT_MODE is read at call time
from .paint_with_words import (LMSDiscreteScheduler, pww_load_tools, preprocess, _pil_from_latents,
                               _encode_text_color_inputs, _sampler_for, _unet_dtype, _broadcast)


def _mode():
    import sys
    return sys.modules[_pw_module_name].DEFAULT_MODE


def _as_uint8_pixels(image, channels):
    """PIL image or numpy array -> uint8 numpy array ([H, W, 3] for channels == 3, [H, W] for 1), or None when the
    values are not 8-bit pixels (float arrays go through the tensor path)."""
    if isinstance(image, Image.Image):
        return np.asarray(image.convert("RGB" if channels == 3 else "L"))
    arr = np.asarray(image)
    return arr if arr.dtype == np.uint8 else None


def _prep_on_device(rgb_u8, mask_u8, device, lat_hw=None):
    """(mask [1,1,H,W], masked_image [1,3,H,W], latent-size mask or None) from uint8 pixels, on `device`."""
    H, W = mask_u8.shape
    h, w = lat_hw if lat_hw is not None else (max(H // 8, 1), max(W // 8, 1))
    m, masked, ml = ops.inpaint_prep(torch.from_numpy(np.ascontiguousarray(rgb_u8)).to(device),
                                     torch.from_numpy(np.ascontiguousarray(mask_u8)).to(device), h, w)
    return m, masked, ml


def prepare_mask_and_masked_image(image, mask, device=None):
    """reference :20-106: (image, mask) -> (mask [B,1,H,W] in {0,1}, masked_image [B,3,H,W] in [-1,1] with the masked
    pixels zeroed). 8-bit inputs (PIL images / uint8 arrays -- what the reference's callers pass, runner_inpaint.py:
    44-46) are converted by pww_inpaint_prep when `device` is a HIP device; tensors (the reference's second input form:
    image in [-1,1], mask in [0,1], optional batch / channel axes) are checked and thresholded where they live."""
    image_is_tensor, mask_is_tensor = torch.is_tensor(image), torch.is_tensor(mask)
    if image_is_tensor != mask_is_tensor:
        which, other, bad = ("image", "mask", mask) if image_is_tensor else ("mask", "image", image)
        raise TypeError(f"`{which}` is a torch.Tensor but `{other}` (type: {type(bad)} is not")
    if image_is_tensor:
        image = image[None] if image.ndim == 3 else image
        if mask.ndim == 2:
            mask = mask[None, None]
        elif mask.ndim == 3:       # [1, H, W] is one mask with its channel axis, [B, H, W] a batch without one
            mask = mask[None] if mask.shape[0] == 1 else mask[:, None]
        if image.ndim != 4 or mask.ndim != 4 or image.shape[1] != 3:
            raise AssertionError("Image must be (3, H, W) or (B, 3, H, W) and Mask (H, W), (1, H, W), (B, H, W) or (B, 1, H, W)")
        if image.shape[-2:] != mask.shape[-2:] or image.shape[0] != mask.shape[0]:
            raise AssertionError("Image and Mask must have the same spatial dimensions and batch size")
        if image.min() < -1 or image.max() > 1:
            raise ValueError("Image should be in [-1, 1] range")
        if mask.min() < 0 or mask.max() > 1:
            raise ValueError("Mask should be in [0, 1] range")
        mask = (mask >= 0.5).to(mask.dtype)
        image = image.to(torch.float32)
        return mask, image * (mask < 0.5)
    rgb, m8 = _as_uint8_pixels(image, 3), _as_uint8_pixels(mask, 1)
    if rgb is not None and m8 is not None and device is not None and torch.device(device).type == "cuda":
        m, masked, _ = _prep_on_device(rgb, m8, device)
        return m, masked
    # float arrays (already scaled masks) or no HIP device named: the same arithmetic as host tensors
    rgb = np.asarray(image.convert("RGB")) if isinstance(image, Image.Image) else np.asarray(image)
    pix = torch.from_numpy(np.ascontiguousarray(rgb)).permute(2, 0, 1)[None].to(torch.float32) / 127.5 - 1.0
    m = np.asarray(mask.convert("L")).astype(np.float32) / 255.0 if isinstance(mask, Image.Image) else np.asarray(mask, dtype=np.float32)
    m = torch.from_numpy((m >= 0.5).astype(np.float32))[None, None]
    return m, pix * (m < 0.5)


def prepare_mask_latents(vae, mask, masked_image, batch_size, height, width, dtype, device, generator,
                         do_classifier_free_guidance):
    """reference :109-134: mask to latent resolution (nearest), masked image through the VAE encoder, both repeated
    per batch row (and doubled for a CFG-concatenated batch)."""
    mask = F.interpolate(mask, size=(height // 8, width // 8)).to(device=device, dtype=dtype)
    latent = 0.18215 * vae.encode(masked_image.to(device=device, dtype=vae.dtype)).latent_dist.sample()
    reps = batch_size * (2 if do_classifier_free_guidance else 1)
    return mask.repeat(reps, 1, 1, 1), latent.repeat(reps, 1, 1, 1).to(device=device, dtype=dtype)


def _inpaint_inputs(vae, init_image, mask_image, seed, scheduler, timesteps, device, mask_hw=None):
    """One request's (noised latents [1,4,h,w], extra channels [1,5,h,w]) -- reference :201-228. `mask_hw`: the pipeline
    class sizes the latent mask by its `height` / `width` arguments (:498-508) instead of by the init image."""
    width, height = init_image.size
    if mask_hw is not None:
        height, width = mask_hw
    generator = torch.manual_seed(seed)
    image = preprocess(init_image).to(device=device)
    init_latents = 0.18215 * vae.encode(image.to(vae.dtype)).latent_dist.sample().float()
    noise = torch.randn(init_latents.shape, generator=generator).to(device)
    latents = scheduler.add_noise(init_latents, noise, timesteps[:1])

    rgb, m8 = _as_uint8_pixels(init_image, 3), _as_uint8_pixels(mask_image, 1)
    _, masked_image, mask_lat = _prep_on_device(rgb, m8, device, lat_hw=(height // 8, width // 8))
    masked_latents = 0.18215 * vae.encode(masked_image.to(vae.dtype)).latent_dist.sample().to(latents.dtype)
    if mask_hw is not None and (mask_lat.shape[-2:] != latents.shape[-2:] or masked_latents.shape[-2:] != latents.shape[-2:]):
        # the reference's pipeline fails in torch.cat here (:532): `height` / `width` must be the size of `image`
        raise ValueError(f"`height` x `width` = {height} x {width} gives a {tuple(mask_lat.shape[-2:])} latent mask but `image` gives "
                         f"{tuple(latents.shape[-2:])} latents: pass the size of `image` (a multiple of 32)")
    if mask_lat.shape[-2:] != latents.shape[-2:]:      # sizes that `preprocess` rounds down to a multiple of 32 (:213-214)
        mask_lat = F.interpolate(mask_lat, size=latents.shape[-2:], mode="nearest")
        masked_latents = F.interpolate(masked_latents, size=latents.shape[-2:], mode="nearest")
    return latents, torch.cat([mask_lat.to(latents.dtype), masked_latents], dim=1)


def _generate_inpaint(tools, device, color_contexts, color_map_images, mask_images, init_images, prompts, seeds,
                      num_inference_steps, guidance_scale, weight_function, unconditional_input_prompt, strength, shared,
                      on_step=None, mask_hw=None, resize_inputs=True, use_region_sigma=True):
    """Shared body of paint_with_words_inpaint / _batch / the inpaint pipeline class. The function API resizes color map and
    mask to the init image (:172-173, `resize_inputs`); the pipeline class does not and sizes the latent mask by `mask_hw`."""
    vae, unet, text_encoder, tokenizer, scheduler = tools
    n = len(seeds)
    sampler = _sampler_for(unet, scheduler, _mode())
    conds, unconds = [], []
    for i in range(1 if shared else n):
        width, height = init_images[i].size
        color_map = color_map_images[i]
        if resize_inputs and color_map is not None:
            color_map = color_map.resize((width, height), Image.NEAREST)            # :172
        _, _, cond, uncond = _encode_text_color_inputs(text_encoder, tokenizer, device, color_map, color_contexts[i], prompts[i],
                                                       unconditional_input_prompt, dtype=_unet_dtype(unet), use_sigma=use_region_sigma)
        conds.append(cond), unconds.append(uncond)
    if shared:
        conds, unconds = conds[0], unconds[0]

    scheduler.set_timesteps(num_inference_steps)
    offset = scheduler.config.get("steps_offset", 0)
    init_timestep = min(int(num_inference_steps * strength) + offset, num_inference_steps)
    t_start = max(num_inference_steps - init_timestep + offset, 0)
    timesteps = scheduler.timesteps[t_start:]

    lats, extras = [], []
    for i in range(n):
        width, height = init_images[i].size
        mask_image = mask_images[i].resize((width, height), Image.NEAREST) if resize_inputs else mask_images[i]      # :173
        if mask_image.size != init_images[i].size:
            raise AssertionError("Image and Mask must have the same spatial dimensions")                            # :72-73
        lat, extra = _inpaint_inputs(vae, init_images[i], mask_image, seeds[i], scheduler, timesteps, device, mask_hw=mask_hw)
        lats.append(lat), extras.append(extra)
    latents, extra = torch.cat(lats, dim=0), torch.cat(extras, dim=0)

    n_lat, n_extra = latents.shape[1], extra.shape[1]
    if n_lat + n_extra != unet.in_channels:
        config = getattr(unet, "config", None)
        raise ValueError(
            f"Incorrect configuration settings! The config of `pipeline.unet`: {config} expects {unet.in_channels} input "
            f"channels but received {n_lat} latent + 1 mask + {n_extra - 1} masked-image latent channels = {n_lat + n_extra}. "
            "Please verify the config of `pipeline.unet` or your `mask_image` or `image` input.")
    with pww_hip.miopen_find():
        return sampler.sample(conds, unconds, latents, timesteps, guidance_scale, weight_function, extra_channels=extra, on_step=on_step)


@torch.no_grad()
def paint_with_words_inpaint(
    color_context: Dict[Tuple[int, int, int], str] = {},
    color_map_image: Optional[Image.Image] = None,
    mask_image: Optional[Image.Image] = None,
    init_image: Image.Image = None,
    input_prompt: str = "",
    num_inference_steps: int = 150,
    guidance_scale: float = 7.5,
    seed: int = 0,
    scheduler_type=LMSDiscreteScheduler,
    device: str = "cuda:0",
    weight_function: Callable = lambda w, sigma, qk: 0.1 * w * math.log(sigma + 1) * qk.max(),
    local_model_path: Optional[str] = None,
    hf_model_path: Optional[str] = "runwayml/stable-diffusion-inpainting",
    preloaded_utils: Optional[Tuple] = None,
    unconditional_input_prompt: str = "",
    model_token: Optional[str] = None,
    strength: float = 1.0,
    return_latents: bool = False,
):
    """reference :137-270."""
    tools = (pww_load_tools(device, scheduler_type, local_model_path=local_model_path, hf_model_path=hf_model_path,
                            model_token=model_token) if preloaded_utils is None else preloaded_utils)
    latents = _generate_inpaint(tools, device, [color_context], [color_map_image], [mask_image], [init_image], [input_prompt],
                                [seed], num_inference_steps, guidance_scale, weight_function, unconditional_input_prompt,
                                strength, shared=True)
    if return_latents:
        return _sampler_for(tools[1], tools[4], _mode()).checked(latents)
    image = _pil_from_latents(tools[0], latents)[0]
    _sampler_for(tools[1], tools[4], _mode()).check_errors()
    return image


@torch.no_grad()
def paint_with_words_inpaint_batch(
    color_contexts: Union[Dict, Sequence[Dict]],
    color_map_images: Union[Image.Image, Sequence[Image.Image]],
    mask_images: Union[Image.Image, Sequence[Image.Image]],
    init_images: Union[Image.Image, Sequence[Image.Image]],
    input_prompts: Union[str, Sequence[str]],
    seeds: Sequence[int],
    num_inference_steps: int = 150,
    guidance_scale: float = 7.5,
    scheduler_type=LMSDiscreteScheduler,
    device: str = "cuda:0",
    weight_function: Callable = lambda w, sigma, qk: 0.1 * w * math.log(sigma + 1) * qk.max(),
    local_model_path: Optional[str] = None,
    hf_model_path: Optional[str] = "runwayml/stable-diffusion-inpainting",
    preloaded_utils: Optional[Tuple] = None,
    unconditional_input_prompt: str = "",
    model_token: Optional[str] = None,
    strength: float = 1.0,
    return_latents: bool = False,
):
    """len(seeds) inpainting requests through one denoise loop (see paint_with_words_batch): each of the image-like
    arguments and the prompt is one shared value or a sequence with one entry per seed; all init images of a call must
    have the same size. Image i equals the single-request call on request i."""
    seeds = list(seeds)
    n = len(seeds)
    if n == 0:
        return []
    ctxs, s1 = _broadcast(color_contexts, n, "color_context")
    maps, s2 = _broadcast(color_map_images, n, "color_map_images")
    masks, _ = _broadcast(mask_images, n, "mask_images")
    inits, _ = _broadcast(init_images, n, "init_images")
    prompts, s3 = _broadcast(input_prompts, n, "input_prompts")
    if len({im.size for im in inits}) != 1:
        raise ValueError("paint_with_words_inpaint_batch: all init images of one call must have the same size")
    if not (s1 and s2 and s3):
        ctxs = [dict(c) for c in ctxs]
    tools = (pww_load_tools(device, scheduler_type, local_model_path=local_model_path, hf_model_path=hf_model_path,
                            model_token=model_token) if preloaded_utils is None else preloaded_utils)
    latents = _generate_inpaint(tools, device, ctxs, maps, masks, inits, prompts, seeds, num_inference_steps, guidance_scale,
                                weight_function, unconditional_input_prompt, strength, shared=s1 and s2 and s3)
    if return_latents:
        return _sampler_for(tools[1], tools[4], _mode()).checked(latents)
    images = _pil_from_latents(tools[0], latents)
    _sampler_for(tools[1], tools[4], _mode()).check_errors()
    return images


def __getattr__(name):
    """reference paint_with_words_inpaint.py:273 defines the inpaint pipeline class in this module"""
    if name == "PaintWithWord_StableDiffusionInpaintPipeline":
        from .pipelines import PaintWithWord_StableDiffusionInpaintPipeline
        return PaintWithWord_StableDiffusionInpaintPipeline
    raise AttributeError(name)
