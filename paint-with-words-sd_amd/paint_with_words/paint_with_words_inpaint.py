"""Drop-in for the reference module paint_with_words/paint_with_words_inpaint.py (function API):
the inpainting variant feeds a 9-channel UNet input cat([latents, mask, masked_image_latents])
(reference :237, :250); the attention path is the same fused HIP op."""
import math
from typing import Callable, Dict, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F
from PIL import Image

from .paint_with_words import (LMSDiscreteScheduler, pww_load_tools, preprocess, _pil_from_latents,
                               _encode_text_color_inputs, _sampler_for, _unet_dtype, DEFAULT_MODE)


def prepare_mask_and_masked_image(image, mask):
    """reference :20-106: (image, mask) -> (mask [B,1,H,W] binarised at 0.5, masked_image [B,3,H,W] in [-1,1])."""
    if isinstance(image, torch.Tensor):
        if not isinstance(mask, torch.Tensor):
            raise TypeError(f"`image` is a torch.Tensor but `mask` (type: {type(mask)} is not")
        if image.ndim == 3:
            assert image.shape[0] == 3, "Image outside a batch should be of shape (3, H, W)"
            image = image.unsqueeze(0)
        if mask.ndim == 2:
            mask = mask.unsqueeze(0).unsqueeze(0)
        if mask.ndim == 3:
            mask = mask.unsqueeze(0) if mask.shape[0] == 1 else mask.unsqueeze(1)
        assert image.ndim == 4 and mask.ndim == 4, "Image and Mask must have 4 dimensions"
        assert image.shape[-2:] == mask.shape[-2:], "Image and Mask must have the same spatial dimensions"
        assert image.shape[0] == mask.shape[0], "Image and Mask must have the same batch size"
        if image.min() < -1 or image.max() > 1:
            raise ValueError("Image should be in [-1, 1] range")
        if mask.min() < 0 or mask.max() > 1:
            raise ValueError("Mask should be in [0, 1] range")
        mask = (mask >= 0.5).to(mask.dtype)
        image = image.to(dtype=torch.float32)
    elif isinstance(mask, torch.Tensor):
        raise TypeError(f"`mask` is a torch.Tensor but `image` (type: {type(image)} is not")
    else:
        if isinstance(image, Image.Image):
            image = np.array(image.convert("RGB"))
        image = torch.from_numpy(image[None].transpose(0, 3, 1, 2)).to(dtype=torch.float32) / 127.5 - 1.0
        if isinstance(mask, Image.Image):
            mask = np.array(mask.convert("L")).astype(np.float32) / 255.0
        mask = torch.from_numpy((mask[None, None] >= 0.5).astype(np.float32))
    masked_image = image * (mask < 0.5)
    return mask, masked_image


def prepare_mask_latents(vae, mask, masked_image, batch_size, height, width, dtype, device, generator,
                         do_classifier_free_guidance):
    """reference :109-134."""
    mask = F.interpolate(mask, size=(height // 8, width // 8)).to(device=device, dtype=dtype)
    masked_image = masked_image.to(device=device, dtype=vae.dtype)
    masked_image_latents = 0.18215 * vae.encode(masked_image).latent_dist.sample().to(dtype)
    mask = mask.repeat(batch_size, 1, 1, 1)
    masked_image_latents = masked_image_latents.repeat(batch_size, 1, 1, 1)
    if do_classifier_free_guidance:
        mask, masked_image_latents = torch.cat([mask] * 2), torch.cat([masked_image_latents] * 2)
    return mask, masked_image_latents.to(device=device, dtype=dtype)


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
    vae, unet, text_encoder, tokenizer, scheduler = (
        pww_load_tools(device, scheduler_type, local_model_path=local_model_path, hf_model_path=hf_model_path,
                       model_token=model_token)
        if preloaded_utils is None else preloaded_utils)
    sampler = _sampler_for(unet, scheduler, DEFAULT_MODE)

    width, height = init_image.size
    color_map_image = color_map_image.resize((width, height), Image.NEAREST)
    mask_image = mask_image.resize((width, height), Image.NEAREST)
    _, _, encoder_hidden_states, uncond_encoder_hidden_states = _encode_text_color_inputs(
        text_encoder, tokenizer, device, color_map_image, color_context, input_prompt, unconditional_input_prompt,
        dtype=_unet_dtype(unet))
    mask, masked_image = prepare_mask_and_masked_image(init_image, mask_image)

    scheduler.set_timesteps(num_inference_steps)
    offset = scheduler.config.get("steps_offset", 0)
    init_timestep = min(int(num_inference_steps * strength) + offset, num_inference_steps)
    t_start = max(num_inference_steps - init_timestep + offset, 0)
    timesteps = scheduler.timesteps[t_start:]
    latent_timestep = timesteps[:1]

    generator = torch.manual_seed(seed)
    image = preprocess(init_image).to(device=device)
    init_latents = 0.18215 * vae.encode(image.to(vae.dtype)).latent_dist.sample().float()
    noise = torch.randn(init_latents.shape, generator=generator).to(device)
    latents = scheduler.add_noise(init_latents, noise, latent_timestep)

    mask, masked_image_latents = prepare_mask_latents(vae, mask, masked_image, 1, height, width, latents.dtype,
                                                      device, generator=generator, do_classifier_free_guidance=False)
    mask = F.interpolate(mask, size=latents.shape[-2:], mode="nearest")
    masked_image_latents = F.interpolate(masked_image_latents, size=latents.shape[-2:], mode="nearest")

    n_lat, n_mask, n_img = latents.shape[1], mask.shape[1], masked_image_latents.shape[1]
    if n_lat + n_mask + n_img != unet.in_channels:
        raise ValueError(
            f"Incorrect configuration settings! The config of `pipeline.unet`: {unet.config} expects"
            f" {unet.in_channels} but received `num_channels_latents`: {n_lat} +"
            f" `num_channels_mask`: {n_mask} + `num_channels_masked_image`: {n_img}"
            f" = {n_lat + n_img + n_mask}. Please verify the config of"
            " `pipeline.unet` or your `mask_image` or `image` input.")

    latents = sampler.sample(encoder_hidden_states, uncond_encoder_hidden_states, latents, timesteps, guidance_scale,
                             weight_function, extra_channels=torch.cat([mask, masked_image_latents], dim=1))
    if return_latents:
        return latents
    return _pil_from_latents(vae, latents)[0]
