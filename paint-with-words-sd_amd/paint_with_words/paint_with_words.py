"""Drop-in for the reference module paint_with_words/paint_with_words.py (function API).

Same public names, signatures, defaults and context protocol as the reference (file:line cited per
function) so its runner.py works unchanged (`device="cuda:0"` is the HIP device under PyTorch-ROCm);
the attention arithmetic and the mask preparation run in hand-written gfx950 kernels
(libpww_hip.so, through pww_hip). This file is host glue: it owns no arithmetic of the hot path.
"""
import math
import os
from typing import Callable, Dict, Optional, Tuple

import numpy as np
import torch
from PIL import Image

import pww_hip
from pww_hip.attention import inj_forward  # noqa: F401  (same import path as the reference's symbol)
from pww_hip.conditioning import (always_round, _extract_seed_and_sigma_from_context, _encode_text_color_inputs,
                                  _get_binary_mask, gaussian_blur_mask)
from pww_hip.sampler import PwWSampler, initial_latents

try:  # the reference's dependency; absent in the offline build image
    from diffusers import AutoencoderKL, LMSDiscreteScheduler, UNet2DConditionModel
    _HAVE_DIFFUSERS = True
except Exception:  # pragma: no cover - depends on the environment
    from sd_standin import LMSDiscreteScheduler
    _HAVE_DIFFUSERS = False

# Execution mode of the denoise loop (see pww_hip/sampler.py): "eager" reproduces the reference's two
# batch-1 UNet calls per step; "folded"/"graph" batch cond+uncond (and replay hipGraphs).
DEFAULT_MODE = os.environ.get("PWW_MODE", "graph")


def preprocess(image):
    """reference :28-35"""
    w, h = image.size
    w, h = map(lambda x: x - x % 32, (w, h))
    image = image.resize((w, h), resample=Image.LANCZOS)
    image = np.array(image).astype(np.float32) / 255.0
    image = image[None].transpose(0, 3, 1, 2)
    return 2.0 * torch.from_numpy(image) - 1.0


def _pil_from_latents(vae, latents):
    """reference :48-57"""
    _latents = 1 / 0.18215 * latents.clone()
    image = vae.decode(_latents.to(vae.dtype)).sample
    image = (image / 2 + 0.5).clamp(0, 1)
    image = image.detach().float().cpu().permute(0, 2, 3, 1).numpy()
    images = (image * 255).round().astype("uint8")
    return [Image.fromarray(im) for im in images]


def pww_load_tools(device: str = "cuda:0", scheduler_type=LMSDiscreteScheduler, local_model_path: Optional[str] = None,
                   hf_model_path: Optional[str] = None, model_token: Optional[str] = None):
    """reference :128-204: load vae / unet / text encoder / tokenizer / scheduler and install the
    attention plug (:193-195). Needs diffusers + transformers and a model on disk or the hub; in an
    environment without them build the modules yourself and pass `preloaded_utils`."""
    assert local_model_path or hf_model_path, "either local_model_path or hf_model_path must be provided"
    if not _HAVE_DIFFUSERS:
        raise ImportError("pww_load_tools needs `diffusers` (the reference pins diffusers==0.10.0); it is not "
                          "installed here. Pass preloaded_utils=(vae, unet, text_encoder, tokenizer, scheduler).")
    from transformers import CLIPTextModel, CLIPTokenizer
    dtype = torch.float16
    model_path = local_model_path if local_model_path is not None else hf_model_path
    local_only = local_model_path is not None
    print(model_path)
    vae = AutoencoderKL.from_pretrained(model_path, subfolder="vae", use_auth_token=model_token, torch_dtype=dtype,
                                        local_files_only=local_only)
    tokenizer = CLIPTokenizer.from_pretrained(model_path, subfolder="tokenizer")
    text_encoder = CLIPTextModel.from_pretrained(model_path, subfolder="text_encoder")
    unet = UNet2DConditionModel.from_pretrained(model_path, subfolder="unet", use_auth_token=model_token,
                                                torch_dtype=dtype, local_files_only=local_only)
    vae.to(device), unet.to(device), text_encoder.to(device)
    if pww_hip.install(unet) == 0 and hasattr(unet, "set_attn_processor"):   # diffusers >= 0.12
        unet.set_attn_processor(pww_hip.PwWAttnProcessor())
    scheduler = scheduler_type(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
                               num_train_timesteps=1000)
    return vae, unet, text_encoder, tokenizer, scheduler


def _unet_dtype(unet):
    return unet.dtype if hasattr(unet, "dtype") else next(unet.parameters()).dtype


def _sampler_for(unet, scheduler, mode):
    """One PwWSampler (and its captured graphs) per (unet, scheduler, mode), kept on the unet."""
    cache = unet.__dict__.setdefault("_pww_samplers", {})
    key = (id(scheduler), mode)
    if key not in cache:
        pww_hip.enable_miopen_find()
        cache[key] = PwWSampler(unet, scheduler, mode)
    return cache[key]


@torch.no_grad()
def paint_with_words(
    color_context: Dict[Tuple[int, int, int], str] = {},
    color_map_image: Optional[Image.Image] = None,
    input_prompt: str = "",
    num_inference_steps: int = 30,
    guidance_scale: float = 7.5,
    seed: int = 0,
    scheduler_type=LMSDiscreteScheduler,
    device: str = "cuda:0",
    weight_function: Callable = lambda w, sigma, qk: 0.1 * w * math.log(sigma + 1) * qk.max(),
    local_model_path: Optional[str] = None,
    hf_model_path: Optional[str] = "CompVis/stable-diffusion-v1-4",
    preloaded_utils: Optional[Tuple] = None,
    unconditional_input_prompt: str = "",
    model_token: Optional[str] = None,
    init_image: Optional[Image.Image] = None,
    strength: float = 0.5,
    return_latents: bool = False,
):
    """reference :391-510. `return_latents=True` (extension) returns the final latent tensor instead
    of decoding it -- the quantity parity is checked on."""
    width, height = color_map_image.size
    vae, unet, text_encoder, tokenizer, scheduler = (
        pww_load_tools(device, scheduler_type, local_model_path=local_model_path, hf_model_path=hf_model_path,
                       model_token=model_token)
        if preloaded_utils is None else preloaded_utils)
    sampler = _sampler_for(unet, scheduler, DEFAULT_MODE)   # also installs the attention plug

    extra_seeds, region_info, encoder_hidden_states, uncond_encoder_hidden_states = _encode_text_color_inputs(
        text_encoder, tokenizer, device, color_map_image, color_context, input_prompt, unconditional_input_prompt,
        dtype=_unet_dtype(unet))

    scheduler.set_timesteps(num_inference_steps)
    if init_image is None:
        timesteps = scheduler.timesteps
    else:   # img2img, :434-441
        offset = scheduler.config.get("steps_offset", 0)
        init_timestep = min(int(num_inference_steps * strength) + offset, num_inference_steps)
        t_start = max(num_inference_steps - init_timestep + offset, 0)
        timesteps = scheduler.timesteps[t_start:]
        latent_timestep = timesteps[:1]

    if init_image is None:   # txt2img, :444-457
        latents = initial_latents(seed, unet.in_channels, height, width,
                                  region_masks=lambda dtype, size: _get_binary_mask(region_info, extra_seeds, dtype, size),
                                  extra_seeds=extra_seeds)
        latents = latents.to(device) * scheduler.init_noise_sigma
    else:                    # :459-468
        image = preprocess(init_image).to(device=device)
        init_latents = 0.18215 * vae.encode(image.to(vae.dtype)).latent_dist.sample().float()
        noise = torch.randn(init_latents.shape).to(device)
        latents = scheduler.add_noise(init_latents, noise, latent_timestep)

    latents = sampler.sample(encoder_hidden_states, uncond_encoder_hidden_states, latents, timesteps, guidance_scale,
                             weight_function)
    if return_latents:
        return latents
    return _pil_from_latents(vae, latents)[0]
