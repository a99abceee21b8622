"""Drop-in for the reference module paint_with_words/paint_with_words.py (function API).

Same public names, signatures, defaults and context protocol as the reference (file:line cited per
function) so its runner.py works unchanged (`device="cuda:0"` is the HIP device under PyTorch-ROCm);
the attention arithmetic and the mask preparation run in hand-written gfx950 kernels
(libpww_hip.so, through pww_hip). This file is host glue: it owns no arithmetic of the hot path.

Beyond the reference's one-image-per-call API there is `paint_with_words_batch`: many
requests -- each with its own color map, color_context, prompt and seed -- through ONE denoise loop, CFG-folded and
hipGraph-replayed; image i of the batch equals the single-image call on request i (the reference generates several
samples by a sequential loop over seeds that reloads the model each time, gradio_pww.py:24-45).
"""
import math
import os
from typing import Callable, Dict, List, Optional, Sequence, Tuple, Union

import numpy as np
import torch
from PIL import Image

import pww_hip
from pww_hip.attention import inj_forward  # noqa: F401  (same import path as the reference's symbol)
from pww_hip.conditioning import (always_round, _extract_seed_and_sigma_from_context, _encode_text_color_inputs,  # noqa: F401
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
    """reference :28-35: PIL image -> [-1, 1] NCHW tensor at the next lower multiple of 32."""
    width, height = (side - side % 32 for side in image.size)
    pixels = np.asarray(image.resize((width, height), resample=Image.LANCZOS), dtype=np.float32) / 255.0
    return 2.0 * torch.from_numpy(pixels[None].transpose(0, 3, 1, 2)) - 1.0


def _pil_from_latents(vae, latents):
    """reference :48-57: decode the latents, one PIL image per batch row."""
    decoded = vae.decode((latents.clone() / 0.18215).to(vae.dtype)).sample
    pixels = (decoded / 2 + 0.5).clamp(0, 1).detach().float().cpu().permute(0, 2, 3, 1).numpy()
    return [Image.fromarray(im) for im in (pixels * 255).round().astype("uint8")]


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
    # the reference's two loading branches (:145-188): half weights from the checkpoint's `fp16` revision everywhere but on Apple's `mps`
    # device, where it loads fp32 from the default revision (kept for signature / behaviour parity: this package's kernels need a HIP device)
    half = device != "mps"
    model_path = local_model_path if local_model_path is not None else hf_model_path
    common = dict(use_auth_token=model_token, torch_dtype=torch.float16 if half else torch.float32, local_files_only=local_model_path is not None)
    if half:
        common["revision"] = "fp16"
    print(model_path)
    vae = AutoencoderKL.from_pretrained(model_path, subfolder="vae", **common)
    tokenizer = CLIPTokenizer.from_pretrained(model_path, subfolder="tokenizer")
    text_encoder = CLIPTextModel.from_pretrained(model_path, subfolder="text_encoder")
    unet = UNet2DConditionModel.from_pretrained(model_path, subfolder="unet", **common)
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
        cache[key] = PwWSampler(unet, scheduler, mode)
    return cache[key]


def _broadcast(value, n, name):
    """A per-request argument of paint_with_words_batch: one value for every request, or a sequence of n."""
    if isinstance(value, (list, tuple)) and not (name == "color_context" and isinstance(value, dict)):
        if len(value) != n:
            raise ValueError("%s has %d entries for %d requests" % (name, len(value), n))
        return list(value), False
    return [value] * n, True


def _generate(tools, device, color_contexts, color_map_images, prompts, seeds, num_inference_steps, guidance_scale,
              weight_function, unconditional_input_prompt, init_images=None, strength=0.5, latent_hw=None,
              on_step=None, use_region_sigma=True, shared=False):
    """Shared body of paint_with_words / paint_with_words_batch / the pipeline class (reference :414-506): conditioning
    per request (once if every request shares map, context and prompt), CPU-generated latents per seed exactly as :446,
    one denoise loop over all images. Returns the final latents [n, 4, h, w]."""
    vae, unet, text_encoder, tokenizer, scheduler = tools
    n = len(seeds)
    sampler = _sampler_for(unet, scheduler, DEFAULT_MODE)   # also installs the attention plug
    conds, unconds, seeds_info = [], [], []
    for i in range(1 if shared else n):
        extra_seeds, region_info, cond, uncond = _encode_text_color_inputs(
            text_encoder, tokenizer, device, color_map_images[i], color_contexts[i], prompts[i], unconditional_input_prompt,
            dtype=_unet_dtype(unet), use_sigma=use_region_sigma)
        conds.append(cond), unconds.append(uncond), seeds_info.append((extra_seeds, region_info))
    if shared:
        conds, unconds, seeds_info = conds[0], unconds[0], seeds_info * n

    scheduler.set_timesteps(num_inference_steps)
    if init_images is None:          # txt2img, :444-457
        timesteps = scheduler.timesteps
        lats = []
        for i in range(n):
            width, height = color_map_images[i].size if latent_hw is None else (latent_hw[1], latent_hw[0])
            extra_seeds, region_info = seeds_info[i]
            lats.append(initial_latents(seeds[i], unet.in_channels, height, width,
                                        region_masks=lambda dtype, size, ri=region_info, es=extra_seeds: _get_binary_mask(ri, es, dtype, size),
                                        extra_seeds=extra_seeds))
        latents = torch.cat(lats, dim=0).to(device) * scheduler.init_noise_sigma
    else:                            # img2img, :434-441, :459-468
        offset = scheduler.config.get("steps_offset", 0)
        init_timestep = min(int(num_inference_steps * strength) + offset, num_inference_steps)
        t_start = max(num_inference_steps - init_timestep + offset, 0)
        timesteps = scheduler.timesteps[t_start:]
        lats = []
        for i in range(n):
            image = preprocess(init_images[i]).to(device=device)
            init_latents = 0.18215 * vae.encode(image.to(vae.dtype)).latent_dist.sample().float()
            noise = torch.randn(init_latents.shape).to(device)
            lats.append(scheduler.add_noise(init_latents, noise, timesteps[:1]))
        latents = torch.cat(lats, dim=0)

    with pww_hip.miopen_find():
        return sampler.sample(conds, unconds, latents, timesteps, guidance_scale, weight_function, on_step=on_step)


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
    color_map_image.size     # the reference dereferences it unconditionally (:414): None raises here too
    tools = (pww_load_tools(device, scheduler_type, local_model_path=local_model_path, hf_model_path=hf_model_path,
                            model_token=model_token) if preloaded_utils is None else preloaded_utils)
    latents = _generate(tools, device, [color_context], [color_map_image], [input_prompt], [seed], num_inference_steps,
                        guidance_scale, weight_function, unconditional_input_prompt,
                        init_images=None if init_image is None else [init_image], strength=strength, shared=True)
    if return_latents:
        return _sampler_for(tools[1], tools[4], DEFAULT_MODE).checked(latents)
    image = _pil_from_latents(tools[0], latents)[0]
    _sampler_for(tools[1], tools[4], DEFAULT_MODE).check_errors()     # (the decode above synchronised already)
    return image


@torch.no_grad()
def paint_with_words_batch(
    color_contexts: Union[Dict, Sequence[Dict]],
    color_map_images: Union[Image.Image, Sequence[Image.Image]],
    input_prompts: Union[str, Sequence[str]],
    seeds: Sequence[int],
    num_inference_steps: int = 30,
    guidance_scale: float = 7.5,
    scheduler_type=LMSDiscreteScheduler,
    device: str = "cuda:0",
    weight_function: Callable = lambda w, sigma, qk: 0.1 * w * math.log(sigma + 1) * qk.max(),
    local_model_path: Optional[str] = None,
    hf_model_path: Optional[str] = "CompVis/stable-diffusion-v1-4",
    preloaded_utils: Optional[Tuple] = None,
    unconditional_input_prompt: str = "",
    model_token: Optional[str] = None,
    init_images: Union[None, Image.Image, Sequence[Image.Image]] = None,
    strength: float = 0.5,
    return_latents: bool = False,
):
    """len(seeds) requests through ONE denoise loop (the reference's multi-sample path is a
    sequential loop of paint_with_words calls, gradio_pww.py:24-45). `color_contexts`, `color_map_images`,
    `input_prompts` and `init_images` are either one value shared by every request or a sequence with one entry per
    seed; each image has its own weight maps (kernel argument bias_stride[0]), prompt embedding, per-image score
    statistic and region seeds, so image i equals `paint_with_words(color_contexts[i], color_map_images[i],
    input_prompts[i], seed=seeds[i], ...)`. All color maps of one call must have the same size. The caller's
    color_context dicts are mutated like the single-image call mutates its dict (:296). Returns a list of PIL images
    (or the [n, 4, h, w] latents with return_latents=True)."""
    seeds = list(seeds)
    n = len(seeds)
    if n == 0:
        return []
    ctxs, s1 = _broadcast(color_contexts, n, "color_context")
    maps, s2 = _broadcast(color_map_images, n, "color_map_images")
    prompts, s3 = _broadcast(input_prompts, n, "input_prompts")
    inits = None if init_images is None else _broadcast(init_images, n, "init_images")[0]
    if len({m.size for m in maps}) != 1:
        raise ValueError("paint_with_words_batch: all color maps of one call must have the same size, got %s"
                         % sorted({m.size for m in maps}))
    shared = s1 and s2 and s3
    originals = ctxs
    if not shared:       # one dict may serve several requests: parse a private copy per request, strip the caller's afterwards
        ctxs = [dict(c) for c in ctxs]
    tools = (pww_load_tools(device, scheduler_type, local_model_path=local_model_path, hf_model_path=hf_model_path,
                            model_token=model_token) if preloaded_utils is None else preloaded_utils)
    latents = _generate(tools, device, ctxs, maps, prompts, seeds, num_inference_steps, guidance_scale, weight_function,
                        unconditional_input_prompt, init_images=inits, strength=strength, shared=shared)
    if not shared:
        for c in {id(c): c for c in originals}.values():
            _extract_seed_and_sigma_from_context(c)
    if return_latents:
        return _sampler_for(tools[1], tools[4], DEFAULT_MODE).checked(latents)
    images = _pil_from_latents(tools[0], latents)
    _sampler_for(tools[1], tools[4], DEFAULT_MODE).check_errors()
    return images


def __getattr__(name):
    """The reference defines its pipeline class in this module (:513); here it lives in pipelines.py (which imports this
    module), so the name resolves lazily."""
    if name == "PaintWithWord_StableDiffusionPipeline":
        from .pipelines import PaintWithWord_StableDiffusionPipeline
        return PaintWithWord_StableDiffusionPipeline
    raise AttributeError(name)
