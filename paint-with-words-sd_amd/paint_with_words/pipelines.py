"""Pipeline-class face of the same algorithm (reference paint_with_words.py:513-842 and
paint_with_words_inpaint.py:273-575). The reference subclasses diffusers' StableDiffusionPipeline (not installable
offline); these classes keep the reference's constructor / `from_pretrained` / `plugin_cross_attention` / `__call__`
signatures -- argument names, ORDER and defaults -- and run the shared generation body of the function API.

Behaviour the reference's pipeline variant has and the function API has not, kept on purpose:
  * height / width default to `unet.config.sample_size * 8` and size the LATENT; the color map only sizes the weight
    maps (:700-701, :756);
  * the per-region blur sigma is parsed and dropped (:574); `eta` doubles as the img2img strength (:735);
  * `latents`, `generator` and `num_images_per_prompt` are accepted and not used (:744-753 is commented out in the
    reference; the latents always come from `seed`): passing a non-default value warns once instead of failing;
  * the inpaint class takes color map and mask as given (no resize to the init image, unlike the function API :172-173),
    sizes the latent mask by `height` / `width` (paint_with_words_inpaint.py:427-432, :498-508 -- they must be the size of
    `image`) and calls `callback(i, t, latents)` every `callback_steps` steps (:555-559);
  * the safety checker never runs (`nsfw_content_detected` is False, :829).
"""
import math
import warnings
from types import SimpleNamespace
from typing import Callable, Dict, List, Optional, Tuple, Union

import numpy as np
import torch
from PIL import Image

import pww_hip
import importlib
_pw = importlib.import_module(__name__.rsplit(".", 1)[0] + ".paint_with_words")
from .paint_with_words import (pww_load_tools, LMSDiscreteScheduler, _generate, _pil_from_latents, _sampler_for)
from . import paint_with_words_inpaint as _inp

_warned = set()


def _unused(name, value, default):
    if value is not default and value != default and name not in _warned:
        _warned.add(name)
        warnings.warn("%s is accepted for signature compatibility and not used (the reference's pipeline ignores it too)" % name)


def _decode(vae, latents, output_type):
    """decode_latents + numpy_to_pil of the diffusers pipeline (:821-833): [n, H, W, 3] float array or PIL images."""
    decoded = vae.decode((latents / 0.18215).to(vae.dtype)).sample
    pixels = (decoded / 2 + 0.5).clamp(0, 1).float().cpu().permute(0, 2, 3, 1).numpy()
    if output_type == "pil":
        return [Image.fromarray(im) for im in (pixels * 255).round().astype("uint8")]
    return pixels


class PaintWithWord_StableDiffusionPipeline:
    def __init__(self, vae, text_encoder, tokenizer, unet, scheduler=None, safety_checker=None, feature_extractor=None,
                 requires_safety_checker: bool = False):
        self.vae, self.text_encoder, self.tokenizer, self.unet = vae, text_encoder, tokenizer, unet
        self.safety_checker, self.feature_extractor = safety_checker, feature_extractor
        # the reference replaces whatever scheduler it is given by LMSDiscrete (:533-538)
        self.scheduler = LMSDiscreteScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
                                              num_train_timesteps=1000)
        self.vae_scale_factor = 8
        self.device = next(unet.parameters()).device
        self.plugin_cross_attention()

    @classmethod
    def from_pretrained(cls, save_dir, **kwargs):
        """reference :541-554. `save_dir` is a local directory or a hub id; `torch_dtype`, `revision`, `use_auth_token`,
        `local_files_only` ... go to diffusers' loader untouched when diffusers is installed."""
        try:
            from diffusers import StableDiffusionPipeline
        except Exception as e:
            raise ImportError("from_pretrained needs `diffusers` (the reference pins diffusers==0.10.0); build the modules "
                              "yourself and call %s(vae, text_encoder, tokenizer, unet, ...)" % cls.__name__) from e
        sd = StableDiffusionPipeline.from_pretrained(save_dir, **kwargs)
        return cls(vae=sd.vae, text_encoder=sd.text_encoder, tokenizer=sd.tokenizer, unet=sd.unet, scheduler=sd.scheduler,
                   safety_checker=getattr(sd, "safety_checker", None), feature_extractor=getattr(sd, "feature_extractor", None),
                   requires_safety_checker=getattr(sd, "requires_safety_checker", False))

    def to(self, device):
        for m in (self.vae, self.text_encoder, self.unet):
            m.to(device)
        self.device = torch.device(device)
        return self

    def plugin_cross_attention(self):
        """reference :556-559"""
        if pww_hip.install(self.unet) == 0 and hasattr(self.unet, "set_attn_processor"):
            self.unet.set_attn_processor(pww_hip.PwWAttnProcessor())

    def _tools(self):
        return (self.vae, self.unet, self.text_encoder, self.tokenizer, self.scheduler)

    def _default_side(self):
        cfg = getattr(self.unet, "config", None)
        size = cfg.get("sample_size") if isinstance(cfg, dict) else getattr(cfg, "sample_size", None)
        return (size or 64) * self.vae_scale_factor

    @staticmethod
    def _callback_adapter(callback, callback_steps):
        """the reference calls `callback(i, t, latents)` every `callback_steps` steps (:815-816)"""
        if callback is None:
            return None
        return lambda i, t, latents: callback(i, t, latents) if i % callback_steps == 0 else None

    @torch.no_grad()
    def __call__(
        self,
        prompt: Union[str, List[str]],
        color_map_image: Optional[Image.Image] = None,
        color_context: Dict[Tuple[int, int, int], str] = {},
        weight_function: Callable = lambda w, sigma, qk: 0.1 * w * math.log(sigma + 1) * qk.max(),
        height: Optional[int] = None,
        width: Optional[int] = None,
        num_inference_steps: int = 30,
        guidance_scale: float = 7.5,
        negative_prompt: Optional[Union[str, List[str]]] = "",
        num_images_per_prompt: Optional[int] = 1,
        eta: float = 0.5,
        seed: Optional[int] = 0,
        generator: Optional[torch.Generator] = None,
        image: Optional[Image.Image] = None,
        latents: Optional[torch.FloatTensor] = None,
        output_type: Optional[str] = "pil",
        return_dict: bool = True,
        callback: Optional[Callable[[int, int, torch.FloatTensor], None]] = None,
        callback_steps: Optional[int] = 1,
    ):
        """reference :629-842 (same parameter list, order and defaults)."""
        height = height or self._default_side()
        width = width or self._default_side()
        if height % 8 or width % 8:                                   # check_inputs of the diffusers base class
            raise ValueError(f"`height` and `width` have to be divisible by 8 but are {height} and {width}.")
        if callback_steps is None or not isinstance(callback_steps, int) or callback_steps <= 0:
            raise ValueError(f"`callback_steps` has to be a positive integer but is {callback_steps} of type {type(callback_steps)}.")
        if not isinstance(prompt, str):
            if not isinstance(prompt, (list, tuple)) or len(prompt) != 1:
                raise ValueError("`prompt` has to be a str (the reference's pipeline generates one image per call)")
            prompt = prompt[0]
        if isinstance(negative_prompt, (list, tuple)):
            negative_prompt = negative_prompt[0] if negative_prompt else ""
        _unused("num_images_per_prompt", num_images_per_prompt, 1)
        _unused("generator", generator, None)
        _unused("latents", latents, None)

        lat = _generate(self._tools(), str(self.device), [color_context], [color_map_image], [prompt], [seed], num_inference_steps,
                        guidance_scale, weight_function, negative_prompt or "", init_images=None if image is None else [image],
                        strength=eta, latent_hw=(height, width), use_region_sigma=False, shared=True,
                        on_step=self._callback_adapter(callback, callback_steps))
        images = _decode(self.vae, lat, output_type)
        _sampler_for(self.unet, self.scheduler, _pw.DEFAULT_MODE).check_errors()
        if not return_dict:
            return (images, False)
        return SimpleNamespace(images=images, nsfw_content_detected=False)


class PaintWithWord_StableDiffusionInpaintPipeline(PaintWithWord_StableDiffusionPipeline):
    @torch.no_grad()
    def __call__(
        self,
        prompt: Union[str, List[str]],
        image: Image.Image = None,
        mask_image: Optional[Image.Image] = None,
        color_map_image: Optional[Image.Image] = None,
        color_context: Dict[Tuple[int, int, int], str] = {},
        weight_function: Callable = lambda w, sigma, qk: 0.1 * w * math.log(sigma + 1) * qk.max(),
        height: Optional[int] = None,
        width: Optional[int] = None,
        num_inference_steps: int = 30,
        guidance_scale: float = 7.5,
        negative_prompt: Optional[Union[str, List[str]]] = "",
        num_images_per_prompt: Optional[int] = 1,
        eta: float = 1.0,
        seed: Optional[int] = 0,
        generator: Optional[torch.Generator] = None,
        latents: Optional[torch.FloatTensor] = None,
        output_type: Optional[str] = "pil",
        return_dict: bool = True,
        callback: Optional[Callable[[int, int, torch.FloatTensor], None]] = None,
        callback_steps: Optional[int] = 1,
    ):
        """reference paint_with_words_inpaint.py:340-575 (same parameter list, order and defaults)."""
        if image is None or mask_image is None:
            raise ValueError("`image` and `mask_image` are required for inpainting")
        height = height or self._default_side()                      # :427-428
        width = width or self._default_side()
        if height % 8 or width % 8:                                   # check_inputs of the diffusers base class (:431)
            raise ValueError(f"`height` and `width` have to be divisible by 8 but are {height} and {width}.")
        if callback_steps is None or not isinstance(callback_steps, int) or callback_steps <= 0:
            raise ValueError(f"`callback_steps` has to be a positive integer but is {callback_steps} of type {type(callback_steps)}.")
        if not isinstance(prompt, str):
            if not isinstance(prompt, (list, tuple)) or len(prompt) != 1:
                raise ValueError("`prompt` has to be a str (the reference's pipeline generates one image per call)")
            prompt = prompt[0]
        if isinstance(negative_prompt, (list, tuple)):
            negative_prompt = negative_prompt[0] if negative_prompt else ""
        for name, value, default in (("num_images_per_prompt", num_images_per_prompt, 1), ("generator", generator, None), ("latents", latents, None)):
            _unused(name, value, default)
        lat = _inp._generate_inpaint(self._tools(), str(self.device), [color_context], [color_map_image], [mask_image], [image], [prompt],
                                     [seed], num_inference_steps, guidance_scale, weight_function, negative_prompt or "", eta, shared=True,
                                     on_step=self._callback_adapter(callback, callback_steps), mask_hw=(height, width), resize_inputs=False,
                                     use_region_sigma=False)
        images = _decode(self.vae, lat, output_type)
        _sampler_for(self.unet, self.scheduler, _pw.DEFAULT_MODE).check_errors()
        if not return_dict:
            return (images, False)
        return SimpleNamespace(images=images, nsfw_content_detected=False)
