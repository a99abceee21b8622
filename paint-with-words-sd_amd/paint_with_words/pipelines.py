"""Pipeline-class face of the same algorithm (reference paint_with_words.py:513-842 and
paint_with_words_inpaint.py:273-575). The reference subclasses diffusers' StableDiffusionPipeline,
which is not installable offline; these classes keep the reference's constructor / `from_pretrained`
/ `plugin_cross_attention` / `__call__` names and delegate to the function API (one code path)."""
import math
from types import SimpleNamespace

import torch

import pww_hip
from .paint_with_words import paint_with_words, pww_load_tools, LMSDiscreteScheduler
from . import paint_with_words_inpaint as _inp

_default_weight = lambda w, sigma, qk: 0.1 * w * math.log(sigma + 1) * qk.max()  # noqa: E731


class PaintWithWord_StableDiffusionPipeline:
    def __init__(self, vae, text_encoder, tokenizer, unet, scheduler=None, safety_checker=None, feature_extractor=None,
                 requires_safety_checker=False):
        self.vae, self.text_encoder, self.tokenizer, self.unet = vae, text_encoder, tokenizer, unet
        self.scheduler = scheduler or LMSDiscreteScheduler(beta_start=0.00085, beta_end=0.012,
                                                           beta_schedule="scaled_linear", num_train_timesteps=1000)
        self.device = next(unet.parameters()).device
        self.plugin_cross_attention()

    @classmethod
    def from_pretrained(cls, model_path, **kwargs):
        vae, unet, text_encoder, tokenizer, scheduler = pww_load_tools(
            kwargs.get("device", "cuda:0"), LMSDiscreteScheduler, hf_model_path=model_path)
        return cls(vae, text_encoder, tokenizer, unet, scheduler)

    def to(self, device):
        for m in (self.vae, self.text_encoder, self.unet):
            m.to(device)
        self.device = torch.device(device)
        return self

    def plugin_cross_attention(self):
        """reference :556-559"""
        pww_hip.install(self.unet)

    def _tools(self):
        return (self.vae, self.unet, self.text_encoder, self.tokenizer, self.scheduler)

    def __call__(self, prompt, color_context={}, color_map_image=None, num_inference_steps=50, guidance_scale=7.5,
                 negative_prompt="", weight_function=_default_weight, seed=0, init_image=None, eta=0.5, **kwargs):
        img = paint_with_words(color_context=color_context, color_map_image=color_map_image, input_prompt=prompt,
                               num_inference_steps=num_inference_steps, guidance_scale=guidance_scale, seed=seed,
                               device=str(self.device), weight_function=weight_function,
                               preloaded_utils=self._tools(), unconditional_input_prompt=negative_prompt or "",
                               init_image=init_image, strength=eta)
        return SimpleNamespace(images=[img], nsfw_content_detected=None)


class PaintWithWord_StableDiffusionInpaintPipeline(PaintWithWord_StableDiffusionPipeline):
    def __call__(self, prompt, image=None, mask_image=None, color_context={}, color_map_image=None,
                 num_inference_steps=50, guidance_scale=7.5, negative_prompt="", weight_function=_default_weight,
                 seed=0, eta=1.0, **kwargs):
        img = _inp.paint_with_words_inpaint(
            color_context=color_context, color_map_image=color_map_image, mask_image=mask_image, init_image=image,
            input_prompt=prompt, num_inference_steps=num_inference_steps, guidance_scale=guidance_scale, seed=seed,
            device=str(self.device), weight_function=weight_function, preloaded_utils=self._tools(),
            unconditional_input_prompt=negative_prompt or "", strength=eta)
        return SimpleNamespace(images=[img], nsfw_content_detected=None)
