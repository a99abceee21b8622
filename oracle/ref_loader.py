"""Load the REAL reference functions for golden generation (works only where /root/reference exists).

TEST INFRASTRUCTURE -- not product code. `import paint_with_words` from /root/reference is
impossible here (diffusers / torchvision are not installed, transformers 5.x lacks
CLIPFeatureExtractor), but the hot-path code itself is plain torch/numpy: the top-level
``FunctionDef`` nodes of the reference files are AST-extracted and exec'd UNMODIFIED in a namespace
that supplies the names their bodies use. Nothing is copied into this repo:
the source is read from /root/reference at run time and only its OUTPUTS are stored as fixtures
(oracle/make_golden.py -> tests/golden/).

The GPU box has no /root/reference: nothing under tests marked gpu, smoke() or bench.py may call
this module (they use oracle/pww_oracle.py, the restatement that is pinned against these outputs).
"""
import ast
import math
import os
import sys
from types import SimpleNamespace
from typing import Callable, Dict, List, Optional, Tuple, Union

import numpy as np
import PIL
import torch
import torch.nn.functional as F
from PIL import Image

REFERENCE_ROOT = os.environ.get("PWW_REFERENCE_ROOT", "/root/reference")

_REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_PKG = os.path.join(_REPO, "paint-with-words-sd_amd")
if _PKG not in sys.path:
    sys.path.insert(0, _PKG)


def available():
    return os.path.isfile(os.path.join(REFERENCE_ROOT, "paint_with_words", "paint_with_words.py"))


class _GaussianBlur:
    """torchvision.transforms.GaussianBlur stand-in (torchvision is not installed): separable
    Gaussian, reflect padding, kernel exp(-0.5 (x/sigma)^2) normalised -- the published algorithm of
    torchvision.transforms.functional.gaussian_blur for a float tensor [..., H, W]."""

    def __init__(self, kernel_size, sigma):
        self.kernel_size = tuple(kernel_size)
        self.sigma = tuple(float(s) for s in sigma)

    @staticmethod
    def _kernel1d(ksize, sigma):
        half = (ksize - 1) * 0.5
        x = torch.linspace(-half, half, steps=ksize)
        pdf = torch.exp(-0.5 * (x / sigma).pow(2))
        return pdf / pdf.sum()

    def __call__(self, img):
        kx = self._kernel1d(self.kernel_size[0], self.sigma[0])
        ky = self._kernel1d(self.kernel_size[1], self.sigma[1])
        kernel = torch.mm(ky[:, None], kx[None, :]).to(img.dtype)
        shape = img.shape
        x = img.reshape(-1, 1, shape[-2], shape[-1])
        pad = [self.kernel_size[0] // 2, self.kernel_size[0] // 2, self.kernel_size[1] // 2, self.kernel_size[1] // 2]
        x = F.pad(x, pad, mode="reflect")
        x = F.conv2d(x, kernel[None, None])
        return x.reshape(shape)


class _StableDiffusionPipelineStub:
    """The members of diffusers==0.10.0's `StableDiffusionPipeline` that the reference's two pipeline classes call on `self`
    (paint_with_words.py:513-842, paint_with_words_inpaint.py:273-575), restated from the published 0.10.0 sources (the package
    is not installable offline): module registration, `vae_scale_factor`, `_execution_device`, `check_inputs`,
    `prepare_extra_step_kwargs`, `progress_bar`, `decode_latents`, `numpy_to_pil`. The reference's classes are exec'd UNMODIFIED
    with this class bound to the name `StableDiffusionPipeline`."""

    def __init__(self, vae, text_encoder, tokenizer, unet, scheduler, safety_checker, feature_extractor, requires_safety_checker=True):
        self.vae, self.text_encoder, self.tokenizer, self.unet, self.scheduler = vae, text_encoder, tokenizer, unet, scheduler
        self.safety_checker, self.feature_extractor = safety_checker, feature_extractor
        self.vae_scale_factor = 2 ** (len(self.vae.config.block_out_channels) - 1)

    @property
    def device(self):
        return next(self.unet.parameters()).device

    @property
    def _execution_device(self):
        return self.device

    def check_inputs(self, prompt, height, width, callback_steps):
        if not isinstance(prompt, str) and not isinstance(prompt, list):
            raise ValueError(f"`prompt` has to be of type `str` or `list` but is {type(prompt)}")
        if height % 8 != 0 or width % 8 != 0:
            raise ValueError(f"`height` and `width` have to be divisible by 8 but are {height} and {width}.")
        if (callback_steps is None) or (callback_steps is not None and (not isinstance(callback_steps, int) or callback_steps <= 0)):
            raise ValueError(f"`callback_steps` has to be a positive integer but is {callback_steps} of type {type(callback_steps)}.")

    def prepare_extra_step_kwargs(self, generator, eta):
        import inspect
        accepted = set(inspect.signature(self.scheduler.step).parameters.keys())
        kw = {}
        if "eta" in accepted:
            kw["eta"] = eta
        if "generator" in accepted:
            kw["generator"] = generator
        return kw

    def progress_bar(self, iterable=None, total=None):
        import contextlib

        class _Bar:
            def update(self, n=1):
                pass
        return contextlib.nullcontext(_Bar())

    def decode_latents(self, latents):
        latents = 1 / 0.18215 * latents
        image = self.vae.decode(latents).sample
        image = (image / 2 + 0.5).clamp(0, 1)
        return image.cpu().permute(0, 2, 3, 1).float().numpy()

    @staticmethod
    def numpy_to_pil(images):
        if images.ndim == 3:
            images = images[None, ...]
        images = (images * 255).round().astype("uint8")
        return [Image.fromarray(image) for image in images]


def load_reference(filename="paint_with_words.py", extra=None, classes=False):
    """Return a namespace dict with the reference's top-level functions of `filename` (and, with `classes`, its classes:
    the pipeline classes subclass diffusers' StableDiffusionPipeline, bound here to _StableDiffusionPipelineStub)."""
    from sd_standin import LMSDiscreteScheduler
    path = os.path.join(REFERENCE_ROOT, "paint_with_words", filename)
    src = open(path).read()
    kinds = (ast.FunctionDef, ast.ClassDef) if classes else (ast.FunctionDef,)
    defs = [n for n in ast.parse(src).body if isinstance(n, kinds)]
    ns = dict(math=math, np=np, torch=torch, F=F, PIL=PIL, Image=Image, tqdm=lambda it, **kw: it,
              Callable=Callable, Dict=Dict, List=List, Optional=Optional, Tuple=Tuple, Union=Union,
              LMSDiscreteScheduler=LMSDiscreteScheduler, UNet2DConditionModel=None, CLIPTextModel=None,
              CLIPTokenizer=None, AutoencoderKL=None, PNDMScheduler=None, CLIPFeatureExtractor=None,
              StableDiffusionPipeline=_StableDiffusionPipelineStub,
              StableDiffusionPipelineOutput=lambda images, nsfw_content_detected: SimpleNamespace(images=images, nsfw_content_detected=nsfw_content_detected),
              T=SimpleNamespace(GaussianBlur=_GaussianBlur))
    if extra:
        ns.update(extra)
    exec(compile(ast.Module(body=defs, type_ignores=[]), path, "exec"), ns)
    return ns


def load_reference_inpaint(classes=False):
    base = load_reference("paint_with_words.py", classes=classes)
    names = ("pww_load_tools", "preprocess", "_pil_from_latents", "_encode_text_color_inputs") + (("PaintWithWord_StableDiffusionPipeline",) if classes else ())
    extra = {k: base[k] for k in names}
    ns = load_reference("paint_with_words_inpaint.py", extra=extra, classes=classes)
    ns["_base_namespace"] = base
    return ns
