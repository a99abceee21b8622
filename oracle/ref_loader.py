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


def load_reference(filename="paint_with_words.py", extra=None):
    """Return a namespace dict with the reference's top-level functions of `filename`."""
    from sd_standin import LMSDiscreteScheduler
    path = os.path.join(REFERENCE_ROOT, "paint_with_words", filename)
    src = open(path).read()
    defs = [n for n in ast.parse(src).body if isinstance(n, ast.FunctionDef)]
    ns = dict(math=math, np=np, torch=torch, F=F, PIL=PIL, Image=Image, tqdm=lambda it, **kw: it,
              Callable=Callable, Dict=Dict, List=List, Optional=Optional, Tuple=Tuple, Union=Union,
              LMSDiscreteScheduler=LMSDiscreteScheduler, UNet2DConditionModel=None, CLIPTextModel=None,
              CLIPTokenizer=None, AutoencoderKL=None, PNDMScheduler=None,
              T=SimpleNamespace(GaussianBlur=_GaussianBlur))
    if extra:
        ns.update(extra)
    exec(compile(ast.Module(body=defs, type_ignores=[]), path, "exec"), ns)
    return ns


def load_reference_inpaint():
    base = load_reference("paint_with_words.py")
    extra = {k: base[k] for k in ("pww_load_tools", "preprocess", "_pil_from_latents", "_encode_text_color_inputs")}
    return load_reference("paint_with_words_inpaint.py", extra=extra)
