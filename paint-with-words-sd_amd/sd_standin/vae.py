"""Minimal AutoencoderKL stand-in: the hot path ends at the final latent (parity is checked there);
the reference only needs ``vae.decode(z).sample`` (paint_with_words/paint_with_words.py:50) and, for
img2img / inpaint, ``vae.encode(x).latent_dist.sample()`` (:461-462)."""
from types import SimpleNamespace

import torch
import torch.nn as nn
import torch.nn.functional as F


class TinyVAE(nn.Module):
    def __init__(self, latent_channels=4, seed=1236):
        super().__init__()
        state = torch.random.get_rng_state()
        torch.manual_seed(seed)
        try:
            self.dec = nn.Conv2d(latent_channels, 3, 3, padding=1)
            self.enc = nn.Conv2d(3, latent_channels, 8, stride=8)
        finally:
            torch.random.set_rng_state(state)
        self.requires_grad_(False)

    @property
    def dtype(self):
        return self.dec.weight.dtype

    def decode(self, z):
        x = F.interpolate(self.dec(z.to(self.dtype)), scale_factor=8.0, mode="nearest")
        return SimpleNamespace(sample=torch.tanh(x))

    def encode(self, x):
        mean = self.enc(x.to(self.dtype))
        return SimpleNamespace(latent_dist=SimpleNamespace(sample=lambda generator=None: mean, mode=lambda: mean))
