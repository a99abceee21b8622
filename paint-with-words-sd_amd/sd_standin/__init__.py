"""Stand-ins for the reference's un-vendored dependencies (diffusers==0.10.0, CLIP, torchvision):
only the pieces the Paint-with-Words path drives. See each module's docstring."""
from .unet import (CrossAttention, BasicTransformerBlock, Transformer2DModel, UNet2DConditionModel, build_unet,
                   SD15_CONFIG, SD15_INPAINT_CONFIG, SD21_CONFIG, TINY_CONFIG, TINY_SD2_CONFIG)
from .schedulers import LMSDiscreteScheduler, PLMSScheduler
from .text import HashTokenizer, TinyTextEncoder, build_clip_text_encoder
from .vae import TinyVAE
