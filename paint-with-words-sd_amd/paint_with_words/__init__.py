"""Same public surface as the reference package (paint_with_words/__init__.py:1-3)."""
from .paint_with_words import paint_with_words, pww_load_tools, inj_forward
from .pipelines import PaintWithWord_StableDiffusionPipeline, PaintWithWord_StableDiffusionInpaintPipeline
from .paint_with_words_inpaint import paint_with_words_inpaint
from .utils import fig_from_settings
