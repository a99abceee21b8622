"""Same public surface as the reference package (paint_with_words/__init__.py:1-3), plus the batched entry points
(SURVEY.md 8 row f-2)."""
from .paint_with_words import paint_with_words, pww_load_tools, inj_forward, paint_with_words_batch
from .pipelines import PaintWithWord_StableDiffusionPipeline, PaintWithWord_StableDiffusionInpaintPipeline
from .paint_with_words_inpaint import paint_with_words_inpaint, paint_with_words_inpaint_batch
from .utils import fig_from_settings
