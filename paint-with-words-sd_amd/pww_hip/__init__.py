"""pww_hip: Python face of libpww_hip.so (hand-written gfx950 kernels for the Paint-with-Words
attention path). `import pww_hip` never touches the GPU; the shared library is loaded on first use
and a missing library raises PwwHipError (no fallback)."""
from ._lib import PwwHipError, load as load_library, device_arch, LIB_PATH, EXPORTS
from . import ops
from . import blocks
from .attention import QKProxy, ScaledW, inj_forward, install, uninstall, PwWAttnProcessor, pww_attention



def _miopen_find_wanted():
    import os
    return os.environ.get("PWW_MIOPEN_FIND", "1") != "0"


def enable_miopen_find():
    """Process-wide switch (what bench.py uses): let MIOpen time its solvers per convolution shape and keep the fastest
    (`torch.backends.cudnn.benchmark = True`, a stock PyTorch-ROCm setting: the UNet's convolutions stay stock ops).
    Costs a one-off search per new shape (~25 s for the SD1.5 UNet) and buys ~8 % end to end on MI355X.
    `PWW_MIOPEN_FIND=0` leaves PyTorch's default (immediate mode) alone. Library callers use `miopen_find()` below,
    which restores the caller's setting."""
    import torch
    if _miopen_find_wanted():
        torch.backends.cudnn.benchmark = True


class miopen_find:
    """Scoped form used by the drop-in API entry points (paint_with_words, paint_with_words_inpaint, the pipeline
    classes): MIOpen's find mode is on for the duration of the call and the caller's
    `torch.backends.cudnn.benchmark` value is put back afterwards -- calling the drop-in API has no process-global
    side effect. `PWW_MIOPEN_FIND=0` makes it a no-op."""

    def __enter__(self):
        import torch
        self._prev = torch.backends.cudnn.benchmark
        if _miopen_find_wanted():
            torch.backends.cudnn.benchmark = True
        return self

    def __exit__(self, *exc):
        import torch
        torch.backends.cudnn.benchmark = self._prev
        return False


__all__ = ["PwwHipError", "enable_miopen_find", "miopen_find", "load_library", "device_arch", "ops", "QKProxy", "ScaledW", "inj_forward", "install", "uninstall",
           "PwWAttnProcessor", "pww_attention", "LIB_PATH", "EXPORTS"]
