"""pww_hip: Python face of libpww_hip.so (hand-written gfx950 kernels for the Paint-with-Words
attention path). `import pww_hip` never touches the GPU; the shared library is loaded on first use
and a missing library raises PwwHipError (no fallback)."""
from ._lib import PwwHipError, load as load_library, device_arch, LIB_PATH, EXPORTS
from . import ops
from .attention import QKProxy, ScaledW, inj_forward, install, uninstall, PwWAttnProcessor, pww_attention



def enable_miopen_find():
    """Let MIOpen time its solvers per convolution shape and keep the fastest (`torch.backends.cudnn.benchmark = True`,
    a stock PyTorch-ROCm setting: the UNet's convolutions stay stock ops). Costs a one-off search per new shape
    (~25 s for the SD1.5 UNet) and buys ~6 % end to end on MI355X (2.48 -> 2.64 images/s). `PWW_MIOPEN_FIND=0`
    leaves PyTorch's default (immediate mode) alone. Called by the drop-in API entry points and by bench.py."""
    import os
    import torch
    if os.environ.get("PWW_MIOPEN_FIND", "1") != "0":
        torch.backends.cudnn.benchmark = True


__all__ = ["PwwHipError", "enable_miopen_find", "load_library", "device_arch", "ops", "QKProxy", "ScaledW", "inj_forward", "install", "uninstall",
           "PwWAttnProcessor", "pww_attention", "LIB_PATH", "EXPORTS"]
