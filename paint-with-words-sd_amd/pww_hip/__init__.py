"""pww_hip: Python face of libpww_hip.so (hand-written gfx950 kernels for the Paint-with-Words
attention path). `import pww_hip` never touches the GPU; the shared library is loaded on first use
and a missing library raises PwwHipError (no fallback)."""
from ._lib import PwwHipError, load as load_library, device_arch, LIB_PATH, EXPORTS
from . import ops
from .attention import QKProxy, ScaledW, inj_forward, install, uninstall, PwWAttnProcessor, pww_attention

__all__ = ["PwwHipError", "load_library", "device_arch", "ops", "QKProxy", "ScaledW", "inj_forward", "install", "uninstall",
           "PwWAttnProcessor", "pww_attention", "LIB_PATH", "EXPORTS"]
