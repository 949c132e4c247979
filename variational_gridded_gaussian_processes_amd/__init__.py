"""MI355X-native engine for the Kronecker-structured collapsed-ELBO hot path of
maxnorman569/Variational-Gridded-Gaussian-Processes (include/vggp.h).

The package imports on a CPU-only host (so the C-ABI can be checked), but every numeric entry
point requires the HIP library and a gfx950 GPU -- there is no CPU fallback.
"""
from . import _lib
from ._lib import VggpError
from .engine import Engine

__all__ = ["Engine", "VggpError", "_lib"]
__version__ = "0.1.0"
