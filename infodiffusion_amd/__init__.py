"""infodiffusion_amd -- MI355X-native (gfx950) implementation of InfoDiffusion's
data-parallel hot path behind the reference's Python surface.  The HIP extension
(libinfodiff_hip.so) is required; there is no CPU / eager fallback."""
from . import _lib

__all__ = ['_lib']
