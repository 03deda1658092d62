"""compv_amd -- MI355X (gfx950) implementation of CompV's Sobel -> Canny -> Hough hot path.

The product is the C-ABI shared library ``compv_amd/lib/libcompv_hip.so`` (include/compv_hip.h, sources in
``compv_amd/csrc``) plus the CompV-side plugin classes in ``integration/compv_hip_plugin.cxx``; ``compv_amd.capi`` is a ctypes
binding used by the tests and the benchmark.  There is no CPU fallback anywhere in this package.
"""
from . import capi  # noqa: F401

__all__ = ["capi"]
