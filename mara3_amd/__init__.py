"""mara3_amd — MI355X-native engine for the Mara3 per-step hydro hot path.

The product is libmara_hip.so (hand-written HIP for gfx950 behind the C ABI in
include/mara_hip.h). This package is the thin Python host used by the tests and
bench.py: ctypes bindings, slab decomposition and halo exchange over
torch.distributed. There is no CPU fallback: importing the bindings without the
built library raises.
"""
from ._lib import load_library, library_path, MaraHipError  # noqa: F401
from . import setups  # noqa: F401

__all__ = ["load_library", "library_path", "MaraHipError", "setups"]
