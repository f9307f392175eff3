"""Loader for the in-tree HIP library (fails loudly; there is no fallback)."""
import ctypes
import os

from . import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
# AGX_LIB: a diagnostic build of the same library (e.g. -DAGX_KP_TRACE)
LIB_PATH = os.environ.get("AGX_LIB") or os.path.join(_HERE, "libaither_gfx950.so")
_api = None


def load():
    """Return the bound C-ABI of libaither_gfx950.so."""
    global _api
    if _api is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with "
                "`python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
        _api = abi.Api(ctypes.CDLL(LIB_PATH), "agx_")
    return _api
