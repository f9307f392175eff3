"""Loader for the in-tree HIP library (fails loudly; there is no fallback)."""
import ctypes
import os

from . import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
# AGX_LIB: a diagnostic build of the same library (e.g. -DAGX_KP_TRACE)
LIB_PATH = os.environ.get("AGX_LIB") or os.path.join(_HERE, "libaither_gfx950.so")
# the same sources built for the 7-equation set (rans: + k, omega), same C-ABI
RANS_LIB_PATH = os.environ.get("AGX_RANS_LIB") or os.path.join(_HERE, "libaither_gfx950_rans.so")
_api = {}


def load(n_eq=5):
    """Return the bound C-ABI of libaither_gfx950.so (n_eq = 5: euler /
    navierStokes) or libaither_gfx950_rans.so (n_eq = 7: rans)."""
    if n_eq not in (5, 7):
        raise ValueError("n_eq is 5 or 7")
    if n_eq not in _api:
        path = LIB_PATH if n_eq == 5 else RANS_LIB_PATH
        if not os.path.exists(path):
            raise RuntimeError(
                f"{path} is missing: build it with "
                "`python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
        _api[n_eq] = abi.Api(ctypes.CDLL(path), "agx_")
    return _api[n_eq]
