"""aither_amd -- MI355X (gfx950) implementation of AITHER's per-iteration
residual + implicit-sweep hot path behind a C-ABI (include/aither_gfx950.h).

  aither_amd.csrc     hand-written HIP kernels + the C-ABI (libaither_gfx950.so)
  aither_amd.abi      ctypes mirror of the C-ABI structs (plumbing)
  aither_amd.solver   host mirror of the reference's time-step loop
  aither_amd.case     case setup used by tests/bench (input deck, Plot3D
                      metrics, ghost geometry, connections)
"""
from ._lib import load, LIB_PATH, RANS_LIB_PATH  # noqa: F401
