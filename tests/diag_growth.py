"""Diagnostic (not a test): per-iteration growth of GPU-vs-oracle differences."""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import aither_amd
from aither_amd import abi
from aither_amd.solver import Solver
from conftest import golden_case, _oracle_lib
from parity_utils import rel_err
agx = aither_amd.load()
ora = abi.Api(ctypes.CDLL(_oracle_lib()), "ora_")
for name, steps in [("subsonicCylinder", 12), ("multiblockCylinder", 6), ("viscousFlatPlate", 6), ("shockTube", 3), ("couette", 6)]:
    case = golden_case(name)
    sg, so = Solver(agx, case), Solver(ora, case)
    ng = case.ng
    for nn in range(steps):
        sg.step(nn); so.step(nn)
        es = max(rel_err(sg.download("state", b)[ng:-ng, ng:-ng, ng:-ng], so.download("state", b)[ng:-ng, ng:-ng, ng:-ng]) for b in sg.block_ids)
        er = max(rel_err(sg.download("residual", b), so.download("residual", b)) for b in sg.block_ids)
        el = rel_err(sg.history[-1]["l2"][None], so.history[-1]["l2"][None])
        print(f"{name} nn={nn} state {es:.2e} resid {er:.2e} l2 {el:.2e} |r|max {np.abs(so.download('residual', 0)).max():.3e}")
    sg.close(); so.close()
