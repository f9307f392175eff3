"""Diagnostic (not a test): where do GPU and oracle residuals differ most."""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import aither_amd
from aither_amd import abi
from aither_amd.solver import Solver
from conftest import golden_case, _oracle_lib
agx = aither_amd.load()
ora = abi.Api(ctypes.CDLL(_oracle_lib()), "ora_")
name = sys.argv[1] if len(sys.argv) > 1 else "subsonicCylinder"
case = golden_case(name)
sg, so = Solver(agx, case), Solver(ora, case)
ng = case.ng
sg.step(0); so.step(0)
for b in sg.block_ids:
    rg, ro = sg.download("residual", b), so.download("residual", b)
    d = np.abs(rg - ro)
    idx = np.unravel_index(d.argmax(), d.shape)
    print("block", b, "max abs resid diff", d.max(), "at (k,j,i,e)", idx, "gpu", rg[idx], "ora", ro[idx], "max|r|", np.abs(ro).max())
    for e in range(5):
        ie = np.unravel_index(d[..., e].argmax(), d[..., e].shape)
        print("  eq", e, "maxdiff", d[..., e].max(), "at", ie, "ref", ro[ie + (e,)], "max|r_e|", np.abs(ro[..., e]).max())
    stg, sto = sg.download("state", b), so.download("state", b)
    ds = np.abs(stg - sto)
    ds[~np.isfinite(ds)] = 0
    idx = np.unravel_index(ds.argmax(), ds.shape)
    print("  state (incl ghosts) max diff", ds.max(), "at", idx, stg[idx], sto[idx])
    dtg, dto = sg.download("dt", b), so.download("dt", b)
    print("  dt rel diff", (np.abs(dtg - dto) / np.abs(dto)).max())
    g = case.blocks[b].geom
    print("  areas i max", g.farea['i'].a[..., 3].max(), "j max", g.farea['j'].a[..., 3].max(), "k max", g.farea['k'].a[..., 3].max(), "min", g.farea['i'].a[ng:-ng,ng:-ng,ng:-ng, 3].min())
