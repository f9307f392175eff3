"""Residual history of bench.py's rans4 case on the GPU (diagnostic)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import aither_amd, bench
from aither_amd.solver import Solver

n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
its = int(sys.argv[2]) if len(sys.argv) > 2 else 30
api = aither_amd.load(7)
case = bench.rank_local_chain_case(0, 1, n, "rans4")
if len(sys.argv) > 3:
    case.deck.cfl_start = case.deck.cfl_max = float(sys.argv[3])
sol = Solver(api, case)
for it in range(its):
    sol.store_time_n(it)
    l2, linf, mres = sol.iterate(0, case.deck.cfl(it))
    if it % 5 == 4 or not np.all(np.isfinite(l2)): print(it, " ".join(f"{v:.3e}" for v in np.sqrt(np.asarray(l2))), f"mres {mres:.3e}", flush=True)
    if not np.all(np.isfinite(l2)):
        break
sol.close()
