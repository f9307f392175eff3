#!/usr/bin/env python3
"""Time of a multigrid iteration against a single-grid one on the GPU (synthetic box, Euler,
MUSCL + vanAlbada + Roe; scalar DPLUR with 4 sweeps, or `lusgs` with 2):
python tools/mg_timing.py [n] [levels] [cycle] [solver].
Under `rocprofv3 --kernel-trace --stats` the k_mg_* rows are the transfer kernels."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import aither_amd
from aither_amd.case import synthetic
from aither_amd.solver import MultigridSolver

n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
levels = int(sys.argv[2]) if len(sys.argv) > 2 else 3
cycle = sys.argv[3] if len(sys.argv) > 3 else "V"
solver = sys.argv[4] if len(sys.argv) > 4 else "dplur"
api = aither_amd.load(5)
kw = dict(n=(n, n, n), stretch=1.02, time_integration="implicitEuler", matrix_solver=solver,
          matrix_sweeps=4 if solver == "dplur" else 2, cfl=50.0, amplitude=0.02)
for lev in (1, levels):
    t0 = time.time()
    cases, trs = synthetic.multigrid_levels(levels=lev, cycle=cycle, **kw)
    s = MultigridSolver(api, cases, trs)
    tb = time.time() - t0
    for nn in range(3):
        s.step(nn)
    api.check(api.sync(s.levels[0].ctx), "sync")
    t0 = time.time()
    its = 10
    for nn in range(3, 3 + its):
        out = s.step(nn)
    api.check(api.sync(s.levels[0].ctx), "sync")
    dt = (time.time() - t0) / its
    print(f"{solver} {n}^3 levels {lev} ({cycle}): {dt * 1e3:.2f} ms per iteration, matrix residual {out['matrix']:.3e}, "
          f"L2 {out['norm'][0]:.3e} (set-up {tb:.1f} s)", flush=True)
    s.close()
