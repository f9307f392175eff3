"""Shared helpers for the parity tests (HIP path vs CPU oracle)."""
import numpy as np

from aither_amd.solver import Solver

# fp64 tolerance stated by BASELINE.json north_star: residuals and updated
# state within 1e-10 relative of the CPU reference.
RTOL = 1.0e-10


def rel_err(got, ref):
    """max |got-ref| per last-axis component, relative to that component's
    max |ref| (floored at 1e-3 of the global max so that components that are
    identically ~0, e.g. w in a 2-D case, are judged on the global scale)."""
    got = np.asarray(got, dtype=float)
    ref = np.asarray(ref, dtype=float)
    comp_axes = tuple(range(ref.ndim - 1))
    gmax = np.abs(ref).max()
    scale = np.maximum(np.abs(ref).max(axis=comp_axes), 1.0e-3 * gmax)
    scale = np.where(scale > 0, scale, 1.0)
    return (np.abs(got - ref).max(axis=comp_axes) / scale).max()


def run_pair(agx, oracle, case, steps, fields=("state", "residual", "dt")):
    """Advance `steps` time steps with both backends and compare everything
    that crosses the boundary; returns the two solvers for extra checks."""
    sg, so = Solver(agx, case), Solver(oracle, case)
    for nn in range(steps):
        og, oo = sg.step(nn), so.step(nn)
    # every nonlinear iteration's L2 norms
    assert len(sg.history) == len(so.history)
    for hg, ho in zip(sg.history, so.history):
        e = rel_err(hg["l2"][None, :], ho["l2"][None, :])
        assert e < RTOL, ("L2 residual norm", hg["nn"], hg["mm"], e, hg["l2"], ho["l2"])
        if ho["matrix"] > 0:
            assert abs(hg["matrix"] - ho["matrix"]) <= 1e-8 * ho["matrix"] + 1e-300, \
                ("matrix residual", hg["matrix"], ho["matrix"])
    # L-infinity location and value of the last iteration
    lg, lo = sg.history[-1]["linf"], so.history[-1]["linf"]
    assert abs(lg[0] - lo[0]) <= RTOL * abs(lo[0]) + 1e-300, (lg, lo)
    for gb in sg.block_ids:
        for f in fields:
            a, b = sg.download(f, gb), so.download(f, gb)
            if f == "state":          # corners are never assigned by either
                ng = case.ng
                a = a[ng:-ng, ng:-ng, ng:-ng]
                b = b[ng:-ng, ng:-ng, ng:-ng]
            e = rel_err(a, b)
            assert e < RTOL, (f, gb, e)
    return sg, so
