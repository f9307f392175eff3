"""Shared helpers for the parity tests (HIP path vs CPU oracle)."""
import numpy as np

from aither_amd.solver import Solver

# fp64 tolerance stated by BASELINE.json north_star: residuals and updated
# state within 1e-10 relative of the CPU reference.
RTOL = 1.0e-10
MATRIX_RTOL = 1.0e-6   # cap of the bound run_pair DERIVES per case (matrix_tolerance:
                       # 5e-9 .. 1e-7 for the synthetic decks, cancellation factors 12 .. 270)
MATRIX_FLOOR = 1.0e-14  # below this the matrix residual of a converged / uniform state
                        # is the round-off of O(1) operands


def matrix_tolerance(so):
    """Relative tolerance of the matrix residual, derived: f - (A x - b) is what is LEFT
    after its operands A x, the off-diagonal terms and b cancel, so its error is the 1e-10
    parity of those operands (x, state, residual: asserted field by field in run_pair)
    amplified by |operands| / |remainder|.  The oracle reports both sums of squares of its
    last matrix residual (ora_debug_matrix_operands, a test hook); a factor 4 covers the
    three operands and the square root.  Returns (tolerance, cancellation factor)."""
    import ctypes as C
    fn = so.api.lib.ora_debug_matrix_operands
    fn.restype, fn.argtypes = C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    ops, res = C.c_double(0.0), C.c_double(0.0)
    so.api.check(fn(so.ctx, C.byref(ops), C.byref(res)), "debug_matrix_operands")
    if res.value <= 0.0:
        return MATRIX_RTOL, float("inf")
    amp = (ops.value / res.value) ** 0.5
    return 4.0 * RTOL * amp, amp


def rel_err(got, ref, floor=0.0):
    """max |got-ref| per last-axis component, relative to the largest of
      * that component's max |ref|,
      * 10 % of the whole field's max |ref|  (components that are physically
        zero, e.g. the w-momentum residual of a 2-D case, are pure round-off),
      * `floor`: an absolute scale below which the quantity is round-off of
        the terms it is built from (see flux_scale)."""
    got = np.asarray(got, dtype=float)
    ref = np.asarray(ref, dtype=float)
    comp_axes = tuple(range(ref.ndim - 1))
    gmax = np.abs(ref).max()
    scale = np.maximum(np.abs(ref).max(axis=comp_axes), max(0.1 * gmax, floor))
    scale = np.where(scale > 0, scale, 1.0)
    return (np.abs(got - ref).max(axis=comp_axes) / scale).max()


def flux_scale(case):
    """Magnitude of one face flux: largest face area times rho*c^2 ~ O(1) in the
    reference's nondimensionalisation.  A residual is the sum of six such
    fluxes, so its round-off floor is ~1e-16 * flux_scale; residuals smaller
    than 1e-3 * flux_scale (e.g. the exactly-zero residual of a uniform flow
    at iteration 0) are compared on that scale."""
    amax = 0.0
    for blk in case.blocks:
        g = blk.geom
        for d in "ijk":
            amax = max(amax, float(g.farea[d].a[..., 3].max()))
    return amax


SYNC_FIELDS = ("state", "cons_n", "cons_nm1", "update")


def run_pair(agx, oracle, case, steps, fields=("state", "residual", "dt"),
             resync=True, check_from=0):
    """Advance `steps` time steps with both backends and compare everything
    that crosses the boundary after every step.

    resync=True (default) measures parity "on identical inputs" (BASELINE.json
    north_star): before each step after the first, the oracle's state, time
    levels and update are uploaded into the HIP solver, so each comparison is
    one time step (all its nonlinear iterations) from bit-identical inputs.
    A residual is a small difference of large fluxes, so without the resync
    its error relative to its own (shrinking) magnitude grows like
    |flux| / |residual| times the state error although the state itself stays
    within 1e-13 (see test_free_running_drift).

    check_from: first time step that is compared (earlier ones run, resynchronised
    as usual, but are not asserted on) -- for cases whose first step is
    ill-conditioned in the reference's own formulas, see the caller."""
    sg, so = Solver(agx, case), Solver(oracle, case)
    ng = case.ng
    n_hist = 0
    rfloor = 1.0e-3 * flux_scale(case)
    nfloor = rfloor * np.sqrt(case.total_cells)
    for nn in range(steps):
        if resync and nn > 0:
            for gb in sg.block_ids:
                for f in SYNC_FIELDS:
                    sg.upload(f, gb, so.download(f, gb))
                # (a state upload re-derives what the library keeps from the state at
                # start-up -- the rans viscosity_ of AuxillaryAndWidths, main.cpp:169 --
                # so the oracle gets the same call)
                so.upload("state", gb, so.download("state", gb))
            sg.l2_first = None if so.l2_first is None else so.l2_first.copy()
        sg.step(nn), so.step(nn)
        assert len(sg.history) == len(so.history)
        if nn < check_from:
            n_hist = len(so.history)
            continue
        for hg, ho in zip(sg.history[n_hist:], so.history[n_hist:]):
            e = rel_err(hg["l2"][None, :], ho["l2"][None, :], nfloor)
            assert e < RTOL, ("L2 residual norm", hg["nn"], hg["mm"], e,
                              hg["l2"], ho["l2"])
            if ho["matrix"] > 0:
                # the last nonlinear iteration of the step against the DERIVED bound (the
                # hook describes the oracle's last matrix residual), the earlier ones
                # against its cap
                last = ho is so.history[-1]
                tol, amp = matrix_tolerance(so) if last else (MATRIX_RTOL, None)
                tol = min(tol, MATRIX_RTOL)
                assert abs(hg["matrix"] - ho["matrix"]) <= tol * ho["matrix"] + MATRIX_FLOOR, \
                    ("matrix residual", hg["matrix"], ho["matrix"], tol, amp)
        n_hist = len(so.history)
        lg, lo = sg.history[-1]["linf"], so.history[-1]["linf"]
        assert abs(lg[0] - lo[0]) <= RTOL * max(abs(lo[0]), rfloor), (lg, lo)
        for gb in sg.block_ids:
            for f in fields:
                a, b = sg.download(f, gb), so.download(f, gb)
                if f == "state":      # corners are never assigned by either
                    a = a[ng:-ng, ng:-ng, ng:-ng]
                    b = b[ng:-ng, ng:-ng, ng:-ng]
                e = rel_err(a, b, rfloor if f == "residual" else 0.0)
                assert e < RTOL, (f, gb, nn, e)
    return sg, so
