"""Pin the CPU oracle against the reference's own regression truths.

The reference's tests (testCases/regressionTests.py) run each case for a fixed
number of iterations and compare the last line of <case>.resid with hard-coded
normalised L2 residuals to 1 %.  The same inputs (grid + .inp, copied as data
under tests/golden/cases) are run here through the oracle; nine cases
reproduce every printed digit (5 significant figures; transonicBump through the
three-level W cycle of the multigrid driver), couette and
convectingVortex agree to 1e-3 or better, inside the reference's own 1 %
tolerance.
"""
import json
import os

import pytest

from conftest import GOLDEN, golden_case, golden_solver
from aither_amd.solver import Solver

with open(os.path.join(GOLDEN, "regression_truths.json")) as fh:
    TRUTH = {k: v for k, v in json.load(fh).items() if not k.startswith("_")}


@pytest.mark.parametrize("name", sorted(TRUTH))
def test_oracle_reproduces_reference_truth(oracle, name):
    spec = TRUTH[name]
    sol = golden_solver(oracle, name)
    out = sol.run(spec["iterations"])
    got = out["norm"]
    for idx, (g, t) in enumerate(zip(got, spec["truth"])):
        if idx in spec["ignore"]:
            continue
        # the reference's own acceptance test (regressionTests.py:108-112)
        assert abs(g - t) <= 0.01 * t, (name, idx, g, t)
        if spec.get("digits_exact", True):
            assert f"{g:.4e}" == f"{t:.4e}", (name, idx, g, t)
        else:       # couette, convectingVortex: what is actually reached
            assert abs(g - t) <= spec["rtol"] * t, (name, idx, g, t)
    sol.close()


@pytest.mark.parametrize("kind", ["weno_visc_lusgs", "rans_blusgs", "dplur"])
def test_oracle_does_not_depend_on_the_thread_count(oracle, kind):
    """The oracle's loops are threaded over k-planes / hyperplane cells for the bench's
    cpu_baseline on all host cores; every sum is still formed in the serial order, so one
    thread and several give bit-identical states and norms."""
    import ctypes
    import numpy as np
    from aither_amd.case import synthetic
    gomp = ctypes.CDLL("libgomp.so.1")
    wall = {3: ("viscousWall", 2), 1: ("characteristic", 1), 2: ("characteristic", 1),
            4: ("characteristic", 1), 5: ("characteristic", 1), 6: ("characteristic", 1)}
    kw = {
        "weno_visc_lusgs": dict(bcs=wall, equation_set="navierStokes",
                                face_reconstruction="weno", limiter="none",
                                inviscid_flux="ausm", time_integration="implicitEuler",
                                matrix_sweeps=2, cfl=10.0),
        "rans_blusgs": dict(bcs=wall, equation_set="rans", turbulence_model="sst2003",
                            time_integration="implicitEuler", matrix_solver="blusgs",
                            cfl=10.0),
        "dplur": dict(inviscid_flux="ausm", time_integration="implicitEuler",
                      matrix_solver="dplur", matrix_sweeps=3, cfl=5.0),
    }[kind]
    out = []
    os.environ["ORA_OMP_MIN_CELLS"] = "0"      # thread even these small blocks
    for nthreads in (1, 5):
        gomp.omp_set_num_threads(nthreads)
        case = synthetic.single_block_case((13, 11, 9), stretch=1.2, **kw)
        sol = Solver(oracle, case)
        for nn in range(2):
            sol.step(nn)
        out.append((sol.download("state", 0).copy(), np.array(sol.history[-1]["l2"]),
                    sol.history[-1]["matrix"]))
        sol.close()
    os.environ.pop("ORA_OMP_MIN_CELLS")
    gomp.omp_set_num_threads(int(os.environ.get("OMP_NUM_THREADS", "8")))
    assert np.array_equal(out[0][0], out[1][0])
    assert np.array_equal(out[0][1], out[1][1])
    assert out[0][2] == out[1][2]


def test_golden_tree_is_reproduced_by_its_script():
    """tests/golden/make_fixtures.py regenerates every file under tests/golden/cases and
    every transcribed truth from the reference (only where the reference is mounted: it
    does not travel to the GPU box)."""
    import subprocess
    import sys
    if not os.path.isdir("/root/reference/testCases"):
        pytest.skip("the reference is not mounted here")
    out = subprocess.run([sys.executable, os.path.join(GOLDEN, "make_fixtures.py")],
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "12 case directories" in out.stdout and "11 truth vectors" in out.stdout
