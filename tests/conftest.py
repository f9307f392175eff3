import ctypes
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
# the oracle is threaded (OpenMP) for blocks of bench size; a test box may show far more
# cores than it may use, so the tests fix the thread count (read when libgomp loads)
os.environ.setdefault("OMP_NUM_THREADS", "8")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def _oracle_lib():
    path = os.path.join(ROOT, "oracle", "liboracle.so")
    src = os.path.join(ROOT, "oracle", "oracle.c")
    if not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")],
                              stdout=subprocess.DEVNULL)
    return path


@pytest.fixture(scope="session")
def oracle():
    """CPU oracle (test infrastructure): same entry points, prefix ora_."""
    from aither_amd import abi
    return abi.Api(ctypes.CDLL(_oracle_lib()), "ora_")


@pytest.fixture(scope="session")
def agx():
    """The product library; fails (not skips) when it is missing."""
    import aither_amd
    return aither_amd.load()


def golden_case(name):
    from aither_amd.case.builder import build_case
    return build_case(os.path.join(GOLDEN, "cases", name, name + ".inp"))


def golden_solver(api, name):
    """A Solver of a golden case -- a MultigridSolver over its grid levels where the deck
    asks for more than one (transonicBump)."""
    from aither_amd.case import builder, geometry, multigrid
    from aither_amd.case.inputfile import parse_input
    from aither_amd.solver import MultigridSolver, Solver
    base = os.path.join(GOLDEN, "cases", name)
    deck = parse_input(os.path.join(base, name + ".inp"))
    if deck.multigrid_levels == 1:
        return Solver(api, golden_case(name))
    coords = geometry.read_plot3d(os.path.join(base, deck.grid_name + ".xyz"), deck.l_ref)
    cases, transfers = multigrid.build_levels(deck, coords, deck.multigrid_levels,
                                              builder.build_case)
    return MultigridSolver(api, cases, transfers)
