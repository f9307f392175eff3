"""Host side of the geometric multigrid (aither_amd/case/multigrid.py, the cycle driver
aither_amd.solver.MultigridSolver) and the oracle's agx_mg_* counterparts, on the CPU."""
import numpy as np
import pytest

from aither_amd.case import multigrid as mg, synthetic
from aither_amd.case.inputfile import Surface
from aither_amd.solver import MultigridSolver


def _surfs(ni, nj, nk, splits=()):
    s = [Surface("slipWall", 0, 0, 0, nj, 0, nk, 0), Surface("slipWall", ni, ni, 0, nj, 0, nk, 0),
         Surface("slipWall", 0, ni, nj, nj, 0, nk, 0),
         Surface("slipWall", 0, ni, 0, nj, 0, 0, 0), Surface("slipWall", 0, ni, 0, nj, nk, nk, 0)]
    edges = [0] + list(splits) + [ni]
    for a, b in zip(edges[:-1], edges[1:]):       # the lower j side in patches
        s.append(Surface("slipWall", a, b, 0, 0, 0, nk, 0))
    return s


def test_kept_indices_every_other_node_and_every_surface_boundary():
    """procBlock::GetCoarseMeshAndBCs: boundaries of surface patches are always kept, every
    other node between them; the coarse surfaces are renumbered; an odd cell count leaves a
    coarse cell of one fine cell."""
    surfs = _surfs(12, 5, 1, splits=(5,))
    new = [Surface(**vars(s)) for s in surfs]
    keep = mg.kept_indices(13, surfs, new, "i")
    assert keep == [0, 2, 4, 5, 7, 9, 11, 12]
    assert sorted({s.imax for s in new}) == [0, 3, 7] and {s.imin for s in new} == {0, 3, 7}
    new = [Surface(**vars(s)) for s in surfs]
    assert mg.kept_indices(6, surfs, new, "j") == [0, 2, 4, 5]
    assert mg.kept_indices(2, surfs, new, "k") == [0, 1]


def test_transfer_maps_weights_and_coefficients():
    nodes = synthetic.box_nodes(12, 5, 2, 1.15, skew=0.02)
    surfs = _surfs(12, 5, 2, splits=(5,))
    cx, cs, tc, kept = mg.coarsen_block(nodes, surfs)
    assert cx.shape[:3] == (2, 4, 8) and tc.shape == (2, 5, 12, 3)
    assert tc[0, 0, :, 0].tolist() == [0, 0, 1, 1, 2, 3, 3, 4, 4, 5, 5, 6]
    assert tc[0, :, 0, 1].tolist() == [0, 0, 1, 1, 2] and tc[:, 0, 0, 2].tolist() == [0, 0]
    # a coarse node IS the fine node it was kept from
    assert np.array_equal(cx[1, 2, 3], nodes[2, 4, 5])
    from aither_amd.case import geometry
    m = geometry.interior_metrics(nodes)
    vf = mg.volume_weights(m["vol"][..., 0], tc, (1, 3, 7))
    flat = (tc[..., 2] * 3 + tc[..., 1]) * 7 + tc[..., 0]
    sums = np.zeros(21)
    np.add.at(sums, flat.ravel(), vf.ravel())
    assert np.allclose(sums, 1.0, rtol=0, atol=4e-16)
    # the trilinear coefficients locate the fine centre in its coarse cell: interpolating
    # the coarse NODE coordinates with them gives the centre back
    cf = mg.trilinear_coeffs(m["center"], cx, tc)
    assert cf.min() > 0.0 and cf.max() < 1.0
    ci, cj, ck = tc[..., 0], tc[..., 1], tc[..., 2]
    n = lambda di, dj, dk: cx[ck + dk, cj + dj, ci + di]
    lin = lambda a, b, c: (1.0 - c)[..., None] * a + c[..., None] * b
    d04, d15 = lin(n(0, 0, 0), n(0, 0, 1), cf[..., 0]), lin(n(1, 0, 0), n(1, 0, 1), cf[..., 1])
    d26, d37 = lin(n(0, 1, 0), n(0, 1, 1), cf[..., 2]), lin(n(1, 1, 0), n(1, 1, 1), cf[..., 3])
    back = lin(lin(d04, d15, cf[..., 4]), lin(d26, d37, cf[..., 5]), cf[..., 6])
    assert np.abs(back - m["center"]).max() < 2e-3      # (exact on a box, close on a skewed one)


@pytest.mark.parametrize("cycle,nblocks", [("V", 1), ("W", 2)])
def test_oracle_cycle_converges_the_linear_system(oracle, cycle, nblocks):
    """The cycle driver on the oracle: two and three levels, one block and two blocks joined
    by a connection (the coarse levels find their own connections); the matrix residual after
    a cycle is below that of the same number of fine sweeps alone, everything finite."""
    kw = dict(n=(12, 10, 8), nblocks=nblocks, axis="i", stretch=1.1,
              time_integration="implicitEuler", matrix_solver="dplur", matrix_sweeps=4, cfl=40.0)
    cases, trs = synthetic.multigrid_levels(levels=3 if cycle == "W" else 2, cycle=cycle, **kw)
    s = MultigridSolver(oracle, cases, trs)
    out = [s.step(nn) for nn in range(3)]
    base_cases, _ = synthetic.multigrid_levels(levels=1, cycle=cycle, **kw)
    b = MultigridSolver(oracle, base_cases, [])
    ref = [b.step(nn) for nn in range(3)]
    for o, r in zip(out, ref):
        assert np.all(np.isfinite(o["l2"])) and np.isfinite(o["matrix"])
        assert o["matrix"] < r["matrix"]
    assert len(cases[-1].connections) == (1 if nblocks == 2 else 0)
    s.close(), b.close()


RANS_MG = dict(n=(12, 10, 8), stretch=1.15,
               bcs={3: ("viscousWall", 2), 1: ("characteristic", 1), 2: ("characteristic", 1),
                    4: ("characteristic", 1), 5: ("characteristic", 1), 6: ("characteristic", 1)},
               equation_set="rans", turbulence_model="sst2003",
               time_integration="implicitEuler", matrix_sweeps=2, cfl=10.0)


@pytest.mark.parametrize("solver", ["lusgs", "blusgs"])
def test_oracle_cycle_seven_equations(oracle, solver):
    """k-omega SST under the cycle: the two turbulence equations are restricted, forced and
    prolonged like the five flow equations (the reference's transfers are written on varArray);
    their part of the diagonal accumulates over the visits of a coarse level like the flow
    part.  Everything finite, turbulence variables positive, and a different answer from the
    single grid (the coarse levels act on equations 6 and 7 as well)."""
    cases, trs = synthetic.multigrid_levels(levels=2, cycle="V", matrix_solver=solver, **RANS_MG)
    s = MultigridSolver(oracle, cases, trs)
    out = [s.step(nn) for nn in range(3)]
    base_cases, _ = synthetic.multigrid_levels(levels=1, cycle="V", matrix_solver=solver, **RANS_MG)
    b = MultigridSolver(oracle, base_cases, [])
    ref = [b.step(nn) for nn in range(3)]
    ng = cases[0].ng
    st, st1 = s.download("state", 0), b.download("state", 0)
    core = lambda a: a[ng:-ng, ng:-ng, ng:-ng]
    assert np.all(np.isfinite(core(st))) and core(st)[..., 5:].min() > 0.0
    for o in out:
        assert np.all(np.isfinite(o["l2"])) and np.isfinite(o["matrix"])
    assert np.abs(core(st)[..., 5:] - core(st1)[..., 5:]).max() > 0.0
    assert len(out[0]["l2"]) == 7 and len(ref[0]["l2"]) == 7
    s.close(), b.close()
