"""Set-up helpers: plot3dBlock's metrics and the nearest-wall search of
CalcWallDistance.  CPU: the oracle's C restatement against the numpy one of
aither_amd.case.geometry (two independent statements of plot3d.cpp:35-360) and against
scipy's k-d tree; GPU: the HIP library against the oracle, and a whole case built through
the library's helpers against the host-built one."""
import numpy as np
import pytest

from aither_amd.case import geometry, synthetic
from aither_amd.solver import DeviceSetup, Solver

KEYS = ("vol", "center", "farea_i", "farea_j", "farea_k", "fcen_i", "fcen_j", "fcen_k")
WALL = {3: ("viscousWall", 2), 1: ("characteristic", 1), 2: ("characteristic", 1),
        4: ("characteristic", 1)}


def _nodes():
    return synthetic.box_nodes(13, 9, 7, stretch=1.3, skew=0.02)


def test_oracle_metrics_against_numpy_restatement(oracle):
    x = _nodes()
    a, b = DeviceSetup(oracle).metrics(x), geometry.interior_metrics(x)
    for k in KEYS:
        assert a[k].shape == b[k].shape, k
        np.testing.assert_allclose(a[k], b[k], rtol=1e-13, atol=1e-16, err_msg=k)
    # unit normals, positive areas; the closed cell: sum of outward area vectors = 0
    for k in ("farea_i", "farea_j", "farea_k"):
        np.testing.assert_allclose(np.linalg.norm(a[k][..., :3], axis=-1), 1.0, rtol=1e-14)
        assert np.all(a[k][..., 3] > 0.0)
    av = {d: a["farea_" + d][..., :3] * a["farea_" + d][..., 3:] for d in "ijk"}
    closed = (av["i"][:, :, 1:] - av["i"][:, :, :-1]) + (av["j"][:, 1:] - av["j"][:, :-1]) + \
        (av["k"][1:] - av["k"][:-1])
    assert np.abs(closed).max() < 1e-15
    assert abs(a["vol"].sum() - 1.0) < 1e-12            # the unit box
    # a block turned inside out is refused as the reference refuses it
    with pytest.raises(RuntimeError, match="negative volume"):
        DeviceSetup(oracle).metrics(x[:, :, ::-1].copy())


def test_oracle_nearest_wall_against_kdtree(oracle):
    from scipy.spatial import cKDTree
    rng = np.random.default_rng(20261003)
    cells, walls = rng.random((3000, 3)), rng.random((700, 3)) * [1.0, 0.01, 1.0]
    d = DeviceSetup(oracle).nearest(cells, walls)
    np.testing.assert_allclose(d, cKDTree(walls).query(cells)[0], rtol=1e-14)


@pytest.mark.gpu
def test_setup_helpers_parity_and_whole_case(agx, oracle):
    x = _nodes()
    dev, ora = DeviceSetup(agx), DeviceSetup(oracle)
    a, b = dev.metrics(x), ora.metrics(x)
    for k in KEYS:
        scale = np.abs(b[k]).max()
        assert np.abs(a[k] - b[k]).max() <= 1e-13 * scale, k
    rng = np.random.default_rng(7)
    cells, walls = rng.random((5001, 3)), rng.random((777, 3))
    np.testing.assert_allclose(dev.nearest(cells, walls), ora.nearest(cells, walls), rtol=1e-14)
    with pytest.raises(RuntimeError, match="negative volume"):
        dev.metrics(x[:, :, ::-1].copy())
    # a viscous case set up through the library (metrics + wall distance) runs to the same
    # residuals as the host-built one
    kw = dict(stretch=1.2, skew=0.01, bcs=WALL, equation_set="navierStokes",
              time_integration="implicitEuler", matrix_solver="lusgs", cfl=5.0)
    c_dev = synthetic.single_block_case((14, 9, 8), setup=dev, **kw)
    c_host = synthetic.single_block_case((14, 9, 8), **kw)
    g0, g1 = c_dev.blocks[0].geom, c_host.blocks[0].geom
    np.testing.assert_allclose(g0.wall_dist.a, g1.wall_dist.a, rtol=1e-13)
    np.testing.assert_allclose(g0.vol.a, g1.vol.a, rtol=1e-13)
    s0, s1 = Solver(agx, c_dev), Solver(agx, c_host)
    for nn in range(2):
        s0.step(nn), s1.step(nn)
    np.testing.assert_allclose(s0.history[-1]["l2"], s1.history[-1]["l2"], rtol=1e-10)
    s0.close(), s1.close(), dev.close(), ora.close()
