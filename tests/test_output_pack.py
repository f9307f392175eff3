"""Device-side output: the payloads of the reference's function file
(WriteFunFile, output.cpp:209-437) and restart file (WriteRestart, :591-755) formed and
re-dimensionalised by the library.  CPU: the oracle's packers against values put together
by hand from the downloaded fields with the reference's scale factors; GPU: the HIP
library against the oracle, variable by variable."""
import numpy as np
import pytest

from aither_amd import abi
from aither_amd.case import synthetic
from aither_amd.solver import Solver

WALL = {3: ("viscousWall", 2), 1: ("characteristic", 1), 2: ("characteristic", 1),
        4: ("characteristic", 1), 5: ("characteristic", 1), 6: ("characteristic", 1)}
LAMINAR = [n for n in abi.OUT if n not in ("tkeGrad_x", "tkeGrad_y", "tkeGrad_z",
                                            "omegaGrad_x", "omegaGrad_y", "omegaGrad_z")]


def _case(kind):
    if kind == "rans":
        return synthetic.single_block_case((11, 9, 7), stretch=1.15, bcs=WALL,
                                           equation_set="rans", turbulence_model="sst2003",
                                           time_integration="bdf2", matrix_solver="lusgs",
                                           nonlinear_iterations=2, dt=2.0e-6, cfl=-1.0)
    return synthetic.single_block_case((12, 9, 7), stretch=1.2, skew=0.01, bcs=WALL,
                                       equation_set="navierStokes", face_reconstruction="weno",
                                       limiter="none", inviscid_flux="ausm",
                                       time_integration="bdf2", matrix_solver="lusgs",
                                       nonlinear_iterations=2, dt=2.0e-6, cfl=-1.0)


def _refs(sol):
    c = sol.cfg.gas
    mu_ref = c.visc_c1 * c.t_ref ** 1.5 / (c.t_ref + c.visc_s)
    return c.rho_ref, c.a_ref, c.l_ref, c.t_ref, mu_ref


def test_oracle_packers_against_hand_made_values(oracle):
    case = _case("laminar")
    s = Solver(oracle, case)
    s.step(0), s.step(1)
    g = case.ng
    rR, aR, lR, tR, muR = _refs(s)
    st = s.download("state", 0)[g:-g, g:-g, g:-g]
    res = s.download("residual", 0)
    got = dict(zip(LAMINAR, s.output_pack(0, LAMINAR)))
    gam = (s.cfg.gas.n + 1.0) / s.cfg.gas.n
    cs = np.sqrt(gam * st[..., 4] / st[..., 0])
    vmag = np.sqrt((st[..., 1:4] ** 2).sum(-1))
    np.testing.assert_allclose(got["density"], st[..., 0] * rR, rtol=1e-15)
    np.testing.assert_allclose(got["vel_y"], st[..., 2] * aR, rtol=1e-15)
    np.testing.assert_allclose(got["pressure"], st[..., 4] * rR * aR * aR, rtol=1e-15)
    np.testing.assert_allclose(got["mach"], vmag / cs, rtol=1e-14)
    np.testing.assert_allclose(got["sos"], cs * aR, rtol=1e-14)
    temp = st[..., 4] / (st[..., 0] * s.cfg.gas.gas_constant)
    np.testing.assert_allclose(got["temperature"], temp * tR, rtol=1e-14)
    e = s.cfg.gas.n * st[..., 4] / st[..., 0] + 0.5 * vmag ** 2
    np.testing.assert_allclose(got["energy"], e * aR * aR, rtol=1e-13)
    np.testing.assert_allclose(got["enthalpy"], (e + st[..., 4] / st[..., 0]) * aR * aR, rtol=1e-13)
    np.testing.assert_allclose(got["cp"], s.cfg.gas.gas_constant * (s.cfg.gas.n + 1) * aR * aR / tR)
    np.testing.assert_allclose(got["dt"], s.download("dt", 0)[..., 0] / (aR * lR), rtol=1e-15)
    np.testing.assert_allclose(got["resid_energy"], res[..., 4] * rR * aR ** 3 * lR * lR, rtol=1e-15)
    np.testing.assert_allclose(got["resid_mass"], res[..., 0] * rR * aR * lR * lR, rtol=1e-15)
    vg = s.download("vel_grad", 0)
    np.testing.assert_allclose(got["velGrad_vx"], vg[..., 1] * aR / lR, rtol=1e-14)     # XY
    np.testing.assert_allclose(got["velGrad_uz"], vg[..., 6] * aR / lR, rtol=1e-14)     # ZX
    np.testing.assert_allclose(got["pressGrad_y"],
                               s.download("press_grad", 0)[..., 1] * rR * aR * aR / lR, rtol=1e-14)
    assert np.all(got["tke"] == 0.0) and np.all(got["viscosityRatio"] == 0.0)
    assert np.all(got["rank"] == 0.0) and np.all(got["wallDistance"] > 0.0)
    # restart payload: cell-major, n_eq + 1 entries, the mass fraction last
    r0, r1 = s.restart_pack(0, 0), s.restart_pack(0, 1)
    assert r0.shape == st.shape[:3] + (6,)
    np.testing.assert_allclose(r0[..., :5], st * np.array([rR, aR, aR, aR, rR * aR * aR]), rtol=1e-15)
    assert np.all(r0[..., 5] == 1.0) and np.all(r1[..., 5] == 1.0)
    nm1 = s.download("cons_nm1", 0)
    np.testing.assert_allclose(r1[..., :5], nm1 * np.array([rR, aR * rR, aR * rR, aR * rR, aR * aR * rR]),
                               rtol=1e-15)
    with pytest.raises(RuntimeError, match="unknown output variable"):
        s.api.check(s.api.output_pack(s.ctx, 0, 1, (abi.C.c_int32 * 1)(99), None), "output_pack")
    s.close()


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["laminar", "rans"])
def test_output_and_restart_pack_parity(kind, oracle):
    import aither_amd
    agx = aither_amd.load(7 if kind == "rans" else 5)
    case = _case(kind)
    sg, so = Solver(agx, case), Solver(oracle, case)
    for nn in range(2):
        sg.step(nn), so.step(nn)
    names = list(abi.OUT) if kind == "rans" else LAMINAR
    a, b = sg.output_pack(0, names), so.output_pack(0, names)
    for n, x, y in zip(names, a, b):
        scale = np.abs(y).max()
        if scale == 0.0:
            assert np.all(x == 0.0), n
            continue
        # gradients and residuals are differences of O(1) quantities: the floor is what
        # parity of the fields they are formed from allows
        tol = 1e-10 if "Grad" not in n and not n.startswith("resid") else 1e-8
        assert np.abs(x - y).max() <= tol * scale, (n, np.abs(x - y).max() / scale)
    for which in (0, 1):
        x, y = sg.restart_pack(0, which), so.restart_pack(0, which)
        assert x.shape == y.shape
        assert np.abs(x - y).max() <= 1e-10 * np.abs(y).max(), which
    # a subset in another order: variable-major, in the caller's order
    sub = ["pressure", "density", "mach"]
    z = sg.output_pack(0, sub)
    for q, n in enumerate(sub):
        assert np.array_equal(z[q], a[names.index(n)])
    sg.close(), so.close()
