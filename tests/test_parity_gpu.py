"""Parity of the HIP path (through the C-ABI) with the CPU oracle.

Run on a real MI355X:  python -m pytest tests -m gpu
Tolerance: 1e-10 relative (fp64), stated in parity_utils.RTOL.
"""
import ctypes
import os

import numpy as np
import pytest

from conftest import golden_case
from parity_utils import run_pair, rel_err, RTOL
from aither_amd.case import synthetic
from aither_amd.solver import Solver

pytestmark = pytest.mark.gpu


def _close(*solvers):
    for s in solvers:
        s.close()


# ---- the reference's own regression inputs ---------------------------------
@pytest.mark.parametrize("name,steps", [
    ("supersonicWedge", 20),      # explicit Euler, MUSCL+vanAlbada+Roe
    ("subsonicCylinder", 10),     # LU-SGS, stagnationInlet, pressureOutlet
    ("multiblockCylinder", 10),   # AUSMPW+, characteristic, interblock, CFL ramp
    ("shockTube", 4),             # WENO5, bdf2 + dual time, 5 nonlinear its
    ("viscousFlatPlate", 10),     # Navier-Stokes, viscousWall, LU-SGS
    ("couette", 10),              # periodic, moving isothermal walls
])
def test_golden_case_parity(agx, oracle, name, steps):
    case = golden_case(name)
    _close(*run_pair(agx, oracle, case, steps))


@pytest.mark.parametrize("name,steps", [("subsonicCylinder", 30),
                                        ("shockTube", 6),
                                        ("viscousFlatPlate", 20)])
def test_free_running_drift(agx, oracle, name, steps):
    """No resync: after `steps` time steps the states still agree to 1e-10."""
    case = golden_case(name)
    sg, so = Solver(agx, case), Solver(oracle, case)
    for nn in range(steps):
        sg.step(nn), so.step(nn)
    ng = case.ng
    for gb in sg.block_ids:
        a = sg.download("state", gb)[ng:-ng, ng:-ng, ng:-ng]
        b = so.download("state", gb)[ng:-ng, ng:-ng, ng:-ng]
        assert rel_err(a, b) < RTOL
    _close(sg, so)


@pytest.mark.gpu
@pytest.mark.parametrize("ti", ["rk4", "explicitEuler"])
def test_free_running_explicit_with_time_n(agx, oracle, ti):
    """Explicit steps without resync: exercises the stage-0 launch that also
    forms consVarsN (AssignSolToTimeN folded into the fused kernel) and the
    deferred-store flush on download."""
    kw = dict(n=(70, 9, 8), stretch=1.15, skew=0.01, time_integration=ti, cfl=0.5)
    case = synthetic.single_block_case(**kw)
    sg, so = Solver(agx, case), Solver(oracle, case)
    for nn in range(3):
        sg.step(nn), so.step(nn)
    sg.store_time_n(3), so.store_time_n(3)      # left pending on the GPU side
    ng = case.ng
    for f in ("state", "cons_n", "residual", "dt"):
        a, b = sg.download(f, 0), so.download(f, 0)
        if f == "state":
            a, b = a[ng:-ng, ng:-ng, ng:-ng], b[ng:-ng, ng:-ng, ng:-ng]
        assert rel_err(a, b) < RTOL, f
    _close(sg, so)


@pytest.mark.gpu
@pytest.mark.parametrize("perturb", [0.0, 0.05])
def test_uniformflow_all_orientations_parity(agx, oracle, perturb):
    """Halo exchange over all eight patch orientations (the reference's
    uniformFlow grid, run as an Euler LU-SGS deck with two sweeps): HIP path vs
    oracle, and -- unperturbed -- the uniform stream stays uniform on the GPU."""
    from parity_utils import flux_scale
    case = golden_case("uniformFlow")
    if perturb:
        synthetic.perturbed_state(case, perturb)
    sg, so = run_pair(agx, oracle, case, 3)
    if not perturb:
        for gb in sg.block_ids:
            assert np.abs(sg.download("residual", gb)).max() < 1e-10 * flux_scale(case)
    _close(sg, so)


def _run_with_env(agx, case, steps, env):
    """State after `steps` time steps with the given AGX_* switches (they are
    read when the context is created)."""
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        s = Solver(agx, case)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    for nn in range(steps):
        s.step(nn)
    g = case.ng        # ghost cells hold whatever the last ghost fill left there
    out = s.download("state", 0)[g:-g, g:-g, g:-g]
    s.close()
    return out


@pytest.mark.gpu
def test_kernel_variants_agree(agx):
    """The production tile kernel, the register-window form and the gather form
    are three statements of the same residual."""
    case = synthetic.single_block_case(n=(70, 13, 9), stretch=1.15, skew=0.01,
                                       time_integration="rk4", cfl=0.5)
    ref = _run_with_env(agx, case, 2, {"AGX_KERNEL": "tile"})
    for kind in ("march", "gather"):
        got = _run_with_env(agx, case, 2, {"AGX_KERNEL": kind})
        assert rel_err(got, ref) < 1e-12, kind


@pytest.mark.gpu
def test_viscous_kernel_forms_agree(agx):
    """The LDS-staged viscous kernel (default), the face-once form without staging
    (the fallback of blocks too large for 32-bit plane offsets) and the
    one-thread-per-cell form are three statements of the same viscous residual;
    ragged tile edges in i and j (62 x 6 owned cells per workgroup)."""
    wall = {3: ("viscousWall", 2), 1: ("characteristic", 1),
            2: ("characteristic", 1), 4: ("characteristic", 1)}
    case = synthetic.single_block_case(n=(70, 15, 9), stretch=1.1, skew=0.01, bcs=wall,
                                       equation_set="navierStokes",
                                       time_integration="rk4", cfl=0.3)
    ref = _run_with_env(agx, case, 2, {"AGX_VISC": "tile"})
    for kind in ("march", "gather"):
        got = _run_with_env(agx, case, 2, {"AGX_VISC": kind})
        assert rel_err(got, ref) < 1e-12, kind


@pytest.mark.gpu
@pytest.mark.parametrize("n", [(70, 15, 9), (131, 20, 40)])
def test_viscous_central_fourth_tile_agrees_with_gather(agx, n):
    """viscousFaceReconstruction centralFourth on the LDS-staged kernel (60 x 6 owned cells
    per workgroup: the four-cell face state takes two data lanes on the low side, the rows
    beyond the window come from memory, plane k+2 of the own column is requested a step
    early) against the one-thread-per-cell form; ragged tile edges in i and j, and with the
    larger box workgroups that cross from one column tile into the next.  A reference length
    of 20 um makes the box's Reynolds number a few hundred, so that the four-cell terms are
    five orders above the rounding of the residual."""
    wall = {3: ("viscousWall", 2), 1: ("characteristic", 1),
            2: ("characteristic", 1), 4: ("characteristic", 1)}
    case = synthetic.single_block_case(n=n, stretch=1.1, skew=0.01, bcs=wall,
                                       equation_set="navierStokes",
                                       viscous_face_reconstruction="centralFourth",
                                       time_integration="rk4", cfl=0.3, l_ref=2.0e-5)
    ref = _run_with_env(agx, case, 2, {"AGX_VISC": "gather"})
    got = _run_with_env(agx, case, 2, {"AGX_VISC": "tile"})
    assert rel_err(got, ref) < 1e-12


@pytest.mark.gpu
@pytest.mark.parametrize("sweeps", [1, 2])
def test_lusgs_sweep_forms_agree(agx, sweeps):
    """The pipelined k-plane sweep on the diagonal-ordered arrays (default) and
    the launch-per-hyperplane form on the SoA planes order the same dependency
    graph differently and must give the same update (sweeps = 2 also runs the
    both-triangle branches)."""
    wall = {3: ("viscousWall", 2), 1: ("characteristic", 1),
            2: ("characteristic", 1), 4: ("characteristic", 1)}
    case = synthetic.single_block_case(n=(21, 19, 17), stretch=1.1, bcs=wall,
                                       equation_set="navierStokes",
                                       time_integration="implicitEuler",
                                       matrix_solver="lusgs", matrix_sweeps=sweeps,
                                       cfl=5.0)
    ref = _run_with_env(agx, case, 2, {"AGX_LUSGS": "kp"})
    got = _run_with_env(agx, case, 2, {"AGX_LUSGS": "plane"})
    assert rel_err(got, ref) < 1e-12


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["single", "stacked"])
def test_matrix_residual_forms_agree(agx, kind):
    """The matrix residual marching along k (default: the k-neighbours of a plane position
    stay in registers) and with one plane position per thread (AGX_MRESID=plane) are the
    same sum in the same order per cell; 70 planes = three k-chunks of the marching form, a
    ragged last chunk of plane positions, and -- stacked -- ghost cells across a connection
    as k-neighbours."""
    wall = {3: ("viscousWall", 2), 1: ("characteristic", 1),
            2: ("characteristic", 1), 4: ("characteristic", 1)}
    kw = dict(stretch=1.1, bcs=wall, equation_set="navierStokes",
              time_integration="implicitEuler", matrix_solver="lusgs", matrix_sweeps=2, cfl=5.0)
    if kind == "single":
        case = synthetic.single_block_case(n=(19, 11, 70), **kw)
    else:
        case = synthetic.stacked_blocks_case(n=(13, 9, 35), nblocks=2, axis="k", **kw)
    out = {}
    for form in ("march", "plane"):
        old = os.environ.get("AGX_MRESID")
        os.environ["AGX_MRESID"] = form
        try:
            s = Solver(agx, case)
        finally:
            if old is None:
                os.environ.pop("AGX_MRESID", None)
            else:
                os.environ["AGX_MRESID"] = old
        s.step(0), s.step(1)
        out[form] = [h["matrix"] for h in s.history]
        s.close()
    assert all(m > 0.0 for m in out["march"])
    assert np.allclose(out["march"], out["plane"], rtol=1e-12, atol=0.0), out


@pytest.mark.gpu
def test_lusgs_spin_limit_error_path(agx, oracle):
    """A k-plane that waits longer than AGX_SPIN_LIMIT polls for its predecessor
    raises the error flag, the grid drains, agx_iterate returns the error -- and a
    fresh context works again (the limit is read when the context is created)."""
    case = synthetic.single_block_case(n=(40, 36, 12), stretch=1.1,
                                       time_integration="implicitEuler",
                                       matrix_solver="lusgs", cfl=5.0)
    old = os.environ.get("AGX_SPIN_LIMIT")
    os.environ["AGX_SPIN_LIMIT"] = "1"
    try:
        s = Solver(agx, case)
    finally:
        if old is None:
            os.environ.pop("AGX_SPIN_LIMIT", None)
        else:
            os.environ["AGX_SPIN_LIMIT"] = old
    with pytest.raises(RuntimeError, match="spin limit"):
        for nn in range(20):      # one poll is not always too few: repeat
            s.step(nn)
    s.close()
    _close(*run_pair(agx, oracle, case, 2))


@pytest.mark.gpu
def test_cube_of_blocks_dplur_parity(agx, oracle):
    """BASELINE configs[3] in small: 2 x 2 x 2 blocks (12 interblock connections),
    Euler MUSCL + AUSMPW+, DPLUR with 4 sweeps, all blocks on one GPU."""
    case = synthetic.cube_blocks_case(n=(9, 7, 6), splits=(2, 2, 2), inviscid_flux="ausm",
                                      limiter="none", time_integration="implicitEuler",
                                      matrix_solver="dplur", matrix_sweeps=4, cfl=5.0)
    run_pair(agx, oracle, case, steps=3)


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["dplur", "rk4", "lusgs"])
def test_batched_local_halo_exchange_is_bitwise_the_sequential_one(agx, kind):
    """The local connections exchanged level by level -- every slice of a level in one launch,
    then every insert, a connection in a later level than the earlier ones whose inserts it
    reads or overwrites (the cube's patches border each other: edge ghost cells travel on) --
    against a launch pair per connection in the reference's order (AGX_HALO_BATCH=0; `require`
    makes the set-up fail if no batch forms): 2 x 2 x 2 blocks, twelve connections; DPLUR
    (x and xold change roles every sweep), the fused RK4 stages (the two state buffers change
    roles every stage) and LU-SGS (x in the diagonal-ordered arrays)."""
    kw = {"dplur": dict(inviscid_flux="ausm", limiter="none", time_integration="implicitEuler",
                        matrix_solver="dplur", matrix_sweeps=3, cfl=5.0),
          "rk4": dict(time_integration="rk4", cfl=0.5),
          "lusgs": dict(time_integration="implicitEuler", matrix_solver="lusgs",
                        matrix_sweeps=2, cfl=5.0)}[kind]
    case = synthetic.cube_blocks_case(n=(9, 7, 6), splits=(2, 2, 2), **kw)
    out = {}
    for mode in ("require", "0"):
        old = os.environ.get("AGX_HALO_BATCH")
        os.environ["AGX_HALO_BATCH"] = mode
        try:
            s = Solver(agx, case)
        finally:
            if old is None:
                os.environ.pop("AGX_HALO_BATCH", None)
            else:
                os.environ["AGX_HALO_BATCH"] = old
        for nn in range(2):
            s.step(nn)
        out[mode] = [s.download("state", gb) for gb in range(8)]
        s.close()
    for a, b in zip(out["require"], out["0"]):
        assert np.array_equal(a, b)


@pytest.mark.gpu
def test_cube_of_blocks_equals_single_block_gpu(agx):
    """Explicit RK4 on 2 x 2 x 2 blocks reproduces the single-block state: the
    halo slabs carry exactly what the fused stage kernel reads."""
    from aither_amd.case import builder as _bld
    kw = dict(time_integration="rk4", cfl=0.5)
    c8 = synthetic.cube_blocks_case(n=(40, 7, 6), splits=(2, 2, 2), **kw)
    deck = synthetic.make_deck(**kw)
    deck.bcs = [synthetic.box_surfaces(80, 14, 12, None)]
    c1 = _bld.build_case(None, deck=deck,
                         coords=[synthetic.box_nodes(80, 14, 12, 1.0, lengths=(2.0, 2.0, 2.0))])
    synthetic.perturbed_state(c1, 0.05)
    s8, s1 = Solver(agx, c8), Solver(agx, c1)
    for nn in range(2):
        s8.step(nn), s1.step(nn)
    g = c1.ng
    full = s1.download("state", 0)[g:-g, g:-g, g:-g]
    for bk in range(2):
        for bj in range(2):
            for bi in range(2):
                a = s8.download("state", bi + 2 * (bj + 2 * bk))[g:-g, g:-g, g:-g]
                ref = full[bk * 6:(bk + 1) * 6, bj * 7:(bj + 1) * 7, bi * 40:(bi + 1) * 40]
                assert rel_err(a, ref) < 1e-12
    _close(s8, s1)


# ---- synthetic 3-D cases: every scheme combination on the hot path ----------
SLIP = None
FARFIELD = {s: ("characteristic", 1) for s in range(1, 7)}
WALL_J = {3: ("viscousWall", 2), 1: ("characteristic", 1), 2: ("pressureOutlet", 3),
          4: ("characteristic", 1)}
WALL_HEATFLUX = {3: ("viscousWall", 5), 1: ("characteristic", 1), 2: ("pressureOutlet", 3),
                 4: ("characteristic", 1)}
WALL_ISO = {3: ("viscousWall", 4), 4: ("viscousWall", 2),
            1: ("characteristic", 1), 2: ("characteristic", 1)}
# nonreflecting (LODI) inlet and pressure outlet, ghostStates.cpp:435-462, :614-643
NONREFLECTING = {1: ("inlet", 6), 2: ("pressureOutlet", 7), 3: ("characteristic", 1),
                 4: ("pressureOutlet", 7), 5: ("inlet", 6), 6: ("characteristic", 1)}
NONREFLECTING_WALL = {1: ("inlet", 6), 2: ("pressureOutlet", 7), 3: ("viscousWall", 2),
                      4: ("characteristic", 1)}

CASES = {
    "cfg2_muscl_roe_rk4": dict(n=(14, 12, 10), stretch=1.2, skew=0.01, bcs=SLIP,
                               time_integration="rk4", cfl=0.5),
    "minmod_ausm_euler": dict(n=(12, 10, 9), stretch=1.15, bcs=FARFIELD,
                              limiter="minmod", inviscid_flux="ausm",
                              time_integration="explicitEuler", cfl=0.4),
    "constant_roe": dict(n=(10, 9, 8), face_reconstruction="constant",
                         limiter="none", time_integration="explicitEuler",
                         cfl=0.4, bcs=FARFIELD),
    "upwind_none_rk4": dict(n=(10, 9, 8), stretch=1.1, face_reconstruction="upwind",
                            limiter="none", time_integration="rk4", cfl=0.5),
    "cfg3_weno_ausm_visc_lusgs": dict(
        n=(12, 11, 10), stretch=1.2, bcs=WALL_J, equation_set="navierStokes",
        face_reconstruction="weno", limiter="none", inviscid_flux="ausm",
        time_integration="implicitEuler", matrix_solver="lusgs", cfl=10.0),
    "wenoz_roe_bdf2_dual": dict(
        n=(10, 9, 8), stretch=1.1, face_reconstruction="wenoZ", limiter="none",
        time_integration="bdf2", nonlinear_iterations=3, dt=2.0e-5,
        dual_time_cfl=100.0, matrix_sweeps=2),
    "visc_iso_crank_lusgs": dict(
        n=(10, 9, 8), stretch=1.15, bcs=WALL_ISO, equation_set="navierStokes",
        time_integration="crankNicholson", nonlinear_iterations=2, cfl=20.0,
        limiter="none"),
    "dplur_muscl_ausm": dict(
        n=(11, 10, 9), stretch=1.1, bcs=FARFIELD, inviscid_flux="ausm",
        limiter="none", time_integration="implicitEuler",
        matrix_solver="dplur", matrix_sweeps=4, cfl=50.0),
    "visc_explicit_rk4": dict(
        n=(9, 10, 8), stretch=1.2, bcs=WALL_J, equation_set="navierStokes",
        time_integration="rk4", cfl=0.3),
    # constant-heat-flux wall (ghostStates.cpp:228-242): no reference truth holds
    # this branch, so it is HIP-vs-oracle parity only
    "visc_heatflux_wall_lusgs": dict(
        n=(10, 9, 8), stretch=1.15, bcs=WALL_HEATFLUX, equation_set="navierStokes",
        time_integration="implicitEuler", matrix_solver="lusgs", cfl=10.0),
    # inviscidFluxJacobian: approximateRoe (RoeOffDiagonal fluxJacobian.cpp:240-291)
    "roe_jacobian_lusgs2": dict(
        n=(11, 10, 9), stretch=1.15, bcs=FARFIELD, inv_flux_jac="approximateRoe",
        time_integration="implicitEuler", matrix_solver="lusgs", matrix_sweeps=2, cfl=10.0),
    "roe_jacobian_dplur": dict(
        n=(10, 9, 8), stretch=1.1, inviscid_flux="ausm", inv_flux_jac="approximateRoe",
        time_integration="implicitEuler", matrix_solver="dplur", matrix_sweeps=3, cfl=5.0),
    # viscousFaceReconstruction: centralFourth (FaceReconCentral4th
    # reconstruction.hpp:335-379): four-cell face state and viscosity
    "visc_central4th_lusgs": dict(
        n=(11, 10, 9), stretch=1.15, bcs=WALL_J, equation_set="navierStokes",
        viscous_face_reconstruction="centralFourth",
        time_integration="implicitEuler", matrix_solver="lusgs", cfl=10.0),
    # no reference truth holds these branches (HIP-vs-oracle parity only); they read
    # dt, the state at time n, the last residual's cell gradients and the surface's
    # Mach mean / maximum (procBlock.cpp:2497-2515, :6233-6262)
    "nonreflecting_bdf2_lusgs": dict(
        n=(11, 10, 9), stretch=1.1, skew=0.01, bcs=NONREFLECTING, time_integration="bdf2",
        nonlinear_iterations=2, dt=2.0e-5, dual_time_cfl=100.0, matrix_solver="lusgs"),
    "nonreflecting_rk4": dict(
        n=(10, 9, 8), stretch=1.1, bcs=NONREFLECTING, time_integration="rk4", cfl=0.5),
    "nonreflecting_visc_lusgs": dict(
        n=(10, 9, 8), stretch=1.15, bcs=NONREFLECTING_WALL, equation_set="navierStokes",
        face_reconstruction="weno", limiter="none", inviscid_flux="ausm",
        time_integration="implicitEuler", matrix_solver="lusgs", cfl=10.0),
    # block-matrix solvers (matMultiArray3d, RusanovBlockOffDiagonal, ApproxTSLJacobian):
    # no single-species reference truth runs them; the oracle's Jacobians are pinned
    # by tests/test_block_matrix.py, these are HIP-vs-oracle parity
    "blusgs_muscl_roe": dict(
        n=(11, 10, 9), stretch=1.15, skew=0.01, bcs=FARFIELD, time_integration="implicitEuler",
        matrix_solver="blusgs", cfl=20.0),
    "blusgs_weno_ausm_visc_2sweeps": dict(
        n=(10, 9, 8), stretch=1.2, bcs=WALL_J, equation_set="navierStokes",
        face_reconstruction="weno", limiter="none", inviscid_flux="ausm",
        time_integration="implicitEuler", matrix_solver="blusgs", matrix_sweeps=2, cfl=10.0),
    "bdplur_minmod_bdf2_dual": dict(
        n=(10, 9, 8), stretch=1.1, bcs=FARFIELD, limiter="minmod", time_integration="bdf2",
        nonlinear_iterations=2, dt=2.0e-5, dual_time_cfl=100.0, matrix_solver="bdplur",
        matrix_sweeps=4, matrix_relaxation=1.2),
    "bdplur_visc_iso_wall": dict(
        n=(9, 9, 8), stretch=1.15, bcs=WALL_ISO, equation_set="navierStokes",
        time_integration="implicitEuler", matrix_solver="bdplur", matrix_sweeps=3, cfl=5.0),
    "visc_central4th_weno_rk4": dict(
        n=(9, 10, 8), stretch=1.2, bcs=WALL_ISO, equation_set="navierStokes",
        face_reconstruction="weno", limiter="none",
        viscous_face_reconstruction="centralFourth", time_integration="rk4", cfl=0.3),
}


@pytest.mark.parametrize("name", sorted(CASES))
def test_synthetic_single_block_parity(agx, oracle, name):
    case = synthetic.single_block_case(**CASES[name])
    _close(*run_pair(agx, oracle, case, 3))


@pytest.mark.gpu
def test_stacked_blocks_parity_blusgs(agx, oracle):
    """Block-matrix LU-SGS across interblock connections (inviscid: the ghost update
    of the neighbour block enters RusanovBlockOffDiagonal)."""
    case = synthetic.stacked_blocks_case(n=(8, 7, 5), nblocks=2, axis="j", stretch=1.1,
                                         bcs=FARFIELD, time_integration="implicitEuler",
                                         matrix_solver="blusgs", matrix_sweeps=2, cfl=10.0)
    _close(*run_pair(agx, oracle, case, 2))


@pytest.mark.gpu
@pytest.mark.parametrize("solver,axis", [("blusgs", "i"), ("bdplur", "k")])
def test_stacked_blocks_parity_block_viscous(agx, oracle, solver, axis):
    """Viscous block-matrix solvers across connections: the thin-shear-layer part of
    RusanovBlockOffDiagonal reads the velocity gradient of the cell across the
    connection, exchanged after the residual (gridLevel.cpp:343-368, :386-388)."""
    case = synthetic.stacked_blocks_case(n=(7, 8, 6), nblocks=2, axis=axis, stretch=1.15,
                                         bcs=WALL_J, equation_set="navierStokes",
                                         time_integration="implicitEuler",
                                         matrix_solver=solver, matrix_sweeps=3, cfl=10.0)
    _close(*run_pair(agx, oracle, case, 2))


@pytest.mark.parametrize("axis", ["i", "j", "k"])
def test_stacked_blocks_parity_dplur(agx, oracle, axis):
    """BASELINE config 4 in miniature: interblock connections + DPLUR."""
    case = synthetic.stacked_blocks_case(
        (8, 7, 6), nblocks=3, axis=axis, stretch=1.1, bcs=FARFIELD,
        inviscid_flux="ausm", limiter="none", time_integration="implicitEuler",
        matrix_solver="dplur", matrix_sweeps=4, cfl=20.0)
    _close(*run_pair(agx, oracle, case, 3))


def test_stacked_blocks_parity_lusgs_weno(agx, oracle):
    case = synthetic.stacked_blocks_case(
        (8, 7, 6), nblocks=2, axis="k", face_reconstruction="weno",
        limiter="none", time_integration="implicitEuler", cfl=10.0)
    _close(*run_pair(agx, oracle, case, 3))


# ---- edge cases --------------------------------------------------------------
@pytest.mark.parametrize("n", [(1, 1, 40), (33, 1, 2), (2, 3, 1), (65, 5, 3)])
def test_thin_and_ragged_blocks(agx, oracle, n):
    """Blocks thinner than the ghost depth and sizes that are not multiples of
    the 64x4 thread tile."""
    case = synthetic.single_block_case(n=n, stretch=1.0, bcs=None,
                                       time_integration="rk4", cfl=0.4)
    _close(*run_pair(agx, oracle, case, 2))


def test_download_roundtrip_and_determinism(agx):
    case = synthetic.single_block_case(n=(20, 9, 7), stretch=1.1)
    s1, s2 = Solver(agx, case), Solver(agx, case)
    assert np.array_equal(s1.download("state", 0), case.blocks[0].state)
    for nn in range(2):
        s1.step(nn), s2.step(nn)
    assert np.array_equal(s1.download("state", 0), s2.download("state", 0))
    assert np.array_equal(s1.history[-1]["l2"], s2.history[-1]["l2"])
    _close(s1, s2)


# ---- BASELINE sizes: oracle at 128^3, size-independent properties at 256^3 ----
def test_config2_128cubed_one_rk_step(agx, oracle):
    """configs[1]: single 128^3 block, MUSCL + Roe, RK4 (one full step)."""
    case = synthetic.single_block_case(n=(128, 128, 128), stretch=1.2,
                                       time_integration="rk4", cfl=0.5)
    _close(*run_pair(agx, oracle, case, 1, fields=("state", "residual")))


def test_config3_88cubed_lusgs_iterations(agx, oracle):
    """configs[2] at 88^3: WENO5 + AUSMPW+ + viscous, LU-SGS: 88 pipelined
    k-planes of 175 diagonals each, against the oracle."""
    wall = {3: ("viscousWall", 2), 1: ("characteristic", 1),
            2: ("characteristic", 1), 4: ("characteristic", 1)}
    case = synthetic.single_block_case(n=(88, 88, 88), stretch=1.2, bcs=wall,
                                       equation_set="navierStokes",
                                       face_reconstruction="weno", limiter="none",
                                       inviscid_flux="ausm",
                                       time_integration="implicitEuler",
                                       matrix_solver="lusgs", cfl=10.0)
    _close(*run_pair(agx, oracle, case, 2, fields=("state", "residual")))


def test_256cubed_freestream_and_conservation(agx):
    """Full-size properties that need no oracle:
    (a) a uniform state on a stretched grid with far-field BCs has zero
        residual (free-stream preservation / metric closure);
    (b) with slip walls all round, the mass residual sums to zero over the
        block (every interior face flux appears once with each sign and the
        wall mass flux vanishes)."""
    n = (256, 256, 256)
    far = {s: ("characteristic", 1) for s in range(1, 7)}
    case = synthetic.single_block_case(n=n, stretch=1.2, bcs=far, amplitude=0.0,
                                       time_integration="rk4", cfl=0.5)
    s = Solver(agx, case)
    s.step(0)
    r = s.download("residual", 0)
    st = s.download("state", 0)[2:-2, 2:-2, 2:-2]
    flux_scale = 1.0 / (256 * 256)            # |A| * rho * u ~ 1e-5
    assert np.abs(r).max() < 1e-9 * flux_scale
    assert np.abs(st - case.blocks[0].state[2:-2, 2:-2, 2:-2]).max() < 1e-12
    s.close()
    del r, st
    case = synthetic.single_block_case(n=n, stretch=1.2, bcs=None, amplitude=0.05,
                                       time_integration="rk4", cfl=0.5)
    s = Solver(agx, case)
    s.step(0)
    r = s.download("residual", 0)[..., 0]
    assert abs(r.sum()) < 1e-9 * np.abs(r).sum()
    s.close()


@pytest.mark.gpu
def test_256cubed_implicit_freestream(agx):
    """configs[2]'s implicit machinery at full size without an oracle: a uniform
    state with far-field boundaries is a fixed point of the viscous LU-SGS
    iteration -- residual and update at round-off -- and the single-launch sweep
    gets through its 256 pipelined k-planes (every wait satisfied, no spin-limit
    error).
    MUSCL + Roe here: the reference's WENO5 does not preserve a free stream to
    round-off (its residual is ~1e-8 of a face flux on any grid, oracle and HIP
    alike), so it cannot carry this property."""
    far = {s: ("characteristic", 1) for s in range(1, 7)}
    case = synthetic.single_block_case(n=(256, 256, 256), stretch=1.2, bcs=far, amplitude=0.0,
                                       equation_set="navierStokes", limiter="none",
                                       time_integration="implicitEuler",
                                       matrix_solver="lusgs", cfl=10.0)
    s = Solver(agx, case)
    s.step(0)
    s.step(1)
    g = case.ng
    st = s.download("state", 0)[g:-g, g:-g, g:-g]
    ref = case.blocks[0].state[g:-g, g:-g, g:-g]
    assert np.all(np.isfinite(st))
    assert np.abs(st - ref).max() < 1e-11 * np.abs(ref).max()
    s.close()


@pytest.mark.gpu
def test_temperature_viscosity_fields(agx, oracle):
    """AGX_FIELD_TEMPERATURE / VISCOSITY (temperature_, viscosity_ of
    UpdateAuxillaryVariables procBlock.cpp:6171): formed on demand from the state
    the device holds; compared with the oracle's arrays for the same state."""
    wall = {3: ("viscousWall", 2), 1: ("characteristic", 1),
            2: ("characteristic", 1), 4: ("characteristic", 1)}
    case = synthetic.single_block_case(n=(13, 11, 9), stretch=1.1, bcs=wall,
                                       equation_set="navierStokes",
                                       time_integration="implicitEuler",
                                       matrix_solver="lusgs", cfl=5.0)
    sg, so = Solver(agx, case), Solver(oracle, case)
    sg.step(0), so.step(0)
    # the oracle's aux arrays belong to the state its last residual saw: give the
    # device exactly that state (ghost cells included)
    so2 = Solver(oracle, case)
    oracle.check(oracle.phase_bc_faces(so2.ctx)); oracle.check(oracle.phase_bc_edges(so2.ctx))
    oracle.check(oracle.phase_residual(so2.ctx, 0, case.deck.cfl(0)))
    sg.upload("state", 0, so2.download("state", 0))
    g = case.ng
    for f in ("temperature", "viscosity"):
        a, b = sg.download(f, 0), so2.download(f, 0)
        inner = (slice(g, -g),) * 3
        assert rel_err(a[inner], b[inner]) < RTOL, f
        # ghost layers except the block's corner lines, which nobody assigns
        assert rel_err(a[g:-g, g:-g, :], b[g:-g, g:-g, :]) < RTOL, f
    # cell-centre gradients (velocityGrad_ ... pressureGrad_), on demand from that state
    for f in ("vel_grad", "temp_grad", "dens_grad", "press_grad"):
        a, b = sg.download(f, 0), so2.download(f, 0)
        assert np.abs(b).max() > 0 and rel_err(a, b) < RTOL, f
    _close(sg, so, so2)


@pytest.mark.gpu
def test_config3_256cubed_real_scheme_fast_vs_simple_forms(agx):
    """BASELINE configs[2] at FULL size with its own scheme (WENO5 + AUSMPW+ +
    viscous fluxes, scalar LU-SGS): two iterations on the production kernels
    (LDS-tiled inviscid and viscous residual, pipelined k-plane sweeps on the
    diagonal-ordered arrays) against the same library's simple forms (one thread
    per cell and six faces, one launch per hyperplane on the SoA planes), which
    are the forms the oracle parity tests pin at small sizes.  Also: everything
    stays finite and no k-plane runs into the spin limit (iterate returns 0)."""
    wall = {3: ("viscousWall", 2), 1: ("characteristic", 1),
            2: ("characteristic", 1), 4: ("characteristic", 1)}
    case = synthetic.single_block_case(n=(256, 256, 256), stretch=1.2, bcs=wall,
                                       amplitude=0.05, equation_set="navierStokes",
                                       face_reconstruction="weno", limiter="none",
                                       inviscid_flux="ausm",
                                       time_integration="implicitEuler",
                                       matrix_solver="lusgs", cfl=10.0)
    fast = _run_with_env(agx, case, 2, {})
    assert np.all(np.isfinite(fast))
    simple = _run_with_env(agx, case, 2, {"AGX_LUSGS": "plane", "AGX_VISC": "gather",
                                           "AGX_KERNEL": "gather"})
    assert rel_err(fast, simple) < 1e-11


@pytest.mark.gpu
def test_config4_eight_128cubed_blocks(agx):
    """BASELINE configs[3] at FULL size on one GPU: 2 x 2 x 2 blocks of 128^3.
    (a) one explicit RK4 step on the eight blocks reproduces the single 256^3
    block (the 12 halo slabs carry exactly what the stage kernel reads);
    (b) its own scheme -- Euler MUSCL + AUSMPW+, DPLUR with 4 sweeps -- runs an
    iteration that stays finite and changes the state by O(dt)."""
    from aither_amd.case import builder as _bld
    kw = dict(time_integration="rk4", cfl=0.5)
    c8 = synthetic.cube_blocks_case(n=(128, 128, 128), splits=(2, 2, 2), **kw)
    s8 = Solver(agx, c8)
    s8.step(0)
    g = c8.ng
    parts = [s8.download("state", b)[g:-g, g:-g, g:-g] for b in range(8)]
    s8.close()
    deck = synthetic.make_deck(**kw)
    deck.bcs = [synthetic.box_surfaces(256, 256, 256, None)]
    c1 = _bld.build_case(None, deck=deck,
                         coords=[synthetic.box_nodes(256, 256, 256, 1.0, lengths=(2.0, 2.0, 2.0))])
    synthetic.perturbed_state(c1, 0.05)
    s1 = Solver(agx, c1)
    s1.step(0)
    full = s1.download("state", 0)[g:-g, g:-g, g:-g]
    s1.close()
    del c1
    for bk in range(2):
        for bj in range(2):
            for bi in range(2):
                ref = full[bk * 128:(bk + 1) * 128, bj * 128:(bj + 1) * 128,
                           bi * 128:(bi + 1) * 128]
                assert rel_err(parts[bi + 2 * (bj + 2 * bk)], ref) < 1e-12
    del full, parts
    c8 = synthetic.cube_blocks_case(n=(128, 128, 128), splits=(2, 2, 2),
                                    inviscid_flux="ausm", time_integration="implicitEuler",
                                    matrix_solver="dplur", matrix_sweeps=4, cfl=10.0)
    s8 = Solver(agx, c8)
    out = s8.step(0)
    assert np.all(np.isfinite(out["l2"])) and np.isfinite(out["matrix"])
    for b in (0, 7):
        st = s8.download("state", b)[g:-g, g:-g, g:-g]
        ini = c8.blocks[b].state[g:-g, g:-g, g:-g]
        assert np.all(np.isfinite(st))
        d = np.abs(st - ini).max() / np.abs(ini).max()
        assert 0.0 < d < 0.2
    s8.close()



@pytest.mark.gpu
def test_convecting_vortex_nonreflecting_parity_and_truth(agx, oracle):
    """The reference's convectingVortex case (bdf2, 10 nonlinear iterations with dual
    time stepping, non-reflecting inlet and pressure outlet, a periodic pair, initial
    state from a point cloud): HIP vs oracle per nonlinear iteration, and the HIP library
    alone within 1e-3 of the reference's truth after 100 time steps
    (regressionTests.py:508-509; the oracle reaches the same 6e-5..9e-4).

    The FIRST time step is not compared: in its first iteration dt is still zero, the
    non-reflecting outlet returns the interior state, and FaceReconMUSCL without a
    limiter (reconstruction.hpp:131-153) multiplies the upwind difference -- exactly 0,
    or 3e-16 after the conserved -> primitive round trip of the state at time n -- by
    r = (EPS + dw) / (EPS + uw) ~ 1e24: the downwind term is kept or dropped as a whole
    depending on the last bit of the ghost state.  The two implementations round that
    round trip differently (one flux at the outlet differs by 1e-6 of its size); any
    build of the reference would, too.  From the second step on dt > 0 separates ghost
    and interior state and the comparison is the usual 1e-10."""
    import json
    from conftest import GOLDEN
    case = golden_case("convectingVortex")
    _close(*run_pair(agx, oracle, case, 4, fields=("state", "residual", "dt"), check_from=1))
    with open(os.path.join(GOLDEN, "regression_truths.json")) as fh:
        spec = json.load(fh)["convectingVortex"]
    sol = Solver(agx, golden_case("convectingVortex"))
    out = sol.run(spec["iterations"])
    for idx, (got, t) in enumerate(zip(out["norm"], spec["truth"])):
        if idx in spec["ignore"]:
            continue
        assert abs(got - t) <= spec["rtol"] * t, (idx, got, t)
    sol.close()


# ---- rans: k-omega SST 2003, 7 equations (libaither_gfx950_rans.so) ------------------
@pytest.fixture(scope="module")
def agx_rans():
    import aither_amd
    return aither_amd.load(7)


@pytest.mark.gpu
def test_rae2822_rans_parity(agx_rans, oracle):
    """The reference's rae2822 case (SST 2003, scalar LU-SGS, C-grid cut as a
    connection of the block with itself, adiabatic wall, characteristic farfield):
    HIP vs the oracle that reproduces the reference's truth digits, every iteration
    from identical inputs."""
    case = golden_case("rae2822")
    sg, so = run_pair(agx_rans, oracle, case, 4, fields=("state", "residual", "dt"))
    # k and omega are orders of magnitude away from the flow variables: every
    # component against its OWN scale as well
    g = case.ng
    a = sg.download("state", 0)[g:-g, g:-g, g:-g]
    b = so.download("state", 0)[g:-g, g:-g, g:-g]
    for e in range(7):
        scale = np.abs(b[..., e]).max() or 1.0
        assert np.abs(a[..., e] - b[..., e]).max() <= 1e-10 * scale, e
    _close(sg, so)


@pytest.mark.gpu
def test_rae2822_gpu_reproduces_reference_truth(agx_rans):
    """The HIP path alone, 20 iterations free-running: the normalised residuals of
    all seven equations equal the reference's own regression truth for this case
    (testCases/regressionTests.py:401-403, 1 process) to the printed digits."""
    import json
    from conftest import GOLDEN
    with open(os.path.join(GOLDEN, "regression_truths.json")) as fh:
        spec = json.load(fh)["rae2822"]
    case = golden_case("rae2822")
    sol = Solver(agx_rans, case)
    out = sol.run(spec["iterations"])
    for idx, (got, t) in enumerate(zip(out["norm"], spec["truth"])):
        if idx in spec["ignore"]:
            continue
        assert f"{got:.4e}" == f"{t:.4e}", (idx, got, t)
    sol.close()


@pytest.mark.gpu
def test_turbflatplate_wilcox_parity_and_truth(agx_rans, oracle):
    """The reference's turbFlatPlate case (k-omega Wilcox 2006, stagnation inlet,
    pressure outlet, viscous + slip wall on one surface): HIP vs oracle per iteration,
    and the HIP library alone reproduces the reference's truth digits after 20
    iterations (regressionTests.py:378-380)."""
    import json
    from conftest import GOLDEN
    case = golden_case("turbFlatPlate")
    _close(*run_pair(agx_rans, oracle, case, 3, fields=("state", "residual", "dt")))
    with open(os.path.join(GOLDEN, "regression_truths.json")) as fh:
        spec = json.load(fh)["turbFlatPlate"]
    sol = Solver(agx_rans, golden_case("turbFlatPlate"))
    out = sol.run(spec["iterations"])
    for idx, (got, t) in enumerate(zip(out["norm"], spec["truth"])):
        if idx in spec["ignore"]:
            continue
        assert f"{got:.4e}" == f"{t:.4e}", (idx, got, t)
    sol.close()


@pytest.mark.gpu
def test_walllaw_parity_and_truth(agx_rans, oracle):
    """The reference's wallLaw case (SST 2003, wall functions on the plate, BLU-SGS,
    two blocks): HIP vs oracle per iteration -- the wall data come from a Ridder root
    of the White-Christoph profile evaluated on the device -- and the HIP library
    alone reproduces the reference's truth digits after 20 iterations
    (regressionTests.py:443-445)."""
    import json
    from conftest import GOLDEN
    case = golden_case("wallLaw")
    _close(*run_pair(agx_rans, oracle, case, 3, fields=("state", "residual", "dt")))
    with open(os.path.join(GOLDEN, "regression_truths.json")) as fh:
        spec = json.load(fh)["wallLaw"]
    sol = Solver(agx_rans, golden_case("wallLaw"))
    out = sol.run(spec["iterations"])
    for idx, (got, t) in enumerate(zip(out["norm"], spec["truth"])):
        if idx in spec["ignore"]:
            continue
        assert f"{got:.4e}" == f"{t:.4e}", (idx, got, t)
    sol.close()


@pytest.mark.gpu
def test_walllaw_refused_by_the_five_equation_library(agx):
    """wallTreatment=wallLaw is a rans feature: the 5-equation library refuses the
    surface instead of ignoring the field."""
    from aither_amd import abi
    surf = (abi.BcSurface * 1)()
    surf[0].bc_type = abi.BC["viscousWall"]
    surf[0].imin = surf[0].imax = 0
    surf[0].jmax = surf[0].kmax = 2
    surf[0].state.is_wall_law = 1
    case = golden_case("uniformFlow")
    sol = Solver(agx, case)
    rc = agx.block_set_bcs(sol.ctx, 0, 1, surf)
    assert rc != 0
    assert b"wallLaw" in agx.last_error()
    sol.close()


@pytest.mark.gpu
def test_plane_sweep_forms_agree_bitwise(agx_rans):
    """The hyperplane sweeps of the 7-equation / block-matrix builds: ONE launch per half
    sweep with a workgroup per k-plane, pipelined (k_lusgs_pipe, default), and with
    AGX_SWEEP_PIPE=0 a launch per hyperplane -- all blocks of a step in one launch reading
    cell-major records with three lanes per cell, one lane per cell (AGX_SWEEP_THREE=0), one graph per block on branch
    streams (AGX_SWEEP_ALL=0), plane-major loads (AGX_SWEEP_RECORDS=0) and launches
    without graphs (AGX_GRAPHS=0) are the same arithmetic in a different order of memory
    accesses: bit-identical states on the reference's wallLaw case
    (two blocks, BLU-SGS with four sweeps, wall functions)."""
    ref = None
    off = {"AGX_SWEEP_PIPE": "0"}
    for env in ({}, off, {**off, "AGX_SWEEP_ALL": "0"}, {"AGX_SWEEP_RECORDS": "0"},
                {**off, "AGX_GRAPHS": "0"}, {**off, "AGX_SWEEP_THREE": "0"},
                {**off, "AGX_SWEEP_THREE": "0", "AGX_SWEEP_ALL": "0"},
                {"AGX_SWEEP_ALL": "0", "AGX_SWEEP_RECORDS": "0"}):
        got = _run_with_env(agx_rans, golden_case("wallLaw"), 3, env)
        if ref is None:
            ref = got
        assert np.array_equal(got, ref), env


@pytest.mark.gpu
def test_pipelined_sweep_spin_limit_error_path_and_recovery(agx_rans):
    """k_lusgs_pipe: a k-plane that polls longer than AGX_SPIN_LIMIT for the plane below
    raises the error flag, every workgroup leaves, agx_iterate returns the error; the SAME
    context then sets its pipeline up afresh -- with a sane limit a fresh context gives the
    states of the launch-per-hyperplane form (wallLaw: two blocks, BLU-SGS)."""
    old = os.environ.get("AGX_SPIN_LIMIT")
    os.environ["AGX_SPIN_LIMIT"] = "1"
    try:
        s = Solver(agx_rans, golden_case("wallLaw"))
    finally:
        if old is None:
            os.environ.pop("AGX_SPIN_LIMIT", None)
        else:
            os.environ["AGX_SPIN_LIMIT"] = old
    with pytest.raises(RuntimeError, match="spin limit"):
        for nn in range(20):
            s.step(nn)
    s.close()
    ref = _run_with_env(agx_rans, golden_case("wallLaw"), 2, {"AGX_SWEEP_PIPE": "0"})
    got = _run_with_env(agx_rans, golden_case("wallLaw"), 2, {})
    assert np.array_equal(got, ref)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["wallLaw", "rae2822", "turbFlatPlate"])
def test_rans_viscous_forms_agree(agx_rans, name):
    """The rans viscous residual face-once (k_rans_faces<D> + k_rans_cells, production)
    against the one-thread-per-cell gather form (AGX_VISC=gather), which evaluates every
    face twice: the same device functions, so the states agree to round-off (a product that
    passes through memory is rounded where the gather form may contract it into an fma) --
    on the reference's cases: wall functions + BLU-SGS on two blocks, SST + LU-SGS, Wilcox."""
    ref = _run_with_env(agx_rans, golden_case(name), 3, {"AGX_VISC": "gather"})
    got = _run_with_env(agx_rans, golden_case(name), 3, {})
    assert rel_err(got, ref) < 1e-12


@pytest.mark.gpu
def test_rans4_at_bench_size_against_the_simple_forms(agx_rans):
    """BASELINE configs[4] at the size `bench.py --workload rans4` times (4 blocks of
    128 x 128 x 64, k-omega SST 2003, BLU-SGS, the flat-plate start): the production sweep
    form (one pipelined launch per half sweep, 256 k-planes side by side) against the
    launch-per-hyperplane ones -- three lanes per cell, one lane per cell, and plane-major
    loads -- bit for bit over four iterations;
    every norm finite, the run repeatable."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    import bench
    from aither_amd.solver import DeviceSetup
    setup = DeviceSetup(agx_rans)
    case = bench.rank_local_chain_case(0, 1, 256, "rans4", setup=setup)
    setup.close()
    assert case.n_eq == 7 and case.total_cells == 4 * 128 * 128 * 64

    def run(env, steps):
        old = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        try:
            s = Solver(agx_rans, case)
        finally:
            for k, v in old.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
        for nn in range(steps):
            s.step(nn)
        hist = np.array([np.append(h["l2"], h["matrix"]) for h in s.history])
        g = case.ng
        out = [s.download("state", gb)[g:-g, g:-g, g:-g] for gb in range(4)]
        s.close()
        return hist, out

    h0, s0 = run({}, 4)
    assert np.all(np.isfinite(h0)) and np.all(h0[:, -1] > 0.0)
    for st in s0:
        assert np.all(np.isfinite(st)) and np.all(st[..., 0] > 0.0) and np.all(st[..., 4] > 0.0)
    h1, s1 = run({"AGX_SWEEP_PIPE": "0"}, 4)
    assert np.array_equal(h0, h1)
    for a, b in zip(s0, s1):
        assert np.array_equal(a, b)
    h1, s1 = run({"AGX_SWEEP_PIPE": "0", "AGX_SWEEP_THREE": "0"}, 2)
    assert np.array_equal(h0[:2], h1)
    h2, s2 = run({"AGX_SWEEP_RECORDS": "0", "AGX_SWEEP_ALL": "0"}, 2)
    assert np.array_equal(h0[:2], h2)
    h3, _ = run({}, 4)
    assert np.array_equal(h0, h3)


@pytest.mark.gpu
@pytest.mark.parametrize("tag,solver", [(2, "lusgs"), (4, "lusgs"), (5, "lusgs"),
                                        (4, "blusgs"), (5, "blusgs")])
def test_rans_wall_function_variants_parity(agx_rans, oracle, tag, solver):
    """Wall functions on an adiabatic (tag 2), an isothermal moving (4) and a constant
    heat flux wall (5): wallLaw::AdiabaticBCs / IsothermalBCs / HeatFluxBCs solved per wall
    face on the device, the ghost density each implies, the wall-law flux.  (Only the
    adiabatic law has a reference truth -- wallLaw; the other two are HIP vs oracle.)"""
    wall = {3: ("viscousWall", tag), 1: ("characteristic", 1), 2: ("characteristic", 1),
            4: ("characteristic", 1), 5: ("characteristic", 1), 6: ("characteristic", 1)}
    case = synthetic.single_block_case(n=(9, 8, 7), stretch=1.2, bcs=wall, equation_set="rans",
                                       turbulence_model="sst2003", matrix_solver=solver,
                                       time_integration="implicitEuler", cfl=10.0,
                                       wall_treatment="wallLaw")
    _close(*run_pair(agx_rans, oracle, case, 3))


@pytest.mark.gpu
@pytest.mark.parametrize("lib", ["agx", "rans"])
def test_user_stream_and_stream_change(agx, agx_rans, lib):
    """agx_ctx_set_stream: the whole iteration on a stream of the caller (given at set-up),
    and a change of stream between two iterations, give the states of the default stream
    bit for bit (5-equation LU-SGS on the diagonal-ordered path; wallLaw on the
    hyperplane graphs)."""
    # the HIP runtime the libraries are linked to -- the copy already mapped into this
    # process (a second copy loaded by name would be a second runtime with its own devices)
    # (torch, when imported, maps a private copy of its own beside it)
    with open("/proc/self/maps") as f:
        paths = sorted({ln.split()[-1] for ln in f if "libamdhip64.so" in ln},
                       key=lambda p: "/torch/" in p)
    assert paths, "no HIP runtime mapped"
    hip = ctypes.CDLL(paths[0])

    def new_stream():
        st = ctypes.c_void_p()
        assert hip.hipStreamCreate(ctypes.byref(st)) == 0
        return st
    api = agx if lib == "agx" else agx_rans
    if lib == "agx":
        wall = {3: ("viscousWall", 2), 1: ("characteristic", 1),
                2: ("characteristic", 1), 4: ("characteristic", 1)}
        make = lambda: synthetic.single_block_case(n=(21, 19, 17), stretch=1.1, bcs=wall,
                                                   equation_set="navierStokes",
                                                   time_integration="implicitEuler",
                                                   matrix_solver="lusgs", cfl=5.0)
    else:
        make = lambda: golden_case("wallLaw")
    g = make().ng
    ref = Solver(api, make())
    for nn in range(3):
        ref.step(nn)
    want = ref.download("state", 0)[g:-g, g:-g, g:-g]
    ref.close()
    s1, s2 = new_stream(), new_stream()
    sol = Solver(api, make(), stream=s1.value)
    sol.step(0)
    api.check(api.ctx_set_stream(sol.ctx, s2), "set_stream")
    sol.step(1)
    api.check(api.ctx_set_stream(sol.ctx, ctypes.c_void_p(0)), "set_stream")
    sol.step(2)
    got = sol.download("state", 0)[g:-g, g:-g, g:-g]
    sol.close()
    for st in (s1, s2):
        hip.hipStreamDestroy(st)
    assert np.array_equal(got, want)


RANS_WALL = {3: ("viscousWall", 2), 1: ("characteristic", 1), 2: ("characteristic", 1),
             4: ("characteristic", 1), 5: ("characteristic", 1), 6: ("characteristic", 1)}


@pytest.mark.gpu
@pytest.mark.parametrize("kw", [
    dict(matrix_solver="lusgs", matrix_sweeps=2, inviscid_flux="roe"),
    dict(matrix_solver="dplur", matrix_sweeps=3, inviscid_flux="ausm", limiter="minmod"),
    dict(matrix_solver="lusgs", time_integration="bdf2", nonlinear_iterations=2, dt=2.0e-5,
         dual_time_cfl=100.0, face_reconstruction="weno", limiter="none"),
    # BASELINE configs[4]'s solver: BLU-SGS on the 7-equation set (5 x 5 flow block with
    # eddy viscosity + diagonal turbulence block); and its point-Jacobi twin
    dict(matrix_solver="blusgs", matrix_sweeps=2),
    dict(matrix_solver="bdplur", matrix_sweeps=3, inviscid_flux="ausm"),
    # k-omega Wilcox 2006 (stress-limited eddy viscosity, unlimited one in the k / omega
    # diffusion, vortex-stretching beta, cross diffusion switch), scalar and block
    dict(turbulence_model="kOmegaWilcox2006", matrix_solver="lusgs", matrix_sweeps=2),
    dict(turbulence_model="kOmegaWilcox2006", matrix_solver="blusgs", inviscid_flux="ausm"),
    # SST-DES (turbSstDes: the k destruction scaled by phi, the reference's source spectral
    # radius with the cell width in phi's place), scalar and block
    # (farfield turbulence chosen so that phi > 1 in most of the box: Lt ~ 0.3 of its size)
    dict(turbulence_model="sstdes", matrix_solver="lusgs", matrix_sweeps=2, turbulence=(0.2, 2.4e4)),
    dict(turbulence_model="sstdes", matrix_solver="blusgs", turbulence=(0.2, 2.4e4)),
])
def test_rans_synthetic_parity(agx_rans, oracle, kw):
    """3-D boxes with a viscous wall: both flux functions, MUSCL and WENO, LU-SGS and
    DPLUR, dual time stepping."""
    deck = dict(n=(9, 8, 7), stretch=1.2, bcs=RANS_WALL, equation_set="rans",
                turbulence_model="sst2003", time_integration="implicitEuler", cfl=10.0)
    deck.update(kw)
    case = synthetic.single_block_case(**deck)
    _close(*run_pair(agx_rans, oracle, case, 3))


@pytest.mark.gpu
@pytest.mark.parametrize("n,solver", [((6, 5, 700), "blusgs"), ((6, 5, 700), "lusgs"),
                                      ((180, 172, 3), "blusgs")])
def test_pipelined_sweep_shapes(agx_rans, oracle, n, solver):
    """k_lusgs_pipe outside the shape it was tuned on: 700 k-planes (more workgroups than the
    GPU holds at once: planes wait for tickets, a plane's predecessor always has a lower one)
    and diagonals of 172 cells (more than seven waves of 21: the cells of a step in two
    passes) -- parity with the oracle, two sweeps (both triangles), block and scalar."""
    case = synthetic.single_block_case(n=n, stretch=1.05, bcs=RANS_WALL, equation_set="rans",
                                       turbulence_model="sst2003",
                                       time_integration="implicitEuler", cfl=10.0,
                                       matrix_solver=solver, matrix_sweeps=2)
    _close(*run_pair(agx_rans, oracle, case, 2))


@pytest.mark.gpu
@pytest.mark.parametrize("vel", [(50.0, 20.0, 10.0), (420.0, 20.0, 10.0)])
def test_rans_inlet_and_supersonic_boundaries_parity(agx_rans, oracle, vel):
    """rans ghost states of inlet (subsonic: characteristics both ways; supersonic: the free
    stream), supersonicInflow and supersonicOutflow with their farfield turbulence
    (ghostStates.cpp:392-533), beside a viscous wall, a characteristic far field and a
    pressure outlet -- at a subsonic and a supersonic free stream."""
    bcs = {1: ("supersonicInflow", 8), 2: ("supersonicOutflow", 9), 3: ("viscousWall", 2),
           4: ("characteristic", 1), 5: ("inlet", 10), 6: ("pressureOutlet", 3)}
    case = synthetic.single_block_case(n=(9, 8, 7), stretch=1.2, bcs=bcs, equation_set="rans",
                                       turbulence_model="sst2003", velocity=list(vel),
                                       time_integration="implicitEuler", cfl=5.0)
    _close(*run_pair(agx_rans, oracle, case, 3))


@pytest.mark.gpu
@pytest.mark.parametrize("solver", ["lusgs", "blusgs"])
def test_rans_stacked_blocks_parity(agx_rans, oracle, solver):
    """rans across interblock connections: the ghost eddy viscosity and blending
    function (and, with BLU-SGS, the velocity gradients) of the off-diagonal terms
    come from the neighbour block -- configs[4] in miniature."""
    case = synthetic.stacked_blocks_case(n=(7, 8, 6), nblocks=2, axis="i", stretch=1.15,
                                         bcs=RANS_WALL, equation_set="rans",
                                         turbulence_model="sst2003",
                                         time_integration="implicitEuler",
                                         matrix_solver=solver, matrix_sweeps=2, cfl=10.0)
    _close(*run_pair(agx_rans, oracle, case, 2))


@pytest.mark.gpu
def test_rans4_config_parity(agx_rans, oracle):
    """BASELINE configs[4] in kind and in small: four blocks in a row, k-omega SST 2003,
    BLU-SGS, viscous wall (the case `bench.py --workload rans4` times)."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    import bench
    case = bench.rank_local_chain_case(0, 1, 24, "rans4")
    assert case.n_eq == 7 and len(case.blocks) == 4
    _close(*run_pair(agx_rans, oracle, case, 2))


@pytest.mark.gpu
def test_transonic_bump_multigrid_parity_and_truth(agx, oracle):
    """Geometric multigrid: the reference's transonicBump -- Euler, DPLUR with
    four sweeps, three grid levels, W cycle -- through the library's agx_mg_* calls and the
    host cycle driver (aither_amd.solver.MultigridSolver).  Against the oracle driven by
    the same driver, iteration by iteration: residual norms, matrix residual, the finest
    level's state and the coarse levels' restricted states; free-running for the
    reference's 100 iterations the HIP path alone reproduces the truth
    (regressionTests.py:333-334) to the printed digits."""
    import json
    from conftest import GOLDEN, golden_solver
    sg, so = golden_solver(agx, "transonicBump"), golden_solver(oracle, "transonicBump")
    g = sg.case.ng
    for nn in range(12):
        og, oo = sg.step(nn), so.step(nn)
        # (the z-momentum residual of this 2-D case is round-off: a floor relative to the others)
        assert np.allclose(og["l2"], oo["l2"], rtol=1e-9, atol=1e-12 * oo["l2"].max()), \
            (nn, og["l2"], oo["l2"])
        assert abs(og["matrix"] - oo["matrix"]) <= 1e-8 * oo["matrix"], (nn, og["matrix"], oo["matrix"])
        for lev in range(3):
            a = sg.download("state", 0, lev)[g:-g, g:-g, g:-g]
            b = so.download("state", 0, lev)[g:-g, g:-g, g:-g]
            assert rel_err(a, b) < 1e-9, (nn, lev)
    so.close()
    with open(os.path.join(GOLDEN, "regression_truths.json")) as fh:
        spec = json.load(fh)["transonicBump"]
    out = None
    for nn in range(12, spec["iterations"]):
        out = sg.step(nn)
    for idx, (got, t) in enumerate(zip(out["norm"], spec["truth"])):
        if idx in spec["ignore"]:
            continue
        assert f"{got:.4e}" == f"{t:.4e}", (idx, got, t)
    sg.close()


@pytest.mark.gpu
@pytest.mark.parametrize("cycle,nblocks,levels", [("V", 1, 2), ("W", 2, 3)])
def test_multigrid_synthetic_parity(agx, oracle, cycle, nblocks, levels):
    """agx_mg_* against the oracle on 3-D boxes: V cycle on two levels, W cycle on three with
    two blocks joined by a connection (coarse levels with their own connections and odd
    cell counts: 10 -> 5 -> 3), stretched grids; norms, matrix residual, the states and the
    updates of every level after every iteration."""
    from aither_amd.solver import MultigridSolver
    kw = dict(n=(12, 10, 8), nblocks=nblocks, axis="i", stretch=1.1, levels=levels, cycle=cycle,
              time_integration="implicitEuler", matrix_solver="dplur", matrix_sweeps=4, cfl=40.0)
    cg, tg = synthetic.multigrid_levels(**kw)
    co, to = synthetic.multigrid_levels(**kw)
    sg, so = MultigridSolver(agx, cg, tg), MultigridSolver(oracle, co, to)
    g = cg[0].ng
    for nn in range(4):
        og, oo = sg.step(nn), so.step(nn)
        assert np.allclose(og["l2"], oo["l2"], rtol=1e-9, atol=1e-12 * oo["l2"].max())
        assert abs(og["matrix"] - oo["matrix"]) <= 1e-8 * oo["matrix"]
        for lev in range(levels):
            for gb in range(nblocks):
                for f in ("state", "update"):
                    a = sg.download(f, gb, lev)[g:-g, g:-g, g:-g]
                    b = so.download(f, gb, lev)[g:-g, g:-g, g:-g]
                    assert rel_err(a, b) < 1e-9, (nn, lev, gb, f)
    sg.close(), so.close()


@pytest.mark.gpu
@pytest.mark.parametrize("solver,env", [("bdplur", {}), ("blusgs", {}), ("blusgs", {"AGX_SWEEP_PIPE": "0"}),
                                        ("lusgs", {"AGX_LUSGS": "plane"}), ("lusgs", {})])
def test_multigrid_other_solvers_parity(agx, oracle, solver, env):
    """The forcing term in the other relaxations: BDPLUR (beside b), BLU-SGS on the pipelined
    and on the launch-per-hyperplane record sweeps, scalar LU-SGS on the hyperplane form
    (inside b, read from the records) and on the production path -- the diagonal-ordered
    pipelined sweeps, where b with the forcing term is formed by k_lusgs_prepare, x crosses
    between the D2 arrays and the planes around every transfer and the matrix residual is
    formed on the planes -- W cycle, three levels, two blocks with a connection, viscous,
    against the oracle."""
    from aither_amd.solver import MultigridSolver
    wall = {3: ("viscousWall", 2)}
    kw = dict(n=(12, 10, 8), nblocks=2, axis="i", stretch=1.1, levels=3, cycle="W", bcs=wall,
              equation_set="navierStokes", time_integration="implicitEuler",
              matrix_solver=solver, matrix_sweeps=2, cfl=20.0)
    cg, tg = synthetic.multigrid_levels(**kw)
    co, to = synthetic.multigrid_levels(**kw)
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        sg = MultigridSolver(agx, cg, tg)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    so = MultigridSolver(oracle, co, to)
    g = cg[0].ng
    for nn in range(3):
        og, oo = sg.step(nn), so.step(nn)
        assert np.allclose(og["l2"], oo["l2"], rtol=1e-9, atol=1e-12 * oo["l2"].max())
        assert abs(og["matrix"] - oo["matrix"]) <= 1e-7 * oo["matrix"]
        for lev in range(3):
            for gb in range(2):
                for f in ("state", "update"):
                    a = sg.download(f, gb, lev)[g:-g, g:-g, g:-g]
                    b = so.download(f, gb, lev)[g:-g, g:-g, g:-g]
                    assert rel_err(a, b) < 1e-9, (nn, lev, gb, f)
    sg.close(), so.close()


@pytest.mark.gpu
@pytest.mark.parametrize("solver,env", [("blusgs", {}), ("blusgs", {"AGX_SWEEP_PIPE": "0"}),
                                        ("lusgs", {}), ("dplur", {}), ("bdplur", {})])
def test_multigrid_seven_equations_parity(agx_rans, oracle, solver, env):
    """k-omega SST 2003 under the cycle (the 7-equation library): the turbulence equations are
    restricted, forced and prolonged with the flow equations, their part of the scalar and of
    the block diagonal accumulates over the visits of a coarse level; V and W cycles on three
    levels, two blocks with a connection, a viscous wall -- norms, matrix residual, state and
    update of every level against the oracle."""
    from aither_amd.solver import MultigridSolver
    kw = dict(n=(12, 10, 8), nblocks=2, axis="i", stretch=1.15, levels=3,
              cycle="W" if solver == "blusgs" else "V", bcs=RANS_WALL, equation_set="rans",
              turbulence_model="sst2003", time_integration="implicitEuler",
              matrix_solver=solver, matrix_sweeps=2, cfl=10.0)
    cg, tg = synthetic.multigrid_levels(**kw)
    co, to = synthetic.multigrid_levels(**kw)
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        sg = MultigridSolver(agx_rans, cg, tg)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    so = MultigridSolver(oracle, co, to)
    g = cg[0].ng
    for nn in range(3):
        og, oo = sg.step(nn), so.step(nn)
        assert np.allclose(og["l2"], oo["l2"], rtol=1e-9, atol=1e-12 * oo["l2"].max())
        assert abs(og["matrix"] - oo["matrix"]) <= 1e-7 * oo["matrix"]
        # (the saved update of a level may be round-off of the finest one's: measured against
        # the largest update of the iteration)
        scale = max(np.abs(so.download("update", gb, lev)).max()
                    for lev in range(3) for gb in range(2))
        for lev in range(3):
            for gb in range(2):
                a = sg.download("state", gb, lev)[g:-g, g:-g, g:-g]
                b = so.download("state", gb, lev)[g:-g, g:-g, g:-g]
                assert rel_err(a, b) < 1e-9, (nn, lev, gb, "state")
                a = sg.download("update", gb, lev)[g:-g, g:-g, g:-g]
                b = so.download("update", gb, lev)[g:-g, g:-g, g:-g]
                assert np.abs(a - b).max() < 1e-9 * scale, (nn, lev, gb, "update")
    sg.close(), so.close()
