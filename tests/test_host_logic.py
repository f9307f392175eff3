"""CPU tests of the host layer: input parsing, metrics, connections, halo
maps, and that both shared libraries export every symbol the header declares.
"""
import ctypes
import os
import re
import sys

import numpy as np
import pytest

from conftest import ROOT, GOLDEN, golden_case
from aither_amd import abi
from aither_amd.case import connections as conn_mod
from aither_amd.case import geometry as geo
from aither_amd.case import synthetic
from aither_amd.case.inputfile import parse_input
from aither_amd.solver import Solver


def test_header_symbols_match_abi_table():
    text = open(os.path.join(ROOT, "include", "aither_gfx950.h")).read()
    declared = set(re.findall(r"\bagx_(\w+)\s*\(", text))
    assert declared == set(abi.SYMBOLS), declared ^ set(abi.SYMBOLS)


def test_product_library_exports_every_symbol():
    """No compute calls here (no GPU): only load + symbol lookup."""
    import aither_amd
    assert os.path.exists(aither_amd.LIB_PATH), \
        "libaither_gfx950.so not built: run __graft_entry__.build()"
    lib = ctypes.CDLL(aither_amd.LIB_PATH)
    for name in abi.SYMBOLS:
        assert hasattr(lib, "agx_" + name), name
    api = aither_amd.load()
    assert b"gfx950" in api.version()
    # the 7-equation (rans) build of the same sources exports the same C-ABI
    assert os.path.exists(aither_amd.RANS_LIB_PATH)
    lib7 = ctypes.CDLL(aither_amd.RANS_LIB_PATH)
    for name in abi.SYMBOLS:
        assert hasattr(lib7, "agx_" + name), name


def test_product_libraries_export_only_the_c_abi():
    """The 5- and 7-equation libraries are one source tree compiled for two state layouts:
    kernel handles, device stubs and C++ internals of one must never resolve into the
    other, so nothing but agx_* may appear in either dynamic symbol table --
    and both can then be loaded RTLD_GLOBAL into one process."""
    import subprocess
    import aither_amd
    for path in (aither_amd.LIB_PATH, aither_amd.RANS_LIB_PATH):
        out = subprocess.check_output(["nm", "-D", "--defined-only", path], text=True)
        names = [ln.split()[-1] for ln in out.splitlines() if ln.strip()]
        assert names and all(n.startswith("agx_") for n in names), \
            [n for n in names if not n.startswith("agx_")][:5]
        assert {n[4:] for n in names} == set(abi.SYMBOLS)
    # in a fresh interpreter (this one may hold torch's own copy of the HIP runtime, which
    # must not be mixed with /opt/rocm's through the global scope)
    code = ("import ctypes, sys\n"
            "a = ctypes.CDLL(sys.argv[1], mode=ctypes.RTLD_GLOBAL)\n"
            "b = ctypes.CDLL(sys.argv[2], mode=ctypes.RTLD_GLOBAL)\n"
            "a.agx_version.restype = b.agx_version.restype = ctypes.c_char_p\n"
            "assert b'gfx950' in a.agx_version() and b'gfx950' in b.agx_version()\n"
            "print('both loaded')\n")
    out = subprocess.check_output([sys.executable, "-c", code, aither_amd.LIB_PATH,
                                   aither_amd.RANS_LIB_PATH], text=True)
    assert "both loaded" in out


def test_product_library_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import aither_amd
    api = aither_amd.load()
    ctx = ctypes.c_void_p()
    rc = api.ctx_create(0, 0, ctypes.byref(ctx))
    assert rc != 0 and b"no HIP device" in api.last_error()


def test_oracle_exports_every_symbol(oracle):
    for name in abi.SYMBOLS:
        assert hasattr(oracle, name)


def test_parse_shocktube_deck():
    d = parse_input(os.path.join(GOLDEN, "cases", "shockTube", "shockTube.inp"))
    assert d.time_integration == "bdf2" and (d.theta, d.zeta) == (1.0, 0.5)
    assert d.face_reconstruction == "weno" and d.num_ghost_layers() == 3
    assert d.nonlinear_iterations == 5 and d.dual_time_cfl == 1000
    assert len(d.bcs) == 2 and d.bcs[0][4].bc_type == "slipWall"
    s = [s for s in d.bcs[0] if s.bc_type == "interblock"][0]
    assert (s.partner_surface(), s.partner_block()) == (5, 1)
    assert d.ic_for_block(1).get("density") == 0.153125


def test_metrics_closed_cells_and_volume():
    x = synthetic.box_nodes(6, 5, 4, stretch=1.3, skew=0.02)
    m = geo.interior_metrics(x)
    av = lambda f: f[..., :3] * f[..., 3:4]
    ai, aj, ak = av(m["farea_i"]), av(m["farea_j"]), av(m["farea_k"])
    closed = (ai[:, :, 1:] - ai[:, :, :-1]) + (aj[:, 1:] - aj[:, :-1]) + \
        (ak[1:] - ak[:-1])
    assert np.abs(closed).max() < 1e-14
    # divergence theorem: sum of x.n A over the boundary = 3 V
    assert m["vol"].sum() > 0
    fc = m["fcen_i"]
    tot = (fc[:, :, -1] * ai[:, :, -1]).sum() - (fc[:, :, 0] * ai[:, :, 0]).sum()
    fc = m["fcen_j"]
    tot += (fc[:, -1] * aj[:, -1]).sum() - (fc[:, 0] * aj[:, 0]).sum()
    fc = m["fcen_k"]
    tot += (fc[-1] * ak[-1]).sum() - (fc[0] * ak[0]).sum()
    assert abs(tot / 3.0 - m["vol"].sum()) < 2e-3 * m["vol"].sum()


def test_uniform_box_ghost_geometry():
    case = synthetic.single_block_case((5, 4, 3), amplitude=0.0)
    g = case.blocks[0].geom
    ng = g.ng
    # widths are uniform everywhere except corners
    for d, n in zip("ijk", (5, 4, 3)):
        w = g.width[d].a[ng:-ng, ng:-ng, :, 0] if d == "i" else \
            g.width[d].a[ng:-ng, :, ng:-ng, 0] if d == "j" else \
            g.width[d].a[:, ng:-ng, ng:-ng, 0]
        assert np.allclose(w, 1.0 / n, rtol=1e-13)
    assert np.allclose(g.vol.a[ng:-ng, ng:-ng, :, 0], 1.0 / 60.0, rtol=1e-13)


def test_connection_detection_shocktube():
    case = golden_case("shockTube")
    assert len(case.connections) == 1
    c = case.connections[0]
    assert c.orientation == 1 and c.is_interblock
    assert sorted(c.boundary) == [5, 6]


def test_periodic_connection_couette():
    case = golden_case("couette")
    (c,) = case.connections
    assert not c.is_interblock and c.block == [0, 0] and c.orientation == 1


def test_halo_maps_python_vs_oracle(oracle):
    """insert_maps (numpy restatement of GetSwapLoc) and the oracle's C
    restatement move the same cells."""
    case = synthetic.stacked_blocks_case((4, 3, 5), nblocks=2, axis="j",
                                         amplitude=0.05)
    sol = Solver(oracle, case)
    ng = case.ng
    for gb, blk in enumerate(case.blocks):
        st = np.arange(blk.state.size, dtype=float).reshape(blk.state.shape) + \
            1e6 * gb
        sol.upload("state", gb, st)
        blk._probe = st
    oracle.check(oracle.halo_swap_local(sol.ctx, abi.HALO_STATE))
    (c,) = case.connections
    b0, b1 = case.blocks
    d0, s1, _ = conn_mod.insert_maps(c, True, ng, b0.geom.n, b1.geom.n)
    d1, s0, _ = conn_mod.insert_maps(c, False, ng, b1.geom.n, b0.geom.n)
    e0 = b0._probe.reshape(-1, 5).copy()
    e1 = b1._probe.reshape(-1, 5).copy()
    e0[d0] = b1._probe.reshape(-1, 5)[s1]
    e1[d1] = b0._probe.reshape(-1, 5)[s0]
    assert np.array_equal(sol.download("state", 0).reshape(-1, 5), e0)
    assert np.array_equal(sol.download("state", 1).reshape(-1, 5), e1)
    sol.close()


def test_uniformflow_all_eight_orientations(oracle):
    """The reference's uniformFlow grid (regressionTests.py:478-495) joins ten
    blocks with every one of the eight patch orientations, lower/lower and
    upper/upper pairs and i<->j / j<->k patch pairs (run here as an Euler deck).
    (1) the numpy restatement of GetSwapLoc (boundaryConditions.cpp:3006-3181)
    and the oracle's C restatement move the same cells for every connection;
    (2) with the ghost geometry swapped by the general PutGeomSlice rules a
    uniform stream stays uniform: any mismatch of cells, faces or flipped normals
    at a connection would show up as a residual there."""
    case = golden_case("uniformFlow")
    assert sorted({c.orientation for c in case.connections}) == list(range(1, 9))
    assert any(c.lower_lower_or_upper_upper() for c in case.connections)
    assert any(c.dir(3, 0) != c.dir(3, 1) for c in case.connections)
    sol = Solver(oracle, case)
    ng = case.ng
    probes = []
    for gb, blk in enumerate(case.blocks):
        st = np.arange(blk.state.size, dtype=float).reshape(blk.state.shape) + 1e6 * gb
        sol.upload("state", gb, st)
        probes.append(st.reshape(-1, 5))
    oracle.check(oracle.halo_swap_local(sol.ctx, abi.HALO_STATE))
    expect = [p.copy() for p in probes]
    for c in case.connections:     # each connection touches ghost cells of its own
        b0, b1 = c.block
        g0, g1 = case.blocks[b0].geom, case.blocks[b1].geom
        d0, s1, _ = conn_mod.insert_maps(c, True, ng, g0.n, g1.n)
        d1, s0, _ = conn_mod.insert_maps(c, False, ng, g1.n, g0.n)
        expect[b0][d0] = probes[b1][s1]
        expect[b1][d1] = probes[b0][s0]
    for gb in range(len(case.blocks)):
        assert np.array_equal(sol.download("state", gb).reshape(-1, 5), expect[gb]), gb
    sol.close()
    # free-stream preservation through all the connections
    case = golden_case("uniformFlow")
    sol = Solver(oracle, case)
    out = sol.step(0)
    from parity_utils import flux_scale
    for gb in range(len(case.blocks)):
        r = sol.download("residual", gb)
        assert np.abs(r).max() < 1e-11 * flux_scale(case), (gb, np.abs(r).max())
    sol.close()


def test_interblock_matches_single_block(oracle):
    """Explicit scheme: two stacked blocks give the same residual as the one
    merged block (ghost cells at the connection equal the neighbour's cells)."""
    kw = dict(time_integration="rk4", cfl=0.5)
    two = synthetic.stacked_blocks_case((6, 5, 4), nblocks=2, axis="k",
                                        amplitude=0.0, **kw)
    deck = synthetic.make_deck(**kw)
    deck.bcs = [synthetic.box_surfaces(6, 5, 8)]
    from aither_amd.case.builder import build_case
    one = build_case(None, deck=deck,
                     coords=[synthetic.box_nodes(6, 5, 8, lengths=(1, 1, 2))])
    synthetic.perturbed_state(two, 0.05)
    synthetic.perturbed_state(one, 0.05)
    s2, s1 = Solver(oracle, two), Solver(oracle, one)
    for s in (s1, s2):
        s.step(0)
    r1 = s1.download("residual", 0)
    r2 = np.concatenate([s2.download("residual", 0),
                         s2.download("residual", 1)], axis=0)
    assert np.allclose(r1, r2, rtol=1e-12, atol=1e-16)
    st1 = s1.download("state", 0)[2:-2, 2:-2, 2:-2]
    st2 = np.concatenate([s2.download("state", 0)[2:-2, 2:-2, 2:-2],
                          s2.download("state", 1)[2:-2, 2:-2, 2:-2]], axis=0)
    assert np.allclose(st1, st2, rtol=1e-12)
    s1.close(); s2.close()


NONREFLECTING = {1: ("inlet", 6), 2: ("pressureOutlet", 7), 3: ("characteristic", 1),
                 4: ("pressureOutlet", 7), 5: ("inlet", 6), 6: ("characteristic", 1)}


def test_oracle_nonreflecting_uniform_flow_is_fixed_point(oracle):
    """Nonreflecting inlet / pressure outlet (ghostStates.cpp:435-462, :614-643):
    with the free stream of the BC states everywhere, dU = 0, the gradients vanish
    and the LODI relaxation terms cancel, so the ghost states equal the free stream
    and the flow does not move.  A perturbed start stays bounded and differs from
    the reflecting variants of the same surfaces (the branch is really taken)."""
    kw = dict(n=(9, 8, 7), stretch=1.1, bcs=NONREFLECTING, time_integration="bdf2",
              nonlinear_iterations=2, dt=2.0e-5, dual_time_cfl=100.0, matrix_solver="lusgs")
    case = synthetic.single_block_case(amplitude=0.0, **kw)
    s = Solver(oracle, case)
    s0 = s.download("state", 0).copy()
    g = case.ng
    for nn in range(2):
        s.step(nn)
    a = s.download("state", 0)
    assert np.abs(a - s0)[g:-g, g:-g, g:-g].max() < 1e-12
    assert np.abs(s.download("residual", 0)).max() < 1e-12
    s.close()

    moved = []
    for nr in (True, False):
        case = synthetic.single_block_case(amplitude=0.05, **kw)
        if not nr:
            for st in case.deck.bc_states:
                st.params["nonreflecting"] = False
        s = Solver(oracle, case)
        for nn in range(3):
            s.step(nn)
        moved.append(s.download("state", 0)[g:-g, g:-g, g:-g].copy())
        s.close()
    assert np.isfinite(moved[0]).all() and moved[0][..., 0].min() > 0.5
    assert np.abs(moved[0] - moved[1]).max() > 1e-6


def test_sstdes_differs_from_sst_only_in_the_k_destruction(oracle):
    """turbSstDes (turbulence.cpp:858-935) is SST 2003 with the k destruction scaled by
    phi = max((1 - f2) Lt / (cdes width), 1): on a box whose farfield turbulence puts Lt at
    a third of its size the first residual differs in the k equation, in many cells, and in
    nothing else."""
    from aither_amd.case import synthetic
    wall = {3: ("viscousWall", 2), 1: ("characteristic", 1), 2: ("characteristic", 1),
            4: ("characteristic", 1), 5: ("characteristic", 1), 6: ("characteristic", 1)}
    res = {}
    for model in ("sst2003", "sstdes"):
        case = synthetic.single_block_case(n=(9, 8, 7), stretch=1.2, bcs=wall, equation_set="rans",
                                           turbulence_model=model, time_integration="implicitEuler",
                                           cfl=10.0, turbulence=(0.2, 2.4e4))
        sol = Solver(oracle, case)
        sol.step(0)
        res[model] = (sol.download("residual", 0), sol.download("dt", 0))
        sol.close()
    d = np.abs(res["sstdes"][0] - res["sst2003"][0])
    for e in (0, 1, 2, 3, 4, 6):
        assert d[..., e].max() == 0.0, e
    assert (d[..., 5] > 0.0).sum() > 50
    # more destruction, never less: the k residual (sources subtracted) only grows
    assert np.all(res["sstdes"][0][..., 5] >= res["sst2003"][0][..., 5])


@pytest.mark.parametrize("tag", [2, 4, 5])
def test_wall_functions_are_active_on_every_thermal_wall_type(oracle, tag):
    """Adiabatic (2), isothermal moving (4) and constant heat flux (5) walls with
    wallTreatment=wallLaw: the first residual differs from the low-Re wall's in the momentum
    and turbulence equations and stays finite (y+ of the box's wall cells is far above 10)."""
    from aither_amd.case import synthetic
    wall = {3: ("viscousWall", tag), 1: ("characteristic", 1), 2: ("characteristic", 1),
            4: ("characteristic", 1), 5: ("characteristic", 1), 6: ("characteristic", 1)}
    res = {}
    for wt in (None, "wallLaw"):
        case = synthetic.single_block_case(n=(9, 8, 7), stretch=1.2, bcs=wall, equation_set="rans",
                                           turbulence_model="sst2003",
                                           time_integration="implicitEuler", cfl=10.0,
                                           wall_treatment=wt)
        sol = Solver(oracle, case)
        sol.step(0)
        res[wt] = sol.download("residual", 0)
        sol.close()
    assert np.isfinite(res["wallLaw"]).all()
    d = np.abs(res[None] - res["wallLaw"]).reshape(-1, 7).max(axis=0)
    assert np.all(d[[1, 2, 3, 5, 6]] > 0.0), d
