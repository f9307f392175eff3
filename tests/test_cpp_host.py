"""The C++ host layer (include/aither_gfx950.hpp) driving the C-ABI.

tests/cpp/host_parity.cpp is the reference-language statement of the time loop
(main.cpp:232-275 reduced to StoreOldSolution + Iterate).  It is built twice,
against the product library and against the CPU oracle; this module writes the
case file both read and compares what they produce.
"""
import ctypes as C
import os
import struct
import subprocess

import numpy as np
import pytest

from parity_utils import rel_err, RTOL
from aither_amd import abi
from aither_amd.case import builder as _b
from aither_amd.case import synthetic
from aither_amd.solver import Solver

HERE = os.path.dirname(os.path.abspath(__file__))
CPP = os.path.join(HERE, "cpp")


def build_drivers():
    subprocess.check_call(["make", "-C", os.path.join(HERE, "..", "oracle")],
                          stdout=subprocess.DEVNULL)
    subprocess.check_call(["make", "-C", CPP], stdout=subprocess.DEVNULL)


def write_case_file(case, path, n_steps):
    d = case.deck
    cfg = _b.config_struct(case)
    with open(path, "wb") as f:
        f.write(struct.pack("<7i", 0x32584741, len(case.blocks), len(case.connections),
                            d.nonlinear_iterations, n_steps,
                            int(d.need_to_store_time_n()),
                            int(d.is_multilevel_in_time())))
        f.write(np.array([d.cfl(nn) for nn in range(n_steps)], dtype="<f8").tobytes())
        f.write(bytes(cfg))
        for gb, blk in enumerate(case.blocks):
            g = blk.geom
            f.write(struct.pack("<7i", g.ni, g.nj, g.nk, g.ng, blk.parent, blk.global_pos,
                                int(blk.rank or 0)))
            for a in (g.farea["i"].a, g.farea["j"].a, g.farea["k"].a, g.vol.a,
                      g.center.a, g.width["i"].a, g.width["j"].a, g.width["k"].a,
                      g.wall_dist.a):
                f.write(np.ascontiguousarray(a, dtype="<f8").tobytes())
            surfs = _b.surface_structs(case, gb)
            f.write(struct.pack("<i", len(surfs)))
            f.write(bytes(surfs))
            f.write(np.ascontiguousarray(blk.state, dtype="<f8").tobytes())
        for conn in case.connections:
            cs = _b.connection_struct(conn)
            for side in range(2):
                cs.local_block[side] = conn.block[side]
            f.write(bytes(cs))


def read_output(case, path, n_steps, rank=None):
    """rank: read <path>.<rank>, which holds only that rank's blocks"""
    if rank is not None:
        path = f"{path}.{rank}"
    n_eq, nonlin = 5, case.deck.nonlinear_iterations
    raw = open(path, "rb").read()
    rec = n_eq * 8 + 16 + 20
    hist, off = [], 0
    for _ in range(n_steps * nonlin):
        l2 = np.frombuffer(raw, "<f8", n_eq, off)
        linf, matrix = np.frombuffer(raw, "<f8", 2, off + n_eq * 8)
        loc = np.frombuffer(raw, "<i4", 5, off + n_eq * 8 + 16)
        hist.append((l2.copy(), float(linf), float(matrix), tuple(int(v) for v in loc)))
        off += rec
    states = []
    for blk in case.blocks:
        if rank is not None and blk.rank != rank:
            continue
        n = blk.state.size
        states.append(np.frombuffer(raw, "<f8", n, off).reshape(blk.state.shape).copy())
        off += n * 8
    assert off == len(raw)
    return hist, states


CASES = {
    "rk4_muscl_roe": dict(n=(20, 9, 8), stretch=1.15, skew=0.01, time_integration="rk4",
                          cfl=0.5),
    "lusgs_viscous": dict(n=(12, 10, 9), stretch=1.1, equation_set="navierStokes",
                          time_integration="implicitEuler", matrix_solver="lusgs",
                          cfl=5.0, bcs={3: ("viscousWall", 2), 1: ("characteristic", 1),
                                        2: ("characteristic", 1), 4: ("characteristic", 1)}),
}


def _run(driver, case_file, out_file):
    subprocess.check_call([os.path.join(CPP, driver), case_file, out_file], timeout=300)


@pytest.mark.parametrize("name", sorted(CASES))
def test_cpp_host_oracle_matches_python_host(oracle, tmp_path, name):
    """Same library (the oracle), two hosts: the C++ layer must reproduce what
    the Python plumbing gets, bit for bit."""
    build_drivers()
    case = synthetic.single_block_case(**CASES[name])
    cf, of = str(tmp_path / "case.bin"), str(tmp_path / "ora.bin")
    write_case_file(case, cf, 3)
    _run("host_parity_ora", cf, of)
    hist, states = read_output(case, of, 3)
    s = Solver(oracle, case)
    k = 0
    for nn in range(3):
        s.store_time_n(nn)
        for mm in range(case.deck.nonlinear_iterations):
            l2, linf, mres = s.iterate(mm, case.deck.cfl(nn))
            assert np.array_equal(l2, hist[k][0])
            assert linf.linf == hist[k][1] and mres == hist[k][2]
            assert (linf.block, linf.i, linf.j, linf.k, linf.eqn) == hist[k][3]
            k += 1
    assert np.array_equal(s.download("state", 0), states[0])
    s.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(CASES))
def test_cpp_host_gpu_vs_oracle(tmp_path, name):
    """The C++ host over libaither_gfx950.so against the C++ host over the oracle."""
    build_drivers()
    case = synthetic.single_block_case(**CASES[name])
    cf = str(tmp_path / "case.bin")
    write_case_file(case, cf, 3)
    _run("host_parity_agx", cf, str(tmp_path / "agx.bin"))
    _run("host_parity_ora", cf, str(tmp_path / "ora.bin"))
    hg, sg = read_output(case, str(tmp_path / "agx.bin"), 3)
    ho, so = read_output(case, str(tmp_path / "ora.bin"), 3)
    ng = case.ng
    for (l2g, _, _, _), (l2o, _, _, _) in zip(hg, ho):
        assert rel_err(l2g[None, :], l2o[None, :]) < RTOL
    for a, b in zip(sg, so):
        assert rel_err(a[ng:-ng, ng:-ng, ng:-ng], b[ng:-ng, ng:-ng, ng:-ng]) < RTOL


MR_KINDS = {
    "rk4": dict(time_integration="rk4", cfl=0.5),
    "lusgs": dict(time_integration="implicitEuler", matrix_solver="lusgs", cfl=5.0),
    "dplur": dict(inviscid_flux="ausm", limiter="none", time_integration="implicitEuler",
                  matrix_solver="dplur", matrix_sweeps=4, cfl=5.0),
}


def _multirank(driver, oracle, tmp_path, kind, world, exact):
    """hotPath::Iterate with an exchange installed, one process per rank joined by
    socket pairs, against the single-process oracle."""
    build_drivers()
    kw = MR_KINDS[kind]
    case = synthetic.stacked_blocks_case((10, 6, 5), nblocks=world, axis="k", stretch=1.1,
                                         ranks=list(range(world)), **kw)
    cf, of = str(tmp_path / "case.bin"), str(tmp_path / "out.bin")
    write_case_file(case, cf, 2)
    subprocess.check_call([os.path.join(CPP, driver), cf, of, str(world)], timeout=600)
    one = synthetic.stacked_blocks_case((10, 6, 5), nblocks=world, axis="k", stretch=1.1, **kw)
    ref = Solver(oracle, one)
    for nn in range(2):
        ref.step(nn)
    l2ref = np.array([h["l2"] ** 2 for h in ref.history])
    ng = case.ng
    for r in range(world):
        hist, states = read_output(case, of, 2, rank=r)
        l2 = np.array([h[0] for h in hist])
        a = states[0][ng:-ng, ng:-ng, ng:-ng]
        b = ref.download("state", r)[ng:-ng, ng:-ng, ng:-ng]
        if exact:
            assert np.allclose(l2, l2ref, rtol=1e-12) and np.array_equal(a, b)
        else:
            assert rel_err(l2, l2ref) < RTOL and rel_err(a, b) < RTOL
    ref.close()


@pytest.mark.parametrize("kind,world", [("rk4", 2), ("lusgs", 2), ("dplur", 3)])
def test_cpp_multirank_oracle(oracle, tmp_path, kind, world):
    _multirank("host_parity_ora", oracle, tmp_path, kind, world, exact=True)


@pytest.mark.gpu
@pytest.mark.parametrize("kind,world", [("rk4", 2), ("lusgs", 2), ("dplur", 3)])
def test_cpp_multirank_gpu(oracle, tmp_path, kind, world):
    """the same with libaither_gfx950.so: the ranks share cuda:0, the slabs are
    staged through pinned host buffers (host_buffers = 1)"""
    _multirank("host_parity_agx", oracle, tmp_path, kind, world, exact=False)



# ---- the multigrid cycle from C++ (aither_gfx950::multigrid) -------------------------------
MG_KW = dict(n=(12, 10, 8), nblocks=2, axis="i", stretch=1.1, levels=3, cycle="W",
             time_integration="implicitEuler", matrix_solver="dplur", matrix_sweeps=4, cfl=40.0)


def write_multigrid_files(cases, transfers, tmp_path, n_steps):
    files = []
    for lev, case in enumerate(cases):
        files.append(str(tmp_path / f"level{lev}.bin"))
        write_case_file(case, files[-1], n_steps)
    tf = str(tmp_path / "transfers.bin")
    with open(tf, "wb") as f:
        f.write(struct.pack("<2i", 0x3247474d, len(transfers)))
        for trs in transfers:
            f.write(struct.pack("<i", len(trs)))
            for t in trs:
                f.write(struct.pack("<q", t.vol_fac.size))
                f.write(np.ascontiguousarray(t.to_coarse, dtype="<i4").tobytes())
                f.write(np.ascontiguousarray(t.vol_fac, dtype="<f8").tobytes())
                f.write(np.ascontiguousarray(t.coeffs, dtype="<f8").tobytes())
    return tf, files


def _run_mg(driver, cycle_index, tf, of, files):
    subprocess.check_call([os.path.join(CPP, driver), str(cycle_index), tf, of] + files, timeout=300)


def test_cpp_multigrid_oracle_matches_python_driver(oracle, tmp_path):
    """mgSolution::CycleAtLevel written twice -- aither_gfx950::multigrid (C++, the reference's
    language) and aither_amd.solver.MultigridSolver (Python) -- over the same library (the
    oracle): three levels, W cycle, two blocks with a connection; bit for bit."""
    from aither_amd.solver import MultigridSolver
    build_drivers()
    cases, transfers = synthetic.multigrid_levels(**MG_KW)
    tf, files = write_multigrid_files(cases, transfers, tmp_path, 3)
    of = str(tmp_path / "ora.bin")
    _run_mg("host_multigrid_ora", 2, tf, of, files)
    hist, states = read_output(cases[0], of, 3)
    s = MultigridSolver(oracle, cases, transfers)
    for nn in range(3):
        for lev in s.levels:
            lev.store_time_n(nn)
        l2, linf, mres = s.iterate(0, cases[0].deck.cfl(nn))
        assert np.array_equal(l2, hist[nn][0])
        assert linf.linf == hist[nn][1] and mres == hist[nn][2]
    for gb in range(2):
        assert np.array_equal(s.download("state", gb), states[gb])
    s.close()


@pytest.mark.gpu
def test_cpp_multigrid_gpu_vs_oracle(tmp_path):
    """The C++ multigrid driver over libaither_gfx950.so against the same driver over the
    oracle."""
    build_drivers()
    cases, transfers = synthetic.multigrid_levels(**MG_KW)
    tf, files = write_multigrid_files(cases, transfers, tmp_path, 3)
    _run_mg("host_multigrid_agx", 2, tf, str(tmp_path / "agx.bin"), files)
    _run_mg("host_multigrid_ora", 2, tf, str(tmp_path / "ora.bin"), files)
    hg, sg = read_output(cases[0], str(tmp_path / "agx.bin"), 3)
    ho, so = read_output(cases[0], str(tmp_path / "ora.bin"), 3)
    ng = cases[0].ng
    for (l2g, _, mg_, _), (l2o, _, mo, _) in zip(hg, ho):
        assert rel_err(l2g[None, :], l2o[None, :]) < 1e-9
        assert abs(mg_ - mo) <= 1e-8 * mo
    for a, b in zip(sg, so):
        assert rel_err(a[ng:-ng, ng:-ng, ng:-ng], b[ng:-ng, ng:-ng, ng:-ng]) < 1e-9
