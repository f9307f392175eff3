"""Multi-process halo exchange (the replacement of the reference's MPI path)
rehearsed on CPU: world_size 2 and 3 over gloo, with the CPU oracle standing
in for the device library behind the same phase API (PhasedSolver).  The GPU
run differs only in the backend (agx_ instead of ora_) and the transport
(RCCL instead of gloo)."""
import ctypes
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, _oracle_lib
from aither_amd import abi
from aither_amd.case import synthetic
from aither_amd.solver import Solver, PhasedSolver, DistExchange, MultigridSolver

FARFIELD = {s: ("characteristic", 1) for s in range(1, 7)}
KW = {
    "rk4": dict(time_integration="rk4", cfl=0.5),
    "dplur": dict(bcs=FARFIELD, inviscid_flux="ausm", limiter="none",
                  time_integration="implicitEuler", matrix_solver="dplur",
                  matrix_sweeps=4, cfl=20.0),
    "lusgs": dict(face_reconstruction="weno", limiter="none",
                  time_integration="implicitEuler", cfl=10.0),
    # viscous block-matrix solver: the velocity gradients cross the ranks as well
    "blusgs_visc": dict(bcs={3: ("viscousWall", 2), 1: ("characteristic", 1),
                             2: ("pressureOutlet", 3), 4: ("characteristic", 1)},
                        equation_set="navierStokes", time_integration="implicitEuler",
                        matrix_solver="blusgs", matrix_sweeps=2, cfl=10.0),
    # rans: eddy viscosity and blending functions cross the ranks after the residual
    "rans": dict(bcs={3: ("viscousWall", 2), 1: ("characteristic", 1),
                      2: ("characteristic", 1), 4: ("characteristic", 1)},
                 equation_set="rans", turbulence_model="sst2003",
                 time_integration="implicitEuler", matrix_sweeps=2, cfl=10.0),
}


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _exchange(items):
    reqs = []
    for peer, tag, send, recv in items:
        reqs.append(dist.isend(send, peer, tag=tag))
        reqs.append(dist.irecv(recv, peer, tag=tag))
    for r in reqs:
        r.wait()


def _alloc(cnt):
    return torch.empty(max(int(cnt), 1), dtype=torch.float64)


def _make_case(kind, world, builder):
    if builder == "golden":      # one of the reference's cases, block b on rank b
        from aither_amd.case.builder import build_case
        inp = os.path.join(ROOT, "tests", "golden", "cases", kind, kind + ".inp")
        return lambda rank: build_case(inp, ranks=list(range(world)))
    if builder == "stacked":
        return lambda rank: synthetic.stacked_blocks_case(
            (6, 5, 4), nblocks=world, axis="k", stretch=1.1,
            ranks=list(range(world)), **KW[kind])
    sys.path.insert(0, ROOT)
    import bench
    return lambda rank: bench.rank_local_chain_case(rank, world, 8, "rk4")


def _worker(rank, world, port, kind, builder, steps, q, in_library=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ora = abi.Api(ctypes.CDLL(_oracle_lib()), "ora_")
    case = _make_case(kind, world, builder)(rank)
    if in_library:     # iterate() drives the remote connections through the exchange table
        sol = Solver(ora, case, rank=rank, exchange=DistExchange(world))
    else:
        sol = PhasedSolver(ora, case, rank, _exchange, _alloc)
    for nn in range(steps):
        sol.step(nn)
    (gb,) = sol.block_ids
    q.put((rank, sol.download("state", gb), sol.download("residual", gb),
           np.array([h["l2"] ** 2 for h in sol.history])))
    dist.barrier()
    sol.close()
    dist.destroy_process_group()


def _run(world, kind, builder, steps=2, in_library=False):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker,
                         args=(r, world, port, kind, builder, steps, q, in_library))
             for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(world):
        rank, st, rs, l2 = q.get(timeout=300)
        res[rank] = (st, rs, l2)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return res


@pytest.mark.parametrize("world,kind", [(2, "rk4"), (2, "dplur"), (2, "lusgs"),
                                        (3, "dplur"), (2, "blusgs_visc"), (2, "rans")])
def test_phased_multiprocess_matches_single_process(oracle, world, kind):
    res = _run(world, kind, "stacked")
    case = synthetic.stacked_blocks_case((6, 5, 4), nblocks=world, axis="k",
                                         stretch=1.1, **KW[kind])
    ref = Solver(oracle, case)
    for nn in range(2):
        ref.step(nn)
    ng = case.ng
    core = lambda a: a[ng:-ng, ng:-ng, ng:-ng]
    l2sum = sum(res[r][2] for r in range(world))
    l2ref = np.array([h["l2"] ** 2 for h in ref.history])
    assert np.allclose(l2sum, l2ref, rtol=1e-12)
    for r in range(world):
        assert np.array_equal(core(res[r][0]), core(ref.download("state", r)))
        assert np.array_equal(res[r][1], ref.download("residual", r))
    ref.close()


@pytest.mark.parametrize("world,kind", [(2, "rk4"), (2, "lusgs"), (3, "dplur"),
                                        (2, "blusgs_visc"), (2, "rans")])
def test_iterate_with_exchange_matches_single_process(oracle, world, kind):
    """The in-library multi-rank path: iterate() itself packs, swaps (exchange
    table on gloo, host buffers) and unpacks the slabs of connections to other
    ranks, and returns the norms already reduced over the ranks (main.cpp:254-264)
    -- identical on every rank and equal to the single-process run."""
    res = _run(world, kind, "stacked", in_library=True)
    case = synthetic.stacked_blocks_case((6, 5, 4), nblocks=world, axis="k",
                                         stretch=1.1, **KW[kind])
    ref = Solver(oracle, case)
    for nn in range(2):
        ref.step(nn)
    ng = case.ng
    core = lambda a: a[ng:-ng, ng:-ng, ng:-ng]
    l2ref = np.array([h["l2"] ** 2 for h in ref.history])
    for r in range(world):
        assert np.allclose(res[r][2], l2ref, rtol=1e-12)       # global on every rank
        assert np.array_equal(res[r][2], res[0][2])
        assert np.array_equal(core(res[r][0]), core(ref.download("state", r)))
        assert np.array_equal(res[r][1], ref.download("residual", r))
    ref.close()


def test_bench_rank_local_chain_matches_full_build(oracle):
    """bench.py builds only its own block at full size (neighbours 4 cells
    thick); the result must equal the fully built 2-block chain."""
    res = _run(2, "rk4", "chain", steps=1)
    case = synthetic.stacked_blocks_case((8, 8, 8), nblocks=2, axis="k",
                                         stretch=1.2, time_integration="rk4",
                                         cfl=0.5)
    ref = Solver(oracle, case)
    ref.step(0)
    for r in range(2):
        assert np.allclose(res[r][1], ref.download("residual", r),
                           rtol=1e-12, atol=1e-18)
    ref.close()


# ---- BASELINE configs[3] style: 2 x 2 x 2 blocks, DPLUR, several blocks per rank
CUBE_KW = dict(inviscid_flux="ausm", limiter="none", time_integration="implicitEuler",
               matrix_solver="dplur", matrix_sweeps=4, cfl=5.0)


def _cube_case(ranks):
    return synthetic.cube_blocks_case(n=(6, 5, 4), splits=(2, 2, 2), ranks=ranks, **CUBE_KW)


def _cube_worker(rank, world, port, steps, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ora = abi.Api(ctypes.CDLL(_oracle_lib()), "ora_")
    case = _cube_case([b * world // 8 for b in range(8)])
    sol = PhasedSolver(ora, case, rank, _exchange, _alloc)
    for nn in range(steps):
        sol.step(nn)
    q.put((rank, {gb: sol.download("state", gb) for gb in sol.block_ids},
           np.array([h["l2"] ** 2 for h in sol.history])))
    dist.barrier()
    sol.close()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_cube_of_blocks_over_ranks(oracle, world):
    """Eight blocks, four or two per rank: local swaps and remote slabs mixed in
    one iteration; the Euler result must not depend on where the blocks live."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_cube_worker, args=(r, world, port, 2, q))
             for r in range(world)]
    for p in procs:
        p.start()
    got, l2sum = {}, 0.0
    for _ in range(world):
        rank, states, l2 = q.get(timeout=300)
        got.update(states)
        l2sum = l2sum + l2
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    case = _cube_case(None)
    ref = Solver(oracle, case)
    for nn in range(2):
        ref.step(nn)
    ng = case.ng
    core = lambda a: a[ng:-ng, ng:-ng, ng:-ng]
    assert np.allclose(l2sum, np.array([h["l2"] ** 2 for h in ref.history]), rtol=1e-12)
    for gb in range(8):
        assert np.allclose(core(got[gb]), core(ref.download("state", gb)), rtol=1e-13, atol=0)
    ref.close()


def test_cube_of_blocks_equals_single_block(oracle):
    """Explicit Euler stages do not see block boundaries: 2 x 2 x 2 blocks give
    the single-block answer (face ghost cells carry everything MUSCL needs)."""
    from aither_amd.case import builder as _bld
    kw = dict(time_integration="rk4", cfl=0.5)
    c8 = synthetic.cube_blocks_case(n=(6, 5, 4), splits=(2, 2, 2), **kw)
    deck = synthetic.make_deck(**kw)
    deck.bcs = [synthetic.box_surfaces(12, 10, 8, None)]
    c1 = _bld.build_case(None, deck=deck,
                         coords=[synthetic.box_nodes(12, 10, 8, 1.0, lengths=(2.0, 2.0, 2.0))])
    synthetic.perturbed_state(c1, 0.05)
    s8, s1 = Solver(oracle, c8), Solver(oracle, c1)
    for nn in range(2):
        s8.step(nn), s1.step(nn)
    g = c1.ng
    full = s1.download("state", 0)[g:-g, g:-g, g:-g]
    for bk in range(2):
        for bj in range(2):
            for bi in range(2):
                a = s8.download("state", bi + 2 * (bj + 2 * bk))[g:-g, g:-g, g:-g]
                ref = full[bk * 4:(bk + 1) * 4, bj * 5:(bj + 1) * 5, bi * 6:(bi + 1) * 6]
                assert np.abs(a - ref).max() <= 1e-13 * np.abs(ref).max()
    s8.close(), s1.close()


@pytest.mark.parametrize("in_library", [False, True])
def test_walllaw_two_ranks_match_single_process(oracle, in_library):
    """The reference's wallLaw case (SST 2003, wall functions, BLU-SGS), its two blocks
    on two ranks: the wall data of the faces stay with the block's rank, velocity
    gradients, eddy viscosity and the update cross the ranks -- bit-identical to the
    single-process run, phase API and iterate-with-exchange."""
    from aither_amd.case.builder import build_case
    res = _run(2, "wallLaw", "golden", steps=3, in_library=in_library)
    case = build_case(os.path.join(ROOT, "tests", "golden", "cases", "wallLaw", "wallLaw.inp"))
    ref = Solver(oracle, case)
    for nn in range(3):
        ref.step(nn)
    ng = case.ng
    core = lambda a: a[ng:-ng, ng:-ng, ng:-ng]
    l2sum = res[0][2] if in_library else sum(res[r][2] for r in range(2))
    l2ref = np.array([h["l2"] ** 2 for h in ref.history])
    assert np.allclose(l2sum, l2ref, rtol=1e-12)
    for r in range(2):
        assert np.array_equal(core(res[r][0]), core(ref.download("state", r)))
        assert np.array_equal(res[r][1], ref.download("residual", r))
    ref.close()


MG_KW = {
    "dplur": dict(bcs=FARFIELD, inviscid_flux="ausm", limiter="none",
                  time_integration="implicitEuler", matrix_solver="dplur",
                  matrix_sweeps=4, cfl=20.0),
    "blusgs": dict(bcs=FARFIELD, time_integration="implicitEuler", matrix_solver="blusgs",
                   matrix_sweeps=2, cfl=10.0),
}


def _mg_levels(world, kind, cycle, ranks):
    return synthetic.multigrid_levels((8, 6, 8), nblocks=world, axis="k", levels=3,
                                      cycle=cycle, stretch=1.1, ranks=ranks, **MG_KW[kind])


def _mg_worker(rank, world, port, kind, cycle, steps, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ora = abi.Api(ctypes.CDLL(_oracle_lib()), "ora_")
    cases, transfers = _mg_levels(world, kind, cycle, list(range(world)))
    sol = MultigridSolver(ora, cases, transfers, rank=rank,
                          exchange=lambda: DistExchange(world))
    for nn in range(steps):
        sol.step(nn)
    (gb,) = sol.levels[0].block_ids
    q.put((rank, sol.download("state", gb), sol.download("update", gb, 1),
           np.array([h["l2"] ** 2 for h in sol.history])))
    dist.barrier()
    sol.close()
    dist.destroy_process_group()


@pytest.mark.parametrize("kind,cycle", [("dplur", "W"), ("blusgs", "V")])
def test_multigrid_across_ranks_matches_single_process(oracle, kind, cycle):
    """Every grid level split over the ranks like the finest one (the reference decomposes
    the finest level and coarsens each rank's blocks, gridLevel.cpp:440-535): the connections
    of the coarse levels cross the ranks through the same exchange table; restriction and
    prolongation stay inside a block."""
    world, steps = 2, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_mg_worker, args=(r, world, port, kind, cycle, steps, q))
             for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(world):
        rank, st, up, l2 = q.get(timeout=300)
        res[rank] = (st, up, l2)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    cases, transfers = _mg_levels(world, kind, cycle, None)
    ref = MultigridSolver(oracle, cases, transfers)
    for nn in range(steps):
        ref.step(nn)
    ng = cases[0].ng
    core = lambda a: a[ng:-ng, ng:-ng, ng:-ng]
    l2sum = sum(res[r][2] for r in range(world))
    l2ref = np.array([h["l2"] ** 2 for h in ref.history])
    assert np.allclose(l2sum, l2ref, rtol=1e-12)
    for r in range(world):
        assert np.array_equal(core(res[r][0]), core(ref.download("state", r)))
        assert np.array_equal(core(res[r][1]), core(ref.download("update", r, 1)))
    ref.close()
