"""The multi-rank path on a real GPU: two processes share cuda:0, each drives its
own block through the HIP library with PhasedSolver, and the halo slabs that
agx_halo_pack leaves in device buffers travel between the processes (staged
through the host and gloo here; bench.py hands the same device buffers to RCCL).
Checked against the single-process CPU oracle at 1e-10."""
import ctypes
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import _oracle_lib
from parity_utils import rel_err, RTOL
from aither_amd import abi
from aither_amd.case import synthetic
from aither_amd.solver import Solver, PhasedSolver, DistExchange

KW = {
    "rk4": dict(time_integration="rk4", cfl=0.5),
    "dplur": dict(inviscid_flux="ausm", limiter="none", time_integration="implicitEuler",
                  matrix_solver="dplur", matrix_sweeps=4, cfl=5.0),
    "lusgs": dict(time_integration="implicitEuler", matrix_solver="lusgs", cfl=5.0),
    # rans (7-equation build): eddy viscosity / blending functions -- and with BLU-SGS the
    # velocity gradients -- of the cells across the ranks
    "rans": dict(bcs={3: ("viscousWall", 2), 1: ("characteristic", 1),
                      2: ("characteristic", 1), 4: ("characteristic", 1)},
                 equation_set="rans", turbulence_model="sst2003",
                 time_integration="implicitEuler", matrix_solver="blusgs", matrix_sweeps=2,
                 cfl=10.0),
}
DIMS = (70, 9, 8)


def _case(kind, ranks):
    if kind == "wallLaw":     # the reference's case: SST 2003, wall functions, BLU-SGS, 2 blocks
        from aither_amd.case.builder import build_case
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        return build_case(os.path.join(root, "tests", "golden", "cases", kind, kind + ".inp"),
                          ranks=ranks)
    dims = (20, 9, 8) if kind == "rans" else DIMS
    return synthetic.stacked_blocks_case(dims, nblocks=2, axis="k", stretch=1.1,
                                         ranks=ranks, **KW[kind])


def _exchange(items):
    """device slab -> host -> gloo -> host -> device"""
    reqs, staged = [], []
    for peer, tag, send, recv in items:
        hs, hr = send.cpu(), torch.empty(recv.shape, dtype=recv.dtype)
        staged.append((recv, hr))
        reqs.append(dist.isend(hs, peer, tag=tag))
        reqs.append(dist.irecv(hr, peer, tag=tag))
    for r in reqs:
        r.wait()
    for recv, hr in staged:
        recv.copy_(hr)
    torch.cuda.synchronize()


def _alloc(cnt):
    return torch.empty(max(int(cnt), 1), dtype=torch.float64, device="cuda")


def _worker(rank, port, kind, q, in_library=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=2)
    import aither_amd
    agx = aither_amd.load(7 if kind in ("rans", "wallLaw") else 5)
    case = _case(kind, [0, 1])
    if in_library:   # agx_iterate drives the remote connection (host-staged slabs over gloo)
        sol = Solver(agx, case, rank=rank, exchange=DistExchange(2))
    else:
        sol = PhasedSolver(agx, case, rank, _exchange, _alloc)
    for nn in range(2):
        sol.step(nn)
    (gb,) = sol.block_ids
    q.put((rank, sol.download("state", gb), np.array([h["l2"] ** 2 for h in sol.history])))
    dist.barrier()
    sol.close()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_rccl_transport_single_rank(agx, oracle):
    """The built-in RCCL transport on the one GPU of this box: communicator
    creation on the library's device and the all-gather of the norm records with
    one rank (the grouped send / recv need a second GPU: the driver's multi-GPU
    run exercises them)."""
    import ctypes
    case = synthetic.stacked_blocks_case(DIMS, nblocks=2, axis="k", stretch=1.1, **KW["lusgs"])
    idbuf = ctypes.create_string_buffer(128)
    agx.check(agx.rccl_unique_id(idbuf), "rccl_unique_id")
    sg = Solver(agx, case, rccl=(idbuf.raw, 1, 0))
    so = Solver(oracle, case)
    for nn in range(2):
        sg.step(nn), so.step(nn)
    assert rel_err(np.array([h["l2"] for h in sg.history]),
                   np.array([h["l2"] for h in so.history])) < RTOL
    ng = case.ng
    for gb in range(2):
        assert rel_err(sg.download("state", gb)[ng:-ng, ng:-ng, ng:-ng],
                       so.download("state", gb)[ng:-ng, ng:-ng, ng:-ng]) < RTOL
    sg.close(), so.close()


@pytest.mark.gpu
@pytest.mark.parametrize("kind,in_library", [(k, False) for k in sorted(KW)] +
                         [("rk4", True), ("lusgs", True), ("dplur", True), ("rans", True),
                          ("wallLaw", False), ("wallLaw", True)])
def test_two_ranks_on_one_gpu(oracle, kind, in_library):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = [ctx.Process(target=_worker, args=(r, port, kind, q, in_library)) for r in range(2)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(2):
        rank, st, l2 = q.get(timeout=300)
        res[rank] = (st, l2)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    case = _case(kind, None)
    ref = Solver(oracle, case)
    for nn in range(2):
        ref.step(nn)
    ng = case.ng
    core = lambda a: a[ng:-ng, ng:-ng, ng:-ng]
    l2ref = np.array([h["l2"] ** 2 for h in ref.history])
    # the phase API returns rank-local norms, iterate-with-exchange the global ones
    got = res[0][1] if in_library else res[0][1] + res[1][1]
    assert rel_err(got, l2ref) < RTOL
    for r in range(2):
        assert rel_err(core(res[r][0]), core(ref.download("state", r))) < RTOL
    ref.close()


def _rccl_worker(rank, port, kind, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=2)
    import aither_amd
    agx = aither_amd.load(5)
    idbuf = ctypes.create_string_buffer(128)
    if rank == 0:
        agx.check(agx.rccl_unique_id(idbuf), "rccl_unique_id")
    box = [idbuf.raw]
    dist.broadcast_object_list(box, src=0)      # (the 128-byte id travels over gloo)
    sol = Solver(agx, _case(kind, [0, 1]), device=rank, rank=rank, rccl=(box[0], 2, rank))
    for nn in range(2):
        sol.step(nn)
    (gb,) = sol.block_ids
    q.put((rank, sol.download("state", gb), np.array([h["l2"] ** 2 for h in sol.history])))
    dist.barrier()
    sol.close()
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.skipif(torch.cuda.device_count() < 2,
                    reason="the grouped ncclSend / ncclRecv need one GPU per rank")
@pytest.mark.parametrize("kind", ["rk4", "lusgs"])
def test_two_gpus_rccl_transport(oracle, kind):
    """The built-in transport with a peer: one rank per GPU, slabs by grouped ncclSend /
    ncclRecv on the library's stream, norm records by ncclAllGather (skipped on the one-GPU
    boxes the suite normally runs on)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = [ctx.Process(target=_rccl_worker, args=(r, port, kind, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(2):
        rank, st, l2 = q.get(timeout=300)
        res[rank] = (st, l2)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    case = _case(kind, None)
    ref = Solver(oracle, case)
    for nn in range(2):
        ref.step(nn)
    ng = case.ng
    core = lambda a: a[ng:-ng, ng:-ng, ng:-ng]
    l2ref = np.array([h["l2"] ** 2 for h in ref.history])
    assert rel_err(res[0][1], l2ref) < RTOL          # (global norms on every rank)
    assert rel_err(res[1][1], l2ref) < RTOL
    for r in range(2):
        assert rel_err(core(res[r][0]), core(ref.download("state", r))) < RTOL
    ref.close()


CUBE_KW = dict(inviscid_flux="ausm", limiter="none", time_integration="implicitEuler",
               matrix_solver="dplur", matrix_sweeps=4, cfl=5.0)


def _cube_worker(rank, port, overlap, solver, div, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["AGX_OVERLAP"] = overlap
    dist.init_process_group("gloo", rank=rank, world_size=2)
    import aither_amd
    agx = aither_amd.load(5)
    kw = dict(CUBE_KW, matrix_solver=solver)
    case = synthetic.cube_blocks_case(n=(12, 10, 9), splits=(2, 2, 2),
                                      ranks=[(b // div) % 2 for b in range(8)], **kw)
    sol = Solver(agx, case, rank=rank, exchange=DistExchange(2))
    for nn in range(2):
        sol.step(nn)
    q.put((rank, {gb: sol.download("state", gb) for gb in sol.block_ids},
           np.array([h["l2"] ** 2 for h in sol.history])))
    dist.barrier()
    sol.close()
    dist.destroy_process_group()


def _cube_run(overlap, solver, div):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = [ctx.Process(target=_cube_worker, args=(r, port, overlap, solver, div, q))
             for r in range(2)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(2):
        rank, st, l2 = q.get(timeout=300)
        res[rank] = (st, l2)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return res


@pytest.mark.gpu
@pytest.mark.parametrize("solver,div", [("dplur", 1), ("dplur", 2), ("dplur", 4), ("bdplur", 1)])
def test_dplur_interior_boundary_split(oracle, solver, div):
    """A 2 x 2 x 2 cube of blocks, four per rank (connections inside a rank and across; div:
    the ranks meet at i-, j- or k-faces): DPLUR / BDPLUR sweeps that relax the cells away from
    the faces towards the other rank while the slabs of x travel on a second stream (default)
    against the sequential form (AGX_OVERLAP=0) bit for bit, and against the single-process
    oracle."""
    split, seq = _cube_run("1", solver, div), _cube_run("0", solver, div)
    kw = dict(CUBE_KW, matrix_solver=solver)
    case = synthetic.cube_blocks_case(n=(12, 10, 9), splits=(2, 2, 2), **kw)
    ref = Solver(oracle, case)
    for nn in range(2):
        ref.step(nn)
    ng = case.ng
    core = lambda a: a[ng:-ng, ng:-ng, ng:-ng]
    l2ref = np.array([h["l2"] ** 2 for h in ref.history])
    assert rel_err(split[0][1], l2ref) < RTOL
    for r in range(2):
        for gb, st in split[r][0].items():
            assert np.array_equal(core(st), core(seq[r][0][gb]))
            assert rel_err(core(st), core(ref.download("state", gb))) < RTOL
    ref.close()
