#!/usr/bin/env python3
"""bench.py -- headline measurement of the accelerated hot path.

Metric (BASELINE.json): Mcell-updates/s per iteration (residual + LU-SGS) on a
256^3 block, % of the HBM roofline.  One iteration = one mgSolution::Iterate
call (src/mgSolution.cpp:246).  The default line is BASELINE configs[2]:
single 256^3 block, WENO5 + AUSMPW+ + viscous fluxes, implicit Euler with one
scalar LU-SGS sweep, state resident in HBM.  The same line carries, under
"extra"."rk4", the explicit-RK4 residual sweep (MUSCL thirdOrder + vanAlbada +
Roe, configs[1] at 256^3) that the north-star ">= 40 % of the HBM roofline"
clause is stated on.  `--workload rk4|dplur8|rans4` make those the headline instead
(rans4: BASELINE configs[4] in kind -- 4 blocks, k-omega SST 2003, BLU-SGS, a flat plate
with a boundary-layer start -- on the 7-equation build of the library).

  python bench.py --gpus N --steps K --warmup W

N > 1 (launched by torch.distributed.run, one rank per GPU): weak scaling, one
256^3 block per rank stacked along k and joined by interblock connections; the
library exchanges the ghost slabs itself with RCCL (grouped send/recv on its own
stream, no host sync between pack and unpack) and all-gathers the norms
(agx_rccl_exchange_create, include/aither_gfx950.h).
"""
import argparse
import ctypes
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.distributed as dist

import aither_amd
from aither_amd import abi
from aither_amd.case import synthetic
from aither_amd.case import builder as _b
from aither_amd.case import geometry as _geo
from aither_amd.case import connections as _conn
from aither_amd.solver import Solver, DistExchange

GLOO_GROUP = None          # second process group of an N > 1 run (fallback transport)
HBM_PEAK = 8.0e12          # B/s, MI355X_MICROARCH.md "HBM3E peak BW"
BYTES_STAGE = 296          # explicit stage, nEq = 5
BYTES_RESID_KERNEL = 216   # of which the residual kernel: state 5 + areas 12 +
                           # widths 3 + vol 1 read, residual 5 + dt 1 written
BYTES_LUSGS_ITER = 1296    # scalar LU-SGS viscous iteration
BYTES_DPLUR4_ITER = 1960   # inviscid DPLUR, 4 sweeps


def deck_kwargs(workload):
    if workload == "rans4":
        # BASELINE configs[4]: turbFlatPlate-style rans, SST 2003, BLU-SGS
        return dict(equation_set="rans", turbulence_model="sst2003",
                    face_reconstruction="thirdOrder", limiter="vanAlbada",
                    inviscid_flux="roe", time_integration="implicitEuler",
                    matrix_solver="blusgs", matrix_sweeps=1, cfl=50.0,
                    velocity=[50.0, 0.0, 0.0])
    if workload == "dplur8":
        return dict(face_reconstruction="thirdOrder", limiter="vanAlbada",
                    inviscid_flux="ausm", time_integration="implicitEuler",
                    matrix_solver="dplur", matrix_sweeps=4, cfl=10.0)
    if workload == "rk4":
        return dict(time_integration="rk4", cfl=0.5,
                    face_reconstruction="thirdOrder", limiter="vanAlbada",
                    inviscid_flux="roe")
    return dict(equation_set="navierStokes", face_reconstruction="weno",
                limiter="none", inviscid_flux="ausm",
                time_integration="implicitEuler", matrix_solver="lusgs",
                matrix_sweeps=1, cfl=10.0)


def rank_local_chain_case(rank, nranks, n, workload, dims=None, setup=None):
    """Block `rank` of a chain of nranks identical n^3 blocks stacked along k.
    Only this rank's block is built at full size; its neighbours are built
    four cells thick, which is all the ghost-geometry exchange reads."""
    kw = deck_kwargs(workload)
    if workload == "rans4":
        # 4 blocks in a row along the plate (i), wall on j-min, farfield elsewhere;
        # 4 / nranks blocks per rank
        bcs = {3: ("viscousWall", 2), 1: ("characteristic", 1), 2: ("characteristic", 1),
               4: ("characteristic", 1), 5: ("characteristic", 1), 6: ("characteristic", 1)}
        nb = (n // 2, n // 2, n // 4)
        case = synthetic.stacked_blocks_case(n=nb, nblocks=4, axis="i", stretch=1.15, bcs=bcs,
                                             ranks=[b * nranks // 4 for b in range(4)],
                                             amplitude=0.01, setup=setup, **kw)
        # a flat plate: free stream along the wall, a 1/7-th power boundary-layer profile
        # of thickness 0.05 over it, 1 % perturbation -- the k-omega model develops the
        # layer without leaving its stable range (oracle: residuals fall monotonically in
        # the mean over 120 iterations at CFL 50), so the bench times one continuous run
        for blk in case.blocks:
            y = blk.geom.center.a[..., 1]
            prof = np.minimum(1.0, (np.maximum(y, 0.0) / 0.05) ** (1.0 / 7.0))
            for q in (1, 2, 3):
                blk.state[..., q] *= prof
        case.total_cells = 4 * nb[0] * nb[1] * nb[2]
        return case
    if workload == "dplur8":
        # BASELINE configs[3]: 2 x 2 x 2 blocks of (n/2)^3 cells, 8 / nranks per rank
        case = synthetic.cube_blocks_case(n=(n // 2,) * 3, splits=(2, 2, 2),
                                          ranks=[b * nranks // 8 for b in range(8)],
                                          setup=setup, **kw)
        case.total_cells = 8 * (n // 2) ** 3
        return case
    bcs = None
    if workload == "lusgs":
        bcs = {3: ("viscousWall", 2), 1: ("characteristic", 1),
               2: ("characteristic", 1), 4: ("characteristic", 1)}
    if nranks == 1:
        return synthetic.single_block_case(dims or (n, n, n), stretch=1.2, bcs=bcs,
                                           amplitude=0.05, setup=setup, **kw)
    deck = synthetic.make_deck(**kw)
    thin = 4
    coords, all_bcs, dims = [], [], []
    for b in range(nranks):
        nk = n if b == rank else thin
        # the block occupies z in [b, b+1]; thin neighbours keep the spacing
        # 1/n next to the shared face
        if b == rank:
            z0, lz = float(b), 1.0
        elif b < rank:
            z0, lz = float(b + 1) - thin / n, thin / n
        else:
            z0, lz = float(b), thin / n
        x = synthetic.box_nodes(n, n, nk, 1.0, lengths=(1.0, 1.0, lz),
                                origin=(0.0, 0.0, z0))
        ax = (np.arange(n + 1) / n) ** 1.2
        x[..., 0] = ax.reshape(1, 1, n + 1)
        x[..., 1] = ax.reshape(1, n + 1, 1)
        coords.append(x)
        blk = dict(bcs or {})
        if b > 0:
            blk[5] = ("interblock", 6000 + (b - 1))
        if b < nranks - 1:
            blk[6] = ("interblock", 5000 + (b + 1))
        all_bcs.append(synthetic.box_surfaces(n, n, nk, blk))
        dims.append(nk)
    deck.bcs = all_bcs
    # only neighbours matter: drop the rest to save host memory
    keep = [b for b in range(nranks) if abs(b - rank) <= 1]
    sub_deck_bcs = [all_bcs[b] for b in keep]
    remap = {b: i for i, b in enumerate(keep)}
    for surfs in sub_deck_bcs:
        for s in surfs:
            if s.bc_type == "interblock":
                pb = s.partner_block()
                if pb in remap:
                    s.tag = 1000 * s.partner_surface() + remap[pb]
                else:      # connection to a block this rank does not hold
                    s.bc_type, s.tag = "slipWall", 0
    deck.bcs = sub_deck_bcs
    case = _b.build_case(None, deck=deck, coords=[coords[b] for b in keep],
                         ranks=keep, setup=setup)
    # rank ids are the global block ids (one block per rank)
    for c in case.connections:
        c.rank = [keep[c.block[0]], keep[c.block[1]]]
    for i, blk in enumerate(case.blocks):
        blk.rank = keep[i]
        blk.parent = blk.global_pos = keep[i]
    synthetic.perturbed_state(case, 0.05)
    case.total_cells = n * n * n * nranks
    return case


# reference's own CPU figures measured during the survey,
# quoted beside the port; they did not run on this box
REFERENCE_SURVEY = {
    "rk4": dict(value=0.058, unit="Mcell-updates/s", cores=1,
                what="reference binary, MUSCL + vanAlbada + Roe, RK4, 32^3, 1 of 8 "
                     "cores of the survey container (BASELINE.md section 2)"),
    "lusgs": dict(value=0.0077, unit="Mcell-updates/s", cores=1,
                  what="reference binary, shipped shockTube (WENO5 + Roe, bdf2, LU-SGS "
                       "1 sweep, 2 x 50 cells), 1 core of the survey container "
                       "(BASELINE.md section 2)"),
}


def cpu_baseline(workload, budget_s=12.0):
    """The CPU oracle (a from-scratch port of the reference algorithm, NOT the reference
    binary) on a bounded sample of the same workload: on ALL host cores of this box
    (OpenMP over k-planes / hyperplane cells, oracle/oracle.c; core count stated) at
    96^3, with the one-thread figure at 48^3 beside it."""
    lib = os.path.join(ROOT, "oracle", "liboracle.so")
    if not os.path.exists(lib):
        import subprocess
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")],
                              stdout=subprocess.DEVNULL)
    ora = abi.Api(ctypes.CDLL(lib), "ora_")
    gomp = ctypes.CDLL("libgomp.so.1")
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else os.cpu_count()

    def timed(n, threads, budget):
        gomp.omp_set_num_threads(threads)
        case = rank_local_chain_case(0, 1, n, workload)
        nonlin = case.deck.nonlinear_iterations
        s = Solver(ora, case)
        s.store_time_n(0)
        s.iterate(0, 0.5)                      # warm-up
        its, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < budget and its < 1 << 30:
            if its % nonlin == 0:
                s.store_time_n(its // nonlin)
            s.iterate(its % nonlin, case.deck.cfl(0))
            its += 1
        dt = time.perf_counter() - t0
        s.close()
        return dict(value=case.total_cells * its / dt / 1e6, unit="Mcell-updates/s",
                    cores=threads,
                    sample=f"{its} iterations of the same scheme on {case.total_cells} cells "
                           f"(size parameter {n}; {dt:.1f} s, oracle/liboracle.so, gcc -O2 "
                           f"-fopenmp, {threads} thread{'s' if threads > 1 else ''})")

    # the visible core count may be far above this job's share of the host (a GPU box shows
    # all cores of an 8-GPU node): probe a few thread counts on a small block, keep the best
    probe = {}
    for th in (8, 16, 32, 64):
        if th <= max(cores, 8):
            probe[th] = timed(64, th, 1.0)["value"]
    used = max(probe, key=probe.get)
    multi = timed(96, used, budget_s)
    multi["threads_probed"] = {str(k): round(v, 4) for k, v in probe.items()}
    multi["cores_visible"] = cores
    single = timed(48, 1, 0.5 * budget_s)
    gomp.omp_set_num_threads(used)
    out = dict(multi, kind="port", single_thread=single)
    if workload in REFERENCE_SURVEY:
        out["reference_survey"] = REFERENCE_SURVEY[workload]
    return out


# algorithmic bytes per cell of the passes of one scalar LU-SGS iteration
# (viscous: R 33 + F 43 + B 31 + M 35 + U 20 doubles = 1296 B) and
# the timing groups (agx_timing_get) that implement each pass
LUSGS_PASSES = [
    ("R  residual + dt + diagonal (k_residual_tile, k_visc_march, k_lusgs_prepare)",
     264, (0, 4, 5)),
    ("F+B  LU-SGS forward + backward sweep (k_lusgs_kp x 2)", 344 + 248, (3,)),
    ("M  matrix residual (k_matrix_resid_d2)", 280, (6,)),
    ("U  update + norms (k_update_d2, k_norm_final)", 160, (1,)),
]
# rans: the viscous residual face-once (k_rans_faces<D> + k_rans_cells), the half sweeps one
# pipelined launch each (k_lusgs_pipe), the rest one thread per cell; no per-pass byte model
# is claimed (bytes_per_cell = the 7-equation analogue of SURVEY 8d)
RANS_PASSES = [
    ("R  residual + sources + dt + block diagonal (k_inv_residual, k_block_diag_inv, "
     "k_rans_faces<0..2>, k_rans_cells)", 8 * (7 + 19 + 7 + 3 + 29), (0, 4, 5)),
    ("F+B  BLU-SGS half sweeps, a workgroup per k-plane (k_lusgs_pipe x 2)",
     2 * 8 * (7 * 7 + 19 + 29 + 7), (3,)),
    ("M  matrix residual", 8 * (7 * 7 + 19 + 29), (6,)),
    ("U  update + norms", 8 * (3 * 7), (1,)),
]
DPLUR_PASSES = [
    ("R  residual + dt + diagonal", 240, (0, 4, 5)),
    ("4 x DPLUR sweep (k_dplur)", 4 * 320, (3,)),
    ("M  matrix residual", 280, (6,)),
    ("U  update + norms", 160, (1,)),
]
WORKLOAD_TEXT = {
    "rk4": ("residual + explicit RK4 stage update",
            "single {n}^3 block per GPU, single-species air, MUSCL thirdOrder + "
            "vanAlbada + Roe, RK4 explicit, slip walls",
            "one mgSolution::Iterate call (one RK stage)"),
    "lusgs": ("residual+LU-SGS",
              "single {n}^3 block per GPU, single-species air, WENO5 + AUSMPW+ inviscid + "
              "viscous fluxes, implicit Euler, scalar LU-SGS 1 sweep, viscous wall + "
              "characteristic",
              "one nonlinear iteration (mgSolution::Iterate)"),
    "rans4": ("residual+BLU-SGS, rans SST 2003",
              "4 blocks of {h}x{h}x{q} cells in a row shared by the GPUs, k-omega SST 2003 "
              "(7 equations), MUSCL thirdOrder + vanAlbada + Roe, viscous wall, implicit "
              "Euler, BLU-SGS 1 sweep (libaither_gfx950_rans.so: face-once viscous kernels, "
              "pipelined record sweeps)",
              "one nonlinear iteration (mgSolution::Iterate)"),
    "dplur8": ("residual + DPLUR",
               "2x2x2 blocks of {h}^3 cells shared by the GPUs, Euler MUSCL + AUSMPW+, "
               "implicit Euler DPLUR 4 sweeps, slip walls",
               "one nonlinear iteration (mgSolution::Iterate)"),
}


def run_workload(args, workload, api, world, rank, local_rank):
    """Set the case up, warm up, time `args.steps` iterations; returns the
    measured quantities of this rank (rank 0 holds the max-over-ranks time)."""
    n = args.size
    dims = tuple(int(v) for v in args.dims.split(",")) if args.dims else None
    # the volume-sized parts of the grid set-up (metrics, wall distance) run in the library
    # too (agx_plot3d_metrics, agx_nearest_wall_distance; outside the timed region)
    from aither_amd.solver import DeviceSetup
    setup = DeviceSetup(api, device=local_rank)
    case = rank_local_chain_case(rank, world, n, workload, dims, setup=setup)
    setup.close()
    nonlin = case.deck.nonlinear_iterations
    transport = "none"
    if world > 1 and args.backend == "nccl":
        # the library's own transport: RCCL on its stream (agx_rccl_exchange_create);
        # torch.distributed only carries the 128-byte id to the other ranks
        idbuf = ctypes.create_string_buffer(128)
        if rank == 0:
            api.check(api.rccl_unique_id(idbuf), "rccl_unique_id")
        t = torch.frombuffer(bytearray(idbuf.raw), dtype=torch.uint8).cuda()
        dist.broadcast(t, src=0)
        sol, ok = None, 1
        try:
            sol = Solver(api, case, device=local_rank, rank=rank,
                         rccl=(bytes(t.cpu().numpy().tobytes()), world, rank))
            sol.store_time_n(0)                      # one iteration proves the transport
            sol.iterate(0, case.deck.cfl(0))
        except RuntimeError as exc:
            ok = 0
            print(f"[bench rank {rank}] in-library RCCL transport failed: {exc}",
                  file=sys.stderr, flush=True)
        flag = torch.tensor([ok], dtype=torch.int32, device="cuda")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        transport = "rccl (in-library, agx_rccl_exchange_create)"
        if int(flag.item()) == 0:
            if sol is not None:
                sol.close()
            if not args.allow_host_fallback:
                # a broken transport is a failure, not a slower number
                raise SystemExit("in-library RCCL transport failed on at least one rank "
                                 "(--allow-host-fallback would stage the slabs over gloo)")
            # every rank falls back together: host-staged slabs over the gloo group
            sol = Solver(api, case, device=local_rank, rank=rank,
                         exchange=DistExchange(world, group=GLOO_GROUP))
            transport = "host buffers over gloo (fallback: RCCL transport failed)"
    elif world > 1:
        transport = "host buffers over gloo (rehearsal)"
        # rehearsal: the same in-library path with host-staged slabs over gloo
        sol = Solver(api, case, device=local_rank, rank=rank, exchange=DistExchange(world))
    else:
        sol = Solver(api, case, device=local_rank)

    def one_step(it):
        mm = it % nonlin
        if mm == 0:
            sol.store_time_n(it // nonlin)
        # with an exchange installed the norms come back reduced over the ranks
        # (main.cpp:254-264): inside iterate, hence inside the timed region
        return sol.iterate(mm, case.deck.cfl(it // nonlin))

    for it in range(args.warmup):
        one_step(it)
    api.check(api.timing_reset(sol.ctx))
    api.check(api.timing_enable(sol.ctx, 1))
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for it in range(args.warmup, args.warmup + args.steps):
        l2, linf, mres = one_step(it)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    api.check(api.timing_enable(sol.ctx, 0))
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64,
                         device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if not np.all(np.isfinite(l2)) and not os.environ.get("AGX_ABLATE"):
        raise SystemExit("non-finite residual")

    groups = []
    for g in range(7):
        ms, cnt = ctypes.c_double(0.0), ctypes.c_int64(0)
        api.check(api.timing_get(sol.ctx, g, ctypes.byref(ms), ctypes.byref(cnt)))
        groups.append((ms.value, cnt.value))
    sol.close()
    cells_rank = dims[0] * dims[1] * dims[2] if dims else n ** 3
    total_cells = cells_rank * world
    if workload == "dplur8":      # strong scaling: the 8 blocks are divided
        total_cells = 8 * (n // 2) ** 3
        cells_rank = total_cells // world
    if workload == "rans4":
        total_cells = case.total_cells
        cells_rank = total_cells // world
    return dict(workload=workload, n=n, elapsed=elapsed, groups=groups,
                cells_rank=cells_rank, total_cells=total_cells, transport=transport)


def traffic_entry(workload, cells_rank):
    """HBM bytes from the PMC counters (separate rocprofv3 --pmc passes,
    FETCH_SIZE x 2 per the gfx950 correction, WRITE_SIZE as read), collected by
    tools/profile_round.sh into profiles/hbm_traffic.json."""
    tp = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    if not os.path.exists(tp):
        return None
    with open(tp) as fh:
        tr = json.load(fh).get(workload)
    if tr and tr.get("cells") == cells_rank:
        return tr
    return None


def build_line(args, res, world):
    workload, n, steps = res["workload"], res["n"], args.steps
    cells_rank, groups = res["cells_rank"], res["groups"]
    value = res["total_cells"] * steps / res["elapsed"] / 1e6
    tr = traffic_entry(workload, cells_rank)
    if workload == "rk4":
        t_res, n_res = groups[0]
        t_upd, n_upd = groups[1]
        fused = (os.environ.get("AGX_KERNEL", "tile") != "gather"
                 and os.environ.get("AGX_NO_FUSE", "0") in ("", "0"))
        # dominant kernel: one launch per RK stage per block.  Fused form
        # (default): k_residual_tile does the whole stage (residual, dt,
        # update, norms) = 296 B/cell; unfused: residual kernel 216 B/cell
        # followed by k_update.
        bpc = BYTES_STAGE if fused else BYTES_RESID_KERNEL
        achieved = bpc * cells_rank / (t_res * 1e-3) / 1e9
        stage_ms = t_res * n_res / max(steps, 1) + t_upd
        roof = dict(bound="hbm",
                    kernel=("k_residual_tile<MUSCL,vanAlbada,Roe,fused>" if fused
                            else "residual kernel (unfused)"),
                    achieved=achieved, peak=HBM_PEAK / 1e9, unit="GB/s",
                    frac=achieved * 1e9 / HBM_PEAK,
                    traffic=tr["bytes_per_launch"] if tr else None,
                    bytes_per_cell=bpc, avg_launch_ms=t_res,
                    stage=dict(bytes_per_cell=BYTES_STAGE, device_ms=stage_ms,
                               frac=BYTES_STAGE * cells_rank /
                               (stage_ms * 1e-3) / HBM_PEAK))
    else:
        passes = (LUSGS_PASSES if workload == "lusgs" else
                  RANS_PASSES if workload == "rans4" else DPLUR_PASSES)
        kern, dev_ms = [], 0.0
        for name, bpc, gs in passes:
            ms = sum(groups[g][0] * groups[g][1] for g in gs) / max(steps, 1)
            dev_ms += ms
            ent = dict(pass_=name, bytes_per_cell=bpc, device_ms=ms,
                       frac=(bpc * cells_rank / (ms * 1e-3) / HBM_PEAK) if ms > 0 else None)
            if tr and "passes" in tr:
                ent["traffic"] = tr["passes"].get(name.split()[0])
            kern.append(ent)
        dev_ms += groups[2][0] * groups[2][1] / max(steps, 1)      # ghost cells / halo
        total_bpc = sum(p[1] for p in passes)
        # the dominant kernel of this iteration: the sweeps
        _, sw_bpc, sw_g = passes[1]
        sw_ms = sum(groups[g][0] * groups[g][1] for g in sw_g) / max(steps, 1)
        achieved = sw_bpc * cells_rank / (sw_ms * 1e-3) / 1e9
        roof = dict(bound="hbm", kernel=passes[1][0],
                    achieved=achieved, peak=HBM_PEAK / 1e9, unit="GB/s",
                    frac=achieved * 1e9 / HBM_PEAK,
                    traffic=(tr["passes"].get(passes[1][0].split()[0])
                             if tr and "passes" in tr else None),
                    bytes_per_cell=sw_bpc, avg_launch_ms=sw_ms,
                    iteration=dict(bytes_per_cell=total_bpc, device_ms=dev_ms,
                                   achieved=total_bpc * cells_rank / (dev_ms * 1e-3) / 1e9,
                                   frac=total_bpc * cells_rank / (dev_ms * 1e-3) / HBM_PEAK,
                                   traffic=tr["bytes_per_iteration"] if tr else None),
                    passes=kern)
    metric, wl, itn = WORKLOAD_TEXT[workload]
    return {
        "metric": f"Mcell-updates/sec per iteration ({metric}), {n}^3 block; % HBM roofline",
        "value": value, "unit": "Mcell-updates/s", "n_gpus": world,
        "steps": steps, "warmup": args.warmup,
        "ms_per_step": res["elapsed"] / steps * 1e3,
        "higher_is_better": True,
        "scaling": "strong" if workload in ("dplur8", "rans4") else "weak",
        "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": wl.format(n=n, h=n // 2, q=n // 4), "iteration": itn,
                   "blocks": 8 if workload == "dplur8" else 4 if workload == "rans4" else world,
                   "cells_per_gpu": cells_rank,
                   "halo": ("none" if world == 1 else
                            "RCCL grouped send/recv + all-gather of the norms on the "
                            "library's stream" if res.get("transport", "").startswith("rccl") else
                            res.get("transport", "host-staged slabs over gloo"))},
        "roofline": roof,
    }


def launch_ranks(args):
    """`python bench.py --gpus N` without a launcher: this process has not touched the
    GPU (importing torch does not); it starts N fresh ranks under
    torch.distributed.run as a CHILD process, relays their output (rank 0 prints the
    JSON line) and exits with the child's code.  Nothing is exec'ed over a process
    that holds the GPU."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
           f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    return subprocess.call(cmd, env=env)


def rendezvous_only(args, world, rank):
    """Launcher / rank-plumbing check that needs no GPU: the ranks meet on gloo, each
    builds its rank-local case, and rank 0 prints a line without a measurement."""
    dist.init_process_group("gloo", rank=rank, world_size=world)
    case = rank_local_chain_case(rank, world, args.size, args.workload)
    mine = sum(1 for b in case.blocks if b.rank == rank)
    t = torch.tensor([mine], dtype=torch.int64)
    dist.all_reduce(t)
    if rank == 0:
        print(json.dumps({"metric": "rendezvous only (no measurement)", "value": None,
                          "n_gpus": world, "rendezvous_only": True,
                          "blocks_held": int(t.item()),
                          "remote_connections": sum(1 for c in case.connections
                                                    if c.rank[0] != c.rank[1])}))
    dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--workload", choices=["rk4", "lusgs", "dplur8", "rans4"], default="lusgs")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true",
                    help="skip the extra.rk4 (N = 1) / extra.dplur8 (N > 1) measurement")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl",
                    help="gloo stages the halo slabs through the host (rehearsal of "
                         "the multi-rank path with several ranks on one GPU)")
    ap.add_argument("--allow-host-fallback", action="store_true",
                    help="if the in-library RCCL transport fails, stage the slabs through "
                         "the host over gloo instead of failing")
    ap.add_argument("--rendezvous-only", action="store_true",
                    help="launch the ranks, meet on gloo, build the rank-local cases and "
                         "stop (launcher check; needs no GPU)")
    ap.add_argument("--dims", default=None,
                    help="ni,nj,nk of a non-cubic block (kernel experiments only)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: run "
                         f"`python bench.py --gpus N` or `python -m torch.distributed.run "
                         f"--nproc-per-node N bench.py --gpus N`")
    if args.rendezvous_only:
        return rendezvous_only(args, world, rank)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback)")
    if args.backend == "gloo":
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    if world > 1 and args.backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world,
                                device_id=torch.device("cuda", local_rank))
        if args.allow_host_fallback:
            global GLOO_GROUP
            GLOO_GROUP = dist.new_group(backend="gloo")
    elif world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)

    api = aither_amd.load(7 if args.workload == "rans4" else 5)
    res = run_workload(args, args.workload, api, world, rank, local_rank)
    extra = None
    if args.workload == "lusgs" and world == 1 and not args.no_extra and not args.dims:
        # the explicit-RK4 residual sweep the ">= 40 % of roofline" target is stated on
        # (48 warm-up stages: the stage kernel runs 1.9 ms right after set-up and settles
        # at 1.46-1.5 ms some 60 ms later -- clocks and TLBs -- see the per-launch times in
        # profiles/r03_rk4_kernel_trace_durations.txt; 8 stages were 13 ms)
        rk_args = argparse.Namespace(**vars(args))
        rk_args.steps, rk_args.warmup = max(args.steps, 80), max(args.warmup, 48)
        extra = ("rk4", rk_args, run_workload(rk_args, "rk4", api, world, rank, local_rank))
    elif (args.workload == "lusgs" and world > 1 and 8 % world == 0 and not args.no_extra
          and not args.dims):
        # the 8-block DPLUR case (BASELINE configs[3], strong scaling) the north-star
        # ">= 6x at 8 GPUs" clause is stated on rides in the same line
        # (a failure of this second case must not cost the line its headline value)
        try:
            extra = ("dplur8", args, run_workload(args, "dplur8", api, world, rank, local_rank))
        except (RuntimeError, SystemExit) as exc:
            extra = ("dplur8", args, None, f"{type(exc).__name__}: {exc}")
            print(f"[bench rank {rank}] extra.dplur8 failed: {exc}", file=sys.stderr, flush=True)
    if rank == 0:
        out = build_line(args, res, world)
        if extra is not None and extra[2] is None:
            out["extra"] = {extra[0]: {"error": extra[3]}}
        elif extra is not None:
            e = build_line(extra[1], extra[2], world)
            out["extra"] = {extra[0]: {k: e[k] for k in (
                "metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step",
                "scaling", "config", "roofline")}}
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(args.workload)
            if extra is not None:
                out["extra"]["rk4"]["cpu_baseline"] = cpu_baseline("rk4", budget_s=8.0)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
