"""Host-side mirror of the reference's time-step loop around the C-ABI.

`Solver` plays the role of main.cpp:231-302 + logFileManager: it owns one
library context, uploads a Case (aither_amd.case.builder.Case), and calls
store_time_n / iterate exactly where the reference calls
mgSolution::StoreOldSolution (main.cpp:239) and mgSolution::Iterate
(main.cpp:249).  Residual normalisation follows PrintResiduals
(src/output.cpp:1028-1087).  PyTorch is not involved here; the same class
drives the product library ("agx_") and, in tests only, the CPU oracle
("ora_").
"""
import ctypes as C
import math
import numpy as np

from . import abi
from .case import builder as _b

EPS = 1.0e-30  # macros.hpp.in:20


class DistExchange:
    """An agx_exchange (include/aither_gfx950.h) on torch.distributed
    point-to-point with HOST buffers: what an MPI adapter would pass
    (MPI_Sendrecv / MPI_Allgather), here over gloo for the multi-process tests.
    The production transport is the library's own RCCL one
    (agx_rccl_exchange_create)."""

    def __init__(self, world, group=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.world, self.group = torch, dist, world, group
        self._swap = abi.SWAP_FN(self.swap)
        self._allgather = abi.ALLGATHER_FN(self.allgather)
        self.struct = abi.Exchange(None, self._swap, self._allgather, world, 1)

    def _view(self, ptr, n, dtype):
        ct = C.c_double if dtype == np.float64 else C.c_uint8
        arr = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ct)), shape=(max(int(n), 1),))
        return self.torch.from_numpy(arr)[:int(n)]

    def swap(self, user, n, slabs, stream):
        reqs = []
        for q in range(n):
            sl = slabs[q]
            if sl.count <= 0:
                continue
            reqs.append(self.dist.isend(self._view(sl.send, sl.count, np.float64),
                                        sl.peer, tag=sl.tag, group=self.group))
            reqs.append(self.dist.irecv(self._view(sl.recv, sl.count, np.float64),
                                        sl.peer, tag=sl.tag, group=self.group))
        for r in reqs:
            r.wait()
        return 0

    def allgather(self, user, send, recv, nbytes, stream):
        mine = self._view(send, nbytes, np.uint8).clone()
        parts = [self.torch.empty(nbytes, dtype=self.torch.uint8) for _ in range(self.world)]
        self.dist.all_gather(parts, mine, group=self.group)
        out = self._view(recv, nbytes * self.world, np.uint8)
        for r, p in enumerate(parts):
            out[r * nbytes:(r + 1) * nbytes] = p
        return 0


class DeviceSetup:
    """The volume-sized parts of the grid set-up in the library:
    plot3dBlock's metrics (agx_plot3d_metrics) and the nearest-wall search of
    CalcWallDistance (agx_nearest_wall_distance).  Handed to case.builder.build_case."""

    def __init__(self, api, device=0):
        self.api = api
        self.ctx = C.c_void_p()
        api.check(api.ctx_create(device, 0, C.byref(self.ctx)), "ctx_create")

    def close(self):
        if self.ctx:
            self.api.ctx_destroy(self.ctx)
            self.ctx = None

    def metrics(self, coords):
        x = np.ascontiguousarray(coords, dtype=np.float64)
        nk, nj, ni = (s - 1 for s in x.shape[:3])
        out = dict(vol=np.empty((nk, nj, ni, 1)), center=np.empty((nk, nj, ni, 3)),
                   farea_i=np.empty((nk, nj, ni + 1, 4)), farea_j=np.empty((nk, nj + 1, ni, 4)),
                   farea_k=np.empty((nk + 1, nj, ni, 4)), fcen_i=np.empty((nk, nj, ni + 1, 3)),
                   fcen_j=np.empty((nk, nj + 1, ni, 3)), fcen_k=np.empty((nk + 1, nj, ni, 3)))
        p = lambda a: a.ctypes.data_as(abi.c_dp)
        self.api.check(self.api.plot3d_metrics(
            self.ctx, ni, nj, nk, p(x), p(out["vol"]), p(out["center"]), p(out["farea_i"]),
            p(out["farea_j"]), p(out["farea_k"]), p(out["fcen_i"]), p(out["fcen_j"]),
            p(out["fcen_k"])), "plot3d_metrics")
        return out

    def nearest(self, cells, walls):
        cells = np.ascontiguousarray(cells, dtype=np.float64)
        walls = np.ascontiguousarray(walls, dtype=np.float64)
        dist = np.empty(len(cells))
        self.api.check(self.api.nearest_wall_distance(
            self.ctx, len(cells), cells.ctypes.data_as(abi.c_dp), len(walls),
            walls.ctypes.data_as(abi.c_dp), dist.ctypes.data_as(abi.c_dp)),
            "nearest_wall_distance")
        return dist


class Solver:
    def __init__(self, api, case, device=0, rank=0, stream=None, exchange=None,
                 rccl=None):
        """exchange: a DistExchange-like object (multi-process, host buffers);
        rccl: (id128 bytes, nranks, rank) for the library's RCCL transport.  With
        either, iterate() handles connections to other ranks and returns the
        globally reduced norms."""
        self.api, self.case, self.rank = api, case, rank
        self.ctx = C.c_void_p()
        api.check(api.ctx_create(device, rank, C.byref(self.ctx)), "ctx_create")
        if stream is not None:
            api.check(api.ctx_set_stream(self.ctx, C.c_void_p(stream)),
                      "ctx_set_stream")
        self._exchange = exchange
        if exchange is not None:
            api.check(api.set_exchange(self.ctx, C.byref(exchange.struct)), "set_exchange")
        if rccl is not None:
            self._rccl_id = C.create_string_buffer(bytes(rccl[0]), 128)
            api.check(api.rccl_exchange_create(self.ctx, self._rccl_id, rccl[1], rccl[2]),
                      "rccl_exchange_create")
        self.cfg = _b.config_struct(case)
        api.check(api.config_set(self.ctx, C.byref(self.cfg)), "config_set")
        self._keep = []
        self.block_ids = {}
        for gb, blk in enumerate(case.blocks):
            if blk.rank != rank:
                continue
            g = blk.geom
            bg = abi.BlockGeom()
            bg.ni, bg.nj, bg.nk, bg.ng = g.ni, g.nj, g.nk, g.ng
            bg.parent_block, bg.global_pos = blk.parent, blk.global_pos
            arrs = dict(farea_i=g.farea["i"].a, farea_j=g.farea["j"].a,
                        farea_k=g.farea["k"].a, vol=g.vol.a, center=g.center.a,
                        width_i=g.width["i"].a, width_j=g.width["j"].a,
                        width_k=g.width["k"].a, wall_dist=g.wall_dist.a)
            for name, a in arrs.items():
                a = np.ascontiguousarray(a, dtype=np.float64)
                self._keep.append(a)
                setattr(bg, name, a.ctypes.data_as(abi.c_dp))
            bid = C.c_int(-1)
            api.check(api.block_create(self.ctx, C.byref(bg), C.byref(bid)),
                      "block_create")
            self.block_ids[gb] = bid.value
            surfs = _b.surface_structs(case, gb)
            api.check(api.block_set_bcs(self.ctx, bid.value, len(surfs), surfs),
                      "block_set_bcs")
        self.conn_ids = []
        for conn in case.connections:
            if rank not in conn.rank:
                self.conn_ids.append(-1)
                continue
            cs = _b.connection_struct(conn)
            for side in range(2):
                if conn.rank[side] == rank:
                    cs.local_block[side] = self.block_ids[conn.block[side]]
            cid = C.c_int(-1)
            api.check(api.conn_create(self.ctx, C.byref(cs), C.byref(cid)),
                      "conn_create")
            self.conn_ids.append(cid.value)
        api.check(api.setup_finalize(self.ctx), "setup_finalize")
        for gb, bid in self.block_ids.items():
            self.upload("state", gb, case.blocks[gb].state)
        self.l2_first = None
        self.history = []

    # ------------------------------------------------------------------
    def close(self):
        if self.ctx:
            self.api.ctx_destroy(self.ctx)
            self.ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _shape(self, field, gb):
        g = self.case.blocks[gb].geom
        ng = g.ng
        n_eq = self.cfg.n_eq
        ghost = field in ("state", "update", "temperature", "viscosity")
        comps = n_eq if field in ("state", "residual", "cons_n", "update",
                                  "cons_nm1") else \
            9 if field == "vel_grad" else \
            3 if field in ("temp_grad", "dens_grad", "press_grad") else 1
        pad = 2 * ng if ghost else 0
        return (g.nk + pad, g.nj + pad, g.ni + pad, comps)

    def upload(self, field, gb, arr):
        a = np.ascontiguousarray(arr, dtype=np.float64)
        assert a.shape == self._shape(field, gb), (a.shape, self._shape(field, gb))
        if field == "state":
            rc = self.api.state_upload(self.ctx, self.block_ids[gb],
                                       a.ctypes.data_as(abi.c_dp))
        else:
            rc = self.api.field_upload(self.ctx, self.block_ids[gb],
                                       abi.FIELD[field],
                                       a.ctypes.data_as(abi.c_dp))
        self.api.check(rc, "upload")

    def output_pack(self, gb, names):
        """WriteFunFile's variables of block gb (reference names, abi.OUT), packed by the
        library: [nvar, nk, nj, ni], dimensional."""
        g = self.case.blocks[gb].geom
        ni, nj, nk = g.n
        ids = (C.c_int32 * len(names))(*[abi.OUT[n] for n in names])
        out = np.empty((len(names), nk, nj, ni))
        self.api.check(self.api.output_pack(self.ctx, self.block_ids[gb], len(names), ids,
                                            out.ctypes.data_as(abi.c_dp)), "output_pack")
        return out

    def restart_pack(self, gb, which=0):
        """WriteRestart's payload of block gb: [nk, nj, ni, n_eq + 1], dimensional."""
        g = self.case.blocks[gb].geom
        ni, nj, nk = g.n
        out = np.empty((nk, nj, ni, self.cfg.n_eq + 1))
        self.api.check(self.api.restart_pack(self.ctx, self.block_ids[gb], which,
                                             out.ctypes.data_as(abi.c_dp)), "restart_pack")
        return out

    def download(self, field, gb):
        out = np.empty(self._shape(field, gb))
        self.api.check(self.api.field_download(
            self.ctx, self.block_ids[gb], abi.FIELD[field],
            out.ctypes.data_as(abi.c_dp)), "field_download")
        return out

    # ------------------------------------------------------------------
    def store_time_n(self, nn):
        d = self.case.deck
        if d.need_to_store_time_n():
            also = int(d.is_multilevel_in_time() and nn == 0)
            self.api.check(self.api.store_time_n(self.ctx, also), "store_time_n")

    def iterate(self, mm, cfl):
        n_eq = self.cfg.n_eq
        l2 = np.zeros(n_eq)
        linf = abi.Linf()
        mres = C.c_double(0.0)
        self.api.check(self.api.iterate(
            self.ctx, mm, cfl, l2.ctypes.data_as(abi.c_dp), C.byref(linf),
            C.byref(mres)), "iterate")
        return l2, linf, mres.value

    def normalized(self, l2sq, nn, mm):
        """PrintResiduals (output.cpp:1028-1087) for the single-species case."""
        l2 = np.sqrt(l2sq)                 # residual::SquareRoot (main.cpp:268)
        if nn == 0 and mm == 0:
            self.l2_first = l2.copy()
        elif nn < 5 and mm == 0:
            if l2[0] > self.l2_first[0]:
                self.l2_first[0] = l2[0]
            self.l2_first[1:] = np.maximum(self.l2_first[1:], l2[1:])
        return (l2 + EPS) / (self.l2_first + EPS)

    def step(self, nn):
        d = self.case.deck
        cfl = d.cfl(nn)
        self.store_time_n(nn)
        out = None
        for mm in range(d.nonlinear_iterations):
            l2, linf, mres = self.iterate(mm, cfl)
            total = self.case.total_cells
            mres = math.sqrt(mres / (total * self.cfg.n_eq))   # main.cpp:271
            out = dict(nn=nn, mm=mm, l2=np.sqrt(l2),
                       norm=self.normalized(l2, nn, mm),
                       linf=(linf.linf, linf.block, linf.i, linf.j, linf.k,
                             linf.eqn), matrix=mres)
            self.history.append(out)
        return out

    def run(self, iterations):
        out = None
        for nn in range(iterations):
            out = self.step(nn)
        return out


class PhasedSolver(Solver):
    """One rank of a multi-process run: blocks whose `rank` matches are local,
    connections to other ranks exchange halo slabs between the phases of
    agx_iterate (include/aither_gfx950.h "phases").

    `exchange(items)` is supplied by the caller: items is a list of
    (peer_rank, tag, send_tensor, recv_tensor); it must complete all transfers
    (torch.distributed batch_isend_irecv over RCCL on GPUs, gloo on CPU).
    `alloc(n)` returns a float64 tensor of n elements on the backend's memory
    (torch.cuda for the product library, CPU for the test oracle).
    This is the drop-in replacement of the reference's MPI path
    (multiArray3d.hpp:830-866 SwapSliceParallel, utility.cpp:400-423).
    """

    def __init__(self, api, case, rank, exchange, alloc, device=0, stream=None):
        super().__init__(api, case, device=device, rank=rank, stream=stream)
        self.exchange, self.alloc = exchange, alloc
        self.remote = []
        for n, conn in enumerate(case.connections):
            if rank in conn.rank and conn.rank[0] != conn.rank[1]:
                side = 0 if conn.rank[0] == rank else 1
                cnt = self.api.halo_count(self.ctx, self.conn_ids[n], 0)
                self.remote.append(dict(
                    cid=self.conn_ids[n], tag=n, peer=conn.rank[1 - side],
                    send=self.alloc(cnt), recv=self.alloc(cnt)))

    def _halo(self, what):
        api = self.api
        api.check(api.halo_swap_local(self.ctx, what), "halo_swap_local")
        if not self.remote:
            return
        for r in self.remote:
            api.check(api.halo_pack(self.ctx, r["cid"], what,
                                    C.c_void_p(r["send"].data_ptr())), "halo_pack")
        api.check(api.sync(self.ctx), "sync")
        self.exchange([(r["peer"], r["tag"], r["send"], r["recv"])
                       for r in self.remote])
        for r in self.remote:
            api.check(api.halo_unpack(self.ctx, r["cid"], what,
                                      C.c_void_p(r["recv"].data_ptr())),
                      "halo_unpack")

    def iterate(self, mm, cfl):
        api, ctx, cfg = self.api, self.ctx, self.cfg
        l2 = np.zeros(cfg.n_eq)
        linf = abi.Linf()
        mres = C.c_double(0.0)
        l2p = l2.ctypes.data_as(abi.c_dp)
        api.check(api.phase_bc_faces(ctx), "phase_bc_faces")
        self._halo(abi.HALO_STATE)
        api.check(api.phase_bc_edges(ctx), "phase_bc_edges")
        api.check(api.phase_residual(ctx, mm, cfl), "phase_residual")
        if self.case.deck.is_implicit():
            block = cfg.matrix_solver in (abi.SOLVER["blusgs"], abi.SOLVER["bdplur"])
            if block and cfg.is_viscous:      # gridLevel.cpp:386-388
                self._halo(abi.HALO_VELGRAD_A)
                self._halo(abi.HALO_VELGRAD_B)
            if cfg.n_eq == 7:                 # SwapTurbVars gridLevel.cpp:389-392
                self._halo(abi.HALO_TURB)
            api.check(api.phase_implicit_begin(ctx), "phase_implicit_begin")
            lusgs = cfg.matrix_solver in (abi.SOLVER["lusgs"], abi.SOLVER["blusgs"])
            for s in range(cfg.matrix_sweeps):
                self._halo(abi.HALO_UPDATE)
                api.check(api.phase_relax_forward(ctx, s), "relax_forward")
                if lusgs:
                    self._halo(abi.HALO_UPDATE)
                    api.check(api.phase_relax_backward(ctx, s), "relax_backward")
            self._halo(abi.HALO_UPDATE)
            api.check(api.phase_matrix_residual(ctx, C.byref(mres)),
                      "matrix_residual")
            api.check(api.phase_implicit_update(ctx, mm, l2p, C.byref(linf)),
                      "implicit_update")
        else:
            api.check(api.phase_explicit_update(ctx, mm, l2p, C.byref(linf)),
                      "explicit_update")
        return l2, linf, mres.value


class MultigridSolver:
    """mgSolution with more than one grid level (mgSolution.cpp:160-262) on one rank: a Solver
    (a context of the library) per level and the transfers of aither_amd.case.multigrid between
    them.  levels: (cases, transfers) of multigrid.build_levels, finest first.  The cycle
    (V: 1 coarse visit, W: 2) is host logic over the library's phases and the agx_mg_* calls;
    the same driver runs the product library and the test oracle."""

    def __init__(self, api, cases, transfers, device=0, stream=None, rank=0, exchange=None):
        """rank / exchange: one rank of a multi-process run -- the blocks of every level live
        on the rank of their finest ancestor; exchange() returns a fresh DistExchange-like
        object per level (the levels' connections to other ranks go through the library's
        exchange table, agx_halo_exchange).  The norms a rank gets back are its own."""
        self.api = api
        self.levels = [Solver(api, c, device=device, rank=rank, stream=stream,
                              exchange=exchange() if exchange else None) for c in cases]
        self.transfers = []
        for trs in transfers:
            keep = []
            for t in trs:
                keep.append(dict(
                    tc=np.ascontiguousarray(t.to_coarse, dtype=np.int32),
                    vf=np.ascontiguousarray(t.vol_fac, dtype=np.float64),
                    cf=np.ascontiguousarray(t.coeffs, dtype=np.float64)))
            self.transfers.append(keep)
        deck = cases[0].deck
        self.case, self.cfg = cases[0], self.levels[0].cfg
        self.cycle_index = {"V": 1, "W": 2}[deck.multigrid_cycle]   # input.cpp (mgCycleIndex_)
        self.history = []
        self.l2_first = None

    # -- plumbing ----------------------------------------------------------
    def close(self):
        for s in self.levels:
            s.close()

    @property
    def block_ids(self):
        return self.levels[0].block_ids

    def download(self, field, gb, level=0):
        return self.levels[level].download(field, gb)

    def _p(self, a, ctype):
        return a.ctypes.data_as(C.POINTER(ctype))

    def _boundary_and_residual(self, s, mm, cfl):
        api = self.api
        api.check(api.phase_bc_faces(s.ctx), "phase_bc_faces")
        api.check(api.halo_exchange(s.ctx, abi.HALO_STATE), "halo_exchange")
        api.check(api.phase_bc_edges(s.ctx), "phase_bc_edges")
        api.check(api.phase_residual(s.ctx, mm, cfl), "phase_residual")

    # -- linearSolver::Relax (linearSolver.cpp:430-470, :509-536) ---------------
    def relax(self, lev, sweeps):
        api, s = self.api, self.levels[lev]
        lusgs = s.cfg.matrix_solver in (abi.SOLVER["lusgs"], abi.SOLVER["blusgs"])
        for ii in range(sweeps):
            api.check(api.halo_exchange(s.ctx, abi.HALO_UPDATE), "halo_exchange")
            api.check(api.phase_relax_forward(s.ctx, ii), "relax_forward")
            if lusgs:
                api.check(api.halo_exchange(s.ctx, abi.HALO_UPDATE), "halo_exchange")
                api.check(api.phase_relax_backward(s.ctx, ii), "relax_backward")
        api.check(api.halo_exchange(s.ctx, abi.HALO_UPDATE), "halo_exchange")
        mres = C.c_double(0.0)
        api.check(api.mg_matrix_residual(s.ctx, C.byref(mres)), "mg_matrix_residual")
        return mres.value

    # -- gridLevel::Restriction (gridLevel.cpp:538-595) --------------------------
    def restriction(self, fl, mm, cfl):
        api, f, c = self.api, self.levels[fl], self.levels[fl + 1]
        trs = self.transfers[fl]
        for gb, bid in f.block_ids.items():
            t = trs[gb]
            api.check(api.mg_restrict(f.ctx, c.ctx, bid, 0, self._p(t["tc"], C.c_int32),
                                      self._p(t["vf"], C.c_double)), "mg_restrict state")
            if mm == 0:          # need the solution at time n for the linear solvers
                api.check(api.store_time_n(c.ctx, 0), "store_time_n")
        self._boundary_and_residual(c, mm, cfl)
        api.check(api.mg_invert_diagonal(c.ctx), "mg_invert_diagonal")
        for gb, bid in f.block_ids.items():
            t = trs[gb]
            api.check(api.mg_restrict(f.ctx, c.ctx, bid, 1, self._p(t["tc"], C.c_int32),
                                      self._p(t["vf"], C.c_double)), "mg_restrict update")
        api.check(api.halo_exchange(c.ctx, abi.HALO_UPDATE), "halo_exchange")
        for gb, bid in f.block_ids.items():
            t = trs[gb]
            api.check(api.mg_restrict(f.ctx, c.ctx, bid, 2, self._p(t["tc"], C.c_int32),
                                      None), "mg_restrict forcing")

    # -- mgSolution::CycleAtLevel (mgSolution.cpp:160-205) ------------------------
    def cycle(self, fl, mm, cfl):
        api = self.api
        sweeps_all = self.cfg.matrix_sweeps
        if fl == len(self.levels) - 1:
            return self.relax(fl, sweeps_all)
        sweeps = max(sweeps_all // 2, 1)
        self.relax(fl, sweeps)
        cl = fl + 1
        self.restriction(fl, mm, cfl)
        api.check(api.mg_save_update(self.levels[cl].ctx), "mg_save_update")
        for _ in range(self.cycle_index):
            self.cycle(cl, mm, cfl)
        f, c = self.levels[fl], self.levels[cl]
        for gb, bid in f.block_ids.items():
            t = self.transfers[fl][gb]
            api.check(api.mg_prolong(c.ctx, f.ctx, bid, self._p(t["tc"], C.c_int32),
                                     self._p(t["cf"], C.c_double)), "mg_prolong")
        return self.relax(fl, sweeps)

    # -- mgSolution::Iterate / ImplicitUpdate (mgSolution.cpp:207-262) -------------
    def iterate(self, mm, cfl):
        api, f = self.api, self.levels[0]
        l2 = np.zeros(self.cfg.n_eq)
        linf = abi.Linf()
        self._boundary_and_residual(f, mm, cfl)
        api.check(api.phase_implicit_begin(f.ctx), "phase_implicit_begin")
        mres = self.cycle(0, mm, cfl)
        api.check(api.phase_implicit_update(f.ctx, mm, l2.ctypes.data_as(abi.c_dp),
                                            C.byref(linf)), "implicit_update")
        for s in self.levels[1:]:      # ResetDiagonal on every level (mgSolution.cpp:236-239)
            api.check(api.mg_reset_diagonal(s.ctx), "mg_reset_diagonal")
        return l2, linf, mres

    normalized = Solver.normalized

    def step(self, nn):
        d = self.case.deck
        cfl = d.cfl(nn)
        for s in self.levels:          # mgSolution::StoreOldSolution: every level
            s.store_time_n(nn)
        out = None
        for mm in range(d.nonlinear_iterations):
            l2, linf, mres = self.iterate(mm, cfl)
            total = self.case.total_cells
            mres = math.sqrt(mres / (total * self.cfg.n_eq))
            out = dict(nn=nn, mm=mm, l2=np.sqrt(l2), norm=self.normalized(l2, nn, mm),
                       linf=(linf.linf, linf.block, linf.i, linf.j, linf.k, linf.eqn),
                       matrix=mres)
            self.history.append(out)
        return out

    def run(self, iterations):
        out = None
        for nn in range(iterations):
            out = self.step(nn)
        return out
