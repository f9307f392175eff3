"""Assemble a runnable case: what the reference's main.cpp:100-203 produces
before its time-step loop (input parsing, nondimensionalisation, grid metrics,
ghost geometry, connections, wall distance, initial state), as plain arrays in
the reference's host layout ready to cross the C-ABI.
"""
from dataclasses import dataclass, field
import math
import os
import numpy as np

from . import fluid as _fluid
from . import geometry as _geo
from . import connections as _conn
from .inputfile import parse_input, InputDeck, Surface, State
from .. import abi


@dataclass
class Block:
    geom: "_geo.BlockGeometry"
    surfaces: list
    state: np.ndarray          # [nk+2g, nj+2g, ni+2g, n_eq]
    parent: int = 0
    global_pos: int = 0
    rank: int = 0
    local_pos: int = 0


@dataclass
class Case:
    deck: InputDeck
    gas: "_fluid.Gas"
    blocks: list
    connections: list
    total_cells: int
    n_eq: int = 5

    @property
    def ng(self):
        return self.deck.num_ghost_layers()


def nondim_state(deck, gas, st):
    """icState::Nondimensionalize (inputStates.cpp:464-473) and the
    per-BC variants (:590-599, :674-681, :775-783)."""
    r_ref, a_ref, t_ref, l_ref = gas.rho_ref, gas.a_ref, gas.t_ref, gas.l_ref
    out = abi.BcState()
    kind = st.kind
    # DEFAULT_TURB_INTENSITY / DEFAULT_EDDY_VISC_RATIO inputStates.hpp:41-42
    out.turb_intensity = st.get("turbulenceIntensity", 0.01)
    out.eddy_visc_ratio = st.get("eddyViscosityRatio", 0.01)
    if kind in ("icState", "characteristic", "supersonicInflow", "inlet",
                "subsonicInflow"):
        vel = st.get("velocity", [0.0, 0.0, 0.0])
        for c in range(3):
            out.velocity[c] = vel[c] / a_ref
        out.density = st.get("density", 0.0) / r_ref
        out.pressure = st.get("pressure", 0.0) / (r_ref * a_ref * a_ref)
        if kind == "inlet":
            out.is_nonreflecting = int(bool(st.get("nonreflecting", False)))
            out.length_scale = st.get("lengthScale", 0.0) / l_ref
    elif kind == "stagnationInlet":
        d = st.get("direction", [0.0, 0.0, 0.0])
        for c in range(3):
            out.direction[c] = d[c]       # Normalize() result is discarded (:594)
        out.stagnation_pressure = st.get("p0") / (r_ref * a_ref * a_ref)
        out.stagnation_temperature = st.get("t0") / t_ref
    elif kind in ("pressureOutlet", "subsonicOutflow"):
        out.pressure = st.get("pressure") / (r_ref * a_ref * a_ref)
        out.is_nonreflecting = int(bool(st.get("nonreflecting", False)))
        out.length_scale = st.get("lengthScale", 0.0) / l_ref
    elif kind == "viscousWall":
        vel = st.get("velocity", [0.0, 0.0, 0.0])
        for c in range(3):
            out.velocity[c] = vel[c] / a_ref
        if st.get("temperature") is not None:
            out.is_isothermal = 1
            out.wall_temperature = st.get("temperature") / t_ref
        if st.get("heatFlux") is not None:
            out.is_heat_flux = 1
            out.wall_heat_flux = st.get("heatFlux") / ((a_ref / l_ref) ** 3.0)
        treatment = st.get("wallTreatment", "lowRe")
        if treatment not in ("lowRe", "wallLaw"):
            raise NotImplementedError(f"wallTreatment {treatment}")
        if treatment == "wallLaw":
            if not deck.is_rans():
                raise NotImplementedError("wall functions: rans runs")
            out.is_wall_law = 1
            out.von_karman = st.get("vonKarmen", 0.41)        # inputStates.hpp:343-344
            out.wall_constant = st.get("wallConstant", 5.5)
    elif kind == "periodic":
        pass
    else:
        raise NotImplementedError(f"boundary state {kind}")
    return out


def cloud_states(deck, gas, path):
    """CalcTreeFromCloud (utility.cpp:521-605): the points and nondimensional primitive
    states of a cloud file (count, species line, then x y z rho u v w p tke omega mf...)."""
    with open(path) as fh:
        lines = [ln.strip() for ln in fh if ln.strip()]
    npts = int(lines[0].split()[0])
    species = lines[1].split()
    if len(species) != 1:
        raise NotImplementedError("multi-species cloud file")
    rows = np.array([[float(t) for t in ln.split()] for ln in lines[2:2 + npts]])
    if rows.shape != (npts, 10 + len(species)):
        raise ValueError("cloud file: wrong number of columns")
    a_ref, r_ref = gas.a_ref, deck.rho_ref
    pts = rows[:, 0:3] / deck.l_ref
    st = np.empty((npts, 7 if deck.is_rans() else 5))
    st[:, 0] = rows[:, 3] / r_ref * rows[:, 10]
    st[:, 1:4] = rows[:, 4:7] / a_ref
    st[:, 4] = rows[:, 7] / (r_ref * a_ref * a_ref)
    if deck.is_rans():
        mu_ref = gas.visc_c1 * gas.t_ref ** 1.5 / (gas.t_ref + gas.visc_s)
        st[:, 5] = rows[:, 8] / (a_ref * a_ref)
        st[:, 6] = rows[:, 9] * mu_ref / (r_ref * a_ref * a_ref)
    return pts, st


def initial_from_cloud(deck, gas, geom, path):
    """procBlock::InitializeStates, file branch (procBlock.cpp:287-320): every physical
    cell takes the state of the cloud point nearest to its centre."""
    from scipy.spatial import cKDTree
    pts, st = cloud_states(deck, gas, path)
    ng = geom.ng
    cen = geom.center.a[ng:ng + geom.nk, ng:ng + geom.nj, ng:ng + geom.ni, :]
    _, idx = cKDTree(pts).query(cen.reshape(-1, 3))
    return st[idx].reshape(geom.nk, geom.nj, geom.ni, -1)


def initial_primitive(deck, gas, block):
    """primitive::NondimensionalInitialize (primitive.cpp:40-64)."""
    ic = deck.ic_for_block(block)
    s = nondim_state(deck, gas, State("icState", ic.params))
    prim = [1.0 * s.density, s.velocity[0], s.velocity[1], s.velocity[2], s.pressure]
    if deck.is_rans():
        prim += list(farfield_turbulence(gas, prim, s.velocity, s.turb_intensity,
                                         s.eddy_visc_ratio))
    return np.array(prim)


TURB_MIN = 1.0e-20      # turbModel::TkeMin / OmegaMin turbulence.hpp:72-73


def farfield_turbulence(gas, prim, vel, intensity, ratio):
    """primitive::ApplyFarfieldTurbBC (primitive.cpp:83-98): k and omega from a
    turbulence intensity and an eddy-viscosity ratio, then LimitTurb."""
    vmag = math.sqrt(vel[0] ** 2 + vel[1] ** 2 + vel[2] ** 2)
    tke = 1.5 * (intensity * vmag) ** 2.0
    t = prim[4] / (prim[0] * gas.gas_constant)              # idealGas::Temperature
    mu_ref = gas.visc_c1 * gas.t_ref ** 1.5 / (gas.t_ref + gas.visc_s)
    td = t * gas.t_ref
    mu = gas.visc_c1 * td ** 1.5 / (td + gas.visc_s) / mu_ref   # sutherland::Viscosity
    omega = prim[0] * tke / (ratio * mu)
    return max(tke, TURB_MIN), max(omega, TURB_MIN)


def _wall_distance(blocks, nearest=None):
    """main.cpp:191-203: nearest viscous-wall face centre (k-d tree), then the
    ghost-cell rule of procBlock::CalcWallDistance (procBlock.cpp:6030-6107).
    nearest(cell_centres, wall_points) -> distances: the library's
    agx_nearest_wall_distance (solver.DeviceSetup); default: scipy's k-d tree."""
    pts = []
    for b in blocks:
        g = b.geom
        for s in b.surfaces:
            if s.bc_type != "viscousWall":
                continue
            f = s.dir3()
            rng = {"i": s.rng("i"), "j": s.rng("j"), "k": s.rng("k")}
            pts.append(g.fcen[f].v(rng["i"], rng["j"], rng["k"]).reshape(-1, 3))
    if not pts:
        return
    walls = np.ascontiguousarray(np.concatenate(pts))
    if nearest is None:
        from scipy.spatial import cKDTree
        tree = cKDTree(walls)
        nearest = lambda cen, _w: tree.query(cen)[0]
    for b in blocks:
        g = b.geom
        ni, nj, nk, ng = g.ni, g.nj, g.nk, g.ng
        cen = np.ascontiguousarray(g.center.phys().reshape(-1, 3))
        dist = nearest(cen, walls)
        wd = g.wall_dist
        wd.phys()[..., 0] = dist.reshape(nk, nj, ni)
        n = {"i": ni, "j": nj, "k": nk}
        for s in b.surfaces:
            d = s.dir3()
            st = s.surface_type()
            r = {q: s.rng(q) for q in "ijk"}
            for layer in range(1, ng + 1):
                if st % 2 == 1:
                    gidx = -layer
                    src = (layer - 1) if s.bc_type == "viscousWall" else 0
                else:
                    gidx = n[d] + layer - 1
                    src = (n[d] - layer) if s.bc_type == "viscousWall" \
                        else n[d] - 1
                sign = -1.0 if s.bc_type == "viscousWall" else 1.0
                rg, rs = dict(r), dict(r)
                rg[d] = (gidx, gidx + 1)
                rs[d] = (src, src + 1)
                wd.v(rg["i"], rg["j"], rg["k"])[...] = \
                    sign * wd.v(rs["i"], rs["j"], rs["k"])


def build_case(inp_path, grid_dir=None, deck=None, coords=None, ranks=None, setup=None):
    """Build a Case from an .inp file (and its .xyz grid).  `deck`/`coords`
    may be given directly for synthetic cases.  setup: a solver.DeviceSetup -- the
    volume-sized parts (metrics, wall distance) then run in the library."""
    if deck is None:
        deck = parse_input(inp_path)
    deck.validate()
    fluid_name = deck.fluids[0] if deck.fluids else "air"
    gas = _fluid.make_gas(fluid_name, deck.t_ref, deck.rho_ref, deck.l_ref)
    if coords is None:
        base = grid_dir or os.path.dirname(os.path.abspath(inp_path))
        coords = _geo.read_plot3d(os.path.join(base, deck.grid_name + ".xyz"),
                                  deck.l_ref)
    if len(coords) != len(deck.bcs):
        raise ValueError("number of grid blocks and BC blocks differ")
    ng = deck.num_ghost_layers()
    nblk = len(coords)
    ranks = ranks or [0] * nblk
    local_pos = []
    counts = {}
    for r in ranks:
        local_pos.append(counts.get(r, 0))
        counts[r] = counts.get(r, 0) + 1
    blocks = []
    total = 0
    for b, x in enumerate(coords):
        g = _geo.BlockGeometry(x, ng, metrics=setup.metrics if setup else None)
        g.assign_ghost_geom(deck.bcs[b])
        ic_file = deck.ic_for_block(b).get("file")
        if ic_file is not None:       # (relative to the case directory, as the reference runs)
            base = grid_dir or os.path.dirname(os.path.abspath(inp_path))
            prim = initial_from_cloud(deck, gas, g, os.path.join(base, ic_file))
        else:
            prim = initial_primitive(deck, gas, b)
        st = np.zeros((g.nk + 2 * ng, g.nj + 2 * ng, g.ni + 2 * ng, prim.shape[-1]))
        st[ng:ng + g.nk, ng:ng + g.nj, ng:ng + g.ni, :] = prim
        blocks.append(Block(g, deck.bcs[b], st, b, b, ranks[b], local_pos[b]))
        total += g.ni * g.nj * g.nk
    conns = _conn.find_connections(deck.bcs, coords, deck, ranks, local_pos)
    geoms = [b.geom for b in blocks]
    for c in conns:
        if c.is_interblock:
            _conn.swap_geom(c, geoms)
    for b in blocks:
        b.geom.assign_ghost_geom_edge()
        b.geom.calc_cell_widths()
    if deck.is_viscous():
        _wall_distance(blocks, setup.nearest if setup else None)
        # SwapWallDist (gridLevel.cpp:261-285)
        for c in conns:
            _swap_cell_field(c, blocks, lambda blk: blk.geom.wall_dist.a, ng)
    return Case(deck, gas, blocks, conns, total, n_eq=7 if deck.is_rans() else 5)


def _swap_cell_field(conn, blocks, getter, ng):
    b0, b1 = blocks[conn.block[0]], blocks[conn.block[1]]
    a0, a1 = getter(b0), getter(b1)
    nc = a0.shape[-1]
    f0, f1 = a0.reshape(-1, nc), a1.reshape(-1, nc)
    dst0, src1, _ = _conn.insert_maps(conn, True, ng, b0.geom.n, b1.geom.n)
    dst1, src0, _ = _conn.insert_maps(conn, False, ng, b1.geom.n, b0.geom.n)
    s1 = f1[src1].copy()
    s0 = f0[src0].copy()
    f0[dst0] = s1
    f1[dst1] = s0


# ---------------------------------------------------------------------------
def config_struct(case):
    d, g = case.deck, case.gas
    cfg = abi.Config()
    cfg.n_eq = case.n_eq
    cfg.n_ghost = d.num_ghost_layers()
    if d.face_reconstruction == "constant":
        cfg.recon = abi.RECON["constant"]
    elif d.using_muscl():
        cfg.recon = abi.RECON["muscl"]
    else:
        cfg.recon = abi.RECON[d.face_reconstruction]
    cfg.limiter = abi.LIMITER[d.limiter]
    cfg.inviscid_flux = abi.FLUX[d.inviscid_flux]
    cfg.is_viscous = int(d.is_viscous())
    cfg.time_integration = abi.TIME[d.time_integration]
    cfg.matrix_solver = abi.SOLVER[d.matrix_solver]
    cfg.matrix_sweeps = d.matrix_sweeps
    cfg.nonlinear_iterations = d.nonlinear_iterations
    cfg.equation_set = abi.EQN[d.equation_set]
    cfg.inv_flux_jacobian = abi.JACOBIAN[d.inv_flux_jac]
    cfg.viscous_recon = abi.VISC_RECON[d.viscous_face_reconstruction]
    cfg.turbulence_model = abi.TURB.get(d.turbulence_model, 4)
    cfg.kappa = d.kappa
    cfg.theta, cfg.zeta = d.theta, d.zeta
    cfg.matrix_relaxation = d.matrix_relaxation
    cfg.dual_time_cfl = d.dual_time_cfl
    cfg.dt_nondim = d.dt * g.a_ref / g.l_ref if d.dt > 0.0 else -1.0
    cfg.viscous_cfl_coeff = d.viscous_cfl_coefficient()
    for name, _ in abi.Gas._fields_:
        setattr(cfg.gas, name, getattr(g, name))
    return cfg


def surface_structs(case, block):
    blk = case.blocks[block]
    arr = (abi.BcSurface * len(blk.surfaces))()
    for n, s in enumerate(blk.surfaces):
        a = arr[n]
        a.bc_type = abi.BC[s.bc_type]
        a.imin, a.imax, a.jmin, a.jmax = s.imin, s.imax, s.jmin, s.jmax
        a.kmin, a.kmax, a.tag = s.kmin, s.kmax, s.tag
        if s.bc_type in ("characteristic", "supersonicInflow", "inlet",
                         "stagnationInlet", "pressureOutlet", "viscousWall"):
            try:
                st = case.deck.bc_data(s.tag)
            except KeyError:
                if s.bc_type == "viscousWall":
                    st = State("viscousWall", {})
                else:
                    raise
            a.state = nondim_state(case.deck, case.gas, st)
    return arr


def connection_struct(conn):
    c = abi.Connection()
    for side in range(2):
        c.rank[side] = conn.rank[side]
        c.block[side] = conn.block[side]
        c.local_block[side] = conn.local_block[side]
        c.boundary[side] = conn.boundary[side]
        c.d1_start[side], c.d1_end[side] = conn.d1s[side], conn.d1e[side]
        c.d2_start[side], c.d2_end[side] = conn.d2s[side], conn.d2e[side]
        c.const_surf[side] = conn.const_surf[side]
    for n in range(8):
        c.patch_border[n] = int(conn.border[n])
    c.orientation = conn.orientation
    c.is_interblock = int(conn.is_interblock)
    return c
