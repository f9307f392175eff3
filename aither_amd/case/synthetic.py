"""Synthetic cases: single- or multi-block boxes with
uniform or smoothly stretched spacing and a closed-form, RNG-free perturbed
initial state.  Used by the parity tests and by bench.py.
"""
import math
import numpy as np

from .inputfile import InputDeck, Surface, State, MUSCL_KAPPA
from . import builder as _b


def box_nodes(ni, nj, nk, stretch=1.0, lengths=(1.0, 1.0, 1.0),
              origin=(0.0, 0.0, 0.0), skew=0.0):
    """Node coordinates [nk+1, nj+1, ni+1, 3]; x_i = L*(i/n)**stretch, plus an
    optional smooth skew so that faces are not axis aligned."""
    ax = [origin[d] + lengths[d] * (np.arange(n + 1) / n) ** stretch
          for d, n in enumerate((ni, nj, nk))]
    z, y, x = np.meshgrid(ax[2], ax[1], ax[0], indexing="ij")
    if skew:
        two_pi = 2.0 * math.pi
        x = x + skew * np.sin(two_pi * y) * np.sin(two_pi * z)
        y = y + skew * np.sin(two_pi * z) * np.sin(two_pi * x)
        z = z + skew * np.sin(two_pi * x) * np.sin(two_pi * y)
    return np.stack([x, y, z], axis=-1)


def perturbed_state(case, amplitude=0.05):
    """q * (1 + a sin(2 pi x) sin(2 pi y) sin(2 pi z)) on the cell centres of
    every block (physical and ghost cells; ghosts are overwritten by BCs)."""
    two_pi = 2.0 * math.pi
    for blk in case.blocks:
        c = blk.geom.center.a
        s = np.sin(two_pi * c[..., 0]) * np.sin(two_pi * c[..., 1]) * \
            np.sin(two_pi * c[..., 2])
        g = blk.geom.ng
        base = blk.state[g, g, g, :].copy()
        fac = 1.0 + amplitude * s
        new = base[None, None, None, :] * fac[..., None]
        # velocity components get phase-shifted perturbations
        new[..., 2] = base[2] * (1.0 + amplitude * np.cos(two_pi * c[..., 0]) * s)
        new[..., 3] = base[3] * (1.0 - amplitude * s)
        ni, nj, nk = blk.geom.n
        blk.state[...] = 0.0
        blk.state[g:g + nk, g:g + nj, g:g + ni, :] = \
            new[g:g + nk, g:g + nj, g:g + ni, :]


def make_deck(**kw):
    d = InputDeck()
    d.rho_ref, d.t_ref, d.l_ref = 1.225, 288.15, kw.get("l_ref", 1.0)
    d.equation_set = kw.get("equation_set", "euler")
    d.time_integration = kw.get("time_integration", "rk4")
    if d.time_integration == "bdf2":
        d.theta, d.zeta = 1.0, 0.5
    elif d.time_integration == "crankNicholson":
        d.theta, d.zeta = 0.5, 0.0
    d.face_reconstruction = kw.get("face_reconstruction", "thirdOrder")
    if d.face_reconstruction in MUSCL_KAPPA:
        d.kappa = MUSCL_KAPPA[d.face_reconstruction]
    d.limiter = kw.get("limiter", "vanAlbada")
    d.inviscid_flux = kw.get("inviscid_flux", "roe")
    d.viscous_face_reconstruction = kw.get("viscous_face_reconstruction", "central")
    d.inv_flux_jac = kw.get("inv_flux_jac", "rusanov")
    d.turbulence_model = kw.get("turbulence_model", "none")
    d.matrix_solver = kw.get("matrix_solver", "lusgs")
    d.matrix_sweeps = kw.get("matrix_sweeps", 1)
    d.matrix_relaxation = kw.get("matrix_relaxation", 1.0)
    d.nonlinear_iterations = kw.get("nonlinear_iterations", 1)
    d.dual_time_cfl = kw.get("dual_time_cfl", -1.0)
    d.dt = kw.get("dt", -1.0)
    cfl = kw.get("cfl", 0.5)
    d.cfl_start = d.cfl_max = cfl
    d.cfl_step = 0.0
    if d.time_integration == "rk4":
        d.nonlinear_iterations = 4
    if d.time_integration == "explicitEuler":
        d.nonlinear_iterations = 1
    # farfield turbulence (turbulenceIntensity, eddyViscosityRatio) of the initial state and
    # the characteristic boundary; the reference's defaults unless given
    turb = {}
    if kw.get("turbulence") is not None:
        turb = dict(turbulenceIntensity=kw["turbulence"][0], eddyViscosityRatio=kw["turbulence"][1])
    # viscous walls (tags 2: adiabatic, 4: isothermal and moving, 5: constant heat flux)
    # with wall functions when wall_treatment="wallLaw"
    wall = {}
    if kw.get("wall_treatment") is not None:
        wall = dict(wallTreatment=kw["wall_treatment"])
    vel = list(kw.get("velocity", [50.0, 20.0, 10.0]))     # free stream, m/s
    d.ics = [State("icState", dict(tag=-1, pressure=101325.0, density=1.225,
                                   velocity=vel, **turb))]
    d.bc_states = [
        State("characteristic", dict(tag=1, pressure=101325.0, density=1.225,
                                     velocity=vel, **turb)),
        State("viscousWall", dict(tag=2, **wall)),
        State("pressureOutlet", dict(tag=3, pressure=101325.0)),
        State("viscousWall", dict(tag=4, temperature=300.0,
                                  velocity=[5.0, 0.0, 0.0], **wall)),
        State("viscousWall", dict(tag=5, heatFlux=2.0e3, **wall)),
        State("inlet", dict(tag=6, pressure=101325.0, density=1.225,
                            velocity=[50.0, 20.0, 10.0], nonreflecting=True,
                            lengthScale=1.0)),
        State("pressureOutlet", dict(tag=7, pressure=101325.0, nonreflecting=True,
                                     lengthScale=1.0)),
        State("supersonicInflow", dict(tag=8, pressure=101325.0, density=1.225,
                                       velocity=vel, **turb)),
        State("supersonicOutflow", dict(tag=9)),
        State("inlet", dict(tag=10, pressure=101325.0, density=1.225, velocity=vel, **turb)),
    ]
    return d


_DEFAULT_BC = ("slipWall", 0)


def box_surfaces(ni, nj, nk, bcs=None):
    """Six surfaces of a box; bcs maps surface type 1..6 -> (name, tag)."""
    bcs = bcs or {}
    get = lambda s: bcs.get(s, _DEFAULT_BC)
    surfs = [
        Surface(get(1)[0], 0, 0, 0, nj, 0, nk, get(1)[1]),
        Surface(get(2)[0], ni, ni, 0, nj, 0, nk, get(2)[1]),
        Surface(get(3)[0], 0, ni, 0, 0, 0, nk, get(3)[1]),
        Surface(get(4)[0], 0, ni, nj, nj, 0, nk, get(4)[1]),
        Surface(get(5)[0], 0, ni, 0, nj, 0, 0, get(5)[1]),
        Surface(get(6)[0], 0, ni, 0, nj, nk, nk, get(6)[1]),
    ]
    surfs.sort(key=Surface.sort_key)
    return surfs


def single_block_case(n=(16, 16, 16), stretch=1.0, skew=0.0, bcs=None,
                      amplitude=0.05, setup=None, **deck_kw):
    ni, nj, nk = n
    deck = make_deck(**deck_kw)
    deck.bcs = [box_surfaces(ni, nj, nk, bcs)]
    coords = [box_nodes(ni, nj, nk, stretch, skew=skew)]
    case = _b.build_case(None, deck=deck, coords=coords, setup=setup)
    if amplitude:
        perturbed_state(case, amplitude)
    return case


def stacked_blocks_case(n=(16, 16, 16), nblocks=2, axis="k", stretch=1.0,
                        bcs=None, amplitude=0.05, ranks=None, setup=None, **deck_kw):
    """nblocks boxes stacked along `axis`, joined by interblock connections
    (orientation 1, lower <-> upper)."""
    deck, coords = _stacked(n, nblocks, axis, stretch, bcs, deck_kw)
    case = _b.build_case(None, deck=deck, coords=coords, ranks=ranks, setup=setup)
    if amplitude:
        perturbed_state(case, amplitude)
    return case


def multigrid_levels(n=(16, 16, 16), nblocks=1, axis="k", levels=2, cycle="V", stretch=1.0,
                     bcs=None, amplitude=0.05, ranks=None, **deck_kw):
    """(cases, transfers) of aither_amd.case.multigrid.build_levels for nblocks boxes stacked
    along `axis` (one box: no connections); the finest level's state is perturbed."""
    from . import multigrid as _mg
    deck, coords = _stacked(n, nblocks, axis, stretch, bcs, deck_kw)
    deck.multigrid_cycle = cycle
    cases, transfers = _mg.build_levels(deck, coords, levels, _b.build_case, ranks=ranks)
    if amplitude:
        perturbed_state(cases[0], amplitude)
    return cases, transfers


def _stacked(n, nblocks, axis, stretch, bcs, deck_kw):
    ni, nj, nk = n
    deck = make_deck(**deck_kw)
    d = "ijk".index(axis)
    lo_s, hi_s = 2 * d + 1, 2 * d + 2
    coords, all_bcs = [], []
    for b in range(nblocks):
        origin = [0.0, 0.0, 0.0]
        origin[d] = float(b)
        # no stretching along the stacking axis so that faces match exactly
        x = box_nodes(ni, nj, nk, 1.0, origin=origin)
        if stretch != 1.0:
            for q in range(3):
                if q != d:
                    nq = (ni, nj, nk)[q]
                    ax = (np.arange(nq + 1) / nq) ** stretch
                    shape = [1, 1, 1]
                    shape[2 - q] = nq + 1
                    x[..., q] = ax.reshape(shape)
        coords.append(x)
        blk_bcs = dict(bcs or {})
        if b > 0:
            blk_bcs[lo_s] = ("interblock", 1000 * hi_s + (b - 1))
        if b < nblocks - 1:
            blk_bcs[hi_s] = ("interblock", 1000 * lo_s + (b + 1))
        all_bcs.append(box_surfaces(ni, nj, nk, blk_bcs))
    deck.bcs = all_bcs
    return deck, coords


def cube_blocks_case(n=(8, 8, 8), splits=(2, 2, 2), bcs=None, amplitude=0.05,
                     ranks=None, setup=None, **deck_kw):
    """splits[0] x splits[1] x splits[2] boxes of n cells each tiling one box
    (BASELINE configs[3] style: 2x2x2 = 8 blocks), joined face to face by
    interblock connections; block id = bi + si * (bj + sj * bk)."""
    ni, nj, nk = n
    si, sj, sk = splits
    deck = make_deck(**deck_kw)
    coords, all_bcs = [], []
    bid = lambda a, b_, c: a + si * (b_ + sj * c)
    for bk in range(sk):
        for bj in range(sj):
            for bi in range(si):
                coords.append(box_nodes(ni, nj, nk, 1.0,
                                        origin=(float(bi), float(bj), float(bk))))
                blk = dict(bcs or {})
                idx, cnt = (bi, bj, bk), (si, sj, sk)
                for d in range(3):
                    lo_s, hi_s = 2 * d + 1, 2 * d + 2
                    if idx[d] > 0:
                        nb = list(idx); nb[d] -= 1
                        blk[lo_s] = ("interblock", 1000 * hi_s + bid(*nb))
                    if idx[d] < cnt[d] - 1:
                        nb = list(idx); nb[d] += 1
                        blk[hi_s] = ("interblock", 1000 * lo_s + bid(*nb))
                all_bcs.append(box_surfaces(ni, nj, nk, blk))
    deck.bcs = all_bcs
    case = _b.build_case(None, deck=deck, coords=coords, ranks=ranks, setup=setup)
    if amplitude:
        perturbed_state(case, amplitude)
    return case
