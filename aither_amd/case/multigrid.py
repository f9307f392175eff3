"""Grid levels of the geometric multigrid: what gridLevel::Coarsen (gridLevel.cpp:440-535)
builds before the first cycle -- the coarse mesh and its boundary surfaces, the fine-to-coarse
cell map, the volume weights of the restriction and the trilinear coefficients of the
prolongation -- as plain arrays (host set-up, like the rest of aither_amd.case).
"""
import copy
from dataclasses import dataclass

import numpy as np


def kept_indices(n_nodes, orig_surfaces, new_surfaces, d):
    """Node indices of direction d the coarse mesh keeps (procBlock::GetCoarseMeshAndBCs
    procBlock.cpp:6477-6527): every index that bounds a surface, every other one between.
    `new_surfaces` (the coarse block's copy) is renumbered on the way, one kept index after
    the other, as boundarySurface::UpdateForCoarseMesh (boundaryConditions.cpp) does -- on
    the COPY, while boundaries are looked up in the original list."""
    lo, hi = d + "min", d + "max"
    bounds = set()
    for s in orig_surfaces:
        bounds.add(getattr(s, lo))
        bounds.add(getattr(s, hi))
    keep, since = [], 0
    for ii in range(n_nodes):
        if ii in bounds:
            keep.append(ii)
            for s in new_surfaces:
                if getattr(s, lo) == ii:
                    setattr(s, lo, len(keep) - 1)
                if getattr(s, hi) == ii:
                    setattr(s, hi, len(keep) - 1)
            since = 0
        elif since > 0:
            keep.append(ii)
            since = 0
        else:
            since += 1
    return keep


@dataclass
class Transfer:
    """Fine level -> next coarser level, one block."""
    to_coarse: np.ndarray      # [nk, nj, ni, 3] int32: coarse cell (i, j, k) of a fine cell
    vol_fac: np.ndarray        # [nk, nj, ni]: fine volume / sum over the coarse cell
    coeffs: np.ndarray         # [nk, nj, ni, 7]: TrilinearInterpCoeff of the fine centre
    kept: tuple                # (iIndex, jIndex, kIndex)


def coarsen_block(coords, surfaces):
    """coords [nk+1, nj+1, ni+1, 3], surfaces: list of inputfile.Surface ->
    (coarse coords, coarse surfaces, to_coarse, kept index lists)."""
    nkn, njn, nin = coords.shape[:3]
    new = [copy.copy(s) for s in surfaces]
    ii = kept_indices(nin, surfaces, new, "i")
    jj = kept_indices(njn, surfaces, new, "j")
    kk = kept_indices(nkn, surfaces, new, "k")
    coarse = coords[np.ix_(kk, jj, ii)]
    # fine cell fi lies in the coarse cell whose lower kept node is the last one <= fi
    # (procBlock.cpp:6556-6581)
    def cell_map(keep, ncell):
        keep = np.asarray(keep)
        return np.searchsorted(keep, np.arange(ncell), side="right") - 1
    ci, cj, ck = cell_map(ii, nin - 1), cell_map(jj, njn - 1), cell_map(kk, nkn - 1)
    to_coarse = np.empty((nkn - 1, njn - 1, nin - 1, 3), dtype=np.int32)
    to_coarse[..., 0] = ci[None, None, :]
    to_coarse[..., 1] = cj[None, :, None]
    to_coarse[..., 2] = ck[:, None, None]
    return coarse, new, to_coarse, (ii, jj, kk)


def volume_weights(vol, to_coarse, coarse_cells):
    """volFac (procBlock.cpp:6584-6602): a fine cell's volume over the sum of the volumes of
    the fine cells of its coarse cell.  vol [nk, nj, ni]; coarse_cells (nk, nj, ni)."""
    cnk, cnj, cni = coarse_cells
    flat = (to_coarse[..., 2].astype(np.int64) * cnj + to_coarse[..., 1]) * cni + to_coarse[..., 0]
    # the reference adds the volumes in the order of its multimap: fine cells of a coarse
    # cell in insertion order (k, j, i ascending) -- np.add.at walks the flattened array in
    # that order
    sums = np.zeros(cnk * cnj * cni)
    np.add.at(sums, flat.ravel(), vol.ravel())
    return vol / sums[flat]


def _lin_coeff(x0, x1, x):
    # LinearInterpCoeff utility.cpp:626-631
    d = x1 - x0
    dist = np.sqrt((d * d).sum(-1))
    return ((x - x0) * (d / dist[..., None])).sum(-1) / dist


def _lin(d0, d1, c):
    # LinearInterp utility.hpp:341-344
    return (1.0 - c)[..., None] * d0 + c[..., None] * d1


def trilinear_coeffs(centers, coarse_coords, to_coarse):
    """TrilinearInterpCoeff (utility.cpp:633-662) of every fine cell centre in the nodes of
    its coarse cell (gridLevel.cpp:501-527).  centers [nk, nj, ni, 3] (physical cells)."""
    ci, cj, ck = to_coarse[..., 0], to_coarse[..., 1], to_coarse[..., 2]
    n = lambda di, dj, dk: coarse_coords[ck + dk, cj + dj, ci + di]
    x0, x1, x2, x3 = n(0, 0, 0), n(1, 0, 0), n(0, 1, 0), n(1, 1, 0)
    x4, x5, x6, x7 = n(0, 0, 1), n(1, 0, 1), n(0, 1, 1), n(1, 1, 1)
    x = centers
    c = np.empty(centers.shape[:3] + (7,))
    c[..., 0] = _lin_coeff(x0, x4, x); x04 = _lin(x0, x4, c[..., 0])
    c[..., 1] = _lin_coeff(x1, x5, x); x15 = _lin(x1, x5, c[..., 1])
    c[..., 2] = _lin_coeff(x2, x6, x); x26 = _lin(x2, x6, c[..., 2])
    c[..., 3] = _lin_coeff(x3, x7, x); x37 = _lin(x3, x7, c[..., 3])
    c[..., 4] = _lin_coeff(x04, x15, x); x0415 = _lin(x04, x15, c[..., 4])
    c[..., 5] = _lin_coeff(x26, x37, x); x2637 = _lin(x26, x37, c[..., 5])
    c[..., 6] = _lin_coeff(x0415, x2637, x)
    return c


def build_levels(deck, coords, nlevels, build_case, **kw):
    """The Cases of the grid levels (finest first) and the Transfers between them.
    build_case: builder.build_case (passed in: builder imports this module's users).
    Every level is a complete case of its own -- geometry, ghost geometry, connections --
    built from its nodes and surfaces like the finest one (gridLevel::Coarsen)."""
    cases, transfers = [], []
    level_deck = copy.copy(deck)
    level_deck.multigrid_levels = 1          # (each level is built as a case of its own)
    level_coords = coords
    for lev in range(nlevels):
        case = build_case(None, deck=level_deck, coords=level_coords, **kw)
        cases.append(case)
        if lev == nlevels - 1:
            break
        next_coords, next_bcs, trs = [], [], []
        for b, (x, surfs) in enumerate(zip(level_coords, level_deck.bcs)):
            cx, cs, to_coarse, kept = coarsen_block(x, surfs)
            g = case.blocks[b].geom
            vol = g.vol.phys()[..., 0]
            cen = g.center.phys()
            cells = tuple(s - 1 for s in cx.shape[:3])
            trs.append(Transfer(to_coarse, volume_weights(vol, to_coarse, cells),
                                trilinear_coeffs(cen, cx, to_coarse), kept))
            next_coords.append(cx)
            next_bcs.append(cs)
        transfers.append(trs)
        level_deck = copy.copy(level_deck)
        level_deck.bcs = next_bcs
        level_coords = next_coords
    return cases, transfers
