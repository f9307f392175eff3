"""Connection (interblock / periodic) pairing and ghost-geometry exchange.

Setup-time, host side.  Restates:
  * GetConnectionBCs / connection::TestPatchMatch  src/boundaryConditions.cpp:500-831
  * patch::patch                                   src/boundaryConditions.cpp:2127-2222
  * boundaryConditions::BordersSurface             src/boundaryConditions.cpp:193-240
  * GetSwapLoc (all 8 orientations)                src/boundaryConditions.cpp:3006-3181
  * connection::First/SecondSliceIndices, AdjustForSlice  :1016-1150, :833-860
  * SwapGeomSlice / procBlock::PutGeomSlice        src/utility.cpp:212-255, src/procBlock.cpp:3165-3600
"""
from dataclasses import dataclass, field
import numpy as np

_DIRS = {1: ("j", "k", "i"), 2: ("j", "k", "i"), 3: ("k", "i", "j"),
         4: ("k", "i", "j"), 5: ("i", "j", "k"), 6: ("i", "j", "k")}


@dataclass
class Patch:
    boundary: int
    block: int
    d1s: int
    d1e: int
    d2s: int
    d2e: int
    const_surf: int
    bc_type: str
    border: list
    origin: np.ndarray = None
    corner1: np.ndarray = None
    corner2: np.ndarray = None
    corner12: np.ndarray = None


def make_patch(surf, nodes, block, border):
    """patch::patch (boundaryConditions.cpp:2127-2222); nodes[k, j, i, 3]."""
    b = surf.surface_type()
    node = lambda i, j, k: nodes[k, j, i].copy()
    if b <= 2:
        d1s, d1e, d2s, d2e, cs = surf.jmin, surf.jmax, surf.kmin, surf.kmax, surf.imin
        pts = [node(cs, d1s, d2s), node(cs, d1e, d2s), node(cs, d1s, d2e),
               node(cs, d1e, d2e)]
    elif b <= 4:
        d1s, d1e, d2s, d2e, cs = surf.kmin, surf.kmax, surf.imin, surf.imax, surf.jmin
        pts = [node(d2s, cs, d1s), node(d2s, cs, d1e), node(d2e, cs, d1s),
               node(d2e, cs, d1e)]
    else:
        d1s, d1e, d2s, d2e, cs = surf.imin, surf.imax, surf.jmin, surf.jmax, surf.kmin
        pts = [node(d1s, d2s, cs), node(d1e, d2s, cs), node(d1s, d2e, cs),
               node(d1e, d2e, cs)]
    return Patch(b, block, d1s, d1e, d2s, d2e, cs, surf.bc_type, list(border),
                 *pts)


def _close(a, b, tol=1.0e-10):
    # vector3d::CompareWithTol
    return bool(np.all(np.abs(a - b) < tol))


def test_patch_match(p1, p2):
    """Return orientation 1..8 or 0 (connection::TestPatchMatch, cpp:729-831)."""
    if p1.bc_type != p2.bc_type:
        return 0
    table = [
        ("origin", [("corner1", "corner1", "corner2", "corner2", 1),
                    ("corner1", "corner2", "corner2", "corner1", 2)]),
        ("corner1", [("corner1", "origin", "corner2", "corner12", 3),
                     ("corner1", "corner12", "corner2", "origin", 4)]),
        ("corner2", [("corner1", "origin", "corner2", "corner12", 5),
                     ("corner1", "corner12", "corner2", "origin", 6)]),
        ("corner12", [("corner1", "corner1", "corner2", "corner2", 7),
                      ("corner1", "corner2", "corner2", "corner1", 8)]),
    ]
    for o2, options in table:
        if _close(p1.origin, getattr(p2, o2)):
            for a1, a2, b1, b2, orient in options:
                if _close(getattr(p1, a1), getattr(p2, a2)):
                    return orient if _close(getattr(p1, b1),
                                            getattr(p2, b2)) else 0
            return 0
    return 0


def borders_surface(surfs, idx):
    """boundaryConditions::BordersSurface (cpp:193-240)."""
    s = surfs[idx]
    rng = lambda q, d: {"i": (q.imin, q.imax), "j": (q.jmin, q.jmax),
                        "k": (q.kmin, q.kmax)}[d]
    border = [False] * 4
    for other in surfs:
        if other.surface_type() != s.surface_type():
            continue
        s1, s2 = rng(s, s.dir1()), rng(s, s.dir2())
        o1, o2 = rng(other, other.dir1()), rng(other, other.dir2())
        if s1[0] == o1[1]:
            border[0] = True
        if s1[1] == o1[0]:
            border[1] = True
        if s2[0] == o2[1]:
            border[2] = True
        if s2[1] == o2[0]:
            border[3] = True
    return border


@dataclass
class Connection:
    """POD mirror of class connection (boundaryConditions.hpp:371-433)."""
    rank: list
    block: list
    local_block: list
    boundary: list
    d1s: list
    d1e: list
    d2s: list
    d2e: list
    const_surf: list
    border: list
    orientation: int
    is_interblock: bool

    def dir(self, which, side):
        return _DIRS[self.boundary[side]][which - 1]

    def is_lower(self, side):
        return self.const_surf[side] == 0

    def lower_lower_or_upper_upper(self):
        return (self.boundary[0] + self.boundary[1]) % 2 == 0

    def swapped(self):
        """connection::SwapOrder (cpp:341-364)."""
        o = self.orientation
        o = {4: 5, 5: 4}.get(o, o)
        sw = lambda a: [a[1], a[0]]
        return Connection(sw(self.rank), sw(self.block), sw(self.local_block),
                          sw(self.boundary), sw(self.d1s), sw(self.d1e),
                          sw(self.d2s), sw(self.d2e), sw(self.const_surf),
                          self.border[4:] + self.border[:4], o,
                          self.is_interblock)

    def slice_indices(self, side, ng):
        """First/SecondSliceIndices (cpp:1016-1150) -> dict of (start, end)."""
        up_low = -ng if self.boundary[side] % 2 == 0 else 0
        d3s = self.const_surf[side] + up_low
        r = {self.dir(3, side): (d3s, d3s + ng),
             self.dir(1, side): (self.d1s[side] - ng, self.d1e[side] + ng),
             self.dir(2, side): (self.d2s[side] - ng, self.d2e[side] + ng)}
        return r

    def adjusted_for_slice(self, blk_first, ng):
        """connection::AdjustForSlice (cpp:833-860): entry 0 is the block that
        receives, entry 1 the slice coming from the partner."""
        c = self if blk_first else self.swapped()
        blk_start = c.const_surf[0] if c.boundary[0] % 2 == 0 else -ng
        return Connection(
            list(c.rank), list(c.block), list(c.local_block), list(c.boundary),
            [c.d1s[0] - ng, 0],
            [c.d1e[0] + ng, c.d1e[1] - c.d1s[1] + 2 * ng],
            [c.d2s[0] - ng, 0],
            [c.d2e[0] + ng, c.d2e[1] - c.d2s[1] + 2 * ng],
            [blk_start, 0], list(c.border), c.orientation, c.is_interblock)


def swap_loc(l1, l2, l3, ng, c, d3, first):
    """GetSwapLoc (boundaryConditions.cpp:3006-3181), vectorised over numpy
    index arrays l1, l2, l3.  Returns dict {'i':..,'j':..,'k':..}."""
    loc = {}
    if first:
        loc[c.dir(1, 0)] = c.d1s[0] + l1
        loc[c.dir(2, 0)] = c.d2s[0] + l2
        loc[c.dir(3, 0)] = (l3 - ng) if c.is_lower(0) else (c.const_surf[0] + l3)
        return loc
    o = c.orientation
    n1, n2, n3 = c.dir(1, 1), c.dir(2, 1), c.dir(3, 1)
    if o in (2, 4, 5, 7):        # direction 1 and 2 swapped
        loc[n2] = (c.d2e[1] - 1 - l1) if o in (5, 7) else (c.d2s[1] + l1)
        loc[n1] = (c.d1e[1] - 1 - l2) if o in (4, 7) else (c.d1s[1] + l2)
    else:
        # the reference reverses with different orientation ids on i-patches
        # than on j/k-patches (cpp:3064-3073 vs :3104-3113, :3146-3155)
        rev1, rev2 = ((6, 8), (3, 8)) if n3 == "i" else ((3, 8), (6, 8))
        loc[n1] = (c.d1e[1] - 1 - l1) if o in rev1 else (c.d1s[1] + l1)
        loc[n2] = (c.d2e[1] - 1 - l2) if o in rev2 else (c.d2s[1] + l2)
    if c.lower_lower_or_upper_upper():
        loc[n3] = (d3 - l3 - 1) if c.is_lower(1) else \
            (c.const_surf[1] + d3 - l3 - 1)
    else:
        loc[n3] = (l3 - ng) if c.is_lower(1) else (c.const_surf[1] + l3)
    return loc


def insert_maps(conn, blk_first, ng, dims_recv, dims_send):
    """Cell index maps for multiArray3d::PutSlice / InsertSlice
    (multiArray3d.hpp:868-918): returns (dst, src) flat cell indices where
    dst indexes the receiving block's ghost-padded cell array and src indexes
    the sending block's ghost-padded cell array (both (n+2ng) dims, i fastest).
    dims_* = (ni, nj, nk) physical cells."""
    c_adj = conn.adjusted_for_slice(blk_first, ng)
    send_side = 1 if blk_first else 0
    sl = conn.slice_indices(send_side, ng)       # where the slice sits in sender
    len1 = c_adj.d1e[0] - c_adj.d1s[0]
    len2 = c_adj.d2e[0] - c_adj.d2s[0]
    a_s1 = ng if c_adj.border[0] else 0
    a_e1 = ng if c_adj.border[1] else 0
    a_s2 = ng if c_adj.border[2] else 0
    a_e2 = ng if c_adj.border[3] else 0
    l3, l2, l1 = np.meshgrid(np.arange(ng), np.arange(a_s2, len2 - a_e2),
                             np.arange(a_s1, len1 - a_e1), indexing="ij")
    ind_a = swap_loc(l1, l2, l3, ng, c_adj, ng, True)
    ind_s = swap_loc(l1, l2, l3, 0, c_adj, ng, False)   # slices have 0 ghosts
    # slice-local -> sender block indices
    src = {d: ind_s[d] + sl[d][0] for d in "ijk"}

    def flat(idx, dims):
        ni, nj, nk = dims
        return ((idx["k"] + ng) * (nj + 2 * ng) + (idx["j"] + ng)) * \
            (ni + 2 * ng) + (idx["i"] + ng)
    meta = dict(l1=l1, l2=l2, l3=l3, ind_a=ind_a, src=src, c_adj=c_adj)
    return flat(ind_a, dims_recv).ravel(), flat(src, dims_send).ravel(), meta


def find_connections(bcs, nodes, deck, ranks=None, local_pos=None):
    """GetConnectionBCs (boundaryConditions.cpp:500-603)."""
    nblk = len(bcs)
    ranks = ranks or [0] * nblk
    local_pos = local_pos or list(range(nblk))
    iso = []
    for b, surfs in enumerate(bcs):
        for j, s in enumerate(surfs):
            if s.is_connection():
                iso.append([s, b, j])
    conns = []
    ii = 0
    while ii < len(iso):
        s_i, b_i, j_i = iso[ii]
        found = False
        for jj in range(ii + 1, len(iso)):
            s_j, b_j, j_j = iso[jj]
            cand = (s_i.bc_type == "periodic" and s_j.bc_type == "periodic") or \
                (s_i.bc_type == "interblock" and
                 s_i.partner_block() == b_j and
                 s_i.partner_surface() == s_j.surface_type())
            if not cand:
                continue
            p1 = make_patch(s_i, nodes[b_i], b_i, borders_surface(bcs[b_i], j_i))
            p2 = make_patch(s_j, nodes[b_j], b_j, borders_surface(bcs[b_j], j_j))
            for p, s in ((p1, s_i), (p2, s_j)):
                if p.bc_type == "periodic":
                    st = deck.bc_data(s.tag)
                    if st.get("startTag") == s.tag:
                        tr = st.get("translation")
                        if tr is None:
                            raise NotImplementedError("periodic rotation")
                        tr = np.array(tr) / deck.l_ref
                        for nm in ("origin", "corner1", "corner2", "corner12"):
                            setattr(p, nm, getattr(p, nm) + tr)
            orient = test_patch_match(p1, p2)
            if orient:
                conns.append(Connection(
                    [ranks[b_i], ranks[b_j]], [b_i, b_j],
                    [local_pos[b_i], local_pos[b_j]],
                    [p1.boundary, p2.boundary], [p1.d1s, p2.d1s],
                    [p1.d1e, p2.d1e], [p1.d2s, p2.d2s], [p1.d2e, p2.d2e],
                    [p1.const_surf, p2.const_surf],
                    list(p1.border) + list(p2.border), orient,
                    p1.bc_type == "interblock" and p2.bc_type == "interblock"))
                iso[jj], iso[ii + 1] = iso[ii + 1], iso[jj]
                found = True
                break
        if not found:
            raise ValueError(f"no partner found for connection surface {s_i} "
                             f"of block {b_i}")
        ii += 2
    return conns


# --------------------------------------------------------------------------
def _put_geom(recv, send, conn, blk_first):
    """procBlock::PutGeomSlice (procBlock.cpp:3165-3920) for any of the eight
    orientations and any pair of patch directions.  Returns adjEdge.

    The reference's rules, direction by direction of the receiving patch
    (d3 normal, d1, d2 in-plane; the sender's d3/d1/d2 are matched in that order):
      * d3 faces take the sender's d3 faces; for lower/lower or upper/upper pairs
        the sender's index runs the other way (the face above the cell is taken
        and the normal flipped, aFac3 = -1);
      * d1 / d2 faces take the sender's d1 / d2 faces; where the orientation
        reverses that direction (aFac1 / aFac2 = -1) the cell's upper face is
        taken and the normal flipped;
      * at the end of a line the receiving cell's upper face is filled too."""
    ng = recv.ng
    dst, src, meta = insert_maps(conn, blk_first, ng, recv.n, send.n)
    c_adj = meta["c_adj"]
    l1, l2, l3 = (meta[k].ravel() for k in ("l1", "l2", "l3"))
    ia = {d: meta["ind_a"][d].ravel() for d in "ijk"}
    isrc = {d: meta["src"][d].ravel() for d in "ijk"}
    g = ng
    svol = send.vol.a[isrc["k"] + g, isrc["j"] + g, isrc["i"] + g, 0]
    ok = svol != 0.0
    adj_edge = [False] * 4
    # T-intersection detection (procBlock.cpp:3213-3262)
    if np.any(~ok):
        ni, nj, nk = recv.n
        ph = {"i": (ia["i"] >= 0) & (ia["i"] < ni),
              "j": (ia["j"] >= 0) & (ia["j"] < nj),
              "k": (ia["k"] >= 0) & (ia["k"] < nk)}
        d1n, d2n = c_adj.dir(1, 0), c_adj.dir(2, 0)
        for d in "ijk":
            others = [o for o in "ijk" if o != d]
            at_edge = (~ok) & ph[d] & (~ph[others[0]]) & (~ph[others[1]])
            if not np.any(at_edge):
                continue
            if d == d1n:
                low = ia[d2n][at_edge] < c_adj.d2s[0] + ng
                adj_edge[2] |= bool(np.any(low))
                adj_edge[3] |= bool(np.any(~low))
            elif d == d2n:
                low = ia[d1n][at_edge] < c_adj.d1s[0] + ng
                adj_edge[0] |= bool(np.any(low))
                adj_edge[1] |= bool(np.any(~low))
    sel = lambda arr: arr[ok]
    A = {d: sel(ia[d]) + g for d in "ijk"}
    S = {d: sel(isrc[d]) + g for d in "ijk"}
    recv.vol.a[A["k"], A["j"], A["i"]] = send.vol.a[S["k"], S["j"], S["i"]]
    recv.center.a[A["k"], A["j"], A["i"]] = send.center.a[S["k"], S["j"], S["i"]]
    d3r, d1r, d2r = c_adj.dir(3, 0), c_adj.dir(1, 0), c_adj.dir(2, 0)
    d3s, d1s, d2s = c_adj.dir(3, 1), c_adj.dir(1, 1), c_adj.dir(2, 1)
    len1 = c_adj.d1e[0] - c_adj.d1s[0]
    len2 = c_adj.d2e[0] - c_adj.d2s[0]
    o = c_adj.orientation
    lluu = c_adj.lower_lower_or_upper_upper()
    afac = {3: -1.0 if lluu else 1.0,
            1: -1.0 if o in (3, 4, 7, 8) else 1.0,
            2: -1.0 if o in (5, 6, 7, 8) else 1.0}

    def put(dr, ds, a_off, s_off, mask, fac):
        """receiver d`dr`-face of cell A (+a_off along dr) <- fac * sender d`ds`-face
        of cell S (+s_off along ds), for the cells selected by mask"""
        ai = {d: A[d][mask] + (a_off if d == dr else 0) for d in "ijk"}
        si = {d: S[d][mask] + (s_off if d == ds else 0) for d in "ijk"}
        area = send.farea[ds].a[si["k"], si["j"], si["i"]].copy()
        if fac < 0.0:          # unitVec3dMag * negative: flip the unit normal only
            area[..., :3] *= -1.0
        recv.farea[dr].a[ai["k"], ai["j"], ai["i"]] = area
        recv.fcen[dr].a[ai["k"], ai["j"], ai["i"]] = send.fcen[ds].a[si["k"], si["j"], si["i"]]

    every = np.ones(A["i"].shape, dtype=bool)
    # direction 3 (procBlock.cpp:3276-3300): with lower/lower or upper/upper the
    # sender's d3 index is incremented for its d3 faces and runs backwards
    s3 = 1 if lluu else 0
    fac3 = -1 if lluu else 1
    put(d3r, d3s, 0, s3, every, afac[3])
    put(d3r, d3s, 1, s3 + fac3, sel(l3) == ng - 1, afac[3])
    for dr, ds, af, lcur, lend in ((d1r, d1s, afac[1], sel(l1), len1 - 1),
                                   (d2r, d2s, afac[2], sel(l2), len2 - 1)):
        if af > 0.0:
            put(dr, ds, 0, 0, every, af)
            put(dr, ds, 1, 1, lcur == lend, af)
        else:                  # reversed: upper / lower faces swap
            put(dr, ds, 0, 1, every, af)
            put(dr, ds, 1, 0, lcur == lend, af)
    return adj_edge


def swap_geom(conn, geoms):
    """SwapGeomSlice (utility.cpp:212-255); updates conn.border in place."""
    g1, g2 = geoms[conn.block[0]], geoms[conn.block[1]]
    # both slices are taken before either insert (utility.cpp:232-233)
    import copy
    s1, s2 = copy.deepcopy(g1), copy.deepcopy(g2)
    adj1 = _put_geom(g1, s2, conn, True)
    adj2 = _put_geom(g2, s1, conn, False)
    for a in range(4):
        if adj1[a]:
            conn.border[a] = True
        if adj2[a]:
            conn.border[a + 4] = True
