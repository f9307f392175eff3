"""Plot3D reading and finite-volume metrics (setup-time, host side, numpy).

This stands in for the arrays the reference's host code hands over at the
drop-in boundary (procBlock::fAreaI_/J_/K_, vol_, center_, fCenterI_/J_/K_,
cellWidthI_/J_/K_; include/procBlock.hpp:65-90).  It restates, vectorised:

  * ReadP3dGrid                      src/plot3d.cpp:363-448
  * plot3dBlock::Volume/Centroid/FaceArea*/FaceCenter*   src/plot3d.cpp:35-360
  * PadWithGhosts + procBlock::AssignGhostCellsGeom      src/procBlock.cpp:2160-2268
  * procBlock::AssignGhostCellsGeomEdge                  src/procBlock.cpp:2270-2425
  * procBlock::CalcCellWidths                            src/procBlock.cpp:6397-6412

All arrays are indexed [k, j, i, component] (i fastest in memory), which is
byte-for-byte the reference's multiArray3d layout (multiArray3d.hpp:104-113).
"""
import numpy as np


# --------------------------------------------------------------------------
def read_plot3d(path, l_ref=1.0):
    """Return a list of node-coordinate arrays [nk, nj, ni, 3]."""
    with open(path, "rb") as fh:
        raw = fh.read()
    nblk = int(np.frombuffer(raw, dtype="<i4", count=1, offset=0)[0])
    dims = np.frombuffer(raw, dtype="<i4", count=3 * nblk, offset=4)
    dims = dims.reshape(nblk, 3)
    off = 4 + 12 * nblk
    blocks = []
    for b in range(nblk):
        ni, nj, nk = (int(v) for v in dims[b])
        n = ni * nj * nk
        xyz = np.frombuffer(raw, dtype="<f8", count=3 * n, offset=off)
        off += 24 * n
        coords = np.empty((nk, nj, ni, 3))
        for c in range(3):
            coords[..., c] = xyz[c * n:(c + 1) * n].reshape(nk, nj, ni) / l_ref
        blocks.append(coords)
    return blocks


def write_plot3d(path, blocks):
    with open(path, "wb") as fh:
        np.array([len(blocks)], dtype="<i4").tofile(fh)
        for c in blocks:
            nk, nj, ni, _ = c.shape
            np.array([ni, nj, nk], dtype="<i4").tofile(fh)
        for c in blocks:
            for comp in range(3):
                np.ascontiguousarray(c[..., comp], dtype="<f8").tofile(fh)


# --------------------------------------------------------------------------
def _cross(a, b):
    # vector3d::CrossProd (vector3d.hpp:330-339)
    out = np.empty_like(a)
    out[..., 0] = a[..., 1] * b[..., 2] - a[..., 2] * b[..., 1]
    out[..., 1] = -1.0 * (a[..., 0] * b[..., 2] - a[..., 2] * b[..., 0])
    out[..., 2] = a[..., 0] * b[..., 1] - a[..., 1] * b[..., 0]
    return out


def _dot(a, b):
    return a[..., 0] * b[..., 0] + a[..., 1] * b[..., 1] + a[..., 2] * b[..., 2]


def _unit_mag(vec):
    """unitVec3dMag(a): {a / |a|, |a|} (vector3d.hpp:133-135)."""
    mag = np.sqrt(_dot(vec, vec))
    out = np.empty(vec.shape[:-1] + (4,))
    with np.errstate(invalid="ignore", divide="ignore"):
        out[..., :3] = vec / mag[..., None]
    out[..., 3] = mag
    return out


def _pyramid_volume(p, a, b, c, d):
    # PyramidVolume (plot3d.cpp:490-498)
    xp = 0.25 * ((a - p) + (b - p) + (c - p) + (d - p))
    xac = c - a
    xbd = d - b
    return 1.0 / 6.0 * _dot(xp, _cross(xac, xbd))


def interior_metrics(x):
    """Metrics of the physical cells/faces from node coordinates x[nk,nj,ni,3]."""
    c = lambda di, dj, dk: x[dk:x.shape[0] - 1 + dk, dj:x.shape[1] - 1 + dj,
                             di:x.shape[2] - 1 + di]
    c000, c100, c010, c110 = c(0, 0, 0), c(1, 0, 0), c(0, 1, 0), c(1, 1, 0)
    c001, c101, c011, c111 = c(0, 0, 1), c(1, 0, 1), c(0, 1, 1), c(1, 1, 1)
    # plot3d.cpp:35-43
    center = 0.125 * (c000 + c100 + c010 + c110 + c001 + c101 + c011 + c111)
    # plot3d.cpp:60-113 (same pyramid order)
    vol = _pyramid_volume(center, c000, c001, c011, c010)
    vol = vol + _pyramid_volume(center, c100, c110, c111, c101)
    vol = vol + _pyramid_volume(center, c000, c100, c101, c001)
    vol = vol + _pyramid_volume(center, c010, c011, c111, c110)
    vol = vol + _pyramid_volume(center, c000, c010, c110, c100)
    vol = vol + _pyramid_volume(center, c001, c101, c111, c011)
    if np.any(vol <= 0):
        raise ValueError("negative volume in PLOT3D block")

    # i-faces (plot3d.cpp:150-181): nodes (ii, jj..jj+1, kk..kk+1)
    n = lambda dj, dk: x[dk:x.shape[0] - 1 + dk, dj:x.shape[1] - 1 + dj, :]
    xac = n(1, 1) - n(0, 0)
    xbd = n(1, 0) - n(0, 1)
    farea_i = _unit_mag(0.5 * _cross(xbd, xac))
    fcen_i = 0.25 * (n(0, 0) + n(1, 0) + n(0, 1) + n(1, 1))
    # j-faces (plot3d.cpp:224-255)
    n = lambda di, dk: x[dk:x.shape[0] - 1 + dk, :, di:x.shape[2] - 1 + di]
    xac = n(0, 1) - n(1, 0)
    xbd = n(0, 0) - n(1, 1)
    farea_j = _unit_mag(0.5 * _cross(xbd, xac))
    fcen_j = 0.25 * (n(0, 0) + n(1, 0) + n(0, 1) + n(1, 1))
    # k-faces (plot3d.cpp:300-331)
    n = lambda di, dj: x[:, dj:x.shape[1] - 1 + dj, di:x.shape[2] - 1 + di]
    xac = n(0, 1) - n(1, 0)
    xbd = n(1, 1) - n(0, 0)
    farea_k = _unit_mag(0.5 * _cross(xbd, xac))
    fcen_k = 0.25 * (n(0, 0) + n(1, 0) + n(0, 1) + n(1, 1))
    for nm, fa in (("i", farea_i), ("j", farea_j), ("k", farea_k)):
        if np.any(fa[..., 3] <= 0):
            raise ValueError(f"negative {nm}-face area in PLOT3D block")
    return dict(vol=vol[..., None], center=center, farea_i=farea_i,
                farea_j=farea_j, farea_k=farea_k, fcen_i=fcen_i,
                fcen_j=fcen_j, fcen_k=fcen_k)


# --------------------------------------------------------------------------
class Field:
    """Ghost-padded array addressed with the reference's signed indices."""

    def __init__(self, ni, nj, nk, ng, ncomp, fill=0.0):
        self.ni, self.nj, self.nk, self.ng = ni, nj, nk, ng
        self.a = np.full((nk + 2 * ng, nj + 2 * ng, ni + 2 * ng, ncomp), fill)

    def v(self, ir, jr, kr):
        g = self.ng
        return self.a[kr[0] + g:kr[1] + g, jr[0] + g:jr[1] + g,
                      ir[0] + g:ir[1] + g]

    def phys(self):
        return self.v((0, self.ni), (0, self.nj), (0, self.nk))


def _pad(arr, ng):
    nk, nj, ni, nc = arr.shape
    f = Field(ni, nj, nk, ng, nc)
    f.phys()[...] = arr
    return f


def _plane(d, ind, r1, r2, fid="cell", stype=0):
    """Index ranges of multiArray3d::Slice(dir, ind, r1, r2, id, type)
    (multiArray3d.hpp:529-573)."""
    r1, r2 = list(r1), list(r2)
    if d == "i":
        if stype == 2 and fid == "i":
            ind += 1
        if fid == "j":
            r1[1] += 1
        elif fid == "k":
            r2[1] += 1
        return (ind, ind + 1), tuple(r1), tuple(r2)
    if d == "j":
        if stype == 4 and fid == "j":
            ind += 1
        if fid == "k":
            r1[1] += 1
        elif fid == "i":
            r2[1] += 1
        return tuple(r2), (ind, ind + 1), tuple(r1)
    if stype == 6 and fid == "k":
        ind += 1
    if fid == "i":
        r1[1] += 1
    elif fid == "j":
        r2[1] += 1
    return tuple(r1), tuple(r2), (ind, ind + 1)


def _line(d, d2, d3, n, fid="cell", upper2=False, upper3=False):
    """Index ranges of multiArray3d::Slice(dir, d2Ind, d3Ind, physOnly=true,
    id, upper2, upper3) (multiArray3d.hpp:473-523).  n = (ni, nj, nk)."""
    two, three = {"i": ("j", "k"), "j": ("k", "i"), "k": ("i", "j")}[d]
    if upper2 and fid == two:
        d2 += 1
    elif upper3 and fid == three:
        d3 += 1
    nd = n["ijk".index(d)] + (1 if fid == d else 0)
    rng = {d: (0, nd), two: (d2, d2 + 1), three: (d3, d3 + 1)}
    return rng["i"], rng["j"], rng["k"]


_AXIS = {"i": 2, "j": 1, "k": 0}


def _grow(arr, d):
    """multiArray3d::GrowI/J/K (multiArray3d.hpp:920-966)."""
    ax = _AXIS[d]
    last = np.take(arr, [arr.shape[ax] - 1], axis=ax)
    return np.concatenate([arr, last], axis=ax)


class BlockGeometry:
    """Ghost-padded metrics of one block."""

    def __init__(self, coords, ng, metrics=None):
        """metrics: a function nodes -> the dict interior_metrics returns (the library's
        agx_plot3d_metrics through solver.DeviceSetup); default: the numpy form here."""
        nk, nj, ni = (s - 1 for s in coords.shape[:3])
        self.ni, self.nj, self.nk, self.ng = ni, nj, nk, ng
        self.nodes = coords
        m = (metrics or interior_metrics)(coords)
        self.vol = _pad(m["vol"], ng)
        self.center = _pad(m["center"], ng)
        self.farea = {d: _pad(m["farea_" + d], ng) for d in "ijk"}
        self.fcen = {d: _pad(m["fcen_" + d], ng) for d in "ijk"}
        self.width = {d: Field(ni, nj, nk, ng, 1) for d in "ijk"}
        self.wall_dist = Field(ni, nj, nk, ng, 1, fill=1.0e10)  # DEFAULT_WALL_DIST

    @property
    def n(self):
        return (self.ni, self.nj, self.nk)

    # -- procBlock::AssignGhostCellsGeom (procBlock.cpp:2160-2268) ----------
    def assign_ghost_geom(self, surfaces):
        n = {"i": self.ni, "j": self.nj, "k": self.nk}
        for layer in range(1, self.ng + 1):
            for s in surfaces:
                d = s.dir3()
                st = s.surface_type()
                r1, r2, r3 = s.rng(s.dir1()), s.rng(s.dir2()), s.rng(d)
                if st % 2 == 0:   # upper
                    g_cell = r3[0] + layer - 1
                    i_cell = max(r3[0] - layer, 0)
                    p_cell = g_cell - 1
                    pi_cell = i_cell + 1
                    i_face = max(r3[0] - layer, 0)
                    pi_face = i_face + 1
                else:
                    g_cell = r3[0] - layer
                    i_cell = min(r3[0] + layer - 1, n[d] - 1)
                    p_cell = g_cell + 1
                    pi_cell = i_cell - 1
                    i_face = min(r3[0] + layer, n[d])
                    pi_face = i_face - 1
                if s.bc_type == "interblock":
                    continue
                self.vol.v(*_plane(d, g_cell, r1, r2))[...] = \
                    self.vol.v(*_plane(d, i_cell, r1, r2))
                for f in "ijk":
                    self.farea[f].v(*_plane(d, g_cell, r1, r2, f, st))[...] = \
                        self.farea[f].v(*_plane(d, i_cell, r1, r2, f, st))
                fc = self.fcen[d]
                dist_f2f = fc.v(*_plane(d, pi_face, r1, r2)) - \
                    fc.v(*_plane(d, i_face, r1, r2))
                if layer > 1:
                    dist_c2c = self.center.v(*_plane(d, pi_cell, r1, r2)) - \
                        self.center.v(*_plane(d, i_cell, r1, r2))
                else:
                    dist_c2c = dist_f2f
                for f in "ijk":
                    dist = dist_f2f if f == d else _grow(dist_c2c, f)
                    new = dist + self.fcen[f].v(*_plane(d, p_cell, r1, r2, f, st))
                    self.fcen[f].v(*_plane(d, g_cell, r1, r2, f, st))[...] = new
                self.center.v(*_plane(d, g_cell, r1, r2))[...] = \
                    self.center.v(*_plane(d, p_cell, r1, r2)) + dist_c2c

    # -- procBlock::AssignGhostCellsGeomEdge (procBlock.cpp:2270-2425) ------
    def assign_ghost_geom_edge(self):
        n = self.n
        for d in "ijk":
            two, three = {"i": ("j", "k"), "j": ("k", "i"),
                          "k": ("i", "j")}[d]
            max2, max3 = n["ijk".index(two)], n["ijk".index(three)]
            for layer3 in range(1, self.ng + 1):
                for layer2 in range(1, self.ng + 1):
                    for cc in range(4):
                        u2, u3 = cc > 1, cc % 2 == 1
                        p2 = max2 + layer2 - 2 if u2 else 1 - layer2
                        g2 = p2 + 1 if u2 else p2 - 1
                        i2 = max2 - layer2 if u2 else layer2 - 1
                        p3 = max3 + layer3 - 2 if u3 else 1 - layer3
                        g3 = p3 + 1 if u3 else p3 - 1
                        ln = lambda a, b, fid="cell": _line(d, a, b, n, fid,
                                                            u2, u3)
                        self.vol.v(*ln(g2, g3))[...] = self.vol.v(*ln(i2, g3))
                        for f in "ijk":
                            self.farea[f].v(*ln(g2, g3, f))[...] = \
                                self.farea[f].v(*ln(i2, g3, f))
                        dist_f2f = self.fcen[two].v(*ln(g2, p3, two)) - \
                            self.fcen[two].v(*ln(p2, p3, two))
                        dist_c2c = self.center.v(*ln(g2, p3)) - \
                            self.center.v(*ln(p2, p3))
                        self.center.v(*ln(g2, g3))[...] = \
                            dist_c2c + self.center.v(*ln(p2, g3))
                        for f in "ijk":
                            if f == d:
                                dist = _grow(dist_c2c, f)
                            elif f == two:
                                dist = dist_f2f
                            else:
                                dist = dist_c2c
                            new = dist + self.fcen[f].v(*ln(p2, g3, f))
                            self.fcen[f].v(*ln(g2, g3, f))[...] = new

    # -- procBlock::CalcCellWidths (procBlock.cpp:6397-6412) ----------------
    def calc_cell_widths(self):
        for d in "ijk":
            fc = self.fcen[d].a
            ax = _AXIS[d]
            lo = np.take(fc, range(0, fc.shape[ax] - 1), axis=ax)
            hi = np.take(fc, range(1, fc.shape[ax]), axis=ax)
            diff = lo - hi
            self.width[d].a[..., 0] = np.sqrt(_dot(diff, diff))
