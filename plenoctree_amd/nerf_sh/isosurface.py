"""Isosurface extraction on a dense scalar grid (host side of nerf_sh/gen_mesh.py:88-130).

The reference hands the sigma volume to PyMCubes (`mcubes.marching_cubes(sigmas, iso)`, gen_mesh.py:124), a third-party
C++ package that is not vendored in the reference and not installed here.  This module is the replacement: marching
cubes with a case table that is *generated* at import (no 256-row literal), not copied from anywhere:

  * a vertex is placed on every grid edge whose two end values straddle `iso`, by linear interpolation - the same
    vertex set PyMCubes produces (index units, vertex = i + t along the edge axis);
  * inside every cell the crossings are joined face by face: walking a face's four corners counter-clockwise as seen
    from outside the cell, each run of "inside" corners (value >= iso) is cut off by one segment, from the edge where
    the walk enters the run to the edge where it leaves it.  The rule only looks at the face's own four corner signs,
    so the two cells sharing a face always agree (ambiguous faces included) and the mesh is closed;
  * the segments of a cell form closed loops; every loop is fan-triangulated.  Segments are directed so that the
    triangle normals (right-hand rule) point towards lower values, i.e. out of the dense region.

The triangulation inside a cell may differ from PyMCubes' table (fan choice, ambiguous cases); the surface crosses the
same grid edges at the same points.  Host numpy: this step is CPU work in the reference too and is off the hot path.
"""
import numpy as np

# corner c = x + 2y + 4z; an edge is (axis, corner offset with offset[axis] == 0)
_EDGES = [(a, o) for a in range(3) for o in range(8) if not (o >> a) & 1]
_EDGE_ID = {e: i for i, e in enumerate(_EDGES)}


def _edge_between(c0, c1):
    a = (c0 ^ c1).bit_length() - 1
    return _EDGE_ID[(a, min(c0, c1))]


def _face_cycles():
    """Corner cycles of the six faces, counter-clockwise seen from outside the cell."""
    cycles = []
    for a in range(3):
        b, c = (a + 1) % 3, (a + 2) % 3
        for side in (0, 1):
            u, v = (b, c) if side else (c, b)  # u x v = outward normal
            base = side << a
            cycles.append([base | (du << u) | (dv << v) for du, dv in ((0, 0), (1, 0), (1, 1), (0, 1))])
    return cycles


def _build_table():
    cycles = _face_cycles()
    tris = [[] for _ in range(256)]
    for case in range(256):
        inside = [(case >> c) & 1 for c in range(8)]
        nxt = {}
        for cyc in cycles:
            s = [inside[c] for c in cyc]
            for j in range(4):
                if s[j] == 0 and s[(j + 1) % 4] == 1:  # the walk enters an inside run on face edge j
                    k = (j + 1) % 4
                    while s[(k + 1) % 4] == 1:
                        k = (k + 1) % 4
                    e_in = _edge_between(cyc[j], cyc[(j + 1) % 4])
                    e_out = _edge_between(cyc[k], cyc[(k + 1) % 4])
                    assert e_in not in nxt
                    nxt[e_in] = e_out
        seen = set()
        for start in sorted(nxt):
            if start in seen:
                continue
            loop, e = [], start
            while e not in seen:
                seen.add(e)
                loop.append(e)
                e = nxt[e]
            assert e == start and len(loop) >= 3
            for i in range(1, len(loop) - 1):
                tris[case].append((loop[0], loop[i], loop[i + 1]))
    n_max = max(len(t) for t in tris)
    table = np.full((256, n_max, 3), -1, np.int8)
    count = np.zeros(256, np.int32)
    for case, t in enumerate(tris):
        count[case] = len(t)
        if t:
            table[case, :len(t)] = np.asarray(t, np.int8)
    return table, count


TRI_TABLE, TRI_COUNT = _build_table()


def marching_cubes(vol, iso):
    """vol [nx, ny, nz] -> (vertices [V, 3] float64 in index units, triangles [T, 3] int64).  Same call shape and
    vertex convention as `mcubes.marching_cubes` (gen_mesh.py:124)."""
    vol = np.ascontiguousarray(vol)
    if vol.ndim != 3:
        raise ValueError("marching_cubes expects a 3-D array")
    shape = vol.shape
    if min(shape) < 2:
        return np.zeros((0, 3)), np.zeros((0, 3), np.int64)
    if not np.isfinite(vol).all():
        raise ValueError("marching_cubes: the volume holds non-finite values (NaN/inf sigma would give NaN vertices)")
    inside = vol >= iso

    # vertices: one per grid edge whose end points straddle iso, numbered axis by axis
    vert_id, verts, n_verts = [], [], 0
    for a in range(3):
        lo = [slice(None)] * 3
        hi = [slice(None)] * 3
        lo[a], hi[a] = slice(0, -1), slice(1, None)
        lo, hi = tuple(lo), tuple(hi)
        cross = inside[lo] != inside[hi]
        idx = np.nonzero(cross)
        ids = np.full(cross.shape, -1, np.int64)
        ids[idx] = n_verts + np.arange(idx[0].size)
        n_verts += idx[0].size
        v0 = vol[lo][idx].astype(np.float64)
        v1 = vol[hi][idx].astype(np.float64)
        p = np.stack(idx, 1).astype(np.float64)
        p[:, a] += (iso - v0) / (v1 - v0)
        vert_id.append(ids)
        verts.append(p)
    vertices = np.concatenate(verts, 0)

    # cells: case index from the eight corner flags
    case = np.zeros(tuple(s - 1 for s in shape), np.uint8)
    for c in range(8):
        sl = tuple(slice(1, None) if (c >> a) & 1 else slice(0, -1) for a in range(3))
        case |= inside[sl].astype(np.uint8) << c
    cells = np.nonzero(TRI_COUNT[case] > 0)
    if cells[0].size == 0:
        return vertices, np.zeros((0, 3), np.int64)
    ccase = case[cells]
    # global vertex id of each of the 12 local edges of every active cell
    cell_edges = np.empty((cells[0].size, 12), np.int64)
    for e, (a, o) in enumerate(_EDGES):
        pos = tuple(cells[d] + ((o >> d) & 1) for d in range(3))
        cell_edges[:, e] = vert_id[a][pos]
    local = TRI_TABLE[ccase]  # [C, n_max, 3]
    valid = local[:, :, 0] >= 0
    ci, ti = np.nonzero(valid)
    tri_local = local[ci, ti].astype(np.int64)
    triangles = cell_edges[ci[:, None], tri_local]
    return vertices, triangles
