"""Host-side template-mesh logic: OBJ parsing, icosphere generation, sparse adjacency (CSR) construction.

Mirrors the behaviour of the reference's ``pterotactyl/utility/utils.py``
(``load_mesh_touch`` :194-200, ``calc_adj`` :134-148, ``normalize_adj`` :47-52, ``adj_fuse_touch`` :75-130,
``adj_init`` :56-71) but never forms the dense (N,N) matrix: the row-normalised adjacency is built
directly as CSR (plus the CSR of its transpose, needed by the backward pass because D^-1 A is not
symmetric).  Pure numpy — this is one-time setup, not the hot path.
"""
import os

import numpy as np

CHART_VERTS = 25          # vertices of one touch chart (policies/replay.py:13 BASE_CHART_SIZE)
CHART_CENTRE = 4          # utils.py:95 central_point


def load_obj(path):
    """(verts float32 (V,3), faces int64 (F,3)) from a Wavefront OBJ; 1-based -> 0-based, fan triangulation."""
    verts, faces = [], []
    with open(path) as f:
        for line in f:
            if line.startswith("v "):
                p = line.split()
                verts.append((float(p[1]), float(p[2]), float(p[3])))
            elif line.startswith("f "):
                idx = [int(tok.split("/")[0]) for tok in line.split()[1:]]
                idx = [i - 1 if i > 0 else len(verts) + i for i in idx]
                for k in range(1, len(idx) - 1):
                    faces.append((idx[0], idx[k], idx[k + 1]))
    return np.asarray(verts, dtype=np.float32), np.asarray(faces, dtype=np.int64)


def icosphere(level=4, radius=0.25, spatial_order=True):
    """Subdivided icosahedron: 10*4^level + 2 vertices (2562 at level 4, 10242 at level 5), 20*4^level faces.
    The synthetic benchmark template of BASELINE.json configs[1]/[4].  Vertices are numbered along a
    space-filling curve by default (``reorder_spatially``); ``spatial_order=False`` keeps subdivision order."""
    t = (1.0 + 5.0 ** 0.5) / 2.0
    v = [(-1, t, 0), (1, t, 0), (-1, -t, 0), (1, -t, 0), (0, -1, t), (0, 1, t),
         (0, -1, -t), (0, 1, -t), (t, 0, -1), (t, 0, 1), (-t, 0, -1), (-t, 0, 1)]
    verts = [np.asarray(p, dtype=np.float64) / np.linalg.norm(p) for p in v]
    faces = [(0, 11, 5), (0, 5, 1), (0, 1, 7), (0, 7, 10), (0, 10, 11), (1, 5, 9), (5, 11, 4), (11, 10, 2),
             (10, 7, 6), (7, 1, 8), (3, 9, 4), (3, 4, 2), (3, 2, 6), (3, 6, 8), (3, 8, 9), (4, 9, 5),
             (2, 4, 11), (6, 2, 10), (8, 6, 7), (9, 8, 1)]
    for _ in range(level):
        cache = {}

        def mid(a, b):
            key = (a, b) if a < b else (b, a)
            if key not in cache:
                m = verts[a] + verts[b]
                verts.append(m / np.linalg.norm(m))
                cache[key] = len(verts) - 1
            return cache[key]

        nf = []
        for a, b, c in faces:
            ab, bc, ca = mid(a, b), mid(b, c), mid(c, a)
            nf += [(a, ab, ca), (b, bc, ab), (c, ca, bc), (ab, bc, ca)]
        faces = nf
    verts, faces = np.asarray(verts), np.asarray(faces, dtype=np.int64)
    if spatial_order:
        verts, faces = reorder_spatially(verts, faces)
    return (verts * radius).astype(np.float32), faces


def reorder_spatially(verts, faces):
    """Renumber vertices along a Morton (Z-order) curve of their positions: neighbours in the mesh get nearby indices, so
    the rows a neighbour gather touches are the rows the same workgroup touched a moment ago (L1/L2 locality of the
    CSR aggregation).  Subdivision order scatters them: a level-4 vertex sits thousands of rows from its level-0
    neighbours.  Pure relabelling — geometry and topology are unchanged."""
    v = np.asarray(verts, dtype=np.float64)
    lo, hi = v.min(axis=0), v.max(axis=0)
    q = np.minimum(((v - lo) / np.maximum(hi - lo, 1e-30) * 1024).astype(np.int64), 1023)

    def spread(x):                      # 10 bits -> every third bit
        x = (x | (x << 16)) & 0x030000FF
        x = (x | (x << 8)) & 0x0300F00F
        x = (x | (x << 4)) & 0x030C30C3
        return (x | (x << 2)) & 0x09249249

    code = spread(q[:, 0]) | (spread(q[:, 1]) << 1) | (spread(q[:, 2]) << 2)
    order = np.argsort(code, kind="stable")
    new_index = np.empty(len(order), dtype=np.int64)
    new_index[order] = np.arange(len(order))
    return v[order].astype(np.asarray(verts).dtype), new_index[np.asarray(faces, dtype=np.int64)]


def _edges_from_faces(faces):
    f = np.asarray(faces, dtype=np.int64)
    a = np.concatenate([f[:, 0], f[:, 0], f[:, 1], f[:, 1], f[:, 2], f[:, 2]])
    b = np.concatenate([f[:, 1], f[:, 2], f[:, 0], f[:, 2], f[:, 0], f[:, 1]])
    return a, b


def _csr_from_pairs(rows, cols, n):
    """Binary pattern -> row-normalised CSR (rowptr int32, col int32 ascending per row, val float32 = 1/deg)."""
    key = np.unique(rows.astype(np.int64) * n + cols.astype(np.int64))
    r = (key // n).astype(np.int64)
    c = (key % n).astype(np.int32)
    counts = np.bincount(r, minlength=n)
    rowptr = np.zeros(n + 1, dtype=np.int32)
    rowptr[1:] = np.cumsum(counts)
    with np.errstate(divide="ignore"):
        r_inv = (np.float32(1.0) / counts.astype(np.float32)).astype(np.float32)  # utils.py:48-50
    r_inv[~np.isfinite(r_inv)] = 0.0
    return rowptr, c, r_inv[r].astype(np.float32)


def csr_transpose(rowptr, col, val, n):
    rows = np.repeat(np.arange(n, dtype=np.int64), np.diff(rowptr))
    order = np.lexsort((rows, col))
    t_rows = col[order].astype(np.int64)
    t_cols = rows[order].astype(np.int32)
    rp = np.zeros(n + 1, dtype=np.int32)
    rp[1:] = np.cumsum(np.bincount(t_rows, minlength=n))
    return rp, t_cols, val[order].astype(np.float32)


def vision_pairs(faces, n):
    """calc_adj (utils.py:134-148): self loops + undirected face edges, as (row, col) index arrays."""
    a, b = _edges_from_faces(faces)
    eye = np.arange(n, dtype=np.int64)
    return np.concatenate([eye, a]), np.concatenate([eye, b])


def fused_pairs(verts, faces, sheet_faces, num_grasps, finger, sheet_size=CHART_VERTS):
    """adj_fuse_touch (utils.py:75-130) as index pairs.  Returns (rows, cols, n_total, faces_total)."""
    verts = np.asarray(verts, dtype=np.float32)
    nv = verts.shape[0]
    fingers = 1 if finger else 4
    k = fingers * num_grasps
    n = nv + k * sheet_size
    rows, cols = vision_pairs(faces, nv)
    rr, cc = [rows], [cols]
    sa, sb = _edges_from_faces(sheet_faces)
    seye = np.arange(sheet_size, dtype=np.int64)
    all_faces = [np.asarray(faces, dtype=np.int64)]
    for i in range(k):
        s = nv + sheet_size * i
        rr += [seye + s, sa + s]
        cc += [seye + s, sb + s]
        all_faces.append(np.asarray(sheet_faces, dtype=np.int64) + nv + i * sheet_size)
    # vertices with bit-identical float32 positions (utils.py:80-84) form seam groups
    keys = np.ascontiguousarray(verts).view(np.dtype((np.void, 12))).ravel()
    _, inv, counts = np.unique(keys, return_inverse=True, return_counts=True)
    seam = np.nonzero(counts[inv] > 1)[0]
    centres = np.asarray([CHART_CENTRE + i * sheet_size + nv for i in range(k)], dtype=np.int64)
    order = np.argsort(inv[seam], kind="stable")
    seam_sorted = seam[order]
    grp = inv[seam_sorted]
    starts = np.flatnonzero(np.r_[True, grp[1:] != grp[:-1]])
    ends = np.r_[starts[1:], len(grp)]
    for s0, e0 in zip(starts, ends):
        members = seam_sorted[s0:e0]
        rr.append(np.repeat(members, len(members)))
        cc.append(np.tile(members, len(members)))
    if len(seam) and k:
        rr += [np.repeat(seam, k), np.tile(centres, len(seam))]
        cc += [np.tile(centres, len(seam)), np.repeat(seam, k)]
    return np.concatenate(rr), np.concatenate(cc), n, np.concatenate(all_faces)


class CSRAdjacency:
    """Row-normalised adjacency D^-1 A as CSR + CSR of the transpose (numpy, host)."""

    def __init__(self, rowptr, col, val, n):
        self.n = int(n)
        self.rowptr, self.col, self.val = rowptr, col, val
        self.t_rowptr, self.t_col, self.t_val = csr_transpose(rowptr, col, val, n)

    @property
    def nnz(self):
        return int(self.col.shape[0])

    @classmethod
    def from_pairs(cls, rows, cols, n):
        return cls(*_csr_from_pairs(rows, cols, n), n)

    @classmethod
    def from_dense(cls, a):
        """From a dense, already normalised matrix (e.g. a reference-made adj_info entry)."""
        a = np.asarray(a, dtype=np.float32)
        r, c = np.nonzero(a)
        n = a.shape[0]
        rowptr = np.zeros(n + 1, dtype=np.int32)
        rowptr[1:] = np.cumsum(np.bincount(r, minlength=n))
        return cls(rowptr, c.astype(np.int32), a[r, c].astype(np.float32), n)

    def split(self, hub_degree=64, max_local_degree=12):
        """The matrix as ``D^-1 (P + J)`` (``a3vt_adj_split``, include/a3vt.h): P a sparse symmetric 0/1 pattern, J the complete
        bipartite block S x C that ``adj_fuse_touch`` writes (utils.py:119-128: every seam vertex of the atlas linked to every
        touch-chart centre).  Found from the CSR alone — C = the hub rows (more than ``hub_degree`` entries), S = the vertices
        linked to ALL of them — so a reference-made dense ``adj_info`` entry splits the same way as one built here.  Returns a
        :class:`SplitAdjacency`, or None when the matrix is not of that form (values not 1/row length, pattern not symmetric,
        rows of P longer than ``max_local_degree``): callers then keep the plain CSR.  A hub-free matrix with short rows
        (the vision templates) is the special case S = C = {}."""
        if getattr(self, "_split", False) is not False:
            return self._split
        self._split = None
        n = self.n
        deg = np.diff(self.rowptr).astype(np.int64)
        if n == 0 or deg.min() < 1:
            return None
        rows = np.repeat(np.arange(n, dtype=np.int64), deg)
        scale = (np.float32(1.0) / deg.astype(np.float32)).astype(np.float32)
        if not np.array_equal(self.val.view(np.uint32), scale[rows].view(np.uint32)):
            return None                                                       # not D^-1 x (0/1 pattern)
        if not (np.array_equal(self.t_rowptr, self.rowptr) and np.array_equal(self.t_col, self.col)):
            return None                                                       # pattern not symmetric
        cls = np.zeros(n, dtype=np.uint8)
        centres = np.flatnonzero(deg > hub_degree)
        keep = np.ones(self.nnz, dtype=bool)
        if len(centres):
            is_c = np.zeros(n, dtype=bool)
            is_c[centres] = True
            # S = vertices outside C whose row holds every centre
            hits = np.bincount(rows[is_c[self.col]], minlength=n)
            seam = np.flatnonzero((hits == len(centres)) & ~is_c)
            if len(seam) == 0:
                return None
            is_s = np.zeros(n, dtype=bool)
            is_s[seam] = True
            cls[seam], cls[centres] = 1, 2
            keep = ~((is_s[rows] & is_c[self.col]) | (is_c[rows] & is_s[self.col]))
            # J must be complete in both directions: |S| x |C| entries removed from each side
            if int((is_s[rows] & is_c[self.col]).sum()) != len(seam) * len(centres) or \
                    int((is_c[rows] & is_s[self.col]).sum()) != len(seam) * len(centres):
                return None
        p_deg = np.bincount(rows[keep], minlength=n)
        if p_deg.max() > max_local_degree:
            return None
        p_rowptr = np.zeros(n + 1, dtype=np.int32)
        p_rowptr[1:] = np.cumsum(p_deg)
        self._split = SplitAdjacency(p_rowptr, self.col[keep].astype(np.int32), scale, cls)
        return self._split

    def to_dense(self):
        out = np.zeros((self.n, self.n), dtype=np.float32)
        rows = np.repeat(np.arange(self.n), np.diff(self.rowptr))
        out[rows, self.col] = self.val
        return out


class SplitAdjacency:
    """``D^-1 (P + J)`` on the host: CSR pattern of P, ``scale`` = 1 / row length of the full matrix, ``cls`` = 0 / 1 (seam
    set S) / 2 (centre set C).  Built by :meth:`CSRAdjacency.split`."""

    def __init__(self, rowptr, col, scale, cls):
        self.rowptr, self.col, self.scale, self.cls = rowptr, col, scale, cls
        self.n = int(len(cls))
        self.max_degree = int(np.diff(rowptr).max()) if self.n else 0
        self.n_seam, self.n_centre = int((cls == 1).sum()), int((cls == 2).sum())

    def to_dense(self):
        """The full row-normalised matrix again (tests)."""
        out = np.zeros((self.n, self.n), dtype=np.float32)
        rows = np.repeat(np.arange(self.n), np.diff(self.rowptr))
        out[rows, self.col] = 1.0
        s, c = np.flatnonzero(self.cls == 1), np.flatnonzero(self.cls == 2)
        out[np.ix_(s, c)] = 1.0
        out[np.ix_(c, s)] = 1.0
        return out * self.scale[:, None]


def load_asset(name):
    """Template geometry shipped as data: ``vision_charts`` (the reference's 1824-vertex / 2304-face chart atlas,
    ``pterotactyl/objects/vision_charts.obj``) and ``touch_chart`` (25 vertices / 32 faces,
    ``pterotactyl/objects/touch_chart.obj``), stored as npz arrays (verts float32, faces int32)."""
    here = os.path.dirname(os.path.abspath(__file__))
    z = np.load(os.path.join(here, "assets", name + ".npz"))
    return z["verts"].astype(np.float32), z["faces"].astype(np.int64)
