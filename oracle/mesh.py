"""oracle.mesh — template mesh IO and adjacency construction (TEST INFRASTRUCTURE ONLY).

Dense, reference-faithful restatement in numpy of
``pterotactyl/utility/utils.py``: ``load_mesh_touch`` (:194-200), ``calc_adj`` (:134-148),
``normalize_adj`` (:47-52), ``adj_fuse_touch`` (:75-130), ``adj_init`` (:56-71),
``load_mesh_vision`` (:30-36).  OBJ parsing restates the part of PyTorch3D's ``load_obj``
the reference uses (``verts``, ``faces.verts_idx``; 1-based → 0-based).
"""
import numpy as np


def load_obj(path):
    """Return (verts float32 (V,3), faces int64 (F,3)).  Polygons are fan-triangulated
    (PyTorch3D ``load_obj`` behaviour); the reference assets are all triangles."""
    verts, faces = [], []
    with open(path) as f:
        for line in f:
            if line.startswith("v "):
                p = line.split()
                verts.append([float(p[1]), float(p[2]), float(p[3])])
            elif line.startswith("f "):
                idx = [int(tok.split("/")[0]) for tok in line.split()[1:]]
                idx = [i - 1 if i > 0 else len(verts) + i for i in idx]
                for k in range(1, len(idx) - 1):
                    faces.append([idx[0], idx[k], idx[k + 1]])
    return np.asarray(verts, dtype=np.float32), np.asarray(faces, dtype=np.int64)


def calc_adj(faces):
    """utils.py:134-148 — binary adjacency with self loops, N = faces.max()+1."""
    n = int(faces.max()) + 1
    adj = np.eye(n, dtype=np.float32)
    v1, v2, v3 = faces[:, 0], faces[:, 1], faces[:, 2]
    adj[v1, v2] = 1
    adj[v1, v3] = 1
    adj[v2, v1] = 1
    adj[v2, v3] = 1
    adj[v3, v1] = 1
    adj[v3, v2] = 1
    return adj


def normalize_adj(mx):
    """utils.py:47-52 — D^-1 A with NaN reciprocals zeroed.  The reference multiplies by a
    dense diagonal matrix; each output entry is then exactly ``r_inv[i] * mx[i, j]`` plus
    exact zeros, so a row scaling is the same arithmetic in float32."""
    mx = np.asarray(mx, dtype=np.float32)
    rowsum = mx.sum(1, dtype=np.float32)
    with np.errstate(divide="ignore", invalid="ignore"):
        r_inv = (np.float32(1.0) / rowsum).astype(np.float32)
    r_inv[r_inv != r_inv] = 0.0
    return (r_inv[:, None] * mx).astype(np.float32)


def adj_fuse_touch(verts, faces, adj, sheet_verts, sheet_faces, num_grasps, finger):
    """utils.py:75-130 — block-diagonal append of ``fingers*num_grasps`` touch-chart graphs,
    links between bit-identical vision vertices, and links seam-vertex <-> chart centre (idx 4)."""
    groups = {}
    for e, v in enumerate(np.asarray(verts, dtype=np.float32)):
        groups.setdefault(v.tobytes(), []).append(e)
    sheet_adj = calc_adj(sheet_faces)
    ns = sheet_adj.shape[0]
    nv = adj.shape[0]
    fingers = 1 if finger else 4
    k = fingers * num_grasps
    central = [4 + i * ns + nv for i in range(k)]
    new_adj = np.zeros((nv + k * ns, nv + k * ns), dtype=np.float32)
    new_adj[:nv, :nv] = adj
    for i in range(k):
        s = nv + ns * i
        new_adj[s:s + ns, s:s + ns] = sheet_adj
    all_faces = [faces]
    for i in range(k):
        all_faces.append(sheet_faces + verts.shape[0] + i * sheet_verts.shape[0])
    faces = np.concatenate(all_faces)
    for cur in groups.values():
        if len(cur) > 1:
            for v1 in cur:
                for v2 in cur:
                    new_adj[v1, v2] = 1
                for c in central:
                    new_adj[v1, c] = 1
                    new_adj[c, v1] = 1
    return new_adj, faces


def adj_init(verts, faces, use_touch=False, num_grasps=5, finger=False,
             sheet_verts=None, sheet_faces=None):
    """utils.py:56-71 — returns {'origional', 'adj', 'faces'} as dense numpy arrays."""
    adj = calc_adj(faces)
    info = {"origional": normalize_adj(adj.copy())}
    if use_touch:
        adj, faces = adj_fuse_touch(verts, faces, adj, sheet_verts, sheet_faces, num_grasps, finger)
    info["adj"] = normalize_adj(adj)
    info["faces"] = faces
    return info


def dense_to_csr(a):
    """Row-major CSR (rowptr int32, col int32, val float32) of the non-zeros of a dense matrix,
    columns ascending within a row."""
    a = np.asarray(a)
    rows, cols = np.nonzero(a)
    rowptr = np.zeros(a.shape[0] + 1, dtype=np.int32)
    np.add.at(rowptr, rows + 1, 1)
    rowptr = np.cumsum(rowptr).astype(np.int32)
    return rowptr, cols.astype(np.int32), a[rows, cols].astype(np.float32)
