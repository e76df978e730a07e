"""CPU: an anchor for the Chamfer / face-area core that does not come from this repository.

The reference takes nearest neighbours, face areas and barycentric weights from PyTorch3D 0.5.0 (utility/utils.py:20-23),
which is neither vendored nor installable here, so the goldens g3/g5/g6/g7/g8/g9 pin that core to the oracle's own
restatement (``pytorch3d_restated=True``).  This file removes the oracle-versus-itself loop for the arithmetic that can
be checked by a third party: SciPy's ``cKDTree`` (an exact k-d tree search, float64) for the nearest neighbour in both
directions — on the g6 clouds, on the bundled ABC object's cloud and at the benchmark size — and float64 Heron /
Lagrange-identity forms for the triangle areas.  It does not make the oracle "PyTorch3D-pinned"; it shows that what the
oracle calls the nearest neighbour and the area is what an independent implementation calls them.
"""
import numpy as np
import pytest
import torch
from scipy.spatial import cKDTree

from golden_util import load
from helpers import random_cloud, template
from oracle import chamfer as och


def _check_nn(x, y, fma, require_clear=True):
    """oracle.nn_sqdist_c(x -> y) against cKDTree: distances to 1e-6 relative, indices wherever the runner-up is not
    within 1e-6 relative of the minimum (a float32 evaluation may order such a near-tie either way)."""
    d_o, i_o = och.nn_sqdist_c(x, y, fma=fma)
    dk, ik = cKDTree(y.astype(np.float64)).query(x.astype(np.float64), k=2 if y.shape[0] > 1 else 1)
    if y.shape[0] == 1:
        dk, ik = dk[:, None], ik[:, None]
    d2 = dk ** 2
    np.testing.assert_allclose(d_o, d2[:, 0], rtol=1e-5, atol=1e-12)
    clear = np.ones(len(x), bool) if y.shape[0] == 1 else (d2[:, 1] - d2[:, 0]) > 1e-6 * np.maximum(d2[:, 1], 1e-30)
    assert np.array_equal(i_o[clear], ik[clear, 0].astype(np.int32))
    # everywhere (near-ties and exact ties between duplicated target points included): the point the oracle picked is
    # as near as the tree's nearest, evaluated in float64
    picked = ((x.astype(np.float64) - y.astype(np.float64)[i_o]) ** 2).sum(1)
    assert np.all(picked <= d2[:, 0] * (1 + 1e-6) + 1e-14)
    # exact ties: among bit-identical target points the lowest index wins (the reference's strict scan)
    _, first, inverse = np.unique(y, axis=0, return_index=True, return_inverse=True)
    lowest = np.minimum.reduceat(np.argsort(inverse, kind="stable"), np.r_[0, np.cumsum(np.bincount(inverse.ravel()))[:-1]])
    assert np.array_equal(i_o, lowest[inverse.ravel()[i_o]].astype(np.int32))
    if require_clear:
        assert clear.mean() > 0.99
    return d_o, i_o


@pytest.mark.parametrize("fma", [False, True])
def test_nn_matches_ckdtree_on_g6_clouds(fma):
    z = load("g6_chamfer.npz")
    for b in range(z["x"].shape[0]):
        _check_nn(z["x"][b], z["y"][b], fma)
        _check_nn(z["y"][b], z["x"][b], fma)


def test_chamfer_pair_matches_ckdtree_on_g6_and_golden_value():
    z = load("g6_chamfer.npz")
    x, y = z["x"].astype(np.float64), z["y"].astype(np.float64)
    cd = []
    for b in range(x.shape[0]):
        dxy = cKDTree(y[b]).query(x[b])[0] ** 2
        dyx = cKDTree(x[b]).query(y[b])[0] ** 2
        cd.append(dxy.mean() + dyx.mean())
    cd = np.array(cd)
    got = och.chamfer_pair(torch.from_numpy(z["x"]), torch.from_numpy(z["y"])).numpy()
    np.testing.assert_allclose(got, cd, rtol=2e-6)
    np.testing.assert_allclose(z["cd"], cd, rtol=2e-6)          # the committed golden value itself
    got_c = och.chamfer_pair(torch.from_numpy(z["x"]), torch.from_numpy(z["y"]), use_c=True).numpy()
    np.testing.assert_allclose(got_c, cd, rtol=2e-6)


def test_nn_matches_ckdtree_on_the_abc_object_cloud():
    """The bundled ABC object's cloud (a thin rod, 2 176 points) against the undeformed atlas — fixture g6 / g1."""
    z, g1 = load("g6_chamfer.npz"), load("g1_adjacency.npz")
    cloud, verts = z["abc_cloud"], g1["verts"].astype(np.float32)
    _check_nn(verts, cloud, False)
    _check_nn(cloud, verts, False, require_clear=False)   # the atlas duplicates its seam vertices: exact ties


def test_nn_matches_ckdtree_at_benchmark_size():
    """10 000 sphere-surface points against 10 000 ellipsoid points (BASELINE configs[1] shape), both directions."""
    y = random_cloud(1, 10000, 5)[0].numpy()
    v, f = template("ico4")
    g = torch.Generator().manual_seed(3)
    fi = torch.randint(0, f.shape[0], (1, 10000), generator=g)
    x = och.sample_points(torch.from_numpy(v)[None], torch.from_numpy(f.astype(np.int64)), fi,
                          torch.rand(1, 10000, generator=g), torch.rand(1, 10000, generator=g))[0].numpy()
    _check_nn(x, y, True)
    _check_nn(y, x, True)


def test_face_areas_match_heron_and_lagrange_in_float64():
    g1 = load("g1_adjacency.npz")
    verts = g1["verts"].astype(np.float64)
    faces = g1["vision_faces"].astype(np.int64) if "vision_faces" in g1 else g1["t_g_faces"].astype(np.int64)
    faces = faces[faces.max(1) < verts.shape[0]]
    a, b, c = verts[faces[:, 0]], verts[faces[:, 1]], verts[faces[:, 2]]
    # Heron from the three side lengths (numerically stable Kahan ordering)
    s = np.sort(np.stack([np.linalg.norm(b - a, axis=1), np.linalg.norm(c - b, axis=1), np.linalg.norm(a - c, axis=1)]), 0)
    z_, y_, x_ = s
    heron = 0.25 * np.sqrt(np.maximum((x_ + (y_ + z_)) * (z_ - (x_ - y_)) * (z_ + (x_ - y_)) * (x_ + (y_ - z_)), 0.0))
    # Lagrange identity: |u x v|^2 = |u|^2 |v|^2 - (u.v)^2
    u, w = b - a, c - a
    lagr = 0.5 * np.sqrt(np.maximum((u * u).sum(1) * (w * w).sum(1) - (u * w).sum(1) ** 2, 0.0))
    got64 = och.face_areas(torch.from_numpy(verts), torch.from_numpy(faces)).numpy()
    np.testing.assert_allclose(got64, heron, rtol=1e-9, atol=1e-15)
    np.testing.assert_allclose(got64, lagr, rtol=1e-7, atol=1e-15)
    got32 = och.face_areas(torch.from_numpy(verts.astype(np.float32)), torch.from_numpy(faces)).numpy()
    np.testing.assert_allclose(got32, heron, rtol=2e-5, atol=1e-9)
    # the sampling distribution the reference builds from them (utils.py:163-168) sums to one per mesh
    p = och.face_probabilities(torch.from_numpy(verts.astype(np.float32))[None], torch.from_numpy(faces)).numpy()
    np.testing.assert_allclose(p.sum(1), 1.0, rtol=1e-5)
    np.testing.assert_allclose(p[0], heron / heron.sum(), rtol=1e-4, atol=1e-9)


def test_barycentric_weights_are_a_partition_of_unity_and_uniform_on_the_triangle():
    """w0 = 1 - sqrt(u), w1 = sqrt(u)(1 - v), w2 = sqrt(u) v: non-negative, sum to one, and the induced density on the
    triangle is uniform — the centroid of many samples is the triangle's centroid and each sub-triangle of a 4-way
    midpoint split receives a quarter of them."""
    g = torch.Generator().manual_seed(0)
    u, v = torch.rand(200000, generator=g, dtype=torch.float64), torch.rand(200000, generator=g, dtype=torch.float64)
    w0, w1, w2 = och.barycentric(u, v)
    assert (w0 >= 0).all() and (w1 >= 0).all() and (w2 >= 0).all()
    np.testing.assert_allclose((w0 + w1 + w2).numpy(), 1.0, rtol=0, atol=1e-15)
    np.testing.assert_allclose([w0.mean().item(), w1.mean().item(), w2.mean().item()], [1 / 3] * 3, atol=3e-3)
    corner = [(w > 0.5).double().mean().item() for w in (w0, w1, w2)]
    middle = ((w0 <= 0.5) & (w1 <= 0.5) & (w2 <= 0.5)).double().mean().item()
    np.testing.assert_allclose(corner + [middle], [0.25] * 4, atol=4e-3)
