"""oracle.chamfer — surface sampling + Chamfer distance on the CPU (TEST INFRASTRUCTURE ONLY).

Follows ``pterotactyl/utility/utils.py``: ``batch_sample`` (:152-187) and ``chamfer_distance`` (:204-217),
and restates the PyTorch3D 0.5.0 functions those call (not vendored in the reference → parity
unpinned at this boundary, see ``oracle/__init__.py``):

* ``mesh_face_areas_normals``  → :func:`face_areas`       (areas = 0.5*|(v1-v0) x (v2-v0)|)
* ``_rand_barycentric_coords`` → :func:`barycentric`      (w0=1-sqrt(u), w1=sqrt(u)(1-v), w2=sqrt(u)v)
* ``chamfer_distance(x, y, batch_reduction=None)`` → :func:`chamfer_pair`
  (squared-L2 K=1 nearest neighbour both ways, mean over points of each cloud, summed)

All functions take / return torch CPU tensors and are differentiable through autograd where the
reference is; :func:`chamfer_grad_x` is the closed-form gradient.
"""
import ctypes
import os
import subprocess

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build_c(force=False):
    """Compile ``oracle/chamfer_nn.c`` → ``oracle/libchamfer_nn.so`` (gcc, OpenMP)."""
    src = os.path.join(_HERE, "chamfer_nn.c")
    out = os.path.join(_HERE, "libchamfer_nn.so")
    if force or not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(src):
        subprocess.check_call(["gcc", "-O3", "-ffp-contract=off", "-mfma", "-fopenmp", "-shared", "-fPIC", src, "-o", out,
                               "-lm"])
    return out


def _lib():
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(build_c())
        _LIB.oracle_nn_sqdist.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int,
                                          ctypes.c_void_p, ctypes.c_void_p]
        _LIB.oracle_nn_sqdist.restype = None
        _LIB.oracle_nn_sqdist_fma.argtypes = _LIB.oracle_nn_sqdist.argtypes
        _LIB.oracle_nn_sqdist_fma.restype = None
    return _LIB


def set_threads(n):
    """OpenMP thread count of the C searches (bench.py's 1-core leg)."""
    _lib().oracle_set_threads(int(n))


def nn_sqdist_c(x, y, fma=False):
    """Brute-force squared-L2 nearest neighbour of every row of x (P,3) in y (Q,3), float32, in C.
    Returns (dist float32 (P,), idx int32 (P,)); first minimum wins ties.  ``fma``: contract the products the way the
    device compiler does (chamfer_nn.c), which makes the result comparable bit for bit with the HIP kernels."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    y = np.ascontiguousarray(y, dtype=np.float32)
    dist = np.empty(x.shape[0], dtype=np.float32)
    idx = np.empty(x.shape[0], dtype=np.int32)
    fn = _lib().oracle_nn_sqdist_fma if fma else _lib().oracle_nn_sqdist
    fn(x.ctypes.data, x.shape[0], y.ctypes.data, y.shape[0], dist.ctypes.data, idx.ctypes.data)
    return dist, idx


def chamfer_grad_from_indices(x, y, ixy, iyx, g):
    """Closed-form gradients of sum_b g_b * cd_b for ONE cloud pair given the nearest-neighbour indices,
    in float64: returns (grad_x (P,3), grad_y (Q,3)).  For sizes where :func:`chamfer_grad_x`'s torch search is slow."""
    x, y = x.double(), y.double()
    ixy, iyx = ixy.long(), iyx.long()
    a = g * (2.0 / x.shape[0]) * (x - y[ixy])            # d/dx of mean_i |x_i - y_nn(i)|^2
    b = g * (2.0 / y.shape[0]) * (x[iyx] - y)            # d/dx_nn(j) of mean_j |y_j - x_nn(j)|^2
    gx = a.clone().index_add_(0, iyx, b)
    gy = (-b).index_add_(0, ixy, -a)
    return gx, gy


def nn_sqdist(x, y, chunk=2048):
    """Same as :func:`nn_sqdist_c` with torch ops (any dtype): direct (x-y)^2 sums, no |x|^2+|y|^2 expansion."""
    dists, idxs = [], []
    for s in range(0, x.shape[0], chunk):
        d = ((x[s:s + chunk, None, :] - y[None, :, :]) ** 2).sum(-1)
        m, i = d.min(dim=1)
        dists.append(m)
        idxs.append(i)
    return torch.cat(dists), torch.cat(idxs)


def face_areas(verts_flat, faces_flat):
    """PyTorch3D ``mesh_face_areas_normals`` areas: 0.5*||(v1-v0) x (v2-v0)|| per face."""
    v0, v1, v2 = verts_flat[faces_flat[:, 0]], verts_flat[faces_flat[:, 1]], verts_flat[faces_flat[:, 2]]
    return 0.5 * torch.linalg.norm(torch.cross(v1 - v0, v2 - v0, dim=1), dim=1)


def face_probabilities(verts, faces):
    """utils.py:163-168 — per-mesh area distribution with the reference's NaN scrubs. verts (B,N,3), faces (F,3)."""
    bs = verts.shape[0]
    with torch.no_grad():
        F = faces.unsqueeze(0).repeat(bs, 1, 1) + verts.shape[1] * torch.arange(bs).view(-1, 1, 1)
        areas = face_areas(verts.reshape(-1, 3), F.reshape(-1, 3))
        Ar = areas.reshape(bs, -1).clone()
        Ar[Ar != Ar] = 0
        Ar = torch.abs(Ar / Ar.sum(1).unsqueeze(1))
        Ar[Ar != Ar] = 1
    return Ar


def barycentric(u, v):
    """PyTorch3D ``_rand_barycentric_coords`` given the uniforms: returns (w0, w1, w2)."""
    su = u.sqrt()
    return 1.0 - su, su * (1.0 - v), su * v


def draw_samples(prob, num):
    """The reference's two RNG calls in order (utils.py:170, then ``torch.rand(2, bs, num)`` inside
    ``_rand_barycentric_coords`` at :179) → (face_idx (B,num) int64, u (B,num), v (B,num))."""
    face_idx = prob.multinomial(num, replacement=True)
    uv = torch.rand(2, prob.shape[0], num, dtype=prob.dtype)
    return face_idx, uv[0], uv[1]


def sample_points(verts, faces, face_idx, u, v):
    """utils.py:174-187 with injected samples: p = w0*A + w1*B + w2*C.  Differentiable w.r.t. verts."""
    tri = verts[:, faces]                                   # (B,F,3,3)
    bidx = torch.arange(verts.shape[0]).unsqueeze(1)
    sel = tri[bidx, face_idx]                               # (B,num,3,3)
    w0, w1, w2 = barycentric(u, v)
    return w0[:, :, None] * sel[:, :, 0] + w1[:, :, None] * sel[:, :, 1] + w2[:, :, None] * sel[:, :, 2]


def batch_sample(verts, faces, num=10000):
    """utils.py:152-187 drawing from torch's global RNG exactly as the reference does."""
    face_idx, u, v = draw_samples(face_probabilities(verts, faces), num)
    return sample_points(verts, faces, face_idx, u, v)


def chamfer_pair(x, y, use_c=False):
    """PyTorch3D ``chamfer_distance(x, y, batch_reduction=None)[0]`` → (B,).  x (B,P,3), y (B,Q,3).
    Differentiable w.r.t. both clouds (gradient flows through the gathered nearest neighbours)."""
    out = []
    for b in range(x.shape[0]):
        if use_c:
            ixy = torch.from_numpy(nn_sqdist_c(x[b].detach().float().numpy(), y[b].detach().float().numpy())[1]).long()
            iyx = torch.from_numpy(nn_sqdist_c(y[b].detach().float().numpy(), x[b].detach().float().numpy())[1]).long()
        else:
            with torch.no_grad():
                ixy = nn_sqdist(x[b], y[b])[1]
                iyx = nn_sqdist(y[b], x[b])[1]
        dxy = ((x[b] - y[b][ixy]) ** 2).sum(-1)
        dyx = ((y[b] - x[b][iyx]) ** 2).sum(-1)
        out.append(dxy.mean() + dyx.mean())
    return torch.stack(out)


def chamfer_grad_x(x, y, g):
    """Closed-form d(sum_b g_b * cd_b)/dx: (2/P)(x_i - y_nn(i)) + sum_{j: nn(j)=i} (2/Q)(x_i - y_j)."""
    grad = torch.zeros_like(x)
    for b in range(x.shape[0]):
        ixy = nn_sqdist(x[b], y[b])[1]
        iyx = nn_sqdist(y[b], x[b])[1]
        grad[b] += g[b] * (2.0 / x.shape[1]) * (x[b] - y[b][ixy])
        grad[b].index_add_(0, iyx, g[b] * (2.0 / y.shape[1]) * (x[b][iyx] - y[b]))
    return grad


def chamfer_distance(verts, faces, gt_points, num=1000, repeat=3, samples=None, use_c=False):
    """utils.py:204-217 — mean over ``repeat`` draws of chamfer(sampled surface, gt).  ``samples`` may
    inject a list of ``repeat`` (face_idx, u, v) triples; otherwise draws come from torch's global RNG
    in the reference's order."""
    cds = []
    for r in range(repeat):
        if samples is None:
            pred = batch_sample(verts, faces, num)
        else:
            pred = sample_points(verts, faces, *samples[r])
        cds.append(chamfer_pair(pred, gt_points, use_c=use_c))
    return torch.stack(cds).mean(dim=0) if repeat > 1 else cds[0]
