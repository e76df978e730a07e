"""Shared builders for the parity tests (seeded inputs, oracle <-> HIP plumbing)."""
import numpy as np
import torch

from a3vt_amd import mesh as amesh
from a3vt_amd.synthetic import make_args  # noqa: F401  (re-exported: the tests build their args through helpers)


def template(name):
    """(verts float32 (V,3), faces int64 (F,3)) for 'ico<level>' or 'atlas'."""
    if name.startswith("ico"):
        return amesh.icosphere(int(name[3:]))
    return amesh.load_asset("vision_charts")


def oracle_adj(verts, faces, args):
    """CSR triple (long,long,float tensors) + faces for oracle.gcn.adj_matmul, built by the ORACLE's dense path."""
    from oracle import mesh as omesh
    sv, sf = amesh.load_asset("touch_chart")
    info = omesh.adj_init(verts, faces, args.use_touch, args.num_grasps, args.finger, sv, sf)
    rp, col, val = omesh.dense_to_csr(info["adj"])
    return (torch.from_numpy(rp).long(), torch.from_numpy(col).long(), torch.from_numpy(val)), \
        torch.from_numpy(info["faces"])


def rel_err(a, b):
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


def rel_l2(a, b):
    """Relative L2 error.  Used for GRADIENTS through ReLU layers: a pre-activation within rounding of 0 can
    switch sign between two correct fp32/fp64 evaluations, which flips that unit's gradient (a jump, not a
    rounding error) for a handful of rows at large M; the max-norm then measures the kink, the L2 norm does not."""
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def assert_grad_close(a, b, what="", tol=1e-3, outlier_frac=1e-3, l2_tol=1e-3):
    """Gradient check robust to ReLU kinks: all but a fraction `outlier_frac` of the elements within
    `tol` * max|b|, and the whole tensor within `l2_tol` in relative L2."""
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    scale = b.abs().max().clamp_min(1e-30)
    bad = ((a - b).abs() > tol * scale).double().mean().item()
    assert bad <= outlier_frac, f"{what}: {bad:.2e} of the elements differ by more than {tol} (relative to max)"
    l2 = ((a - b).norm() / b.norm().clamp_min(1e-30)).item()
    assert l2 < l2_tol, f"{what}: relative L2 error {l2:.2e}"


def random_cloud(batch, n, seed, kind="ellipsoid"):
    g = np.random.default_rng(seed)
    if kind == "cube":
        return torch.from_numpy(g.uniform(-0.16, 0.16, (batch, n, 3)).astype(np.float32))
    d = g.normal(size=(batch, n, 3))
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    ax = g.uniform(0.05, 0.16, (batch, 1, 3))
    return torch.from_numpy((d * ax).astype(np.float32))
