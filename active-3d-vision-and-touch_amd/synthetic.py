"""Synthetic inputs of the benchmark shapes: seeded ground-truth clouds and loader-format batches.

The reference's dataset (``download_data.sh``) is not available offline; these batches have the wire format
of ``mesh_loader_vision.collate`` (``utility/data_loaders.py:249-258``): ``gt_points (B,P,3)``,
``img`` (dummy ``(B,1)`` when ``use_img`` is off), ``touch_charts`` and ``names``.
"""
from types import SimpleNamespace

import numpy as np
import torch


def make_args(**kw):
    """Argument namespace with the reference trainer's defaults for this path (``vision/train.py:375-386``: 20 GCN
    layers x 300 hidden, cut 0.33, loss_coeff 9000, Adam lr 3e-4) — what ``Engine`` / ``Deformation`` read from
    ``args``.  Keyword arguments override; ``num_stages`` and ``gemm_precision`` are this package's own knobs."""
    d = dict(use_img=False, use_touch=False, finger=False, num_grasps=1, num_GCN_layers=20, hidden_GCN_size=300,
             cut=0.33, number_points=1000, loss_coeff=9000.0, lr=3e-4, seed=0, num_stages=3)
    d.update(kw)
    return SimpleNamespace(**d)


def gt_cloud(batch, points, seed=0, kind="ellipsoid"):
    """'ellipsoid': points on random ellipsoid surfaces with semi-axes U(0.05,0.16) (headline, surface-like);
    'cube': uniform in [-0.16,0.16]^3 (dataset objects are scaled to ~0.32 extent, utils.py:348-356)."""
    g = np.random.default_rng(seed)
    if kind == "cube":
        return torch.from_numpy(g.uniform(-0.16, 0.16, (batch, points, 3)).astype(np.float32))
    d = g.normal(size=(batch, points, 3))
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    ax = g.uniform(0.05, 0.16, (batch, 1, 3))
    return torch.from_numpy((d * ax).astype(np.float32))


def touch_charts(batch, args, seed=0):
    """Loader-format touch tensor: (B,G,4,25,4) or (B,G,25,4) when ``finger``; masks in {0,1,2} per chart."""
    g = np.random.default_rng(seed + 1)
    shape = (batch, args.num_grasps, 25, 4) if args.finger else (batch, args.num_grasps, 4, 25, 4)
    t = np.zeros(shape, dtype=np.float32)
    centre = g.uniform(-0.12, 0.12, shape[:-2] + (1, 3))
    t[..., :3] = centre + g.normal(scale=0.004, size=shape[:-1] + (3,))
    t[..., 3] = g.integers(0, 3, shape[:-2] + (1,))
    t[..., :3] *= (t[..., 3:] > 0)  # empty slots are all-zero (data_loaders.py:222-228)
    return torch.from_numpy(t)


class SyntheticLoader:
    """Iterable of `steps` identical-shape batches (fresh clouds per step, deterministic in `seed`)."""

    def __init__(self, args, steps, batch_size, seed=0, kind="ellipsoid"):
        self.args, self.steps, self.batch_size, self.seed, self.kind = args, steps, batch_size, seed, kind

    def __len__(self):
        return self.steps

    def __iter__(self):
        for s in range(self.steps):
            b = {"names": [(f"synthetic_{self.seed}_{s}_{i}", []) for i in range(self.batch_size)],
                 "gt_points": gt_cloud(self.batch_size, self.args.number_points, self.seed * 100003 + s, self.kind),
                 "img": torch.zeros(self.batch_size, 1)}
            b["touch_charts"] = touch_charts(self.batch_size, self.args, self.seed * 7 + s) \
                if self.args.use_touch else torch.ones(self.batch_size, 1)
            yield b
