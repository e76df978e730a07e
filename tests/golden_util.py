"""Helpers shared by the golden-vector tests: load fixtures, rebuild the reference's inputs from seeds."""
import hashlib
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(GOLDEN, name))


def state_from(z, dtype=torch.float32):
    return {k[2:]: torch.from_numpy(z[k]).to(dtype) for k in z.files if k.startswith("w:")}


def grads_from(z):
    return {k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("g:")}


def csr_from(z, tag, key):
    return (torch.from_numpy(z[f"{tag}_{key}_rowptr"]).long(), torch.from_numpy(z[f"{tag}_{key}_col"].astype(np.int64)),
            torch.from_numpy(z[f"{tag}_{key}_val"]))


def g7_cloud(B=2, P=10000, seed=99):
    """The ground-truth cloud make_golden.g7 builds (same torch generator calls)."""
    g = torch.Generator().manual_seed(seed)
    d = torch.randn(B, P, 3, generator=g)
    return d / d.norm(dim=-1, keepdim=True) * (0.05 + 0.11 * torch.rand(B, 1, 3, generator=g))


def state_sha256(sd):
    """SHA-256 over sorted (key, raw bytes) of a state dict — make_golden.state_checksum."""
    h = hashlib.sha256()
    for k in sorted(sd):
        h.update(k.encode())
        h.update(sd[k].detach().cpu().numpy().tobytes())
    return np.frombuffer(h.digest(), dtype=np.uint8)


def write_mini_dataset(root, n=6, seed=0):
    """A miniature dataset in the reference's on-disk layout (utility/data_loaders.py:18-29), deterministic in ``seed``:
    ``n`` objects "0".."n-1" (the first 4 in recon_train, the rest in valid).  Used by make_golden.g11 (read by the
    REFERENCE loader) and by the CPU test that reads it with the mirror."""
    rng = np.random.default_rng(seed)
    ids = [str(i) for i in range(n)]
    for sub in ("point_cloud_info", "images_colourful", "touch_charts", "object_info"):
        os.makedirs(os.path.join(root, sub), exist_ok=True)
    for i in ids:
        np.save(os.path.join(root, "point_cloud_info", f"{i}.npy"), (0.1 * rng.standard_normal((3000, 3))).astype(np.float64))
        np.save(os.path.join(root, "images_colourful", f"{i}.npy"), rng.integers(0, 256, (256, 256, 3), dtype=np.uint8))
        os.makedirs(os.path.join(root, "touch_charts", i), exist_ok=True)
        tc = rng.standard_normal((50, 4, 25, 4)).astype(np.float32)
        tc[..., 3] = rng.integers(0, 3, (50, 4, 1))
        np.save(os.path.join(root, "touch_charts", i, "touch_charts.npy"), tc.reshape(50, 4, 100))
    np.save(os.path.join(root, "data_split.npy"),
            {"recon_train": ids[:4], "valid": ids[4:], "test": ids[4:], "auto_train": []})
    return ids
