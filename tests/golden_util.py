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
