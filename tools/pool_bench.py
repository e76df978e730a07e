#!/usr/bin/env python
"""Fused image-feature pooling (csrc/pooling.hip) vs the torch formulation of the reference (3 x grid_sample + cat +
permute, vision/model.py:70-103), forward + backward, default pyramid (64x23x23, 128x7x7, 256x3x3)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402


def torch_pool(maps, verts, matrix):
    ones = torch.ones_like(verts[..., :1])
    proj = torch.matmul(torch.cat((verts, ones), dim=-1), matrix.t())
    z = torch.where(proj[..., 2] == 0, torch.full_like(proj[..., 2], 0.1), proj[..., 2])
    xs, ys = proj[..., 1] / z / 256.0, proj[..., 0] / z / 256.0
    grid = torch.stack((ys, xs), dim=-1).unsqueeze(2) * 2 - 1
    return torch.cat([F.grid_sample(b, grid, align_corners=True)[..., 0] for b in maps], dim=1).permute(0, 2, 1)


def main():
    from a3vt_amd import ops
    from a3vt_amd.pterotactyl.reconstruction.vision import model
    dev = torch.device("cuda", 0)
    B, N = 64, 1824
    g = torch.Generator().manual_seed(0)
    verts = ((torch.rand(B, N, 3, generator=g) - 0.5) * 0.5).to(dev).requires_grad_(True)
    maps = [torch.randn(B, c, h, h, generator=g).to(dev).requires_grad_(True) for c, h in ((64, 23), (128, 7), (256, 3))]
    gout = torch.randn(B, N, 448, generator=g).to(dev)
    import types
    enc = model.Image_Encoder(types.SimpleNamespace(CNN_ker_size=5, num_CNN_blocks=6, layers_per_block=3)).to(dev)
    res = {"batch": B, "n_vert": N}
    for name, fn in (("torch_ms", lambda: torch_pool(maps, verts, enc.matrix)), ("fused_ms", lambda: enc.pooling(maps, verts))):
        for _ in range(3):
            fn().backward(gout)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            fn().backward(gout)
        torch.cuda.synchronize()
        res[name] = 1e3 * (time.perf_counter() - t0) / 20
    res["speedup"] = res["torch_ms"] / res["fused_ms"]
    print(json.dumps(res))


if __name__ == "__main__":
    main()
