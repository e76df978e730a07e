#!/usr/bin/env python
"""Counters of the pruned nearest-neighbour search (library built by `tools/build_variants.sh nn`): blocks evaluated and
point-to-box tests per wave of 64 queries, share of waves that took the exact-tie slow path.
A3VT_LIB=gpurun_variants/liba3vt_NN_STATS.so python tools/nn_stats.py [--shapes 3x64x10000,...] [--gap 0.05]"""
import argparse
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from a3vt_amd import lib, ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--shapes", default="3x64x10000,3x64x25000,3x8x50000")
ap.add_argument("--geometry", default="synthetic", choices=["synthetic", "bench"],
                help="synthetic: sphere 0.4 (+gap) against the ellipsoid (0.5, 0.3, 0.2); bench: what bench.py's untrained network "
                     "sees — a sphere of radius 0.25 against ellipsoids with semi-axes U(0.05, 0.16) (a3vt_amd.synthetic.gt_cloud)")
ap.add_argument("--gap", type=float, default=0.05)
args = ap.parse_args()
L = lib.load()
L.a3vt_dbg_nn_stats.argtypes = [ctypes.c_void_p]
dev = torch.device("cuda", 0)
torch.manual_seed(0)


def surface(*shape, radii):
    u = torch.randn(*shape, 3, device=dev)
    return u / u.norm(dim=-1, keepdim=True) * torch.tensor(radii, device=dev)


for shape in args.shapes.split(","):
    draws, B, N = (int(v) for v in shape.split("x"))
    if args.geometry == "bench":
        from a3vt_amd.synthetic import gt_cloud
        x, y = surface(draws, B, N, radii=(0.25, 0.25, 0.25)), gt_cloud(B, N, 0).to(dev)
    x = x if args.geometry == "bench" else surface(draws, B, N, radii=(0.4, 0.4, 0.4)) + args.gap
    y = y if args.geometry == "bench" else surface(B, N, radii=(0.5, 0.3, 0.2))
    out = (ctypes.c_ulonglong * 16)()
    torch.cuda.synchronize()
    L.a3vt_dbg_nn_stats(out)
    ops.chamfer_nn(x, y, algo="pruned")
    torch.cuda.synchronize()
    L.a3vt_dbg_nn_stats(out)
    w = max(out[0], 1)
    print(f"{shape:>14s} gap {args.gap}: {out[0]} waves, {out[1] / w:.1f} blocks ({out[5] / w:.1f} groups of 16) evaluated and {out[2] / w:.1f} point-box tests per wave "
          f"(of {(N + 63) // 64} blocks; shader cycles per wave: {out[8] / w:.0f} = seed search {out[9] / w:.0f} + seed evaluation {out[10] / w:.0f} + "
          f"tests {out[11] / w:.0f} + evaluations {out[12] / w:.0f} + rest {(out[8] - out[9] - out[10] - out[11] - out[12]) / w:.0f}; "
          f"waves of the kernel in flight when a wave starts: {out[14] / w:.0f} = {out[14] / w / 1024:.2f} per SIMD; slowest wave {out[13]}); {out[3] / w:.1f} blocks needed by some lane given the final minima, {out[4] / w / 64:.1f} by a lane on average; worst wave: {out[6]} groups")
