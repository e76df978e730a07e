#!/usr/bin/env python
"""Time the Chamfer forward (nearest neighbour both ways + reduce) for each search algorithm on surface-like clouds:
the headline shape (3 draws x 64 clouds x 10,000 points) and the configs[3] / configs[4] cloud sizes.  Development aid:
python tools/chamfer_bench.py [--algos sweep,pruned] [--shapes 3x64x10000,3x64x25000,3x8x50000] [--gap 0.05]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from a3vt_amd import ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--algos", default="sweep,pruned")
ap.add_argument("--shapes", default="3x64x10000,3x64x25000,3x8x50000")
ap.add_argument("--geometry", default="synthetic", choices=["synthetic", "bench"],
                help="synthetic: sphere 0.4 (+gap) against the ellipsoid (0.5, 0.3, 0.2); bench: what bench.py's untrained network "
                     "sees — a sphere of radius 0.25 against ellipsoids with semi-axes U(0.05, 0.16) (a3vt_amd.synthetic.gt_cloud)")
ap.add_argument("--gap", type=float, default=0.05, help="offset between the predicted and the ground-truth surface")
ap.add_argument("--reps", type=int, default=5)
args = ap.parse_args()

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
    x = x if args.geometry == "bench" else surface(draws, B, N, radii=(0.4, 0.4, 0.4)) + args.gap         # the predicted surface: a sphere, a little off
    y = y if args.geometry == "bench" else surface(B, N, radii=(0.5, 0.3, 0.2))                            # ground truth: an ellipsoid
    ref = None
    for algo in args.algos.split(","):
        out = ops.chamfer_nn(x, y, algo=algo)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.reps):
            ops.chamfer_nn(x, y, algo=algo)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / args.reps
        same = "" if ref is None else ("  == first" if all(torch.equal(a, b) for a, b in zip(ref, out)) else "  DIFFERS")
        ref = ref or out
        print(f"{shape:>14s} {algo:9s} {ms:9.3f} ms per call   cd sum {out[4].sum().item():.6f}{same}", flush=True)
