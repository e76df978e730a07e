#!/usr/bin/env python
"""SURVEY §8d timing protocol on one MI355X: >= 20 warm-up and >= 100 timed iterations of BASELINE configs[1], a HIP
event pair around every iteration on the compute stream, no host sync inside; median / p10 / p90 per variant:
  * with and without the Adam step,
  * ground-truth clouds on ellipsoid surfaces (headline, "S") and uniform in a cube ("U").
Prints one JSON object (kept under profiles/)."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--warmup", type=int, default=20)
    p.add_argument("--iters", type=int, default=100)
    p.add_argument("--batch", type=int, default=64)
    p.add_argument("--points", type=int, default=10000)
    a = p.parse_args()
    from a3vt_amd.synthetic import make_args
    from a3vt_amd import distributed as adist, mesh as amesh
    from a3vt_amd.pterotactyl.reconstruction.vision import model
    from a3vt_amd.pterotactyl.utility import utils
    from a3vt_amd.synthetic import gt_cloud
    dev = torch.device("cuda", 0)
    args = make_args(number_points=a.points)
    verts, faces = amesh.icosphere(4)
    vt, ft = torch.from_numpy(verts).to(dev), torch.from_numpy(faces).to(dev)
    info = utils.adj_init(vt, ft, args)
    torch.manual_seed(0)
    net = model.Deformation(info, vt, args).to(dev)
    params = list(net.parameters())
    bucket = adist.FlatGradBucket(params)
    opt = torch.optim.Adam(params, lr=args.lr, weight_decay=0, fused=True)
    charts = model.prepare_mesh({"img": torch.zeros(a.batch, 1)}, vt, args)
    img = torch.zeros(a.batch, 1, device=dev)
    out = {"config": f"icosphere-4, 3-stage GCN 20x300, bs={a.batch}, {a.points}-pt Chamfer x3, fp32",
           "warmup": a.warmup, "iters": a.iters, "unit": "ms per iteration"}
    for kind in ("ellipsoid", "cube"):
        clouds = [gt_cloud(a.batch, a.points, seed=i, kind=kind).to(dev) for i in range(4)]
        for adam in (True, False):
            def step(i):
                bucket.zero()
                v = net(img, charts)[0]
                loss = args.loss_coeff * utils.chamfer_distance(v, info["faces_i32"], clouds[i % 4], num=a.points).mean()
                loss.backward()
                bucket.all_reduce_mean()   # single process: gathers the gradients and re-homes .grad
                if adam:
                    opt.step()
            for i in range(a.warmup):
                step(i)
            ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.iters)]
            for i, (e0, e1) in enumerate(ev):
                e0.record()
                step(i)
                e1.record()
            torch.cuda.synchronize()
            ms = np.array([e0.elapsed_time(e1) for e0, e1 in ev])
            out[f"{kind}_{'adam' if adam else 'no_adam'}"] = {
                "median": float(np.median(ms)), "p10": float(np.percentile(ms, 10)), "p90": float(np.percentile(ms, 90)),
                "iters_per_s_at_median": 1e3 / float(np.median(ms))}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
