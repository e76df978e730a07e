#!/usr/bin/env python
"""PROBE (round 5): can a whole training step of BASELINE configs[3] / configs[4] be captured in a HIP graph, and what would a
replay save?  The captured step freezes the surface sampler's Philox seed (a host value today), so this measures the
POTENTIAL only — not a shippable training loop.  python tools/experiments/graph_step_probe_r05.py [--which 3]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from a3vt_amd import distributed as adist  # noqa: E402
from a3vt_amd.pterotactyl.utility import utils  # noqa: E402
from a3vt_amd.synthetic import named_config  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--which", type=int, default=3)
a = ap.parse_args()
dev = torch.device("cuda", 0)
c = named_config(a.which, dev, "bf16s")
net, args = c["net"], c["args"]
params = list(net.parameters())
bucket = adist.FlatGradBucket(params)
opt = torch.optim.Adam(params, lr=torch.tensor(args.lr, device=dev), fused=True, capturable=True)


def step():
    bucket.zero()
    v = net(c["img"], c["charts"])[0]
    loss = args.loss_coeff * utils.chamfer_distance(v, c["info"]["faces_i32"], c["gt"], num=args.number_points).mean()
    loss.backward()
    bucket.all_reduce_mean()
    opt.step()
    return loss.detach()


def timeit(fn, n=10):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n


for _ in range(8):
    step()
print(f"eager: {timeit(step):.2f} ms/step")
try:
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            step()
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = step()
    print(f"graph replay: {timeit(g.replay):.2f} ms/step (loss {float(out):.3f})")
except Exception as e:  # noqa: BLE001
    print("capture failed:", type(e).__name__, str(e)[:300])
