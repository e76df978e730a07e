#!/usr/bin/env python
"""Developer aid (GPU box): the same training run in the fp32 / bf16 / bf16s gemm modes — loss per step, and the
per-tensor relative L2 distance of the first step's gradients from the fp32 ones.  python tools/mode_compare.py [--steps 12]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

p = argparse.ArgumentParser()
p.add_argument("--steps", type=int, default=12)
p.add_argument("--batch", type=int, default=16)
p.add_argument("--points", type=int, default=4000)
p.add_argument("--level", type=int, default=4)
p.add_argument("--modes", default="fp32,fp32x3,bf16,bf16s")
a = p.parse_args()

from a3vt_amd import mesh as amesh  # noqa: E402
from a3vt_amd.pterotactyl.reconstruction.vision import model  # noqa: E402
from a3vt_amd.pterotactyl.utility import utils  # noqa: E402
from a3vt_amd.synthetic import gt_cloud, make_args  # noqa: E402

dev = torch.device("cuda", 0)
v, f = amesh.icosphere(a.level)
vt, ft = torch.from_numpy(v).to(dev), torch.from_numpy(f).to(dev)
clouds = [gt_cloud(a.batch, a.points, seed=i).to(dev) for i in range(4)]
g = torch.Generator().manual_seed(1)
samples = [(torch.randint(0, f.shape[0], (3, a.batch, a.points), generator=g).to(torch.int32).to(dev),
            torch.rand(3, a.batch, a.points, generator=g).to(dev), torch.rand(3, a.batch, a.points, generator=g).to(dev))
           for _ in range(a.steps)]
first = {}
for mode in a.modes.split(","):
    args = make_args(gemm_precision=mode)
    info = utils.adj_init(vt, ft, args)
    torch.manual_seed(0)
    net = model.Deformation(info, vt, args).to(dev)
    opt = torch.optim.Adam(net.parameters(), lr=3e-4)
    charts = model.prepare_mesh({"img": torch.zeros(a.batch, 1)}, vt, args)
    losses = []
    for s in range(a.steps):
        opt.zero_grad()
        out = net(torch.zeros(a.batch, 1), charts)[0]
        loss = 9000.0 * utils.chamfer_distance(out, info["faces"], clouds[s % 4], num=a.points, samples=samples[s]).mean()
        loss.backward()
        if s == 0:
            first[mode] = {n: q.grad.detach().clone() for n, q in net.named_parameters()}
        opt.step()
        losses.append(loss.item())
    print(f"{mode:6s} losses: " + " ".join(f"{x:10.3f}" for x in losses))
ref = first.get("fp32")
if ref:
    for mode, gr in first.items():
        if mode == "fp32":
            continue
        worst = sorted(((((gr[n] - ref[n]).norm() / ref[n].norm().clamp_min(1e-30)).item(), n) for n in ref), reverse=True)
        print(f"{mode}: first-step gradient vs fp32, relative L2 per tensor — worst 8:")
        for e, n in worst[:8]:
            print(f"   {e:9.3e}  {n}  (|g| {ref[n].norm().item():.3e})")
        print(f"   median {sorted(e for e, _ in worst)[len(worst) // 2]:.3e}")
