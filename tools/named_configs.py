#!/usr/bin/env python
"""One training step timing for BASELINE.json configs[3] and configs[4] on ONE MI355X (synthetic inputs):
  configs[3]  vision + touch: image model (default CNNs) + chart atlas with 4 touch charts (N = 1924), 25 000-point
              Chamfer, bf16 mode (--precision bf16s = bf16 storage, default; bf16 = operands only)
  configs[4]  10 242-vertex icosphere, 50 000-point Chamfer, same mode (per-GPU shard of the 8-GPU config)
Prints one JSON object per config.  These are capability / parity-case configurations, not the headline metric.
--only 3|4 runs one of them (e.g. under rocprofv3)."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def run(name, net, info, charts, img, gt, args, steps=10, warm=8):
    from a3vt_amd import distributed as adist
    from a3vt_amd.pterotactyl.utility import utils
    params = list(net.parameters())
    bucket = adist.FlatGradBucket(params)
    opt = torch.optim.Adam(params, lr=args.lr, fused=True)

    def step():
        bucket.zero()
        v = net(img, charts)[0]
        loss = args.loss_coeff * utils.chamfer_distance(v, info["faces_i32"], gt, num=args.number_points).mean()
        loss.backward()
        bucket.all_reduce_mean()   # single process: gathers the gradients and re-homes .grad
        opt.step()
        return loss

    for _ in range(warm):   # MIOpen's find mode and the allocator's growth take several steps to settle (image mode)
        loss = step()
    torch.cuda.synchronize()
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(steps):
        loss = step()
        marks[i + 1].record()
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / steps
    dev_ms = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(steps))
    print(json.dumps({"config": name, "ms_per_step": ms, "device_ms_median": dev_ms[len(dev_ms) // 2],
                      "iters_per_s": 1e3 / ms, "loss": float(loss), "finite": bool(torch.isfinite(loss))}))


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--batch3", type=int, default=64)
    p.add_argument("--batch4", type=int, default=8)
    p.add_argument("--precision", default="bf16s", choices=["fp32", "bf16", "bf16s", "fp32x3"])
    p.add_argument("--only", type=int, default=0, choices=[0, 3, 4])
    p.add_argument("--steps", type=int, default=10)
    a = p.parse_args()
    from a3vt_amd import mesh as amesh
    from a3vt_amd.pterotactyl.reconstruction.vision import model
    from a3vt_amd.pterotactyl.utility import utils
    from a3vt_amd.synthetic import gt_cloud, make_args
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(0)
    if a.only in (0, 3):
        config3(a, dev, g, model, utils, gt_cloud, make_args)
        torch.cuda.empty_cache()
    if a.only in (0, 4):
        config4(a, dev, amesh, model, utils, gt_cloud, make_args)


def config3(a, dev, g, model, utils, gt_cloud, make_args):
    B = a.batch3
    args = make_args(use_img=True, use_touch=True, finger=False, num_grasps=1, number_points=25000,
                     gemm_precision=a.precision, CNN_ker_size=5, num_CNN_blocks=6, layers_per_block=3)
    info, verts = utils.load_mesh_vision(args, "vision_charts")
    torch.manual_seed(0)
    net = model.Deformation(info, verts, args).to(dev)
    tc = torch.zeros(B, 1, 4, 25, 4)
    tc[..., :3] = (torch.rand(B, 1, 4, 25, 3, generator=g) - 0.5) * 0.3
    tc[..., 3] = 2
    img = torch.rand(B, 3, 256, 256, generator=g).to(dev)
    charts = model.prepare_mesh({"img": img, "touch_charts": tc}, verts, args)
    run(f"configs[3]: image + 4 touch charts (N=1924), 25k-pt Chamfer, {a.precision}, bs={B}", net, info, charts, img,
        gt_cloud(B, 25000, 0).to(dev), args, a.steps)


def config4(a, dev, amesh, model, utils, gt_cloud, make_args):
    B = a.batch4   # one GPU's shard
    args = make_args(number_points=50000, gemm_precision=a.precision)
    v, f = amesh.icosphere(5)
    vt, ft = torch.from_numpy(v).to(dev), torch.from_numpy(f).to(dev)
    info = utils.adj_init(vt, ft, args)
    torch.manual_seed(0)
    net = model.Deformation(info, vt, args).to(dev)
    img = torch.zeros(B, 1, device=dev)
    charts = model.prepare_mesh({"img": img}, vt, args)
    run(f"configs[4] shard: icosphere-5 (N=10242), 50k-pt Chamfer, {a.precision}, bs={B}", net, info, charts, img,
        gt_cloud(B, 50000, 0).to(dev), args, a.steps)


if __name__ == "__main__":
    main()
