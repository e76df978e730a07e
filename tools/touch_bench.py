#!/usr/bin/env python
"""Training-step time on the reference's production topology: chart atlas + 20 touch charts (t_g: N = 2324, hub rows of
1153 neighbours), bs = 64, L = 20, H = 300, 10k-point Chamfer x3, fp32, Adam — the touch counterpart of bench.py."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def main():
    from a3vt_amd.synthetic import make_args
    from a3vt_amd import distributed as adist
    from a3vt_amd.pterotactyl.reconstruction.vision import model
    from a3vt_amd.pterotactyl.utility import utils
    from a3vt_amd.synthetic import gt_cloud
    dev = torch.device("cuda", 0)
    B, P = 64, 10000
    prec = sys.argv[sys.argv.index("--precision") + 1] if "--precision" in sys.argv else "fp32"   # fp32 | fp32x3 | bf16 | bf16s
    args = make_args(use_touch=True, finger=False, num_grasps=5, number_points=P, gemm_precision=prec)
    if "--csr-algo" in sys.argv:   # auto | rows | sliced (ops.dbg_csr_algo: A/B of the aggregation kernels)
        from a3vt_amd import ops
        ops.dbg_csr_algo(sys.argv[sys.argv.index("--csr-algo") + 1])
    info, verts = utils.load_mesh_vision(args, "vision_charts")
    torch.manual_seed(0)
    net = model.Deformation(info, verts, args).to(dev)
    params = list(net.parameters())
    bucket = adist.FlatGradBucket(params)
    from a3vt_amd import optim as a3vt_optim
    opt = a3vt_optim.make_adam(params, args.lr)
    g = torch.Generator().manual_seed(0)
    tc = torch.zeros(B, 5, 4, 25, 4)
    tc[..., :3] = (torch.rand(B, 5, 4, 25, 3, generator=g) - 0.5) * 0.3
    tc[..., 3] = 2
    img = torch.zeros(B, 1, device=dev)
    charts = model.prepare_mesh({"img": img, "touch_charts": tc}, verts, args)
    gt = gt_cloud(B, P, 0).to(dev)

    def step():
        bucket.zero()
        v = net(img, charts)[0]
        loss = args.loss_coeff * utils.chamfer_distance(v, info["faces_i32"], gt, num=P).mean()
        loss.backward()
        bucket.all_reduce_mean()   # single process: gathers the gradients and re-homes .grad
        opt.step()

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 10
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / n
    print(json.dumps({"topology": "atlas + 20 touch charts (N=2324, nnz=60726, max degree 1153)", "batch": B, "gemm_precision": prec,
                      "ms_per_step": ms, "iters_per_s": 1e3 / ms}))


if __name__ == "__main__":
    main()
