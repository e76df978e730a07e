#!/usr/bin/env python
"""Candidate-touch scoring throughput: the reference's sequential loop (one compute_obs per candidate,
policies/environment.py:174-180) vs one batched call (a3vt_amd ... policies/scoring.py), both on the HIP kernels.
Atlas + 5 finger charts (N = 1949), reference hyper-parameters (L=20, H=300), P = 10 000, E env elements, K candidates."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--env", type=int, default=3)
    p.add_argument("--candidates", type=int, default=50)
    p.add_argument("--points", type=int, default=10000)
    p.add_argument("--reps", type=int, default=3)
    a = p.parse_args()
    from a3vt_amd.synthetic import gt_cloud, make_args
    from a3vt_amd.pterotactyl.policies import scoring
    from a3vt_amd.pterotactyl.reconstruction.vision import model
    from a3vt_amd.pterotactyl.utility import utils
    dev = torch.device("cuda", 0)
    args = make_args(use_touch=True, finger=True, num_grasps=5, number_points=a.points)
    info, verts = utils.load_mesh_vision(args, "vision_charts")
    torch.manual_seed(0)
    net = model.Deformation(info, verts, args).to(dev).eval()
    E, K = a.env, a.candidates
    g = torch.Generator().manual_seed(1)
    img = torch.zeros(E, 1)
    gt = gt_cloud(E, a.points, 3).to(dev)
    charts_list = []
    from a3vt_amd.synthetic import surface_touch_charts
    for k in range(K):   # candidate k: five finger charts laid on the object's surface, a different grasp per candidate
        tc = surface_touch_charts(gt.cpu(), 5, g)
        charts_list.append(model.prepare_mesh({"img": img, "touch_charts": tc}, verts, args))

    def sequential():
        out = []
        for c in charts_list:
            with torch.no_grad():
                v, m = net(img, c)
                s = args.loss_coeff * utils.chamfer_distance(v, info["faces"], gt, num=a.points)
            out.append((s.cpu(), torch.cat((v, m), dim=-1).cpu()))   # the D2H copies compute_obs makes
        return out

    def batched():
        s, v, m = scoring.score_actions(net, img, charts_list, gt, info["faces"], a.points, args.loss_coeff)
        return s.cpu(), torch.cat((v, m), dim=-1).cpu()

    res = {}
    for name, fn in (("sequential", sequential), ("batched", batched)):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.reps):
            fn()
        torch.cuda.synchronize()
        res[name + "_ms"] = 1e3 * (time.perf_counter() - t0) / a.reps
    # one forward pass of E meshes: eager launches vs one HIP-graph replay (launch-bound at this size)
    static = {k: v.clone() for k, v in charts_list[0].items()}
    with torch.no_grad():
        net(img, static)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            net(img, static)
        torch.cuda.current_stream().wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            net(img, static)
        for name, fn in (("forward_eager_ms", lambda: net(img, static)), ("forward_graph_ms", graph.replay)):
            fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(20):
                fn()
            torch.cuda.synchronize()
            res[name] = 1e3 * (time.perf_counter() - t0) / 20
    # the batched call taken apart: the no-stash forward of K*E meshes against its two rooflines, then the loss
    charts = scoring.stack_charts(charts_list)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    with torch.no_grad():
        for rep in range(2):
            ev[0].record()
            v, _ = net.deform_with_maps(charts, [], [])
            ev[1].record()
            utils.chamfer_distance(v, info["faces"], gt, num=a.points)
            ev[2].record()
    torch.cuda.synchronize()
    fwd_ms, loss_ms = ev[0].elapsed_time(ev[1]), ev[1].elapsed_time(ev[2])
    n_vert, L, H, I = int(v.shape[1]), args.num_GCN_layers, args.hidden_GCN_size, 50
    dims = [I] + [H] * (L - 1) + [3]
    m = K * E * n_vert
    flop = 3 * sum(2.0 * m * a_ * b_ for a_, b_ in zip(dims[:-1], dims[1:]))
    nbytes = 3 * sum(4.0 * m * (a_ + b_) for a_, b_ in zip(dims[:-1], dims[1:]))
    res["batched_parts"] = {"forward_ms": fwd_ms, "chamfer_ms": loss_ms, "rows": m,
                            "forward_tflops": flop / (fwd_ms * 1e-3) / 1e12, "forward_frac_of_fp32_matrix_peak": flop / (fwd_ms * 1e-3) / 157.3e12,
                            "forward_hbm_GBps": nbytes / (fwd_ms * 1e-3) / 1e9, "forward_frac_of_8TBps": nbytes / (fwd_ms * 1e-3) / 8e12,
                            "how": "forward-only Deformation (3 stages x 20 layers, no activation stash) of K*E meshes; flops = 3 * sum 2 M d_i "
                                   "d_{i+1}, bytes = 3 * sum 4 M (d_i + d_{i+1}) (SURVEY 8d, forward share); then 3 surface draws + the exact search "
                                   "with the E ground-truth clouds shared by the K candidates"}
    res.update({"env": E, "candidates": K, "points": a.points, "n_vert": n_vert,
                "candidates_per_s_batched": 1e3 * E * K / res["batched_ms"],
                "speedup": res["sequential_ms"] / res["batched_ms"]})
    print(json.dumps(res))


if __name__ == "__main__":
    main()
