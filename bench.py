#!/usr/bin/env python
"""bench.py — mesh-recon iterations/s (fwd + 3-draw Chamfer + bwd + Adam) on N MI355X GPUs of one node.

Contract (driver): ``python bench.py --gpus N --steps K --warmup W``; for N > 1 launched by torch.distributed.run,
one rank per GPU over RCCL.  Rank 0 prints ONE JSON line.

Workload at every N (weak scaling): BASELINE.json configs[1] per GPU — 2562-vertex icosphere template, 3-stage
GCN (20 layers x 300 hidden, cut 0.33 — the reference defaults, vision/train.py:375-386), bs = 64 per GPU,
10 000-point Chamfer x 3 draws, fp32, Adam(lr 3e-4).  Synthetic seeded inputs: ellipsoid-surface
ground-truth clouds, reference weight init.  ``value`` = (N * K iterations of bs 64) / wall time, inputs resident
in HBM, barrier + synchronize on both sides, max over ranks.

Extra objects on the JSON line:
* ``roofline``  — the dominant kernel (fp32-MFMA per-vertex product, hidden x hidden): algorithmic
  2*M*300*300 flop per launch / mean launch duration measured with HIP events on the launch stream during
  extra (untimed) profiled steps, against the 157.3 TFLOP/s fp32 matrix peak (MI355X_MICROARCH.md).
* ``cpu_baseline`` — the CPU oracle (a port of the reference path; kind "port") timed on this box's host cores
  on a bounded sample: bs = 4 of the same workload, one iteration (≈10 s), scaled to iterations of bs 64.
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=10)
    p.add_argument("--warmup", type=int, default=3)
    p.add_argument("--batch", type=int, default=64, help="meshes per GPU")
    p.add_argument("--points", type=int, default=10000)
    p.add_argument("--level", type=int, default=4, help="icosphere level (4 -> 2562 vertices)")
    p.add_argument("--layers", type=int, default=20)
    p.add_argument("--hidden", type=int, default=300)
    p.add_argument("--cloud", default="ellipsoid", choices=["ellipsoid", "cube"])
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--profile-steps", type=int, default=2)
    p.add_argument("--gemm-precision", default="fp32", choices=["fp32", "bf16"],
                   help="bf16 = BASELINE configs[3]/[4] operand mode of the per-vertex products; NOT the headline metric")
    return p.parse_args()


def cpu_baseline(level, layers, hidden, points, seed=0):
    """Oracle (CPU port of the reference path) on a bounded sample: bs=4, one fwd+bwd iteration (about 10 s)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import make_args, oracle_adj
    from a3vt_amd import mesh as amesh
    from a3vt_amd.synthetic import gt_cloud
    from oracle import chamfer as och, gcn as og
    bs = 4
    args = make_args(num_GCN_layers=layers, hidden_GCN_size=hidden)
    verts, faces = amesh.icosphere(level)
    adj_o, faces_o = oracle_adj(verts, faces, args)
    st = {k: v.requires_grad_(True) for k, v in og.init_state(50, hidden, layers, seed=seed).items()}
    ch = og.prepare_mesh(None, torch.from_numpy(verts), bs, False)
    gt = gt_cloud(bs, points, seed)
    torch.manual_seed(seed)
    t0 = time.perf_counter()
    out, _ = og.deformation_forward(st, {"adj": adj_o}, ch, False, layers, 0.33)
    cd = och.chamfer_distance(out, faces_o, gt, num=points, use_c=True)
    (9000.0 * cd.mean()).backward()
    dt = time.perf_counter() - t0
    return {"value": (bs / 64.0) / dt, "unit": "iters/s at bs=64", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"bs={bs} of the bs=64 workload, 1 fwd+bwd iteration (no optimizer), {dt:.1f} s of CPU work; "
                      f"CSR aggregation + plain-C brute-force NN, torch CPU fp32"}


def pmc_traffic():
    """HBM bytes per launch of the dominant kernel from the committed PMC collection (tools/collect_traffic.sh:
    separate --pmc passes; FETCH_SIZE doubled per MI355X_MICROARCH.md "HBM": it reports half of a wide coalesced
    stream on gfx950; WRITE_SIZE exact).  None when the summary is absent."""
    path = os.path.join(ROOT, "profiles", "r01_pmc_traffic_summary.json")
    try:
        with open(path) as f:
            d = json.load(f)
        k = next(v for name, v in d.items() if "rowgemm_kernel<19, 2" in name)
        return (2.0 * k["FETCH_SIZE_KiB_max"] + k["WRITE_SIZE_KiB_max"]) * 1024.0
    except Exception:
        return None


def main():
    a = parse()
    from a3vt_amd import distributed as adist, lib
    from a3vt_amd.pterotactyl.reconstruction.vision import model, train
    from a3vt_amd.synthetic import gt_cloud
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import make_args

    rank, world, local = adist.init_from_env("nccl")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    args = make_args(num_GCN_layers=a.layers, hidden_GCN_size=a.hidden, number_points=a.points, seed=0,
                     exp_type="bench", exp_id=f"rank{rank}", eval=False, epochs=1, patience=70, batch_size=a.batch,
                     gemm_precision=a.gemm_precision)
    os.chdir(os.environ.get("TMPDIR", "/tmp"))  # Engine writes config.json under ./experiments
    from a3vt_amd import mesh as amesh
    eng = train.Engine(args, loaders=((), ()))
    verts, faces = amesh.icosphere(a.level)
    # icosphere template instead of the packaged atlas (BASELINE.json configs[1])
    from a3vt_amd.pterotactyl.utility import utils
    vt, ft = torch.from_numpy(verts).to(dev), torch.from_numpy(faces).to(dev)
    eng.mesh_info, eng.initial_mesh = utils.adj_init(vt, ft, args), vt
    eng.n_vision_charts = vt.shape[0]
    torch.manual_seed(0)
    eng.encoder = model.Deformation(eng.mesh_info, vt, args).to(dev)
    adist.broadcast_parameters(eng.encoder)
    params = list(eng.encoder.parameters())
    eng.bucket = adist.FlatGradBucket(params)
    try:
        eng.optimizer = torch.optim.Adam(params, lr=args.lr, weight_decay=0, fused=True)
    except (RuntimeError, TypeError):
        eng.optimizer = torch.optim.Adam(params, lr=args.lr, weight_decay=0, foreach=True)

    nsteps = a.warmup + a.steps + a.profile_steps
    # inputs resident in HBM before the timed region; a few distinct clouds cycled
    clouds = [gt_cloud(a.batch, a.points, seed=1000 * rank + i, kind=a.cloud).to(dev) for i in range(min(nsteps, 4))]
    batch = {"img": torch.zeros(a.batch, 1)}
    img = batch["img"].to(dev)
    charts = model.prepare_mesh(batch, vt, args)
    torch.manual_seed(1234 + rank)

    def step(i):
        return eng.train_step(img, charts, clouds[i % len(clouds)])

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(a.warmup):
        loss = step(i)
    fence()
    t0 = time.perf_counter()
    for i in range(a.steps):
        loss = step(a.warmup + i)
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()
    final_loss = loss.item()

    # roofline leg: per-launch device time of the MFMA kernels, HIP events on the launch stream
    L = lib.load()
    roof = None
    if a.profile_steps > 0:
        L.a3vt_profile_enable(1)
        for i in range(a.profile_steps):
            step(a.warmup + a.steps + i)
        torch.cuda.synchronize()
        tot = (ctypes.c_double * 3)()
        cnt = (ctypes.c_int * 3)()
        lib.check(L.a3vt_profile_read(tot, cnt), "profile_read")
        L.a3vt_profile_enable(0)
        M = a.batch * vt.shape[0]
        flop = 2.0 * M * a.hidden * a.hidden
        # class 0 includes 3 first-layer launches (K=52) per step; class 1/2 are hidden x hidden except dW of layer 0
        n_fwd, n_dx, n_dw = cnt[0], cnt[1], cnt[2]
        per = {"fwd_ms": tot[0] / max(n_fwd, 1), "dx_ms": tot[1] / max(n_dx, 1), "dw_ms": tot[2] / max(n_dw, 1)}
        # dominant kernel = rowgemm (forward + dX launches share the kernel); dX launches are all hidden x hidden
        t_ms = tot[1] / max(n_dx, 1)
        achieved = flop / (t_ms * 1e-3) / 1e12
        peak = 157.3 if a.gemm_precision == "fp32" else 1250.0   # 16x16x16 bf16 form: half the 16x16x32 rate
        roof = {"bound": "mfma", "kernel": "rowgemm_kernel<19,EPI_DX_MASK> (fp32 MFMA 16x16x4, M x 300 x 300, dX = dZ W^T)"
                if a.gemm_precision == "fp32" else "rowgemm_kernel<19,EPI_DX_MASK,bf16> (v_mfma_f32_16x16x16_bf16)",
                "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
                "traffic": pmc_traffic(),
                "avg_launch_ms": t_ms, "launches": n_dx, "flop_per_launch": flop,
                "other_mfma_ms": per,
                "mfma_ms_per_step": (tot[0] + tot[1] + tot[2]) / a.profile_steps}

    default_cfg = (a.level, a.batch, a.points, a.layers, a.hidden, a.gemm_precision) == (4, 64, 10000, 20, 300, "fp32")
    out = {
        "metric": "mesh-recon iters/sec (fwd+bwd, 2562-vert GCN + 10k-pt Chamfer) at bs=64",
        "value": world * a.steps / dt, "unit": "iters/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": 1e3 * dt / a.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32" if a.gemm_precision == "fp32" else "f32 storage, bf16 GEMM operands (reduced precision, not the headline)",
        "data": "synthetic",
        "config": {"workload": f"icosphere-{a.level} template ({vt.shape[0]} verts, {ft.shape[0]} faces), 3-stage GCN "
                               f"{a.layers}x{a.hidden} cut 0.33, bs={a.batch}/GPU, {a.points}-pt Chamfer x3 draws, "
                               f"Adam, {a.cloud} clouds" + (" (BASELINE.json configs[1])" if default_cfg else " (custom sizes)"),
                   "global_batch": a.batch * world, "parallelism": f"dp{world}", "final_loss": final_loss},
    }
    if roof is not None:
        out["roofline"] = roof
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(a.level, a.layers, a.hidden, a.points)
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
