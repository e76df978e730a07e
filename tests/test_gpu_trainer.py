"""-m gpu: the trainer facade end to end on synthetic loader-format batches, the forward-only (policy scoring) path,
and checkpoint round trips in the reference's file layout."""
import os

import numpy as np
import pytest
import torch

from helpers import make_args, random_cloud, rel_err

pytestmark = pytest.mark.gpu


def _args(tmp, **kw):
    a = make_args(num_GCN_layers=3, hidden_GCN_size=64, number_points=500, exp_type="t", exp_id="run", eval=False,
                  epochs=2, patience=70, batch_size=4, log_interval=0, **kw)
    os.chdir(tmp)
    return a


def test_engine_trains_validates_and_resumes(cuda, tmp_path):
    from a3vt_amd.pterotactyl.reconstruction.vision import train
    from a3vt_amd.synthetic import SyntheticLoader
    args = _args(tmp_path, use_touch=True, num_grasps=2)
    loaders = (SyntheticLoader(args, 3, 4, seed=1), SyntheticLoader(args, 2, 4, seed=2))
    eng = train.Engine(args, loaders=loaders)
    best = eng()
    assert np.isfinite(best) and eng.epoch == 1
    ck = eng.checkpoint_dir
    assert all(os.path.exists(os.path.join(ck, f)) for f in ("model", "optim", "epoch.npy", "config.json"))
    sd = torch.load(os.path.join(ck, "model"), map_location="cpu")
    assert "mesh_deform_2.layers.2.bias" in sd and sd["mesh_deform_1.layers.0.weight"].shape == (1, 50, 64)
    # resume: a fresh Engine picks up the weights and the epoch counter (reference train.py:259-267)
    eng2 = train.Engine(args, loaders=loaders)
    eng2.setup()
    eng2.load()
    assert eng2.epoch == int(np.load(os.path.join(ck, "epoch.npy"))[0])
    for k, v in eng2.encoder.state_dict().items():
        assert torch.equal(v.cpu(), sd[k])
    # the loss went down over the two epochs on this fixed synthetic set (sanity of the whole backward path)
    assert eng.current_loss <= 1.5 * eng.best_loss


def test_forward_only_scoring_path_matches_training_forward(cuda):
    """policies/environment.py:221-257 calls deform + chamfer under torch.no_grad(): no activations are saved."""
    from a3vt_amd.pterotactyl.reconstruction.vision import model
    from a3vt_amd.pterotactyl.utility import utils
    args = make_args(use_touch=True, finger=True, num_grasps=5, num_GCN_layers=4, hidden_GCN_size=300)
    info, verts = utils.load_mesh_vision(args, "vision_charts")
    torch.manual_seed(0)
    net = model.Deformation(info, verts, args).to(cuda)
    B = 3
    g = torch.Generator().manual_seed(4)
    tc = torch.zeros(B, 5, 25, 4)
    tc[..., :3] = (torch.rand(B, 5, 25, 3, generator=g) - 0.5) * 0.3
    tc[..., 3] = torch.randint(0, 3, (B, 5, 1), generator=g).float()
    batch = {"img": torch.zeros(B, 1), "touch_charts": tc}
    charts = model.prepare_mesh(batch, verts, args)
    out_train, mask = net(batch["img"], charts)
    with torch.no_grad():
        out_eval, mask2 = net(batch["img"], charts)
        gt = random_cloud(B, 2000, 5).to(cuda)
        score = args.loss_coeff * utils.chamfer_distance(out_eval, info["faces"], gt, num=1000)
    assert torch.equal(out_train.detach(), out_eval) and torch.equal(mask, mask2)
    assert out_eval.shape == (B, 1824 + 125, 3) and score.shape == (B,) and torch.isfinite(score).all()
    assert torch.equal(out_eval[:, 1824:], charts["touch_charts"])          # touch vertices never move (model.py:250)
    obs_mesh = torch.cat((out_eval, mask2), dim=-1).cpu()                     # the env's observation tensor
    assert obs_mesh.shape == (B, 1949, 4)
    # dense adjacency is still available to graph policies (policies/DDQN/model.py:68)
    dense = info["adj"]
    assert dense.shape == (1949, 1949) and torch.allclose(dense.sum(1), torch.ones(1949, device=cuda), atol=1e-5)


@pytest.mark.parametrize("use_img", [False, True])
def test_batched_scoring(cuda, use_img):
    """K candidate touches x E environment elements in one batch give the scores of the reference's
    sequential loop (environment.py:174-180 -> compute_obs -> get_score): against the ORACLE evaluating that loop one
    candidate at a time on the same injected surface samples (1e-4 on positions and scores; image-free model — the
    image model's oracle comparison is test_g8 / test_config3), and bit for bit against the sequential HIP loop."""
    from a3vt_amd.pterotactyl.policies import scoring
    from a3vt_amd.pterotactyl.reconstruction.vision import model
    from a3vt_amd.pterotactyl.utility import utils
    kw = dict(CNN_ker_size=5, num_CNN_blocks=6, layers_per_block=3) if use_img else {}
    args = make_args(use_touch=True, use_img=use_img, finger=True, num_grasps=5, num_GCN_layers=3, hidden_GCN_size=300,
                     number_points=700, **kw)
    info, verts = utils.load_mesh_vision(args, "vision_charts")
    torch.manual_seed(0)
    net = model.Deformation(info, verts, args).to(cuda).eval()
    E, K, P = 2, 5, args.number_points
    g = torch.Generator().manual_seed(8)
    img = torch.rand(E, 3, 256, 256, generator=g) if use_img else torch.zeros(E, 1)
    gt = random_cloud(E, 900, 2).to(cuda)
    charts_list = []
    for k in range(K):                                   # candidate k: a different 5th touch on top of 4 shared ones
        tc = torch.zeros(E, 5, 25, 4)
        gk = torch.Generator().manual_seed(100)
        tc[:, :4, :, :3] = (torch.rand(E, 4, 25, 3, generator=gk) - 0.5) * 0.3
        tc[:, :4, :, 3] = 2
        gk = torch.Generator().manual_seed(200 + k)
        tc[:, 4, :, :3] = (torch.rand(E, 25, 3, generator=gk) - 0.5) * 0.3
        tc[:, 4, :, 3] = 2 if k % 2 == 0 else 1
        charts_list.append(model.prepare_mesh({"img": img, "touch_charts": tc}, verts, args))
    F = info["faces"].shape[0]
    samples = (torch.randint(0, F, (3, E, P), generator=g).to(torch.int32).to(cuda),
               torch.rand(3, E, P, generator=g).to(cuda), torch.rand(3, E, P, generator=g).to(cuda))
    score, v_all, m_all = scoring.score_actions(net, img, charts_list, gt, info["faces"], P, args.loss_coeff, samples=samples)
    assert score.shape == (K, E) and v_all.shape == (K, E, 1949, 3) and m_all.shape == (K, E, 1949, 1)
    if not use_img:   # the reference's loop on the CPU oracle: compute_obs (environment.py:221-249) + get_score (:252-257)
        from a3vt_amd import mesh as amesh
        from helpers import oracle_adj
        from oracle import chamfer as och, gcn as og
        v_np, f_np = amesh.load_asset("vision_charts")
        adj_o, faces_o = oracle_adj(v_np, f_np, args)
        st = {k_: t.detach().cpu() for k_, t in net.state_dict().items()}
        smp = [(samples[0][r].cpu().long(), samples[1][r].cpu(), samples[2][r].cpu()) for r in range(3)]
        for k in range(K):
            ch = {k_: t.cpu() for k_, t in charts_list[k].items()}
            with torch.no_grad():
                v_o, m_o = og.deformation_forward(st, {"adj": adj_o}, ch, True, 3, 0.33)
                s_o = args.loss_coeff * och.chamfer_distance(v_o, faces_o, gt.cpu(), num=P, samples=smp, use_c=True)
            assert rel_err(v_all[k], v_o) < 1e-4 and torch.equal(m_all[k].cpu(), m_o)
            assert rel_err(score[k], s_o) < 1e-4, (k, score[k], s_o)
    for k in range(K):                                   # the reference's loop: one candidate per call
        with torch.no_grad():
            v, m = net(img.to(cuda) if use_img else img, charts_list[k])
            s = args.loss_coeff * utils.chamfer_distance(v, info["faces"], gt, num=P, samples=samples)
        assert torch.equal(m, m_all[k])
        if use_img:   # torch/MIOpen ops of the image branch (448-wide encoders, grid_sample) pick batch-size dependent
            assert rel_err(v_all[k], v) < 1e-5 and rel_err(score[k], s) < 1e-4   # GEMM tilings: rounding-level differences
        else:
            # round 6: the batched call (>= 12 288 rows) aggregates through the P + bipartite split (gcn_csrqs.hip), the
            # few-row calls of the loop walk the full CSR — the same sums in another association
            assert rel_err(v_all[k], v) < 1e-5 and rel_err(score[k], s) < 1e-4
    if not use_img:   # on ONE kernel family the HIP path is batch-position invariant: bit-identical to the loop
        info["csr"].use_split = False
        try:
            with torch.no_grad():
                score_g, v_g, _ = scoring.score_actions(net, img, charts_list, gt, info["faces"], P, args.loss_coeff, samples=samples)
                for k in range(K):
                    v, m = net(img, charts_list[k])
                    s = args.loss_coeff * utils.chamfer_distance(v, info["faces"], gt, num=P, samples=samples)
                    assert torch.equal(v, v_g[k]) and torch.equal(s, score_g[k])
        finally:
            info["csr"].use_split = True
        assert rel_err(v_g, v_all) < 1e-5 and rel_err(score_g, score) < 1e-4
    taken = torch.zeros(E, K)
    taken[0, int(score[:, 0].argmin())] = 1               # best candidate of element 0 already performed
    best = scoring.best_actions(score, taken)
    assert best[0] != score[:, 0].argmin() and best[1] == score[:, 1].argmin()
    # production sampling (Philox draws) also runs batched
    s2, _, _ = scoring.score_actions(net, img, charts_list, gt, info["faces"], P, args.loss_coeff)
    assert s2.shape == (K, E) and torch.isfinite(s2).all()


def test_engine_on_disk_dataset_and_prefetcher(cuda, tmp_path):
    """the trainer on the reference's on-disk dataset layout (written here in miniature), batches uploaded
    one step ahead by the DevicePrefetcher; the prefetched tensors equal the loader's."""
    from test_host_logic import _write_dataset
    from a3vt_amd.pterotactyl.reconstruction.vision import train
    from a3vt_amd.pterotactyl.utility import data_loaders
    from a3vt_amd.synthetic import SyntheticLoader
    root = os.path.join(str(tmp_path), "data")
    _write_dataset(root, list(range(10)), np.random.default_rng(1))
    args = _args(tmp_path, use_touch=True, num_grasps=2, data_root=root, num_workers=0, limit_data=False, val_grasps=-1)
    args.epochs = 1
    eng = train.Engine(args)
    best = eng()
    assert np.isfinite(best) and eng.epoch == 0 and os.path.exists(os.path.join(eng.checkpoint_dir, "model"))
    # prefetcher: same batches, on the device, in order
    loader = SyntheticLoader(args, 5, 3, seed=4)
    got = list(data_loaders.DevicePrefetcher(loader, cuda))
    assert len(got) == 5
    for a, b in zip(loader, got):
        assert b["gt_points"].is_cuda and torch.equal(a["gt_points"], b["gt_points"].cpu())
        assert torch.equal(a["touch_charts"], b["touch_charts"].cpu()) and a["names"] == b["names"]


def test_rccl_backend_single_rank(cuda):
    """The multi-GPU path's transport on this image: torch.distributed 'nccl' (= RCCL) initialises on the GPU box and
    reduces / broadcasts the flat gradient bucket.  (N > 1 ranks are covered on CPU by tests/test_distributed_gloo.py and
    run by the driver's scaling bench.)"""
    import torch.distributed as dist
    from a3vt_amd import distributed as adist
    if dist.is_initialized():
        pytest.skip("a process group is already up in this process")
    dist.init_process_group(backend="nccl", init_method="tcp://127.0.0.1:29533", rank=0, world_size=1)
    try:
        net = torch.nn.Linear(64, 32).to(cuda)
        bucket = adist.FlatGradBucket(list(net.parameters()))
        net(torch.randn(8, 64, device=cuda)).square().mean().backward()
        want = bucket.flat.clone()
        dist.all_reduce(bucket.flat, op=dist.ReduceOp.SUM)      # one rank: the sum is the value itself
        assert torch.equal(bucket.flat, want) and net.weight.grad.data_ptr() == bucket.flat.data_ptr()
        flat = torch.arange(10, device=cuda, dtype=torch.float32)
        dist.broadcast(flat, 0)
        dist.barrier()
        torch.cuda.synchronize()
        assert flat[9].item() == 9.0
    finally:
        dist.destroy_process_group()


def test_forward_capturable_in_hip_graph(cuda):
    """the library calls neither allocate nor synchronise, so a forward-only pass (policy scoring at
    small batch is launch-bound: ~400 launches for 3 meshes) can be captured once in a HIP graph and replayed."""
    from a3vt_amd.pterotactyl.reconstruction.vision import model
    from a3vt_amd.pterotactyl.utility import utils
    args = make_args(use_touch=True, finger=True, num_grasps=5, num_GCN_layers=6, hidden_GCN_size=300)
    info, verts = utils.load_mesh_vision(args, "vision_charts")
    torch.manual_seed(0)
    net = model.Deformation(info, verts, args).to(cuda).eval()
    B = 3
    g = torch.Generator().manual_seed(4)
    tc = torch.zeros(B, 5, 25, 4)
    tc[..., :3] = (torch.rand(B, 5, 25, 3, generator=g) - 0.5) * 0.3
    tc[..., 3] = 2
    img = torch.zeros(B, 1, device=cuda)
    charts = model.prepare_mesh({"img": img, "touch_charts": tc}, verts, args)
    static = {k: v.clone() for k, v in charts.items()}
    with torch.no_grad():
        eager, _ = net(img, static)                     # also the warm-up: one-time attribute / symbol look-ups
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            net(img, static)
        torch.cuda.current_stream().wait_stream(s)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            out, _ = net(img, static)
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, eager)
        # new inputs through the static buffers
        static["touch_charts"].copy_(static["touch_charts"] * 0.5)
        graph.replay()
        ref, _ = net(img, {k: v.clone() for k, v in static.items()})
        assert torch.equal(out, ref)


def test_multi_step_training_tracks_the_oracle(cuda):
    """Six Adam steps of the touch model (L=3, H=64) on the HIP path vs the same six steps of the CPU oracle from the same
    weights, injected surface samples per step: the loss trajectories stay together (drift < 1e-3 relative) and the
    parameters end up at the same place — fwd, Chamfer, every backward kernel and the optimiser in one loop."""
    from a3vt_amd import mesh as amesh
    from a3vt_amd.pterotactyl.reconstruction.vision import model
    from a3vt_amd.pterotactyl.utility import utils
    from helpers import oracle_adj
    from oracle import chamfer as och, gcn as og
    args = make_args(use_touch=True, finger=False, num_grasps=1, num_GCN_layers=3, hidden_GCN_size=64)
    info, verts = utils.load_mesh_vision(args, "vision_charts")
    torch.manual_seed(5)
    net = model.Deformation(info, verts, args).to(cuda)
    opt = torch.optim.Adam(net.parameters(), lr=3e-4, weight_decay=0)
    B, P, steps = 2, 600, 6
    g = torch.Generator().manual_seed(3)
    tc = torch.zeros(B, 1, 4, 25, 4)
    tc[..., :3] = (torch.rand(B, 1, 4, 25, 3, generator=g) - 0.5) * 0.3
    tc[..., 3] = 2
    gt = random_cloud(B, 800, 7)
    v, f = amesh.load_asset("vision_charts")
    adj_o, faces_o = oracle_adj(v, f, args)
    st = {k: t.detach().cpu().clone().requires_grad_(True) for k, t in net.state_dict().items()}
    opt_o = torch.optim.Adam(list(st.values()), lr=3e-4, weight_decay=0)
    ch_o = og.prepare_mesh(tc, torch.from_numpy(v), B, True)
    F = faces_o.shape[0]
    img = torch.zeros(B, 1)
    charts = model.prepare_mesh({"img": img, "touch_charts": tc}, verts, args)
    hip, ora = [], []
    for it in range(steps):
        fi = torch.randint(0, F, (3, B, P), generator=g)
        u, w = torch.rand(3, B, P, generator=g), torch.rand(3, B, P, generator=g)
        opt.zero_grad()
        out = net(img, charts)[0]
        loss = 9000.0 * utils.chamfer_distance(out, info["faces"], gt.to(cuda), num=P,
                                               samples=(fi.to(torch.int32).to(cuda), u.to(cuda), w.to(cuda))).mean()
        loss.backward()
        opt.step()
        hip.append(loss.item())
        opt_o.zero_grad()
        out_o, _ = og.deformation_forward(st, {"adj": adj_o}, ch_o, True, 3, 0.33)
        loss_o = 9000.0 * och.chamfer_distance(out_o, faces_o, gt, num=P, samples=[(fi[r], u[r], w[r]) for r in range(3)]).mean()
        loss_o.backward()
        opt_o.step()
        ora.append(loss_o.item())
    assert hip[-1] < hip[0]                                                   # it trains
    assert max(abs(a - b) / abs(b) for a, b in zip(hip, ora)) < 1e-3, (hip, ora)
    for k, p in net.state_dict().items():
        assert rel_err(p, st[k]) < 2e-3, k


def test_engine_train_step_reproduces_reference_step(cuda, tmp_path):
    """``Engine.train_step`` itself — flat gradient bucket + fused Adam, the path ``bench.py`` and ``Engine.train`` run —
    against fixture g7 (the REAL reference's loss before / after one Adam step and a weight sample, BASELINE configs[0]
    sizes: atlas, bs 2, 10 000-point Chamfer, 3 stages), on the reference's own surface draws."""
    from golden_util import g7_cloud, load
    from a3vt_amd.pterotactyl.reconstruction.vision import model, train
    from oracle import chamfer as och
    z = load("g7_train_step.npz")
    args = make_args(exp_type="t", exp_id="g7", eval=False, epochs=1, patience=70, batch_size=2, log_interval=0,
                     number_points=10000)
    os.chdir(tmp_path)
    eng = train.Engine(args, loaders=((), ()))
    torch.manual_seed(0)                                        # Engine.__init__ seeds with args.seed = 0 as well
    eng.setup()
    assert eng.bucket is not None and type(eng.optimizer).__name__ == "Adam"
    gt = g7_cloud().to(cuda)
    faces_cpu = eng.mesh_info["faces"].cpu()
    charts = model.prepare_mesh({"img": torch.zeros(2, 1)}, eng.initial_mesh, args)
    losses = []
    for it in range(2):
        with torch.no_grad():
            out = eng.encoder(torch.zeros(2, 1), charts)[0]
        torch.manual_seed(1000 + it)                            # the reference's draws for this evaluation (make_golden.g7)
        draws = [och.draw_samples(och.face_probabilities(out.cpu(), faces_cpu), 10000) for _ in range(3)]
        samples = (torch.stack([d[0] for d in draws]).to(torch.int32).to(cuda), torch.stack([d[1] for d in draws]).to(cuda),
                   torch.stack([d[2] for d in draws]).to(cuda))
        if it == 0:
            losses.append(eng.train_step(torch.zeros(2, 1), charts, gt, samples=samples).item())
            assert all(p.grad is not None and p.grad.data_ptr() == v.data_ptr() for p, v in zip(eng.bucket.params, eng.bucket.views))
        else:
            from a3vt_amd.pterotactyl.utility import utils
            with torch.no_grad():
                cd = utils.chamfer_distance(out, eng.mesh_info["faces"], gt, num=10000, samples=samples)
            losses.append(9000.0 * cd.mean().item())
    assert abs(losses[0] - float(z["loss_before_s3"])) < 1e-4 * abs(losses[0])
    assert abs(losses[1] - float(z["loss_after_s3"])) < 1e-3 * abs(losses[1])
    w = eng.encoder.mesh_deform_1.layers[19].weight.detach().cpu().numpy()[0, :16]
    np.testing.assert_allclose(w, z["w_after_sample_s3"], rtol=0, atol=5e-6)


def test_bench_under_torchrun_single_rank(cuda):
    """The driver's multi-GPU launch line at N = 1: ``python -m torch.distributed.run ... bench.py --gpus 1`` must come up
    (env rendezvous on 127.0.0.1, RCCL-capable init path, Engine + flat bucket) and print one JSON line."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", "29547", os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1",
           "--batch", "8", "--layers", "3", "--points", "1000", "--no-cpu-baseline", "--no-traffic", "--profile-steps", "1"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    line = [l for l in res.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["value"] > 0 and d["config"]["parallelism"] == "dp1"
    assert d["roofline"]["traffic"] is None and "step_ms" in d and "cpu_baseline" not in d
    assert d["metric"].startswith("mesh-recon iters/sec (fwd+bwd, 2562-vert GCN + 1k-pt Chamfer) at bs=8")
    keys = set(d)
    # the launcher-less command line (what a driver that does not use torch.distributed.run would type): bench.py starts the
    # same launch line itself as a child process, at N = 1 through --launcher spawn, and the JSON line has the same keys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    plain = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--launcher", "spawn"] + cmd[cmd.index("--steps"):],
                           capture_output=True, text=True, timeout=600, env=env)
    assert plain.returncode == 0, plain.stderr[-2000:]
    assert "torch.distributed.run" in plain.stderr
    d2 = json.loads([l for l in plain.stdout.splitlines() if l.startswith("{")][-1])
    assert set(d2) == keys and d2["n_gpus"] == 1 and d2["value"] > 0 and d2["rccl"]["rank_ms_per_step"]["n"] == 1
    # more GPUs than the node has: refused by the parent before anything is launched (no hang, no half-started group)
    ndev = torch.cuda.device_count()
    bad = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(ndev + 1), "--steps", "1"], capture_output=True,
                         text=True, timeout=300, env=env)
    assert bad.returncode != 0 and f"{ndev} device" in bad.stderr and "visible" in bad.stderr
    # a --gpus that contradicts an existing WORLD_SIZE is refused instead of silently measuring something else
    bad = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1"], capture_output=True,
                         text=True, timeout=300, env=dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"))
    assert bad.returncode != 0 and "WORLD_SIZE" in (bad.stderr + bad.stdout)


def test_gradients_written_into_the_bucket_equal_the_autograd_path(cuda, tmp_path):
    """``Engine.train_step`` lets the library write the GCN gradients straight into the flat all-reduce bucket
    (``ops.GradSink``; the second use of the shared ``mesh_deform_2`` accumulates inside ``a3vt_gcn_stack_bwd_acc``) instead
    of returning tensors for autograd to assign / add and the bucket to copy.  Same bits as the autograd path
    (``FlatGradBucket(..., sinks=False)``: the sum of a step is formed first and added once, as autograd's add does), the
    early chunk still triggers from inside the backward pass, and a second step starts clean."""
    from a3vt_amd import distributed as adist, ops
    from a3vt_amd.pterotactyl.reconstruction.vision import model, train
    args = make_args(exp_type="t", exp_id="sink", eval=False, epochs=1, patience=70, batch_size=3, log_interval=0,
                     number_points=1500, num_GCN_layers=4, hidden_GCN_size=64)
    os.chdir(tmp_path)
    eng = train.Engine(args, loaders=((), ()))
    eng.setup()
    charts = model.prepare_mesh({"img": torch.zeros(3, 1)}, eng.initial_mesh, args)
    gt = random_cloud(3, 1500, 4).to(cuda)
    g = torch.Generator().manual_seed(5)
    nf = eng.mesh_info["faces"].shape[0]
    samples = (torch.randint(0, nf, (3, 3, 1500), generator=g).to(torch.int32).to(cuda),
               torch.rand(3, 3, 1500, generator=g).to(cuda), torch.rand(3, 3, 1500, generator=g).to(cuda))
    params = list(eng.encoder.parameters())

    def grads_of(bucket):
        bucket.zero()
        verts = eng.encoder(torch.zeros(3, 1), charts)[0]
        from a3vt_amd.pterotactyl.utility import utils
        loss = 9000.0 * utils.chamfer_distance(verts, eng.mesh_info["faces_i32"], gt, num=1500, samples=samples).mean()
        loss.backward()
        early_started = bucket._early_done
        bucket.all_reduce_mean()
        return bucket.flat.clone(), early_started

    assert eng.bucket._sinks and all(ops.grad_sink_of(p) is not None for p in eng.bucket.params)
    stash0 = ops.STATS["stack_stash_calls"]
    flat_sink, early = grads_of(eng.bucket)
    assert ops.STATS["stack_stash_calls"] - stash0 == 3
    assert early                                                  # mesh_deform_2's chunk was complete inside the backward
    w = eng.encoder.mesh_deform_2.layers[1].weight
    assert ops.grad_sink_of(w).written and ops.grad_sink_of(w).expect == 0
    assert w.grad.data_ptr() == ops.grad_sink_of(w).view.data_ptr()
    flat_sink2, _ = grads_of(eng.bucket)                          # second step: flags were reset, nothing left over
    assert torch.equal(flat_sink, flat_sink2)
    # a step that skips all_reduce_mean() would leave .grad == None on the sink parameters (the optimiser would skip them):
    # the next zero() refuses to go on
    eng.bucket.zero()
    verts = eng.encoder(torch.zeros(3, 1), charts)[0]
    verts.sum().backward()
    w1 = eng.encoder.mesh_deform_1.layers[1].weight               # (not in the early chunk, which re-homes itself)
    assert w1.grad is None and ops.grad_sink_of(w1).written
    with pytest.raises(RuntimeError, match="all_reduce_mean"):
        eng.bucket.zero()
    eng.bucket.gather()
    assert w1.grad.data_ptr() == ops.grad_sink_of(w1).view.data_ptr()
    # the autograd path on the same parameters, same order in the buffer
    early = list(eng.encoder.mesh_deform_2.parameters())
    eng.bucket.close()                                            # hooks and sinks gone (a replaced bucket must let go)
    assert all(ops.grad_sink_of(p) is None for p in params)
    plain = adist.FlatGradBucket(params, early=early, sinks=False)
    assert [id(p) for p in plain.params] == [id(p) for p in eng.bucket.params] and not plain._sinks
    flat_plain, early_plain = grads_of(plain)
    assert early_plain
    assert torch.equal(flat_sink, flat_plain)


@pytest.mark.parametrize("precision", ["fp32", "bf16s"])
def test_engine_steps_the_image_model(cuda, tmp_path, precision):
    """``Engine.train_step`` on the image + touch model (configs[3] shape at a small batch): the bucket protocol
    (zero -> backward -> all_reduce_mean -> fused Adam) with the convolution stack in either branch — in the bf16 one the
    parameters keep their NCHW strides (the flat bucket's views and Adam's state share one layout) while the convolutions
    run on channels-last copies.  Every used parameter gets a finite gradient and moves."""
    from a3vt_amd.pterotactyl.reconstruction.vision import model, train
    args = make_args(exp_type="t", exp_id="img" + precision, eval=False, epochs=1, patience=70, batch_size=2, log_interval=0,
                     number_points=800, num_GCN_layers=3, hidden_GCN_size=64, use_img=True, use_touch=True, finger=False,
                     num_grasps=1, CNN_ker_size=5, num_CNN_blocks=6, layers_per_block=3, gemm_precision=precision)
    os.chdir(tmp_path)
    eng = train.Engine(args, loaders=((), ()))
    eng.setup()
    g = torch.Generator().manual_seed(2)
    tc = torch.zeros(2, 1, 4, 25, 4)
    tc[..., :3] = (torch.rand(2, 1, 4, 25, 3, generator=g) - 0.5) * 0.3
    tc[..., 3] = 2
    img = torch.rand(2, 3, 256, 256, generator=g).to(cuda)
    charts = model.prepare_mesh({"img": img, "touch_charts": tc}, eng.initial_mesh, args)
    gt = random_cloud(2, 800, 6).to(cuda)
    before = {n: p.detach().clone() for n, p in eng.encoder.named_parameters()}
    losses = [float(eng.train_step(img, charts, gt)) for _ in range(3)]
    assert all(np.isfinite(losses))
    moved = 0
    for n, p in eng.encoder.named_parameters():
        assert p.is_contiguous(), n                       # no parameter was re-laid out behind the bucket's back
        assert p.grad is not None and p.grad.data_ptr() == eng.bucket.views[[id(q) for q in eng.bucket.params].index(id(p))].data_ptr()
        assert torch.isfinite(p.grad).all(), n
        moved += int(not torch.equal(p.detach(), before[n]))
    # the deepest pyramid blocks see maps smaller than their kernel and are skipped (reference model.py:147-164): 5 blocks x
    # 4 tensors in each of the two image encoders stay where they were
    assert moved == len(before) - 40
    for name in ("mesh_deform_1.layers.0.weight", "mesh_deform_2.layers.1.bias", "img_encoder_global.layers.0.0.weight",
                 "img_encoder_local.layers.5.2.bias"):
        assert not torch.equal(dict(eng.encoder.named_parameters())[name].detach(), before[name]), name


@pytest.mark.parametrize("image_model", [False, True])
def test_collective_path_on_one_gpu_is_bit_identical(cuda, tmp_path, image_model):
    """the gradient exchange of vision/train.py:120-157 as it runs on N ranks — ``div_`` -> asynchronous RCCL
    ``all_reduce`` of the early chunk from inside the backward pass -> all-reduce of the rest -> ``wait`` -> fused Adam — on a
    process group of ONE rank (``FlatGradBucket(force_collectives=True)``): three ``Engine.train_step``s with and without the
    collectives give bit-identical weights, on the 20 x 300 model and on the image model (189 MB bucket; its early chunk holds
    parameters that never get a gradient — ADVICE r03: the countdown learns that in step 1 and overlaps from step 2)."""
    import torch.distributed as dist
    from a3vt_amd.pterotactyl.reconstruction.vision import model, train
    if dist.is_initialized():
        pytest.skip("a process group is already up in this process")
    os.chdir(tmp_path)
    if image_model:
        kw = dict(number_points=800, num_GCN_layers=3, hidden_GCN_size=64, use_img=True, use_touch=True, finger=False,
                  num_grasps=1, CNN_ker_size=5, num_CNN_blocks=6, layers_per_block=3, batch_size=2)
    else:
        kw = dict(number_points=1000, batch_size=2)
    g = torch.Generator().manual_seed(2)
    B = 2
    tc = torch.zeros(B, 1, 4, 25, 4)
    tc[..., :3] = (torch.rand(B, 1, 4, 25, 3, generator=g) - 0.5) * 0.3
    tc[..., 3] = 2
    img = torch.rand(B, 3, 256, 256, generator=g).to(cuda) if image_model else torch.zeros(B, 1)
    gt = random_cloud(B, kw["number_points"], 6).to(cuda)

    def three_steps(force, tag=""):
        args = make_args(exp_type="t", exp_id=f"coll{int(force)}{int(image_model)}{tag}", eval=False, epochs=1, patience=70,
                         log_interval=0, force_collectives=force, **kw)
        torch.manual_seed(0)
        eng = train.Engine(args, loaders=((), ()))
        eng.setup()
        batch = {"img": img, "touch_charts": tc} if image_model else {"img": img}
        charts = model.prepare_mesh(batch, eng.initial_mesh, args)
        nf = eng.mesh_info["faces"].shape[0]
        gs = torch.Generator().manual_seed(9)
        P = kw["number_points"]
        early_flags = []
        for _ in range(3):
            samples = (torch.randint(0, nf, (3, B, P), generator=gs).to(torch.int32).to(cuda),
                       torch.rand(3, B, P, generator=gs).to(cuda), torch.rand(3, B, P, generator=gs).to(cuda))
            eng.train_step(img.to(cuda) if image_model else img, charts, gt, samples=samples)
            early_flags.append(eng.bucket._early_done)
        torch.cuda.synchronize()
        flat = torch.cat([p.detach().reshape(-1) for p in eng.encoder.parameters()]).clone()
        started = eng.bucket.early_started_in_backward
        eng.bucket.close()
        return flat, early_flags, started, eng.bucket.early_numel * 4, eng.bucket.flat.numel() * 4

    plain, flags0, started0, _, _ = three_steps(False)
    plain_again = three_steps(False, "b")[0]
    dist.init_process_group(backend="nccl", init_method="tcp://127.0.0.1:29541", rank=0, world_size=1)
    try:
        forced, flags1, started1, early_bytes, total_bytes = three_steps(True)
    finally:
        dist.destroy_process_group()
    if torch.equal(plain, plain_again):
        assert torch.equal(plain, forced)
    else:
        # the image encoders' convolutions are MIOpen's, whose algorithm choice may differ between two runs in one process
        # (DESIGN §2 "Determinism"): then two runs WITHOUT collectives already differ, and the forced run must be as close
        assert image_model
        spread = (plain - plain_again).abs().max().item()
        assert (plain - forced).abs().max().item() <= 4.0 * spread + 1e-7, (spread, (plain - forced).abs().max().item())
    assert early_bytes > 0 and total_bytes > early_bytes
    # the early chunk's reduce was launched from inside the backward pass: every step on the image-free model; from the
    # second step on the image model (20 tensors of img_encoder_local never receive a gradient)
    assert flags1 == ([False, True, True] if image_model else [True, True, True]), flags1
    assert started1 == (2 if image_model else 3) and started0 == started1
    if image_model:
        assert total_bytes > 150e6
