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
    sequential loop (environment.py:174-180 -> compute_obs -> get_score) — bit for bit on the same surface samples."""
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
    for k in range(K):                                   # the reference's loop: one candidate per call
        with torch.no_grad():
            v, m = net(img.to(cuda) if use_img else img, charts_list[k])
            s = args.loss_coeff * utils.chamfer_distance(v, info["faces"], gt, num=P, samples=samples)
        assert torch.equal(m, m_all[k])
        if use_img:   # torch/MIOpen ops of the image branch (448-wide encoders, grid_sample) pick batch-size dependent
            assert rel_err(v_all[k], v) < 1e-5 and rel_err(score[k], s) < 1e-4   # GEMM tilings: rounding-level differences
        else:         # the HIP kernels are batch-position invariant: bit-identical
            assert torch.equal(v, v_all[k]) and torch.equal(s, score[k])
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
