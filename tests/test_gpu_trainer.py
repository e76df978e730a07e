"""-m gpu: the trainer facade end to end on synthetic loader-format batches, the forward-only (policy scoring) path,
and checkpoint round trips in the reference's file layout."""
import os

import numpy as np
import pytest
import torch

from helpers import make_args, random_cloud

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
