"""CPU, build container only: the oracle against the REAL reference imported from /root/reference under the shim
(oracle/ref_shim.py).  Skipped wherever /root/reference is absent (the GPU box)."""
from types import SimpleNamespace as NS

import pytest
import torch

from oracle import ref_shim

pytestmark = pytest.mark.skipif(not ref_shim.available(), reason="/root/reference not present")


@pytest.mark.parametrize("use_touch,finger,grasps", [(False, False, 1), (True, False, 2), (True, True, 3)])
def test_deformation_chamfer_fwd_bwd_live(use_touch, finger, grasps):
    from oracle import chamfer as och, gcn as og
    ref = ref_shim.load_reference()
    torch.manual_seed(0)
    a = NS(use_touch=use_touch, num_grasps=grasps, finger=finger, use_img=False, num_GCN_layers=4, hidden_GCN_size=40, cut=0.33)
    info, verts = ref.utils.load_mesh_vision(a, ref.objects_dir + "/vision_charts.obj")
    net = ref.model.Deformation(info, verts, a)
    B = 2
    shape = (B, grasps, 25, 4) if finger else (B, grasps, 4, 25, 4)
    tc = torch.cat((torch.randn(shape[:-1] + (3,)) * 0.1, torch.randint(0, 3, shape[:-1] + (1,)).float()), -1)
    batch = {"img": torch.zeros(B, 1), "touch_charts": tc}
    out_r, mask_r = net(batch["img"], ref.model.prepare_mesh(batch, verts, a))
    gt = torch.rand(B, 500, 3) * 0.3 - 0.15
    torch.manual_seed(5)
    loss_r = 9000 * ref.utils.chamfer_distance(out_r, info["faces"], gt, num=400).mean()
    loss_r.backward()
    st = {k: v.detach().clone().requires_grad_(True) for k, v in net.state_dict().items()}
    out_o, mask_o = og.deformation_forward(st, {"adj": info["adj"]}, og.prepare_mesh(tc, verts, B, use_touch), use_touch, 4, 0.33)
    torch.manual_seed(5)
    loss_o = 9000 * och.chamfer_distance(out_o, info["faces"], gt, num=400).mean()
    loss_o.backward()
    assert torch.equal(out_r, out_o) and torch.equal(mask_r, mask_o)
    assert loss_r.item() == loss_o.item()
    for k, p in net.named_parameters():
        err = (p.grad - st[k].grad).abs().max() / p.grad.abs().max().clamp_min(1e-30)
        assert err < 5e-6, k


def test_product_csr_equals_reference_dense():
    from a3vt_amd import mesh as amesh
    import numpy as np
    ref = ref_shim.load_reference()
    v, f = amesh.load_asset("vision_charts")
    sv, sf = amesh.load_asset("touch_chart")
    for grasps, finger in ((5, False), (5, True), (1, False)):
        a = NS(use_touch=True, num_grasps=grasps, finger=finger, use_img=False)
        info, _ = ref.utils.load_mesh_vision(a, ref.objects_dir + "/vision_charts.obj")
        r, c, n, faces = amesh.fused_pairs(v, f, sf, grasps, finger)
        A = amesh.CSRAdjacency.from_pairs(r, c, n)
        assert np.array_equal(A.to_dense(), info["adj"].numpy())
        assert np.array_equal(faces, info["faces"].numpy())
