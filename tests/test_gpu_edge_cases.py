"""-m gpu: degenerate and ragged inputs the reference meets in practice — empty touch slots (all-zero charts, mask 0),
"touched nothing" charts collapsed to a point (mask 1), zero-area meshes, duplicate points / exact ties, one-element
batches, row counts below one MFMA tile.  Each case against the CPU oracle on the same inputs."""
import numpy as np
import pytest
import torch

from helpers import assert_grad_close, make_args, oracle_adj, random_cloud, rel_err, template

pytestmark = pytest.mark.gpu


def test_tiny_graph_single_mesh(cuda):
    """5 vertices, batch 1: M = 5 rows is less than one 16-row tile; vertex count not a multiple of 8."""
    from a3vt_amd import mesh as amesh, ops
    from oracle import gcn as og
    faces = np.array([[0, 1, 2], [0, 2, 3], [0, 3, 4]], dtype=np.int64)
    verts = np.random.default_rng(0).standard_normal((5, 3)).astype(np.float32)
    args = make_args()
    adj_o, _ = oracle_adj(verts, faces, args)
    L, H = 3, 300
    st = og.init_state(50, H, L, seed=1)
    g = torch.Generator().manual_seed(0)
    feats = torch.randn(1, 5, 50, generator=g)
    gup = torch.randn(1, 5, 3, generator=g)
    st64 = {k: v.double().requires_grad_(True) for k, v in st.items() if k.startswith("mesh_deform_1")}
    f64 = feats.double().requires_grad_(True)
    out_o = og.gcn(f64, st64, "mesh_deform_1", (adj_o[0], adj_o[1], adj_o[2].double()), L, 0.33)
    (out_o * gup.double()).sum().backward()
    adj = ops.DeviceCSR(amesh.CSRAdjacency.from_pairs(*amesh.vision_pairs(faces, 5), 5), cuda)
    ws = [st[f"mesh_deform_1.layers.{i}.weight"].to(cuda).requires_grad_(True) for i in range(L)]
    bs = [st[f"mesh_deform_1.layers.{i}.bias"].to(cuda).requires_grad_(True) for i in range(L)]
    fd = torch.nn.functional.pad(feats, (0, 2)).to(cuda).requires_grad_(True)
    out = ops.gcn_stack(fd, adj, 50, H, 99, ws, bs)
    (out * gup.to(cuda)).sum().backward()
    assert rel_err(out, out_o) < 1e-4
    assert_grad_close(fd.grad[..., :50], f64.grad, "grad_feats")
    for i in range(L):
        assert_grad_close(ws[i].grad, st64[f"mesh_deform_1.layers.{i}.weight"].grad, f"dW{i}")
        assert_grad_close(bs[i].grad, st64[f"mesh_deform_1.layers.{i}.bias"].grad, f"db{i}")


def test_empty_and_untouched_chart_slots(cuda):
    """Touch slots as the environment fills them (policies/environment.py:305-315): mask 0 = empty slot (chart all
    zeros at the origin), mask 1 = the finger touched nothing (chart collapsed to the finger position), mask 2 = a real
    touch.  Degenerate charts have zero area: they must never be sampled, and the model output must match the oracle."""
    from a3vt_amd.pterotactyl.reconstruction.vision import model
    from a3vt_amd.pterotactyl.utility import utils
    from a3vt_amd import mesh as amesh
    from oracle import chamfer as och, gcn as og, mesh as omesh
    args = make_args(use_touch=True, finger=True, num_grasps=3, num_GCN_layers=3, hidden_GCN_size=64)
    info, verts = utils.load_mesh_vision(args, "vision_charts")
    torch.manual_seed(3)
    net = model.Deformation(info, verts, args).to(cuda)
    B = 2
    g = torch.Generator().manual_seed(1)
    tc = torch.zeros(B, 3, 25, 4)                                          # slot 0: empty (mask 0, zeros)
    tc[:, 1, :, :3] = (torch.rand(B, 1, 3, generator=g) - 0.5) * 0.3       # slot 1: collapsed to one point, mask 1
    tc[:, 1, :, 3] = 1
    tc[:, 2, :, :3] = (torch.rand(B, 25, 3, generator=g) - 0.5) * 0.3      # slot 2: a real touch
    tc[:, 2, :, 3] = 2
    batch = {"img": torch.zeros(B, 1), "touch_charts": tc}
    charts = model.prepare_mesh(batch, verts, args)
    out, mask = net(batch["img"], charts)
    # oracle on the same weights
    v, f = amesh.load_asset("vision_charts")
    sv, sf = amesh.load_asset("touch_chart")
    oinfo = omesh.adj_init(v, f, True, 3, True, sv, sf)
    st = {k: t.detach().cpu() for k, t in net.state_dict().items()}
    ch = og.prepare_mesh(tc, torch.from_numpy(v), B, True)
    out_o, mask_o = og.deformation_forward(st, {"adj": oracle_adj(v, f, args)[0]}, ch, True, 3, 0.33)
    assert rel_err(out, out_o) < 1e-4 and torch.equal(mask.cpu(), mask_o)
    assert torch.equal(out[:, 1824:].cpu(), ch["touch_charts"])            # touch vertices never move
    # sampling: zero-area faces (both degenerate slots: 2 x 32 faces) get probability exactly 0 and are never drawn
    faces = info["faces"]
    prob = och.face_probabilities(out.detach().cpu(), torch.from_numpy(oinfo["faces"]))
    assert prob.shape[1] == 2304 + 3 * 32 and torch.all(prob[:, 2304:2304 + 64] == 0) and torch.all(prob[:, 2304 + 64:] > 0)
    from a3vt_amd import ops
    pts, fi, _, _ = ops.sample_points(out.detach(), info["faces_i32"], 4000, 3, seed=5, offset=0, return_samples=True)
    fi = fi.cpu()
    assert not torch.any((fi >= 2304) & (fi < 2304 + 64))
    assert torch.isfinite(pts).all()
    cd = utils.chamfer_distance(out, faces, random_cloud(B, 500, 2).to(cuda), num=400)
    cd.sum().backward()
    assert torch.isfinite(cd).all() and all(torch.isfinite(p.grad).all() for p in net.parameters())


def test_zero_area_mesh_samples_uniformly(cuda):
    """utils.py:166-168: an all-zero-area mesh (every vertex at one point) turns the NaN probabilities into a uniform
    distribution; all samples land on that point and the face histogram is flat."""
    from a3vt_amd import ops
    verts, faces = template("ico2")
    v = torch.full((2, verts.shape[0], 3), 0.125)
    v[1] = torch.from_numpy(verts)                                         # second mesh is a normal one
    f = torch.from_numpy(faces).to(torch.int32).to(cuda)
    pts, fi, _, _ = ops.sample_points(v.to(cuda), f, 20000, 1, seed=11, offset=0, return_samples=True)
    assert (pts[0, 0].cpu() - 0.125).abs().max() < 1e-7                 # w0 p + w1 p + w2 p, weights sum to 1
    hist = torch.bincount(fi[0, 0].cpu().long(), minlength=faces.shape[0]).float()
    assert hist.min() > 0 and abs(hist.mean().item() - 20000 / faces.shape[0]) < 1e-3
    assert (hist - hist.mean()).abs().max() < 6 * hist.mean().sqrt()       # flat within Poisson noise
    assert torch.isfinite(pts).all()


@pytest.mark.parametrize("P,Q", [(1, 1), (1, 700), (3000, 17), (257, 2049)])
def test_chamfer_ragged_sizes_and_ties(cuda, P, Q):
    """P != Q down to single points; duplicated candidates: the lowest index among exact ties is reported (first
    occurrence, as the oracle's strict '<' scan)."""
    from a3vt_amd import ops
    from oracle import chamfer as och
    x = random_cloud(2, P, 5).reshape(1, 2, P, 3)
    y = random_cloud(2, Q, 6)
    if Q >= 4:
        y[:, Q // 2] = y[:, 1]                                            # duplicate: index 1 must win over Q//2
        y[:, Q - 1] = y[:, 1]
    if P >= 2:
        x[0, :, 0] = y[:, min(1, Q - 1)]                                   # a query sitting exactly on the duplicated point
    dxy, ixy, dyx, iyx, cd = ops.chamfer_nn(x.to(cuda), y.to(cuda))
    for b in range(2):
        dc, ic = och.nn_sqdist_c(x[0, b].numpy(), y[b].numpy())
        assert np.array_equal(ixy[0, b].cpu().numpy(), ic)
        assert np.allclose(dxy[0, b].cpu().numpy(), dc, rtol=1e-5, atol=1e-12)
        dc2, ic2 = och.nn_sqdist_c(y[b].numpy(), x[0, b].numpy())
        assert np.array_equal(iyx[0, b].cpu().numpy(), ic2)
    if P >= 2 and Q >= 4:
        assert int(ixy[0, 0, 0]) == 1 and float(dxy[0, 0, 0]) == 0.0
    cd_o = och.chamfer_pair(x[0].double(), y.double())
    assert rel_err(cd, cd_o) < 1e-5


def test_error_conventions(cuda):
    """C ABI: bad arguments return < 0 with a message in a3vt_last_error (surfaced as RuntimeError by the wrappers);
    nothing is launched, nothing blocks.  The facade refuses CPU tensors instead of falling back."""
    from a3vt_amd import lib, mesh as amesh, ops
    from a3vt_amd.pterotactyl.utility import utils
    L = lib.load()
    verts, faces = template("ico1")
    adj = ops.DeviceCSR(amesh.CSRAdjacency.from_pairs(*amesh.vision_pairs(faces, verts.shape[0]), verts.shape[0]), cuda)
    n = verts.shape[0]
    w = [torch.zeros(1, 50, 308, device=cuda), torch.zeros(1, 308, 3, device=cuda)]
    b = [torch.zeros(308, device=cuda), torch.zeros(3, device=cuda)]
    with pytest.raises(RuntimeError, match="hidden"):                       # hidden > 304 is not supported by the stack
        ops.gcn_stack(torch.zeros(1, n, 52, device=cuda), adj, 50, 308, 100, w, b)
    with pytest.raises(RuntimeError, match="vertices"):                     # adjacency / feature size mismatch
        ops.gcn_stack(torch.zeros(1, n + 1, 52, device=cuda), adj, 50, 300, 99, w, b)
    with pytest.raises(RuntimeError, match="GPU"):                          # no CPU fallback
        utils.chamfer_distance(torch.zeros(1, n, 3), torch.from_numpy(faces), torch.zeros(1, 10, 3), num=10)
    x = torch.zeros(1, 1, 4, 3, device=cuda)
    rc = L.a3vt_chamfer_fwd(lib.ptr(x), lib.ptr(x), 1, 1, 0, 4, None, None, None, None, None, None, None)
    assert rc < 0 and len(L.a3vt_last_error()) > 0
    with pytest.raises(RuntimeError, match="multiple of 4"):                # feature rows must be 16-byte granular
        ops.gcn_layer(torch.zeros(1, n, 50, device=cuda), adj, torch.zeros(1, 50, 16, device=cuda),
                      torch.zeros(16, device=cuda), 5, True)
