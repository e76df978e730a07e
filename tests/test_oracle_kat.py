"""CPU: analytic known-answer tests for the pieces whose reference arithmetic lives in un-vendored PyTorch3D
 — these anchor the oracle where no golden vector can."""
import numpy as np
import torch

from oracle import chamfer as och, gcn as og, mesh as omesh


def test_chamfer_self_is_zero_and_small_shift_is_2t2():
    g = torch.Generator().manual_seed(0)
    x = torch.rand(1, 400, 3, generator=g, dtype=torch.float64)
    assert och.chamfer_pair(x, x).item() == 0.0
    t = torch.tensor([1e-5, -2e-5, 3e-5], dtype=torch.float64)
    cd = och.chamfer_pair(x + t, x).item()
    assert abs(cd - 2 * float((t ** 2).sum())) < 1e-12


def test_chamfer_two_point_hand_computed():
    x = torch.tensor([[[0.0, 0, 0], [1, 0, 0]]])
    y = torch.tensor([[[0.0, 0.5, 0], [1, 0, 2], [5, 5, 5]]])
    # x->y: 0.25, min(1.25, 4)=1.25 -> mean 0.75 ; y->x: 0.25, min(5, 4)=4, min(75, 66)=66 -> mean 70.25/3
    assert abs(och.chamfer_pair(x, y).item() - (0.75 + 70.25 / 3)) < 1e-5


def test_face_area_and_barycentric():
    v = torch.tensor([[0.0, 0, 0], [2, 0, 0], [0, 3, 0]])
    f = torch.tensor([[0, 1, 2]])
    assert och.face_areas(v, f).item() == 3.0
    w0, w1, w2 = och.barycentric(torch.tensor([0.25]), torch.tensor([0.5]))
    assert (w0.item(), w1.item(), w2.item()) == (0.5, 0.25, 0.25)


def test_zero_area_mesh_falls_back_to_uniform():
    v = torch.zeros(1, 4, 3)
    f = torch.tensor([[0, 1, 2], [1, 2, 3]])
    assert torch.equal(och.face_probabilities(v, f), torch.ones(1, 2))     # utils.py:166-168 NaN scrubs


def test_normalized_adjacency_rows_sum_to_one_and_identity_gcn_is_mlp():
    f = np.array([[0, 1, 2], [2, 3, 0]])
    a = omesh.normalize_adj(omesh.calc_adj(f))
    assert np.allclose(a.sum(1), 1.0)
    st = og.init_state(8, 16, 3, seed=1)
    x = torch.randn(2, 4, 8)
    eye = torch.eye(4)
    out = og.gcn(x, st, "mesh_deform_1", eye, 3, 0.33)
    h = x
    for i in range(3):
        h = h @ st[f"mesh_deform_1.layers.{i}.weight"][0]
        b = st[f"mesh_deform_1.layers.{i}.bias"]
        if i < 2:
            c = round(16 * 0.33)
            h = torch.cat((h[..., :c] + b[:c], h[..., c:]), -1).relu()   # bias only on the first c channels
        else:
            h = h + b
    assert torch.allclose(out, h, atol=1e-6)


def test_cut_length_uses_bankers_rounding():
    assert og.cut_length(300, 0.33) == 99 and og.cut_length(100, 0.33) == 33 and og.cut_length(50, 0.5) == 25


def test_nerf_embedding_layout():
    p = torch.tensor([[0.1, 0.2, 0.3]])
    e = og.nerf_embedding(p)
    assert e.shape == (1, 60)
    fr = [np.pi] + [np.pi * 2 * i for i in range(1, 10)]
    for i, f in enumerate(fr):
        assert torch.allclose(e[0, 6 * i:6 * i + 3], torch.sin(f * p[0]))
        assert torch.allclose(e[0, 6 * i + 3:6 * i + 6], torch.cos(f * p[0]))
