#!/usr/bin/env python
"""Generate the committed golden vectors by running the REAL reference (``/root/reference``) on the CPU.

Run in the build container only:  ``python tests/golden/make_golden.py``  (~1-2 min).
The reference source never leaves that container; what is committed are small ``.npz`` files holding
inputs (or the seeds that regenerate them) and the reference's outputs.  ``oracle.ref_shim`` explains the
import shim (``.cuda()`` no-op; PyTorch3D symbols restated — which is why every fixture that crosses the
Chamfer / sampling boundary is marked ``pytorch3d_restated=True``: parity unpinned there).

Fixtures (names follow SURVEY §8c):
  g1_adjacency.npz     CSR of the reference's row-normalised adjacency for the atlas: vision-only, t_p, t_g
  g2_gcn_layer.npz     GCN_layer fwd + grads (B=2, N=1824, 50->300 with the cut; 300->300 without), atlas adjacency
  g3_small_<mode>.npz  reduced Deformation (L=3,H=32): weights, inputs, verts, loss (injected samples), all grads
  g4_full_forward.npz  full-size Deformation (L=20,H=300, seed-0 init) forward verts for B=2 + weight checksum
  g5_sampling.npz      batch_sample probabilities + points for injected (face, u, v)
  g6_chamfer.npz       chamfer_distance + d/dx on random clouds and on the bundled ABC object's cloud
  g7_train_step.npz    one trainer step (bs=2, P=10000, 3 stages, and the 1-stage variant): loss before / after Adam
  g8_image_<mode>.npz  use_img models (default CNN, reduced GCN): verts (eval + train mode), loss, selected grads
  g9_autoencoder.npz   reconstruction/autoencoder AutoEncoder (L=3, H=300): latent, folded points, Chamfer loss with
                       the gradient on the SECOND cloud (autoencoder/train.py:145-150), selected grads
  g10_graph_model.npz  policies/DDQN Graph_Model (3 layers, 300 -> 200 -> 200 -> 50): Q values + selected grads
  g12_atlas_b8.npz     full-size Deformation (L=20, H=300, seed-0 init) on EIGHT differently perturbed atlases (14 592 rows:
                       enough for the product kernels bench.py times — the hybrid-row channel-sliced aggregation, the
                       19-tile products with the A operand in registers, the split-operand kernels of gemm mode 3):
                       forward verts, Chamfer with injected samples, the gradient norm of every parameter tensor and a few
                       whole gradients
  g13_touch_b8.npz     the same on the production topology t_g (atlas + 20 touch charts, N = 2324) at B = 8 = 18 592 rows,
                       touch charts with empty / touched / untouched slots: reaches the P + bipartite split kernels (round 6)
  g11_loader_batch.npz the reference's ``mesh_loader_vision`` (utility/data_loaders.py:132-258) on the miniature dataset
                       ``golden_util.write_mini_dataset`` writes: instance list, seeded validation grasp choices, seeded
                       training draws, one collated batch (touch_charts, gt_points, image samples), finger variant
"""
import hashlib
import os
import sys
from types import SimpleNamespace as NS

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(HERE))

from oracle import mesh as omesh  # noqa: E402
from oracle import ref_shim  # noqa: E402

torch.set_num_threads(8)
ref = ref_shim.load_reference()
assert ref is not None, "needs /root/reference"
OBJ = os.path.join(ref.objects_dir, "vision_charts.obj")


def args_of(**kw):
    d = dict(use_img=False, use_touch=False, finger=False, num_grasps=1, num_GCN_layers=20, hidden_GCN_size=300, cut=0.33)
    d.update(kw)
    return NS(**d)


def save(name, **arrs):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **arrs)
    print(f"{name}: {os.path.getsize(path) / 1024:.1f} KiB")


def state_checksum(sd):
    h = hashlib.sha256()
    for k in sorted(sd):
        h.update(k.encode())
        h.update(sd[k].detach().cpu().numpy().tobytes())
    return np.frombuffer(h.digest(), dtype=np.uint8)


def g1():
    out = {}
    for tag, kw in (("vision", dict(use_touch=False)), ("t_p", dict(use_touch=True, finger=True, num_grasps=5)),
                    ("t_g", dict(use_touch=True, finger=False, num_grasps=5))):
        info, verts = ref.utils.load_mesh_vision(args_of(**kw), OBJ)
        for key in ("origional", "adj"):
            rp, col, val = omesh.dense_to_csr(info[key].numpy())
            out[f"{tag}_{key}_rowptr"], out[f"{tag}_{key}_col"], out[f"{tag}_{key}_val"] = rp, col.astype(np.int16) \
                if col.max() < 32768 else col, val
        out[f"{tag}_faces"] = info["faces"].numpy().astype(np.int16)
    out["verts"] = verts.numpy()
    save("g1_adjacency.npz", **out)


def touch_batch(B, args, seed):
    g = torch.Generator().manual_seed(seed)
    shape = (B, args.num_grasps, 25, 4) if args.finger else (B, args.num_grasps, 4, 25, 4)
    t = torch.zeros(shape)
    t[..., :3] = (torch.rand(shape[:-1] + (3,), generator=g) - 0.5) * 0.3
    t[..., 3] = torch.randint(0, 3, shape[:-2] + (1,), generator=g).float()
    t[..., :3] *= (t[..., 3:] > 0).float()
    return t


def injected(B, F, P, seed, repeat=3):
    g = torch.Generator().manual_seed(seed)
    return [(torch.randint(0, F, (B, P), generator=g), torch.rand(B, P, generator=g), torch.rand(B, P, generator=g))
            for _ in range(repeat)]


def ref_chamfer_injected(verts, faces, gt, samples):
    """utils.chamfer_distance with the sampling replaced by injected draws (same arithmetic after the draw)."""
    from oracle import chamfer as och
    cds = []
    for fi, u, v in samples:
        # reference arithmetic of utils.py:174-185 on the injected indices / uniforms
        bs, vd = verts.shape[0], verts.shape[1]
        F = faces.unsqueeze(0).repeat(bs, 1, 1) + vd * torch.arange(bs).view(-1, 1, 1)
        V = verts.reshape(-1, 3)
        fv = V[F.reshape(-1, 3)]
        idx = fi + faces.shape[0] * torch.arange(bs).unsqueeze(-1)
        w0, w1, w2 = och.barycentric(u, v)
        pts = w0[:, :, None] * fv[:, 0][idx] + w1[:, :, None] * fv[:, 1][idx] + w2[:, :, None] * fv[:, 2][idx]
        cds.append(sys.modules["pytorch3d.loss"].chamfer_distance(pts, gt, batch_reduction=None)[0])
    return torch.stack(cds).mean(0)


def g3():
    for tag, kw in (("vision", dict(use_touch=False)), ("touch", dict(use_touch=True, num_grasps=1, finger=False))):
        a = args_of(num_GCN_layers=3, hidden_GCN_size=32, **kw)
        torch.manual_seed(7)
        info, verts = ref.utils.load_mesh_vision(a, OBJ)
        net = ref.model.Deformation(info, verts, a)
        B, P = 2, 300
        batch = {"img": torch.zeros(B, 1), "touch_charts": touch_batch(B, a, 3)}
        charts = ref.model.prepare_mesh(batch, verts, a)
        out, mask = net(batch["img"], charts)
        g = torch.Generator().manual_seed(5)
        gt = (torch.rand(B, 400, 3, generator=g) - 0.5) * 0.3
        samples = injected(B, info["faces"].shape[0], P, 11)
        cd = ref_chamfer_injected(out, info["faces"], gt, samples)
        loss = 9000.0 * cd.mean()
        loss.backward()
        arrs = {"verts_out": out.detach().numpy(), "mask": mask.numpy(), "cd": cd.detach().numpy(),
                "loss": np.float32(loss.item()), "gt": gt.numpy(), "touch_charts": batch["touch_charts"].numpy(),
                "face_idx": torch.stack([s[0] for s in samples]).numpy().astype(np.int16),
                "u": torch.stack([s[1] for s in samples]).numpy(), "v": torch.stack([s[2] for s in samples]).numpy(),
                "pytorch3d_restated": np.bool_(True)}
        for k, p in net.named_parameters():
            arrs["w:" + k] = p.detach().numpy()
            arrs["g:" + k] = p.grad.numpy()
        save(f"g3_small_{tag}.npz", **arrs)


def g4():
    a = args_of()
    torch.manual_seed(0)
    info, verts = ref.utils.load_mesh_vision(a, OBJ)
    net = ref.model.Deformation(info, verts, a)
    B = 2
    batch = {"img": torch.zeros(B, 1)}
    charts = ref.model.prepare_mesh(batch, verts, a)
    # perturb the two samples differently so the batch dimension is exercised
    g = torch.Generator().manual_seed(1)
    charts["vision_charts"] = charts["vision_charts"] + 0.01 * torch.randn(B, verts.shape[0], 3, generator=g)
    with torch.no_grad():
        out, _ = net(batch["img"], charts)
    save("g4_full_forward.npz", verts_in=charts["vision_charts"].numpy(), verts_out=out.numpy(),
         weight_sha256=state_checksum(net.state_dict()),
         w_first=net.mesh_deform_1.layers[0].weight.detach().numpy()[0, :4, :8],
         w_last=net.mesh_deform_2.layers[19].weight.detach().numpy()[0, :8])


def g12():
    a = args_of()
    torch.manual_seed(0)
    info, verts = ref.utils.load_mesh_vision(a, OBJ)
    net = ref.model.Deformation(info, verts, a)
    B, P, Q = 8, 1000, 1500
    batch = {"img": torch.zeros(B, 1)}
    charts = ref.model.prepare_mesh(batch, verts, a)
    g = torch.Generator().manual_seed(12)
    # every sample perturbed differently (and by a different amount), so rows of different meshes differ
    amp = torch.linspace(0.004, 0.02, B).view(B, 1, 1)
    charts["vision_charts"] = charts["vision_charts"] + amp * torch.randn(B, verts.shape[0], 3, generator=g)
    verts_in = charts["vision_charts"].clone()
    out, _ = net(batch["img"], charts)
    d = torch.randn(B, Q, 3, generator=g)
    gt = d / d.norm(dim=-1, keepdim=True) * (0.05 + 0.11 * torch.rand(B, 1, 3, generator=g))
    samples = injected(B, info["faces"].shape[0], P, 112)
    cd = ref_chamfer_injected(out, info["faces"], gt, samples)
    loss = 9000.0 * cd.mean()
    loss.backward()
    arrs = {"verts_in": verts_in.numpy(), "verts_out": out.detach().numpy(), "cd": cd.detach().numpy(),
            "loss": np.float32(loss.item()), "gt": gt.numpy(),
            "face_idx": torch.stack([s_[0] for s_ in samples]).numpy().astype(np.int16),
            "u": torch.stack([s_[1] for s_ in samples]).numpy(), "v": torch.stack([s_[2] for s_ in samples]).numpy(),
            "weight_sha256": state_checksum(net.state_dict()), "pytorch3d_restated": np.bool_(True)}
    names, norms = [], []
    for k, p in net.named_parameters():
        names.append(k)
        norms.append(0.0 if p.grad is None else float(p.grad.double().norm()))
    arrs["grad_names"] = np.array(names)
    arrs["grad_norms"] = np.array(norms, dtype=np.float64)
    keep = ["mesh_deform_1.layers.0.weight", "mesh_deform_1.layers.10.bias", "mesh_deform_2.layers.7.bias",
            "mesh_deform_2.layers.19.weight", "mesh_deform_2.layers.19.bias", "mesh_deform_1.layers.19.weight",
            "positional_encoder.model.0.weight", "mask_encoder.model.0.weight"]
    sd = dict(net.named_parameters())
    for k in keep:
        arrs["g:" + k] = sd[k].grad.numpy()
    # one hidden-layer weight gradient, subsampled (300 x 300 floats would be 360 KB): every 7th row, every 5th column
    arrs["g:mesh_deform_2.layers.9.weight[::7,::5]"] = sd["mesh_deform_2.layers.9.weight"].grad.numpy()[0, ::7, ::5]
    save("g12_atlas_b8.npz", **arrs)


def g13():
    """Full-size Deformation on the PRODUCTION topology t_g (atlas + 5 grasps x 4 fingers = 20 touch charts, N = 2324, hub rows
    of 1153 entries, utils.py:75-130) at B = 8 = 18 592 rows: enough for the kernels a touch training step runs — since round
    6 the P + bipartite split of the fused adjacency (csrc/gcn_csrqs.hip) on hybrid rows."""
    a = args_of(use_touch=True, finger=False, num_grasps=5)
    torch.manual_seed(0)
    info, verts = ref.utils.load_mesh_vision(a, OBJ)
    net = ref.model.Deformation(info, verts, a)
    B, P, Q = 8, 1000, 1500
    tc = touch_batch(B, a, 13)                 # status 0 / 1 / 2 per chart: empty, touched and untouched slots
    batch = {"img": torch.zeros(B, 1), "touch_charts": tc}
    charts = ref.model.prepare_mesh(batch, verts, a)
    g = torch.Generator().manual_seed(13)
    amp = torch.linspace(0.004, 0.02, B).view(B, 1, 1)
    charts["vision_charts"] = charts["vision_charts"] + amp * torch.randn(B, verts.shape[0], 3, generator=g)
    verts_in = charts["vision_charts"].clone()
    out, mask = net(batch["img"], charts)
    d = torch.randn(B, Q, 3, generator=g)
    gt = d / d.norm(dim=-1, keepdim=True) * (0.05 + 0.11 * torch.rand(B, 1, 3, generator=g))
    samples = injected(B, info["faces"].shape[0], P, 113)
    cd = ref_chamfer_injected(out, info["faces"], gt, samples)
    loss = 9000.0 * cd.mean()
    loss.backward()
    arrs = {"verts_in": verts_in.numpy(), "touch_charts": tc.numpy(), "verts_out": out.detach().numpy(),
            "mask": mask.numpy().astype(np.int8), "cd": cd.detach().numpy(),
            "loss": np.float32(loss.item()), "gt": gt.numpy(),
            "face_idx": torch.stack([s_[0] for s_ in samples]).numpy().astype(np.int16),
            "u": torch.stack([s_[1] for s_ in samples]).numpy(), "v": torch.stack([s_[2] for s_ in samples]).numpy(),
            "weight_sha256": state_checksum(net.state_dict()), "pytorch3d_restated": np.bool_(True)}
    names, norms = [], []
    for k, p in net.named_parameters():
        names.append(k)
        norms.append(0.0 if p.grad is None else float(p.grad.double().norm()))
    arrs["grad_names"] = np.array(names)
    arrs["grad_norms"] = np.array(norms, dtype=np.float64)
    keep = ["mesh_deform_1.layers.0.weight", "mesh_deform_1.layers.10.bias", "mesh_deform_2.layers.7.bias",
            "mesh_deform_2.layers.19.weight", "mesh_deform_2.layers.19.bias", "mesh_deform_1.layers.19.weight",
            "positional_encoder.model.0.weight", "mask_encoder.model.0.weight"]
    sd = dict(net.named_parameters())
    for k in keep:
        arrs["g:" + k] = sd[k].grad.numpy()
    arrs["g:mesh_deform_2.layers.9.weight[::7,::5]"] = sd["mesh_deform_2.layers.9.weight"].grad.numpy()[0, ::7, ::5]
    save("g13_touch_b8.npz", **arrs)


def g14():
    """The IMAGE model at full depth on the configs[3] topology (use_img + use_touch: atlas + 4 touch charts, N = 1924; default
    CNN: k = 5, 6 blocks x 3 layers -> 448-wide vertex features; GCN 20 x 300) at B = 8 = 15 392 rows, training mode (batch
    statistics in the 2 x 13 BatchNorm layers): enough rows for the stacks to run on hybrid rows with the P + bipartite split,
    the 448-wide feature encoder and the timed product kernels.  Weights are NOT stored (47 M): tests re-derive them from
    torch.manual_seed(0) with the product's constructor and check the SHA-256; the image is regenerated from its seed."""
    a = args_of(use_img=True, use_touch=True, finger=False, num_grasps=1, CNN_ker_size=5, num_CNN_blocks=6, layers_per_block=3)
    torch.manual_seed(0)
    info, verts = ref.utils.load_mesh_vision(a, OBJ)
    net = ref.model.Deformation(info, verts, a)
    sha = state_checksum(net.state_dict())       # (before the training forward moves the running statistics)
    net.train()
    B, P, Q = 8, 800, 1200
    g = torch.Generator().manual_seed(14)
    img = torch.rand(B, 3, 256, 256, generator=g)
    tc = touch_batch(B, a, 14)
    batch = {"img": img, "touch_charts": tc}
    charts = ref.model.prepare_mesh(batch, verts, a)
    amp = torch.linspace(0.004, 0.02, B).view(B, 1, 1)
    charts["vision_charts"] = charts["vision_charts"] + amp * torch.randn(B, verts.shape[0], 3, generator=g)
    verts_in = charts["vision_charts"].clone()
    out, mask = net(img, charts)
    d = torch.randn(B, Q, 3, generator=g)
    gt = d / d.norm(dim=-1, keepdim=True) * (0.05 + 0.11 * torch.rand(B, 1, 3, generator=g))
    samples = injected(B, info["faces"].shape[0], P, 114)
    cd = ref_chamfer_injected(out, info["faces"], gt, samples)
    loss = 9000.0 * cd.mean()
    loss.backward()
    arrs = {"verts_in": verts_in.numpy(), "touch_charts": tc.numpy(), "img_seed": np.int64(14), "verts_out": out.detach().numpy(),
            "mask": mask.numpy().astype(np.int8), "cd": cd.detach().numpy(), "loss": np.float32(loss.item()), "gt": gt.numpy(),
            "face_idx": torch.stack([s_[0] for s_ in samples]).numpy().astype(np.int16),
            "u": torch.stack([s_[1] for s_ in samples]).numpy(), "v": torch.stack([s_[2] for s_ in samples]).numpy(),
            "weight_sha256": sha, "pytorch3d_restated": np.bool_(True)}
    names, norms = [], []
    for k, p in net.named_parameters():
        names.append(k)
        norms.append(0.0 if p.grad is None else float(p.grad.double().norm()))
    arrs["grad_names"] = np.array(names)
    arrs["grad_norms"] = np.array(norms, dtype=np.float64)
    sd = dict(net.named_parameters())
    for k in ("mesh_deform_1.layers.10.bias", "mesh_deform_2.layers.7.bias", "mesh_deform_2.layers.19.weight",
              "mesh_deform_1.layers.19.bias", "positional_encoder.model.0.weight", "mask_encoder.model.0.weight",
              "img_encoder_global.layers.0.0.weight", "img_encoder_local.layers.9.2.bias", "img_encoder_local.layers.4.0.weight"):
        arrs["g:" + k] = sd[k].grad.numpy()
    arrs["g:mesh_deform_1.layers.0.weight[::9,::7]"] = sd["mesh_deform_1.layers.0.weight"].grad.numpy()[0, ::9, ::7]
    arrs["g:mesh_deform_2.layers.9.weight[::7,::5]"] = sd["mesh_deform_2.layers.9.weight"].grad.numpy()[0, ::7, ::5]
    arrs["g:img_encoder_global.layers.7.2.weight[::3,::5]"] = sd["img_encoder_global.layers.7.2.weight"].grad.numpy()[::3, ::5]
    # running statistics after the one training forward (momentum 0.1): the BatchNorm side effect callers checkpoint
    st = net.state_dict()
    for k in ("img_encoder_global.layers.1.0.running_mean", "img_encoder_global.layers.1.0.running_var",
              "img_encoder_local.layers.10.0.running_mean", "img_encoder_local.layers.10.0.running_var"):
        arrs["s:" + k] = st[k].numpy()
    save("g14_image_touch_b8.npz", **arrs)


def g5():
    a = args_of(use_touch=True, num_grasps=1)
    info, verts = ref.utils.load_mesh_vision(a, OBJ)
    B, P = 2, 500
    g = torch.Generator().manual_seed(2)
    V = torch.cat((verts[None].repeat(B, 1, 1) + 0.01 * torch.randn(B, verts.shape[0], 3, generator=g),
                   touch_batch(B, a, 4).view(B, -1, 4)[:, :, :3]), dim=1)
    faces = info["faces"]
    from oracle import chamfer as och
    # reference probabilities (utils.py:163-168) through the shimmed mesh_face_areas_normals
    bs, vd = V.shape[0], V.shape[1]
    F = faces.unsqueeze(0).repeat(bs, 1, 1) + vd * torch.arange(bs).view(-1, 1, 1)
    areas, _ = sys.modules["pytorch3d.ops.mesh_face_areas_normals"].mesh_face_areas_normals(V.reshape(-1, 3), F.reshape(-1, 3))
    Ar = areas.reshape(bs, -1)
    Ar[Ar != Ar] = 0
    Ar = torch.abs(Ar / Ar.sum(1).unsqueeze(1))
    Ar[Ar != Ar] = 1
    fi, u, v = injected(B, faces.shape[0], P, 6, repeat=1)[0]
    torch.manual_seed(123)
    pts_rng = ref.utils.batch_sample(V, faces, num=P)  # the reference's own RNG path, regenerated by the oracle
    idx = fi + faces.shape[0] * torch.arange(bs).unsqueeze(-1)
    fv = V.reshape(-1, 3)[F.reshape(-1, 3)]
    w0, w1, w2 = och.barycentric(u, v)
    pts = w0[:, :, None] * fv[:, 0][idx] + w1[:, :, None] * fv[:, 1][idx] + w2[:, :, None] * fv[:, 2][idx]
    save("g5_sampling.npz", verts=V.numpy(), prob=Ar.numpy(), face_idx=fi.numpy().astype(np.int16), u=u.numpy(),
         v=v.numpy(), points=pts.numpy(), points_rng_seed123=pts_rng.numpy(), pytorch3d_restated=np.bool_(True))


def abc_cloud():
    """GT cloud of the bundled ABC object via the reference's own recipe (data_making.py:50-72, scale 3.1)."""
    torch.cuda.FloatTensor = torch.FloatTensor
    loc = os.path.join(ref.objects_dir, "test_objects", "0.obj")
    verts, faces = omesh.load_obj(loc)
    verts = ref.utils.scale_points(verts.copy(), 3.1) if hasattr(ref.utils, "scale_points") else verts
    v = torch.FloatTensor(verts)
    f = torch.LongTensor(faces)
    voxel = ref.utils.mesh_to_voxel(v, f, 128)
    odms = ref.utils.extract_ODMs(voxel)
    voxel = ref.utils.apply_ODMs(odms, 128)
    pts = ref.utils.voxel_to_pointcloud(voxel)
    pts = ref.utils.realign_points(pts, v.clone())
    return pts.float()


def g6():
    g = torch.Generator().manual_seed(3)
    B, P, Q = 2, 600, 450
    x = ((torch.rand(B, P, 3, generator=g) - 0.5) * 0.3).requires_grad_(True)
    y = (torch.rand(B, Q, 3, generator=g) - 0.5) * 0.3
    cd = sys.modules["pytorch3d.loss"].chamfer_distance(x, y, batch_reduction=None)[0]
    (cd * torch.tensor([1.0, 2.0])).sum().backward()
    arrs = dict(x=x.detach().numpy(), y=y.numpy(), cd=cd.detach().numpy(), grad_x=x.grad.numpy(),
                pytorch3d_restated=np.bool_(True))
    try:
        cloud = abc_cloud()
        arrs["abc_cloud"] = cloud.numpy()
        a = args_of(use_touch=True, num_grasps=5, finger=False)
        info, verts = ref.utils.load_mesh_vision(a, OBJ)
        Bq = 2
        V = torch.cat((verts[None].repeat(Bq, 1, 1), torch.zeros(Bq, 500, 3)), dim=1)  # empty touch slots at the origin
        n = min(10000, cloud.shape[0])
        gt = cloud[:n][None].repeat(Bq, 1, 1)
        samples = injected(Bq, info["faces"].shape[0], 2000, 21)
        score = 9000.0 * ref_chamfer_injected(V, info["faces"], gt, samples)
        arrs["abc_score"] = score.numpy()
        arrs["abc_face_idx"] = torch.stack([s[0] for s in samples]).numpy().astype(np.int16)
        arrs["abc_u"] = torch.stack([s[1] for s in samples]).numpy()
        arrs["abc_v"] = torch.stack([s[2] for s in samples]).numpy()
    except Exception as e:  # the voxel recipe needs optional deps; the random-cloud vectors still stand
        print("abc cloud skipped:", repr(e))
    save("g6_chamfer.npz", **arrs)


def g7():
    """cfg-1 (BASELINE.json configs[0]): bs=2, P=10000, reference trainer arithmetic on the CPU."""
    a = args_of()
    P, B = 10000, 2
    out = {}
    from oracle import chamfer as och
    for stages in (3, 1):
        torch.manual_seed(0)
        info, verts = ref.utils.load_mesh_vision(a, OBJ)
        net = ref.model.Deformation(info, verts, a)
        opt = torch.optim.Adam(list(net.parameters()), lr=3e-4, weight_decay=0)
        g = torch.Generator().manual_seed(99)
        d = torch.randn(B, P, 3, generator=g)
        gt = d / d.norm(dim=-1, keepdim=True) * (0.05 + 0.11 * torch.rand(B, 1, 3, generator=g))
        batch = {"img": torch.zeros(B, 1)}

        def forward():
            charts = ref.model.prepare_mesh(batch, verts, a)
            if stages == 3:
                return net(batch["img"], charts)[0]
            # 1-stage variant composed from the reference's own sub-modules (model.py:236-250)
            vtx = charts["vision_charts"].clone()
            feats = net.positional_encoder(vtx) + net.mask_encoder(charts["vision_masks"].clone())
            return vtx + net.mesh_deform_1(feats, net.adj_info)

        losses = []
        for it in range(2):
            torch.manual_seed(1000 + it)  # fixes the reference's multinomial / rand draws for this evaluation
            opt.zero_grad()
            loss = 9000.0 * ref.utils.chamfer_distance(forward(), info["faces"], gt, num=P).mean()
            losses.append(loss.item())
            if it == 0:
                loss.backward()
                opt.step()
        out[f"loss_before_s{stages}"] = np.float64(losses[0])
        out[f"loss_after_s{stages}"] = np.float64(losses[1])
        out[f"weight_sha256_after_s{stages}"] = state_checksum(net.state_dict())
        out[f"w_after_sample_s{stages}"] = net.mesh_deform_1.layers[19].weight.detach().numpy()[0, :16]
    out["gt_seed"] = np.int64(99)
    out["pytorch3d_restated"] = np.bool_(True)
    save("g7_train_step.npz", **out)


def g8():
    """use_img modes (vision-only and vision+touch): default CNN (k=5, 6 blocks x 3 layers -> 448-wide features),
    reduced GCN (L=3, H=300).  Weights are NOT stored (44 M): tests re-derive them from torch.manual_seed(0) with the
    product's constructor and check the SHA-256."""
    for tag, kw in (("vision", dict(use_touch=False)), ("touch", dict(use_touch=True, num_grasps=1, finger=False))):
        a = args_of(use_img=True, num_GCN_layers=3, hidden_GCN_size=300, CNN_ker_size=5, num_CNN_blocks=6,
                    layers_per_block=3, **kw)
        torch.manual_seed(0)
        info, verts = ref.utils.load_mesh_vision(a, OBJ)
        net = ref.model.Deformation(info, verts, a)
        B, P = 2, 300
        g = torch.Generator().manual_seed(17)
        img = torch.rand(B, 3, 256, 256, generator=g)
        batch = {"img": img, "touch_charts": touch_batch(B, a, 3)}
        out = {"weight_sha256": state_checksum(net.state_dict()), "touch_charts": batch["touch_charts"].numpy(),
               "img_seed": np.int64(17), "pytorch3d_restated": np.bool_(True)}
        gt = (torch.rand(B, 400, 3, generator=g) - 0.5) * 0.3
        samples = injected(B, info["faces"].shape[0], P, 11)
        out["gt"] = gt.numpy()
        out["face_idx"] = torch.stack([s_[0] for s_ in samples]).numpy().astype(np.int16)
        out["u"] = torch.stack([s_[1] for s_ in samples]).numpy()
        out["v"] = torch.stack([s_[2] for s_ in samples]).numpy()
        for mode in ("eval", "train"):
            net.train(mode == "train")
            net.zero_grad()
            verts_out, mask = net(batch["img"], ref.model.prepare_mesh(batch, verts, a))
            cd = ref_chamfer_injected(verts_out, info["faces"], gt, samples)
            loss = 9000.0 * cd.mean()
            out[f"verts_out_{mode}"] = verts_out.detach().numpy()
            out[f"cd_{mode}"] = cd.detach().numpy()
            if mode == "train":
                loss.backward()
                for k in ("mesh_deform_1.layers.0.weight", "mesh_deform_2.layers.2.weight", "mesh_deform_2.layers.0.bias",
                          "img_encoder_global.layers.0.0.weight", "img_encoder_local.layers.9.2.bias",
                          "positional_encoder.model.4.bias", "mask_encoder.model.0.weight"):
                    gk = dict(net.named_parameters())[k].grad
                    out["g:" + k] = gk.numpy() if gk.numel() < 40000 else gk.numpy()[..., ::7, ::11]
                out["mask"] = mask.numpy()
        save(f"g8_image_{tag}.npz", **out)


def g2():
    """the reference layer itself, on the real atlas adjacency (dense, as the reference holds it)."""
    info, verts = ref.utils.load_mesh_vision(args_of(), OBJ)
    adj = info["adj"]
    n = adj.shape[0]
    out = {}
    for tag, (kin, nout, do_cut, relu) in (("cut", (50, 300, True, True)), ("nocut", (300, 300, False, False))):
        torch.manual_seed(21)
        layer = ref.model.GCN_layer(kin, nout, 0.33, do_cut)
        g = torch.Generator().manual_seed(kin)
        x = (torch.randn(2, n, kin, generator=g) * 0.5).requires_grad_(True)
        gy = torch.randn(2, n, nout, generator=g)
        y = layer(x, adj, torch.nn.functional.relu if relu else (lambda t: t))
        (y * gy).sum().backward()
        out[f"{tag}_weight_sha256"] = state_checksum(layer.state_dict())   # tests re-derive the weights from seed 21
        out[f"{tag}_x_seed"] = np.int64(kin)
        out[f"{tag}_y"] = y.detach().numpy()[:, ::32]              # every 32nd vertex
        out[f"{tag}_y_sum"] = y.detach().double().sum(dim=(0, 1)).numpy()  # per-channel checksum over all vertices
        out[f"{tag}_gx"] = x.grad.numpy()[:, ::32]
        out[f"{tag}_gw"], out[f"{tag}_gb"] = layer.weight.grad.numpy()[0, ::3, ::5], layer.bias.grad.numpy()
    save("g2_gcn_layer.npz", **out)


def g9():
    import importlib
    am = importlib.import_module("pterotactyl.reconstruction.autoencoder.model")
    a = args_of(use_touch=True, num_grasps=1, finger=False, num_GCN_layers=3, hidden_GCN_size=300, encoding_size=200)
    torch.manual_seed(0)
    info, verts = ref.utils.load_mesh_vision(a, OBJ)
    net = am.AutoEncoder(info, verts, a)
    B, P = 2, 300
    batch = {"img": torch.zeros(B, 1), "touch_charts": touch_batch(B, a, 3)}
    charts = ref.model.prepare_mesh(batch, verts, a)
    g = torch.Generator().manual_seed(23)
    v_in = torch.cat((charts["vision_charts"], charts["touch_charts"]), dim=1)
    v_in = v_in + 0.01 * torch.randn(v_in.shape, generator=g)
    mask = torch.cat((charts["vision_masks"], charts["touch_masks"]), dim=1)
    pred, latent = net(v_in, mask)
    samples = injected(B, info["faces"].shape[0], P, 11)
    cd = ref_chamfer_injected(v_in, info["faces"], pred, samples)   # grad flows to the second cloud only
    loss = 9000.0 * cd.mean()
    loss.backward()
    out = {"weight_sha256": state_checksum(net.state_dict()), "verts_in": v_in.numpy(), "mask": mask.numpy(),
           "latent": latent.detach().numpy(), "pred_points": pred.detach().numpy()[:, ::16], "cd": cd.detach().numpy(),
           "face_idx": torch.stack([s_[0] for s_ in samples]).numpy().astype(np.int16),
           "u": torch.stack([s_[1] for s_ in samples]).numpy(), "v": torch.stack([s_[2] for s_ in samples]).numpy(),
           "pytorch3d_restated": np.bool_(True)}
    params = dict(net.named_parameters())
    for k in ("encoder.layers.0.weight", "encoder.layers.2.bias", "encoder.layers.1.bias", "encoder.mlp.3.0.bias",
              "decoder.initial.bias", "decoder.model.fold2.conv3.weight", "positional_encoder.model.4.bias",
              "mask_encoder.model.0.weight"):
        out["g:" + k] = params[k].grad.numpy()
    out["g:encoder.layers.2.weight"] = params["encoder.layers.2.weight"].grad.numpy()[..., ::7, ::11]
    save("g9_autoencoder.npz", **out)


def g10():
    import importlib
    dm = importlib.import_module("pterotactyl.policies.DDQN.model")
    a = args_of(use_touch=True, num_grasps=5, finger=True, layers=3, hidden_dim=200, num_actions=50)
    torch.manual_seed(0)
    info, verts = ref.utils.load_mesh_vision(a, OBJ)
    net = dm.Graph_Model(a, info)
    B = 3
    g = torch.Generator().manual_seed(29)
    n = info["adj"].shape[0]
    mesh = torch.zeros(B, n, 4)
    mesh[:, :verts.shape[0], :3] = verts + 0.01 * torch.randn(B, verts.shape[0], 3, generator=g)
    mesh[:, :verts.shape[0], 3] = 3
    mesh[:, verts.shape[0]:, :3] = (torch.rand(B, n - verts.shape[0], 3, generator=g) - 0.5) * 0.3
    mesh[:, verts.shape[0]:, 3] = torch.randint(0, 3, (B, n - verts.shape[0]), generator=g).float()
    done = (torch.rand(B, 50, generator=g) < 0.2).float()
    obs = {"mesh": mesh, "mask": done}
    q = net(obs)
    gq = torch.randn(q.shape, generator=g)
    (q * gq).sum().backward()
    out = {"weight_sha256": state_checksum(net.state_dict()), "mesh": mesh.numpy(), "mask": done.numpy(),
           "q": q.detach().numpy(), "gq": gq.numpy()}
    params = dict(net.named_parameters())
    for k in ("layers.0.bias", "layers.1.weight", "layers.2.weight", "layers.2.bias", "action_model.2.0.bias",
              "positional_embedding.model.4.bias", "mask_embedding.model.0.weight"):
        out["g:" + k] = params[k].grad.numpy()
    out["g:layers.0.weight"] = params["layers.0.weight"].grad.numpy()[..., ::3, ::5]
    save("g10_graph_model.npz", **out)


def g11():
    """The reference's vision-trainer dataset class reading the miniature dataset: its module-level location constants
    (data_loaders.py:18-29) are pointed at a temporary directory, ``glob`` is made order-stable (sorted) so that the
    position-derived validation seeds (:160-170) do not depend on the file system."""
    import importlib
    import random
    import tempfile
    from glob import glob as _glob
    from golden_util import write_mini_dataset
    dl = importlib.import_module("pterotactyl.utility.data_loaders")
    out = {}
    with tempfile.TemporaryDirectory() as root:
        write_mini_dataset(root, n=6, seed=0)
        dl.POINT_CLOUD_LOCATION = os.path.join(root, "point_cloud_info") + "/"
        dl.TOUCH_LOCATION = os.path.join(root, "touch_charts") + "/"
        dl.IMAGE_LOCATION = os.path.join(root, "images_colourful") + "/"
        dl.OBJ_LOCATION = os.path.join(root, "object_info") + "/"
        dl.DATA_SPLIT = np.load(os.path.join(root, "data_split.npy"), allow_pickle=True).item()
        dl.glob = lambda pat: sorted(_glob(pat))
        dl.tqdm = lambda it: it
        for tag, finger in (("full", False), ("finger", True)):
            a = NS(use_touch=True, use_img=True, finger=finger, num_grasps=3, number_points=500, eval=False,
                   limit_data=False, val_grasps=-1)
            valid = dl.mesh_loader_vision(a, set_type="valid")
            train = dl.mesh_loader_vision(a, set_type="recon_train")
            if not finger:
                out["valid_names"] = np.array([int(n) for n, _ in valid.object_names], dtype=np.int64)
                out["valid_seeds"] = np.array([s_ for _, s_ in valid.object_names], dtype=np.int64)
                out["train_names"] = np.array([int(n) for n, _ in train.object_names], dtype=np.int64)
                grasps = np.full((len(valid), 3), -1, dtype=np.int64)
                for i in range(len(valid)):
                    g_ = valid.get_validation_instance(i)[1]
                    grasps[i, :len(g_)] = g_
                out["valid_grasps"] = grasps
                random.seed(7)                                   # training instances draw from the global python RNG
                draws = [train.get_training_instance(0) for _ in range(6)]
                out["train_draw_names"] = np.array([int(o) for o, _ in draws], dtype=np.int64)
                tg = np.full((6, 3), -1, dtype=np.int64)
                for i, (_, g_) in enumerate(draws):
                    tg[i, :len(g_)] = g_
                out["train_draw_grasps"] = tg
                a.eval, a.val_grasps = True, 2                    # evaluation mode: fixed grasp count (:183-184)
                test = dl.mesh_loader_vision(a, set_type="test")
                out["test_grasps_val2"] = np.array([test.get_validation_instance(i)[1] for i in range(len(test))], dtype=np.int64)
                a.eval, a.val_grasps = False, -1
            np.random.seed(11)                                    # get_points shuffles with the global numpy RNG (:196)
            batch = valid.collate([valid[i] for i in (0, 3, 7)])
            out[f"{tag}_touch_charts"] = batch["touch_charts"].numpy()
            if not finger:
                out["gt_points"] = batch["gt_points"].numpy()
                out["img_sub"] = batch["img"][:, :, ::16, ::16].numpy()
                out["img_sum"] = batch["img"].double().sum(dim=(1, 2, 3)).numpy()
                out["batch_names"] = np.array([int(os.path.basename(n)) for n, _ in batch["names"]], dtype=np.int64)
                out["batch_name_dir"] = np.array([ord(c) for c in os.path.basename(os.path.dirname(batch["names"][0][0]))], dtype=np.int64)
        a = NS(use_touch=False, use_img=False, finger=False, num_grasps=3, number_points=500, eval=False,
               limit_data=False, val_grasps=-1)
        plain = dl.mesh_loader_vision(a, set_type="valid")
        b = plain.collate([plain[0], plain[1]])
        out["plain_img_shape"] = np.array(b["img"].shape, dtype=np.int64)
        out["plain_touch_shape"] = np.array(b["touch_charts"].shape, dtype=np.int64)
        out["plain_touch_value"] = b["touch_charts"].numpy()
    save("g11_loader_batch.npz", **out)


if __name__ == "__main__":
    which = sys.argv[1:] or ["g1", "g2", "g3", "g4", "g5", "g6", "g7", "g8", "g9", "g10", "g11", "g12", "g13", "g14"]
    for w in which:
        globals()[w]()
