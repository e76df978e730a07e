"""oracle.gcn — functional CPU restatement of the mesh deformation network (TEST INFRASTRUCTURE ONLY).

Follows ``pterotactyl/reconstruction/vision/model.py``:
``GCN_layer`` (:335-363), ``GCN`` (:290-331), ``Positional_Encoder`` (:367-399), ``Mask_Encoder`` (:403-414),
``Deformation.forward`` (:203-286, the image-free modes) and ``prepare_mesh`` (:418-439).

Everything is a pure function of a ``state`` dict that uses the reference's state-dict key layout
(``positional_encoder.model.{0,2,4}.{weight,bias}``, ``mask_encoder.model.0.weight``,
``mesh_deform_{1,2}.layers.{i}.{weight,bias}``), so the same weights can be fed to the reference
module, to this oracle and to the HIP product.  Works in float32 or float64 (dtype of ``state``).
Backward passes come from torch autograd on these functions (CPU).
"""
import math

import numpy as np
import torch


def cut_length(out_features, cut):
    """model.py:355 — Python banker's rounding, e.g. round(300*0.33)=99, round(100*0.33)=33."""
    return round(out_features * cut)


def adj_matmul(adj, x):
    """``torch.matmul(adj, x)`` for a dense (N,N) adjacency, or the same product from a CSR triple
    ``(rowptr, col, val)`` (long/long/float tensors) — used by the larger tests for speed."""
    if isinstance(adj, (tuple, list)):
        rowptr, col, val = adj
        n = rowptr.numel() - 1
        row = torch.repeat_interleave(torch.arange(n), rowptr[1:] - rowptr[:-1])
        contrib = x[:, col, :] * val.to(x.dtype)[None, :, None]
        out = torch.zeros(x.shape[0], n, x.shape[2], dtype=x.dtype)
        return out.index_add(1, row, contrib)
    return torch.matmul(adj.to(x.dtype), x)


def bf16_round(t):
    """Round to bf16 (nearest even) and back: what the bf16 operand mode of the MFMA kernels does to its inputs."""
    return t.to(torch.bfloat16).to(t.dtype)


def gcn_layer(x, weight, bias, adj, cut=0.33, do_cut=True, relu=True, bf16=False):
    """model.py:351-363.  x (B,N,in), weight (1,in,out), bias (out,).  ``bf16`` emulates the product's reduced-precision
    modes (not a reference feature): True = operand mode (operands of X W rounded to bf16, exact products, wide
    accumulation); "storage" = bf16 storage mode (additionally the raw aggregated channels Z[:, :c] and the layer output
    are rounded to bf16 where the device stores them)."""
    storage = bf16 == "storage"
    z = torch.matmul(bf16_round(x), bf16_round(weight)) if bf16 else torch.matmul(x, weight)
    if do_cut:
        length = cut_length(z.shape[-1], cut)
        za = bf16_round(z[:, :, :length]) if storage else z[:, :, :length]
        agg = adj_matmul(adj, za) + bias[:length]
        out = torch.cat((agg, z[:, :, length:]), dim=-1)
    else:
        out = adj_matmul(adj, z) + bias
    out = torch.relu(out) if relu else out
    return bf16_round(out) if storage else out


def gcn(x, state, prefix, adj, num_layers, cut=0.33, collect=None, bf16=False):
    """model.py:316-331 — ``num_layers`` layers, ReLU on all but the last, last layer aggregates all channels.
    ``bf16``: see :func:`gcn_layer`; the 3-channel output layer always runs in full precision (in "storage" mode on the
    bf16-stored activations, and the stack's input features are rounded once at entry, as the device does)."""
    if bf16 == "storage":
        x = bf16_round(x)
    for i in range(num_layers):
        last = i == num_layers - 1
        x = gcn_layer(x, state[f"{prefix}.layers.{i}.weight"], state[f"{prefix}.layers.{i}.bias"],
                      adj, cut, do_cut=not last, relu=not last, bf16=bf16 if not last else False)
        if collect is not None:
            collect.append(x)
    return x


def nerf_embedding(p):
    """model.py:381-391 — [sin(pi p), cos(pi p), sin(2 pi p), cos(2 pi p), sin(4 pi p), ..., cos(18 pi p)].
    The reference multiplies ``np.pi * 2 * i`` (python float64) into the tensor."""
    emb = []
    for i in range(10):
        f = np.pi if i == 0 else np.pi * 2 * i
        emb.append(torch.sin(f * p))
        emb.append(torch.cos(f * p))
    return torch.cat(emb, dim=-1)


def positional_encoder(positions, state, prefix="positional_encoder"):
    """model.py:393-399."""
    shape = positions.shape
    p = positions.contiguous().view(shape[0] * shape[1], -1)
    h = torch.cat((nerf_embedding(p), p), dim=-1)
    lin = torch.nn.functional.linear
    h = torch.relu(lin(h, state[f"{prefix}.model.0.weight"], state[f"{prefix}.model.0.bias"]))
    h = torch.relu(lin(h, state[f"{prefix}.model.2.weight"], state[f"{prefix}.model.2.bias"]))
    h = lin(h, state[f"{prefix}.model.4.weight"], state[f"{prefix}.model.4.bias"])
    return h.view(shape[0], shape[1], -1)


def mask_encoder(mask, state, prefix="mask_encoder"):
    """model.py:410-414."""
    shape = mask.shape
    idx = mask.contiguous().view(-1).long()
    return state[f"{prefix}.model.0.weight"][idx].view(shape[0], shape[1], -1)


# camera used by Image_Encoder.pooling (model.py:50-67): K.RT with f = 221.7025, c = (128, 128)
CAMERA_RT = [[-7.587616579485257e-08, -1.0000001192092896, 0.0, -2.2762851159541242e-08],
             [-0.7071068286895752, 7.587616579485257e-08, -0.7071068286895752, 0.0],
             [0.7071068286895752, 0.0, -0.7071067690849304, 0.4242640733718872]]


def projection_matrix():
    K = np.array([[221.7025, 0, 128.0], [0, 221.7025, 128.0], [0, 0, 1]])
    return torch.FloatTensor(K.dot(np.array(CAMERA_RT)))


def image_encoder(img, state, prefix, ker=5, num_blocks=6, layers_per_block=3, training=False):
    """Image_Encoder.forward (model.py:147-164) over the layer list built at :36-48: layer 0 is a bare conv, every
    other layer BN -> ReLU -> conv; the first layer of each block has stride 2; all convs pad 1.  Collects the maps of
    the layers ``len-1-(i+1)*layers_per_block`` (i=0..2) and the last map reached before the size drops below ``ker``."""
    F = torch.nn.functional
    n = 1 + num_blocks * layers_per_block
    picks = {n - 1 - (i + 1) * layers_per_block for i in range(3)}
    x, maps = img, []
    for e in range(n):
        if x.shape[-1] < ker:
            break
        if e == 0:
            x = F.conv2d(x, state[f"{prefix}.layers.0.0.weight"], state[f"{prefix}.layers.0.0.bias"], stride=1, padding=1)
        else:
            k = f"{prefix}.layers.{e}"
            x = F.batch_norm(x, state[f"{k}.0.running_mean"], state[f"{k}.0.running_var"], state[f"{k}.0.weight"],
                             state[f"{k}.0.bias"], training, 0.1, 1e-5)
            x = torch.relu(x)
            stride = 2 if (e - 1) % layers_per_block == 0 else 1
            x = F.conv2d(x, state[f"{k}.2.weight"], state[f"{k}.2.bias"], stride=stride, padding=1)
        if e in picks:
            maps.append(x)
    maps.append(x)
    return maps


def image_pooling(maps, verts):
    """Image_Encoder.pooling (model.py:70-103)."""
    ext = torch.cat((verts, torch.ones_like(verts[..., :1])), dim=-1)
    ext = torch.matmul(ext, projection_matrix().to(verts.dtype).permute(1, 0)).clone()
    ext[:, :, 2][ext[:, :, 2] == 0] = 0.1
    xs = ext[:, :, 1] / ext[:, :, 2] / 256.0
    xs[torch.isinf(xs)] = 0.5
    ys = ext[:, :, 0] / ext[:, :, 2] / 256.0
    ys[torch.isinf(ys)] = 0.5
    grid = torch.cat([ys.unsqueeze(2).unsqueeze(3), xs.unsqueeze(2).unsqueeze(3)], 3) * 2 - 1
    feats = [torch.nn.functional.grid_sample(m.to(verts.dtype), grid, align_corners=True) for m in maps]
    return torch.cat(feats, dim=1)[:, :, :, 0].permute(0, 2, 1)


def deformation_forward_img(state, adj_info, charts, img, use_touch, num_layers=20, cut=0.33, cnn=(5, 6, 3),
                            training=False, bf16=False):
    """model.py:203-286 for ``use_img=True``: stage 1 on the vision charts with the vision-only adjacency
    ('origional', :198-200,317-318) and the global encoder's maps; touch charts join in stage 2 (:254-259); stages 2-3
    use the local encoder's maps (pooled with the global encoder's projection, :265,277) and the fused adjacency."""
    vc = charts["vision_charts"].shape[1]
    gmaps = image_encoder(img, state, "img_encoder_global", *cnn, training=training)
    lmaps = image_encoder(img, state, "img_encoder_local", *cnn, training=training)
    vertices = charts["vision_charts"].clone()
    mask = charts["vision_masks"].clone()
    mask_features = mask_encoder(mask, state)
    feats = positional_encoder(vertices, state) + mask_features + image_pooling(gmaps, vertices)
    update = gcn(feats, state, "mesh_deform_1", adj_info["origional"], num_layers, cut, bf16=bf16)
    vertices = vertices + update[:, :vc]
    if use_touch:
        vertices = torch.cat((vertices, charts["touch_charts"].clone()), dim=1)
        mask = torch.cat((charts["vision_masks"].clone(), charts["touch_masks"].clone()), dim=1)
        mask_features = mask_encoder(mask, state)
    for _ in range(2):
        feats = positional_encoder(vertices, state) + mask_features + image_pooling(lmaps, vertices)
        update = gcn(feats, state, "mesh_deform_2", adj_info["adj"], num_layers, cut, bf16=bf16)
        vertices = torch.cat((vertices[:, :vc] + update[:, :vc], vertices[:, vc:]), dim=1)
    return vertices, mask


def deformation_forward(state, adj_info, charts, use_touch, num_layers=20, cut=0.33, num_stages=3, bf16=False):
    """model.py:203-286 for ``use_img=False``.  ``adj_info`` holds 'adj' (dense tensor or CSR triple).
    Returns (vertices (B,N,3), mask (B,N,1)).  ``num_stages`` < 3 truncates after that many
    refinement stages (BASELINE.json configs[0] asks for 1).  ``bf16``: the device's bf16 modes emulated (:func:`gcn_layer`)."""
    vc = charts["vision_charts"].shape[1]
    adj = adj_info["adj"]  # mesh_deform_1 uses 'origional' only when use_img (model.py:198-200,317-320)
    if use_touch:
        vertices = torch.cat((charts["vision_charts"].clone(), charts["touch_charts"].clone()), dim=1)
        mask = torch.cat((charts["vision_masks"].clone(), charts["touch_masks"].clone()), dim=1)
    else:
        vertices = charts["vision_charts"].clone()
        mask = charts["vision_masks"].clone()
    mask_features = mask_encoder(mask, state)
    feats = positional_encoder(vertices, state) + mask_features
    update = gcn(feats, state, "mesh_deform_1", adj, num_layers, cut, bf16=bf16)
    vertices = torch.cat((vertices[:, :vc] + update[:, :vc], vertices[:, vc:]), dim=1)
    for _ in range(1, num_stages):
        # stage 2 re-uses stage-1 mask features (model.py:262); stage 3 recomputes them (:274) — same values.
        feats = positional_encoder(vertices, state) + mask_features
        update = gcn(feats, state, "mesh_deform_2", adj, num_layers, cut, bf16=bf16)
        vertices = torch.cat((vertices[:, :vc] + update[:, :vc], vertices[:, vc:]), dim=1)
    return vertices, mask


def prepare_mesh(touch_charts, vision_mesh, batch_size, use_touch):
    """model.py:418-439 — ``touch_charts`` is the loader tensor (B,G,4,25,4)/(B,G,25,4)."""
    vision_charts = vision_mesh.unsqueeze(0).repeat(batch_size, 1, 1)
    vision_masks = 3 * torch.ones(vision_charts.shape[:-1], dtype=vision_mesh.dtype).unsqueeze(-1)
    charts = {"vision_charts": vision_charts, "vision_masks": vision_masks}
    if use_touch:
        info = touch_charts.view(batch_size, -1, 4)
        charts["touch_charts"] = info[:, :, :3]
        charts["touch_masks"] = info[:, :, 3:]
    return charts


def init_state(input_size=50, hidden=300, num_layers=20, seed=0, dtype=torch.float32):
    """Random weights with the reference's distributions (model.py:345-349 for GCN layers; torch
    defaults for Linear/Embedding).  The draw ORDER here is not the reference constructor's — use
    a module's ``state_dict()`` when identical weights are needed."""
    g = torch.Generator().manual_seed(seed)
    st = {}

    def uni(shape, a):
        return (torch.rand(shape, generator=g, dtype=torch.float64) * 2 * a - a).to(dtype)

    dims = [63, input_size // 4, input_size // 2, input_size]
    for k, li in enumerate((0, 2, 4)):
        bound = 1.0 / math.sqrt(dims[k])
        st[f"positional_encoder.model.{li}.weight"] = uni((dims[k + 1], dims[k]), bound)
        st[f"positional_encoder.model.{li}.bias"] = uni((dims[k + 1],), bound)
    st["mask_encoder.model.0.weight"] = torch.randn(4, input_size, generator=g, dtype=torch.float64).to(dtype)
    hv = [input_size] + [hidden] * (num_layers - 1) + [3]
    for name in ("mesh_deform_1", "mesh_deform_2"):
        for i in range(num_layers):
            stdv = 0.3 * 6.0 / math.sqrt(hv[i] + 1)
            st[f"{name}.layers.{i}.weight"] = uni((1, hv[i], hv[i + 1]), stdv)
            st[f"{name}.layers.{i}.bias"] = uni((hv[i + 1],), 0.1)
    return st
