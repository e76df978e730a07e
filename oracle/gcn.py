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


def gcn_layer(x, weight, bias, adj, cut=0.33, do_cut=True, relu=True):
    """model.py:351-363.  x (B,N,in), weight (1,in,out), bias (out,)."""
    z = torch.matmul(x, weight)
    if do_cut:
        length = cut_length(z.shape[-1], cut)
        agg = adj_matmul(adj, z[:, :, :length]) + bias[:length]
        out = torch.cat((agg, z[:, :, length:]), dim=-1)
    else:
        out = adj_matmul(adj, z) + bias
    return torch.relu(out) if relu else out


def gcn(x, state, prefix, adj, num_layers, cut=0.33, collect=None):
    """model.py:316-331 — ``num_layers`` layers, ReLU on all but the last, last layer aggregates all channels."""
    for i in range(num_layers):
        last = i == num_layers - 1
        x = gcn_layer(x, state[f"{prefix}.layers.{i}.weight"], state[f"{prefix}.layers.{i}.bias"],
                      adj, cut, do_cut=not last, relu=not last)
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


def deformation_forward(state, adj_info, charts, use_touch, num_layers=20, cut=0.33, num_stages=3):
    """model.py:203-286 for ``use_img=False``.  ``adj_info`` holds 'adj' (dense tensor or CSR triple).
    Returns (vertices (B,N,3), mask (B,N,1)).  ``num_stages`` < 3 truncates after that many
    refinement stages (BASELINE.json configs[0] asks for 1)."""
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
    update = gcn(feats, state, "mesh_deform_1", adj, num_layers, cut)
    vertices = torch.cat((vertices[:, :vc] + update[:, :vc], vertices[:, vc:]), dim=1)
    for _ in range(1, num_stages):
        # stage 2 re-uses stage-1 mask features (model.py:262); stage 3 recomputes them (:274) — same values.
        feats = positional_encoder(vertices, state) + mask_features
        update = gcn(feats, state, "mesh_deform_2", adj, num_layers, cut)
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
