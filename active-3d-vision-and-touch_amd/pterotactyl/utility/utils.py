"""Drop-in for the hot-path part of ``pterotactyl/utility/utils.py`` (same names, argument meaning, return types).

Reference symbols mirrored (file:line under the reference tree):
``load_mesh_vision`` :30-36, ``set_seeds`` :39-43, ``normalize_adj`` :47-52, ``adj_init`` :56-71,
``adj_fuse_touch`` :75-130, ``calc_adj`` :134-148, ``batch_sample`` :152-187, ``load_mesh_touch`` :194-200,
``chamfer_distance`` :204-217, ``save_config`` :535-544, ``load_model_config`` :547-553.

Differences that callers can see:
* ``adj_info`` is an :class:`AdjInfo` dict.  It still answers ``['origional']``, ``['adj']`` (dense float tensors,
  materialised lazily — the DDQN graph model reads them, ``policies/DDQN/model.py:68``) and ``['faces']``
  (int64), and additionally carries ``['csr']`` / ``['csr_origional']`` device CSR handles that the HIP GCN uses.
* sampling draws come from Philox4x32-10 seeded from torch's global CPU generator (so ``torch.manual_seed``
  still makes a run reproducible) instead of ``torch.multinomial`` / ``torch.rand`` on the device; pass
  ``samples=`` to inject explicit ``(face_idx, u, v)`` draws (parity tests).
"""
import json
import os
import random
from collections import namedtuple

import numpy as np
import torch

from ... import mesh as _mesh
from ... import ops as _ops


def _device():
    if not torch.cuda.is_available():
        raise RuntimeError("a3vt: no ROCm GPU visible — the MI355X path has no CPU fallback")
    return torch.device("cuda", torch.cuda.current_device())


class AdjInfo(dict):
    """adj_info with lazily materialised dense matrices (reference keys: 'origional', 'adj', 'faces')."""

    _LAZY = {"adj": "csr", "origional": "csr_origional"}

    def __missing__(self, key):
        src = self._LAZY.get(key)
        if src is None or not dict.__contains__(self, src):
            raise KeyError(key)
        csr = dict.__getitem__(self, src)
        dense = torch.from_numpy(csr.host.to_dense()).to(csr.device)
        self[key] = dense
        return dense

    def __contains__(self, key):
        return dict.__contains__(self, key) or (key in self._LAZY and dict.__contains__(self, self._LAZY[key]))


def set_seeds(seed):
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed(seed)
    random.seed(seed)


def load_mesh_touch(obj):
    """OBJ -> (verts float32 (V,3), faces int64 (F,3)) on the GPU.  ``obj`` may also be one of the packaged
    asset names 'vision_charts' / 'touch_chart'."""
    if obj in ("vision_charts", "touch_chart"):
        v, f = _mesh.load_asset(obj)
    else:
        v, f = _mesh.load_obj(obj)
    dev = _device()
    return torch.from_numpy(v).to(dev), torch.from_numpy(f).to(dev)


def calc_adj(faces):
    """Dense binary adjacency with self loops (API compatibility; the hot path uses CSR)."""
    f = faces.detach().cpu().numpy()
    n = int(f.max()) + 1
    r, c = _mesh.vision_pairs(f, n)
    adj = torch.zeros((n, n), dtype=torch.float32)
    adj[torch.from_numpy(r), torch.from_numpy(c)] = 1
    return adj.to(faces.device)


def normalize_adj(mx):
    rowsum = mx.sum(1)
    r_inv = (1.0 / rowsum).view(-1)
    r_inv[r_inv != r_inv] = 0.0
    return r_inv[:, None] * mx


def adj_fuse_touch(verts, faces, adj, args):
    """Dense fused adjacency + faces (API compatibility).  ``adj`` is ignored: the pattern is rebuilt from faces."""
    sv, sf = _mesh.load_asset("touch_chart")
    r, c, n, all_faces = _mesh.fused_pairs(verts.detach().cpu().numpy(), faces.detach().cpu().numpy(), sf,
                                           args.num_grasps, args.finger, sv.shape[0])
    out = torch.zeros((n, n), dtype=torch.float32)
    out[torch.from_numpy(r), torch.from_numpy(c)] = 1
    return out.to(verts.device), torch.from_numpy(all_faces).to(verts.device)


def adj_init(verts, faces, args):
    """CSR adjacency info for a template.  verts (V,3) / faces (F,3) tensors on any device."""
    dev = verts.device if verts.is_cuda else _device()
    v = verts.detach().cpu().numpy().astype(np.float32)
    f = faces.detach().cpu().numpy().astype(np.int64)
    nv = int(f.max()) + 1
    info = AdjInfo()
    orig = _mesh.CSRAdjacency.from_pairs(*_mesh.vision_pairs(f, nv), nv)
    info["csr_origional"] = _ops.DeviceCSR(orig, dev)
    if getattr(args, "use_touch", False):
        sv, sf = _mesh.load_asset("touch_chart")
        r, c, n, all_faces = _mesh.fused_pairs(v, f, sf, args.num_grasps, args.finger, sv.shape[0])
        info["csr"] = _ops.DeviceCSR(_mesh.CSRAdjacency.from_pairs(r, c, n), dev)
        f = all_faces
    else:
        info["csr"] = info["csr_origional"]
    info["faces"] = torch.from_numpy(f).to(dev)
    info["faces_i32"] = info["faces"].to(torch.int32).contiguous()
    return info


def load_mesh_vision(args, obj):
    verts, faces = load_mesh_touch(obj)
    return adj_init(verts, faces, args), verts


def _faces_i32(faces):
    return faces if faces.dtype == torch.int32 else faces.to(torch.int32)


def _philox_seed():
    # one draw from torch's global CPU generator: reproducible under torch.manual_seed, no device sync
    return int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64).item())


def batch_sample(verts, faces, num=10000, draws=None, samples=None):
    """Area-weighted surface samples.  Returns (B,num,3) like the reference, or (draws,B,num,3) when ``draws`` is given."""
    f32 = _faces_i32(faces).contiguous()
    d = 1 if draws is None else draws
    if samples is not None:
        fi, u, v = samples
        pts = _ops.SamplePointsFn.apply(verts, f32, num, d, 0, 0, fi.to(torch.int32), u, v)
    else:
        pts = _ops.SamplePointsFn.apply(verts, f32, num, d, _philox_seed(), 0, None, None, None)
    return pts[0] if draws is None else pts


def chamfer_distance(verts, faces, gt_points, num=1000, repeat=3, samples=None):
    """(B,) Chamfer distance between ``repeat`` surface samplings of the meshes and ``gt_points`` (B,Q,3).
    ``samples``: optional (face_idx, u, v) tensors shaped (repeat,B,num) to inject the draws."""
    pred = batch_sample(verts, faces, num=num, draws=repeat, samples=samples)
    return _ops.ChamferFn.apply(pred, gt_points.contiguous())


def save_config(location, args):
    abs_path = os.path.abspath(location)
    args = vars(args)
    args["check_point"] = abs_path
    config_location = f"{location}/config.json"
    with open(config_location, "w") as fp:
        json.dump(args, fp, indent=4)
    return config_location


def load_model_config(location):
    config_location = f"{location}/config.json"
    with open(config_location) as json_file:
        data = json.load(json_file)
    weight_location = data["check_point"] + "/model"
    args = namedtuple("ObjectName", data.keys())(*data.values())
    return args, weight_location
