"""oracle.ref_shim — import the REAL reference on the CPU (TEST INFRASTRUCTURE ONLY, build container only).

``/root/reference`` is read-only and exists only in the build container; nothing here travels to the
GPU box (tests that need it skip when the directory is absent).  The reference hard-codes ``.cuda()``
and imports packages that are not installed, so importing it needs:

1. ``.cuda()`` turned into a no-op (call sites e.g. ``vision/model.py:67,78,181-201,422,427``;
   ``utility/utils.py:102,198-199``),
2. a stand-in for the four PyTorch3D symbols imported at ``utility/utils.py:20-23`` — provided by
   ``oracle.chamfer`` / ``oracle.mesh`` (our restatement of PyTorch3D 0.5.0's published semantics; this is
   why the Chamfer boundary stays "parity unpinned"),
3. empty stubs for ``trimesh``, ``pyrender``, ``submitit``, ``torch.utils.tensorboard``, ``torchvision``.

Use::

    ref = load_reference()          # None when /root/reference is absent
    ref.model, ref.utils, ref.train  # the reference's vision model / utils / trainer modules
"""
import importlib
import os
import sys
import types
from collections import namedtuple

import torch

REFERENCE_ROOT = "/root/reference"
_CACHE = None


def available():
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "pterotactyl"))


def _stub(name, **attrs):
    mod = types.ModuleType(name)
    mod.__dict__.update(attrs)
    sys.modules[name] = mod
    return mod


def _install_pytorch3d_shim():
    from . import chamfer as och
    from . import mesh as omesh

    def cuda_cd(x, y, batch_reduction=None, **kw):
        assert batch_reduction is None
        return och.chamfer_pair(x, y), None

    def mesh_face_areas_normals(verts, faces):
        return och.face_areas(verts, faces), None

    def _rand_barycentric_coords(size1, size2, dtype, device):
        uv = torch.rand(2, size1, size2, dtype=dtype, device=device)
        return och.barycentric(uv[0], uv[1])

    Faces = namedtuple("Faces", ["verts_idx"])

    def load_obj(path):
        v, f = omesh.load_obj(path)
        return torch.from_numpy(v), Faces(torch.from_numpy(f)), None

    def save_obj(*a, **k):
        raise NotImplementedError

    _stub("pytorch3d")
    _stub("pytorch3d.loss", chamfer_distance=cuda_cd)
    _stub("pytorch3d.ops")
    _stub("pytorch3d.ops.mesh_face_areas_normals", mesh_face_areas_normals=mesh_face_areas_normals)
    _stub("pytorch3d.ops.sample_points_from_meshes", _rand_barycentric_coords=_rand_barycentric_coords)
    _stub("pytorch3d.io")
    _stub("pytorch3d.io.obj_io", load_obj=load_obj, save_obj=save_obj)


def load_reference():
    """Import the reference's vision model / utils / trainer under the shim; returns a namespace or None."""
    global _CACHE
    if _CACHE is not None:
        return _CACHE
    if not available():
        return None
    ident = lambda self, *a, **k: self  # noqa: E731
    torch.Tensor.cuda = ident
    torch.nn.Module.cuda = ident
    torch.cuda.manual_seed = lambda *a, **k: None
    _install_pytorch3d_shim()
    for name in ("trimesh", "pyrender", "submitit", "cv2", "pybullet", "rtree", "meshplot"):
        if name not in sys.modules:
            _stub(name)
    _stub("submitit.helpers", Checkpointable=object)
    try:
        importlib.import_module("torch.utils.tensorboard")
    except Exception:
        class SummaryWriter:  # minimal no-op
            def __init__(self, *a, **k):
                pass

            def add_scalars(self, *a, **k):
                pass

        _stub("torch.utils.tensorboard", SummaryWriter=SummaryWriter)
    try:
        importlib.import_module("torchvision")
    except Exception:
        tv = _stub("torchvision")
        mk = lambda *a, **k: None  # noqa: E731  (data_loaders.py:32 only builds a pipeline object)
        tv.transforms = _stub("torchvision.transforms", Compose=mk, Resize=mk, ToTensor=mk)
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    ns = types.SimpleNamespace()
    ns.model = importlib.import_module("pterotactyl.reconstruction.vision.model")
    ns.utils = importlib.import_module("pterotactyl.utility.utils")
    try:
        ns.train = importlib.import_module("pterotactyl.reconstruction.vision.train")
    except Exception as e:  # trainer needs data dirs at import of data_loaders; optional
        ns.train = None
        ns.train_error = e
    ns.objects_dir = os.path.join(REFERENCE_ROOT, "pterotactyl", "objects")
    _CACHE = ns
    return ns
