"""active-3d-vision-and-touch_amd — MI355X-native reconstruction hot path of `pterotactyl`
(facebookresearch/Active-3D-Vision-and-Touch): mesh GCN deformation network + sampled-surface Chamfer loss.

The directory name is not an importable identifier; import it as ``a3vt_amd`` (alias package at the repo
root).  Layout:

* ``csrc/``         hand-written HIP kernels for gfx950 + the C ABI (``include/a3vt.h``) → ``liba3vt.so``
* ``lib.py``        ctypes binding / in-tree build (no fallback when the library is missing)
* ``ops.py``        torch.autograd wrappers (one C call per op)
* ``mesh.py``       host-side template logic: OBJ, icosphere, CSR adjacency (+ touch fusion)
* ``pterotactyl/``  mirror of the reference's module API for this path
                    (``reconstruction.vision.model`` / ``.train``, ``utility.utils``)
* ``distributed.py`` one-process-per-GPU data parallelism: flat-bucket gradient all-reduce over RCCL
* ``synthetic.py``  synthetic batches of the benchmark shapes
"""
from . import lib, mesh  # noqa: F401

__version__ = "0.2.0"


def install_as_pterotactyl():
    """Register the mirror modules under the reference's import names
    (``pterotactyl.reconstruction.vision.model`` etc.) so existing callers pick up the HIP path.
    Only names not already importable are installed for parent packages; the listed modules are always
    overridden."""
    import importlib
    import sys
    import types

    mapping = {
        "pterotactyl.reconstruction.vision.model": ".pterotactyl.reconstruction.vision.model",
        "pterotactyl.reconstruction.vision.train": ".pterotactyl.reconstruction.vision.train",
        "pterotactyl.utility.utils": ".pterotactyl.utility.utils",
        # consumers of the same kernels
        "pterotactyl.utility.data_loaders": ".pterotactyl.utility.data_loaders",
        "pterotactyl.reconstruction.autoencoder.model": ".pterotactyl.reconstruction.autoencoder.model",
        "pterotactyl.policies.DDQN.model": ".pterotactyl.policies.DDQN.model",
        "pterotactyl.policies.scoring": ".pterotactyl.policies.scoring",
    }
    for parent in ("pterotactyl", "pterotactyl.reconstruction", "pterotactyl.reconstruction.vision",
                   "pterotactyl.reconstruction.autoencoder", "pterotactyl.policies", "pterotactyl.policies.DDQN",
                   "pterotactyl.utility"):
        if parent not in sys.modules:
            try:
                importlib.import_module(parent)
            except Exception:
                sys.modules[parent] = types.ModuleType(parent)
    for target, rel in mapping.items():
        mod = importlib.import_module(rel, __name__)
        sys.modules[target] = mod
        parent, _, leaf = target.rpartition(".")
        setattr(sys.modules[parent], leaf, mod)
