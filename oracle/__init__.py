"""oracle/ — CPU restatement of the reference hot path.  TEST INFRASTRUCTURE ONLY.

This package restates, on the CPU, the arithmetic of the mesh-reconstruction hot
path of facebookresearch/Active-3D-Vision-and-Touch (``pterotactyl``):

* ``oracle.mesh``     – OBJ parsing, adjacency build / row-normalisation / touch fusion
                        (reference ``pterotactyl/utility/utils.py:30-36,47-71,75-130,134-148,194-200``)
* ``oracle.gcn``      – ``GCN_layer`` / ``GCN`` / ``Positional_Encoder`` / ``Mask_Encoder`` /
                        ``Deformation`` / ``prepare_mesh``
                        (reference ``pterotactyl/reconstruction/vision/model.py:168-439``)
* ``oracle.chamfer``  – ``batch_sample`` and ``chamfer_distance``
                        (reference ``pterotactyl/utility/utils.py:152-187,204-217``) plus a restatement
                        of the PyTorch3D 0.5.0 semantics those call into (see below)
* ``oracle/chamfer_nn.c`` – plain-C brute-force nearest neighbour used for full-size checks
                        and for the ``cpu_baseline`` leg of ``bench.py``
* ``oracle.ref_shim`` – imports the real reference from ``/root/reference`` under a
                        ``.cuda()`` no-op patch (build container only; it never travels)

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
anything from here — and only as the checker.  The product package
(``active-3d-vision-and-touch_amd`` / ``a3vt_amd``) never imports ``oracle`` and has no CPU fallback.

Pinning status
--------------
* GCN / encoders / Deformation / adjacency / batch_sample (with injected samples) /
  trainer loss scaling: PINNED against the reference itself, imported in the build container
  (``tests/golden/make_golden.py`` → ``tests/golden/*.npz``; ``tests/test_oracle_vs_reference.py``
  re-runs the comparison live whenever ``/root/reference`` is present).
* Chamfer nearest-neighbour arithmetic, face areas, barycentric weights: these live in
  **PyTorch3D 0.5.0** (``README.md:33``; call sites ``utility/utils.py:20-23,164,179,207,212``), which is
  not vendored in the reference and not installable here, and the reference has no tests.
  Their restatement follows PyTorch3D's published semantics (squared-L2 K=1 ``knn_points``,
  ``point_reduction="mean"``, ``batch_reduction=None``; ``areas = 0.5*|(v1-v0)x(v2-v0)|``;
  ``w0=1-sqrt(u), w1=sqrt(u)(1-v), w2=sqrt(u)v``) → **parity unpinned** for that boundary.
  Analytic known-answer tests (``tests/test_oracle_kat.py``) anchor it instead.
"""
