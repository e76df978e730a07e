"""CPU: host-side logic of the product (no GPU, no compute calls into the library)."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

import a3vt_amd
from a3vt_amd import lib, mesh as amesh

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_icosphere_sizes_and_topology():
    for level, nv, nf, nnz in ((0, 12, 20, 72), (4, 2562, 5120, 17922), (5, 10242, 20480, 71682)):
        v, f = amesh.icosphere(level)
        assert v.shape == (nv, 3) and f.shape == (nf, 3)
        assert np.allclose(np.linalg.norm(v, axis=1), 0.25, atol=1e-6)
        A = amesh.CSRAdjacency.from_pairs(*amesh.vision_pairs(f, nv), nv)
        assert A.nnz == nnz                                     # nnz incl. self loops
        deg = np.diff(A.rowptr)
        assert deg.min() == 6 and deg.max() == (6 if level == 0 else 7) and (deg == 6).sum() == 12   # 12 five-valent verts


def test_csr_rows_sum_to_one_and_transpose():
    v, f = amesh.load_asset("vision_charts")
    sv, sf = amesh.load_asset("touch_chart")
    r, c, n, faces = amesh.fused_pairs(v, f, sf, 5, False)
    A = amesh.CSRAdjacency.from_pairs(r, c, n)
    assert n == 2324 and A.nnz == 60726 and faces.shape == (2944, 3)
    rows = np.repeat(np.arange(n), np.diff(A.rowptr))
    assert np.allclose(np.bincount(rows, weights=A.val, minlength=n), 1.0, atol=1e-6)
    assert np.diff(A.rowptr).max() == 1153                      # chart-centre hub rows
    T = amesh.CSRAdjacency(A.t_rowptr, A.t_col, A.t_val, n)
    assert np.array_equal(T.to_dense(), A.to_dense().T)
    assert np.array_equal(amesh.CSRAdjacency.from_dense(A.to_dense()).col, A.col)


@pytest.mark.parametrize("grasps,finger", [(5, False), (1, False), (5, True), (1, True)])
def test_fused_adjacency_splits_into_local_pattern_plus_bipartite_block(grasps, finger):
    """Round 6: utils.py:119-128 links every seam vertex to every chart centre, so the fused matrix is D^-1 (P + J) with J
    complete bipartite.  The split is found from the CSR alone, reproduces the matrix exactly, and the library's host-side
    validator (a3vt_adj_split_validate) proves it — and refutes every corruption of it."""
    v, f = amesh.load_asset("vision_charts")
    sv, sf = amesh.load_asset("touch_chart")
    r, c, n, _ = amesh.fused_pairs(v, f, sf, grasps, finger)
    A = amesh.CSRAdjacency.from_pairs(r, c, n)
    sp = A.split()
    k = (1 if finger else 4) * grasps
    assert sp is not None and sp.n_centre == k and sp.max_degree == 10
    assert sp.n_seam == (1146 if k > 1 else 1152)       # one centre: its chart ring is linked to "all centres" too
    assert np.array_equal(sp.to_dense(), A.to_dense())
    assert A.nnz == len(sp.col) + 2 * sp.n_seam * sp.n_centre
    # the same split from a dense matrix as the reference's adj_info holds it
    sp2 = amesh.CSRAdjacency.from_dense(A.to_dense()).split()
    assert np.array_equal(sp2.col, sp.col) and np.array_equal(sp2.cls, sp.cls)
    if not os.path.exists(lib.LIB_PATH):
        pytest.skip("library not built")
    L = lib.load()

    def check(rowptr=sp.rowptr, col=sp.col, scale=sp.scale, cls=sp.cls):
        arrs = [np.ascontiguousarray(a) for a in (A.rowptr, A.col, A.val, rowptr, col, scale, cls)]
        return L.a3vt_adj_split_validate(arrs[0].ctypes.data, arrs[1].ctypes.data, arrs[2].ctypes.data, n,
                                         arrs[3].ctypes.data, arrs[4].ctypes.data, arrs[5].ctypes.data, arrs[6].ctypes.data)
    assert check() == 0
    bad = sp.cls.copy()
    bad[np.flatnonzero(sp.cls == 1)[3]] = 0                    # a seam vertex dropped from S
    assert check(cls=bad) != 0 and b"adj_split" in L.a3vt_last_error()
    bad = sp.scale.copy()
    bad[7] = np.nextafter(bad[7], np.float32(1))               # one ulp off
    assert check(scale=bad) != 0
    bad = sp.col.copy()
    bad[sp.rowptr[100]] = (bad[sp.rowptr[100]] + 1) % n         # a wrong column in P
    assert check(col=bad) != 0


def test_split_is_refused_for_matrices_of_another_form():
    v, f = amesh.load_asset("vision_charts")
    A = amesh.CSRAdjacency.from_pairs(*amesh.vision_pairs(f, len(v)), len(v))
    sp = A.split()
    assert sp is not None and sp.n_seam == 0 and sp.n_centre == 0 and np.array_equal(sp.to_dense(), A.to_dense())
    B = amesh.CSRAdjacency(A.rowptr, A.col, (A.val * np.float32(0.5)).astype(np.float32), A.n)   # not D^-1 x pattern
    assert B.split() is None
    rows = np.repeat(np.arange(A.n), np.diff(A.rowptr))
    keep = ~((rows == 0) & (A.col == A.col[A.rowptr[0] + 1]))                                   # one direction of an edge removed
    rp = np.zeros(A.n + 1, dtype=np.int32)
    rp[1:] = np.cumsum(np.bincount(rows[keep], minlength=A.n))
    deg = np.diff(rp)
    C = amesh.CSRAdjacency(rp, A.col[keep], (1.0 / deg[rows[keep]]).astype(np.float32), A.n)
    assert C.split() is None                                                                    # pattern not symmetric


def test_obj_roundtrip(tmp_path):
    p = tmp_path / "t.obj"
    p.write_text("v 0 0 0\nv 1 0 0\nv 0 1 0\nv 1 1 0\nvt 0 0\nf 1/1/1 2/1/1 3/1/1\nf 2 4 3\nf 1 2 4 3\n")
    v, f = amesh.load_obj(str(p))
    assert v.shape == (4, 3) and f.tolist() == [[0, 1, 2], [1, 3, 2], [0, 1, 3], [0, 3, 2]]


def test_header_symbols_all_bound_and_exported():
    """Every function declared in include/a3vt.h has a ctypes signature and is exported by liba3vt.so."""
    hdr = open(os.path.join(ROOT, "include", "a3vt.h")).read()
    declared = set(re.findall(r"\b(a3vt_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(lib.SIGNATURES), declared ^ set(lib.SIGNATURES)
    if not os.path.exists(lib.LIB_PATH):
        pytest.skip("library not built (run __graft_entry__.build())")
    dll = ctypes.CDLL(lib.LIB_PATH)
    for name in declared:
        assert hasattr(dll, name), name
    L = lib.load()
    assert L.a3vt_version() == 163
    assert L.a3vt_posenc_param_count(50) == 12 * 63 + 12 + 25 * 12 + 25 + 50 * 25 + 50 + 200
    assert L.a3vt_wt_rows(300) >= 304 and L.a3vt_wt_ld(300) == 304
    # host-only entry point: CSR validation
    rp = np.array([0, 1, 3], dtype=np.int32)
    col = np.array([0, 0, 1], dtype=np.int32)
    assert L.a3vt_csr_validate(rp.ctypes.data, col.ctypes.data, 2, 3) == 0
    col[2] = 7
    assert L.a3vt_csr_validate(rp.ctypes.data, col.ctypes.data, 2, 3) != 0
    assert b"out of range" in L.a3vt_last_error()


def test_no_cpu_fallback():
    """The product fails loudly without a GPU / without the native library."""
    from a3vt_amd import ops
    with pytest.raises(RuntimeError, match="GPU"):
        ops.ChamferFn.apply(torch.zeros(1, 1, 4, 3), torch.zeros(1, 4, 3))
    import importlib
    src = open(os.path.join(ROOT, "active-3d-vision-and-touch_amd", "ops.py")).read() + \
        open(os.path.join(ROOT, "active-3d-vision-and-touch_amd", "lib.py")).read()
    assert "oracle" not in src.replace("no CPU", "")          # the product never imports the checker
    assert importlib.util.find_spec("a3vt_amd.oracle") is None


def test_state_dict_layout_matches_reference():
    from helpers import make_args
    from a3vt_amd.pterotactyl.reconstruction.vision import model
    net = model.Deformation({}, torch.zeros(4, 3), make_args())
    sd = net.state_dict()
    assert sd["positional_encoder.model.0.weight"].shape == (12, 63)
    assert sd["positional_encoder.model.4.bias"].shape == (50,)
    assert sd["mask_encoder.model.0.weight"].shape == (4, 50)
    assert sd["mesh_deform_1.layers.0.weight"].shape == (1, 50, 300)
    assert sd["mesh_deform_2.layers.19.weight"].shape == (1, 300, 3) and sd["mesh_deform_2.layers.19.bias"].shape == (3,)
    # image model (47 349 642 parameters; encoder keys layers.<k>.{0|2}.*)
    net = model.Deformation({}, torch.zeros(4, 3), make_args(use_img=True, CNN_ker_size=5, num_CNN_blocks=6,
                                                              layers_per_block=3))
    assert sum(p.numel() for p in net.parameters()) == 47349642
    sd = net.state_dict()
    assert sd["mesh_deform_1.layers.0.weight"].shape == (1, 448, 300)
    assert "img_encoder_global.layers.0.0.weight" in sd and "img_encoder_local.layers.1.2.weight" in sd


def test_consumer_modules_reproduce_reference_init():
    """Auto-encoder and DDQN graph model: same constructor order as the reference, so seed 0 gives
    bit-identical weights (SHA-256 recorded by tests/golden/make_golden.py from the real reference)."""
    from golden_util import load, state_sha256
    from helpers import make_args
    from a3vt_amd.pterotactyl.policies.DDQN import model as dm
    from a3vt_amd.pterotactyl.reconstruction.autoencoder import model as am
    torch.manual_seed(0)
    net = am.AutoEncoder({}, torch.zeros(4, 3), make_args(num_GCN_layers=3, hidden_GCN_size=300, encoding_size=200))
    assert np.array_equal(state_sha256(net.state_dict()), load("g9_autoencoder.npz")["weight_sha256"])
    assert am.GridSamplingLayer(2, [[-0.5, 0.5, 80], [-0.5, 0.5, 80]]).shape == (2, 6400, 2)
    torch.manual_seed(0)
    net = dm.Graph_Model(make_args(layers=3, hidden_dim=200, num_actions=50), {"adj": torch.eye(4)})
    assert np.array_equal(state_sha256(net.state_dict()), load("g10_graph_model.npz")["weight_sha256"])


def test_config_roundtrip(tmp_path):
    from argparse import Namespace
    from a3vt_amd.pterotactyl.utility import utils
    a = Namespace(lr=3e-4, num_GCN_layers=20, exp_id="x")
    utils.save_config(str(tmp_path), a)
    cfg, w = utils.load_model_config(str(tmp_path))
    assert cfg.lr == 3e-4 and cfg.num_GCN_layers == 20 and w == str(tmp_path) + "/model"


def test_synthetic_batches_have_loader_format():
    from helpers import make_args
    from a3vt_amd.synthetic import SyntheticLoader
    a = make_args(use_touch=True, num_grasps=2, number_points=64)
    b = next(iter(SyntheticLoader(a, 2, 3)))
    assert b["gt_points"].shape == (3, 64, 3) and b["touch_charts"].shape == (3, 2, 4, 25, 4) and len(b["names"]) == 3
    assert a3vt_amd.__version__


def test_shard_range_partitions_batch():
    from a3vt_amd.distributed import shard_range
    for gb, w in ((512, 8), (10, 3), (7, 8)):
        spans = [shard_range(gb, r, w) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == gb and all(a[1] == b[0] for a, b in zip(spans, spans[1:]))


def _write_dataset(root, ids, rng):
    """A miniature dataset in the reference's on-disk layout (utility/data_loaders.py:18-29)."""
    for sub in ("point_cloud_info", "images_colourful", "touch_charts", "object_info"):
        os.makedirs(os.path.join(root, sub), exist_ok=True)
    for i in ids:
        np.save(os.path.join(root, "point_cloud_info", f"{i}.npy"), (0.1 * rng.standard_normal((3000, 3))).astype(np.float64))
        np.save(os.path.join(root, "images_colourful", f"{i}.npy"), rng.integers(0, 256, (256, 256, 3), dtype=np.uint8))
        os.makedirs(os.path.join(root, "touch_charts", str(i)), exist_ok=True)
        tc = rng.standard_normal((50, 4, 25, 4)).astype(np.float32)
        tc[..., 3] = rng.integers(0, 3, (50, 4, 1))
        np.save(os.path.join(root, "touch_charts", str(i), "touch_charts.npy"), tc.reshape(50, 4, 100))
    np.save(os.path.join(root, "data_split.npy"),
            {"recon_train": [str(i) for i in ids[:4]], "valid": [str(i) for i in ids[4:]], "test": [], "auto_train": []})


@pytest.mark.parametrize("finger", [False, True])
def test_dataset_wire_format(tmp_path, finger):
    from helpers import make_args
    from a3vt_amd.pterotactyl.reconstruction.vision import model
    from a3vt_amd.pterotactyl.utility import data_loaders
    _write_dataset(str(tmp_path), list(range(6)), np.random.default_rng(0))
    args = make_args(use_touch=True, use_img=True, finger=finger, num_grasps=3, number_points=500, eval=False,
                     data_root=str(tmp_path), limit_data=False, val_grasps=-1)
    train = data_loaders.mesh_loader_vision(args, set_type="recon_train")
    valid = data_loaders.mesh_loader_vision(args, set_type="valid")
    assert len(train) == 4 and len(valid) == 2 * 5            # validation objects x 5 grasp subsets (:160-170)
    batch = valid.collate([valid[i] for i in range(3)])
    assert batch["gt_points"].shape == (3, 500, 3) and batch["gt_points"].dtype == torch.float32
    assert batch["img"].shape == (3, 3, 256, 256) and 0.0 <= batch["img"].min() and batch["img"].max() <= 1.0
    assert batch["touch_charts"].shape == ((3, 3, 25, 4) if finger else (3, 3, 4, 25, 4))
    name, grasps = batch["names"][0]
    oid = os.path.basename(name)                              # glob order is file-system dependent: object 4 or 5
    assert os.path.dirname(name).endswith("object_info") and oid in ("4", "5") and len(grasps) <= 3
    # validation instances are deterministic (seeded by position), unused grasp slots are all-zero charts
    again = valid[0]
    assert again["names"][1] == valid[0]["names"][1]
    assert torch.equal(batch["touch_charts"][0][len(grasps):], torch.zeros_like(batch["touch_charts"][0][len(grasps):]))
    raw = np.load(os.path.join(str(tmp_path), "touch_charts", oid, "touch_charts.npy")).reshape(50, 4, 25, 4)
    if grasps:
        want = raw[grasps[0]][1] if finger else raw[grasps[0]]
        assert np.array_equal(batch["touch_charts"][0][0].numpy(), want)
    # the batch feeds prepare_mesh exactly as the synthetic loader's does
    charts = model.prepare_mesh(batch, torch.zeros(7, 3), args)
    assert charts["touch_charts"].shape == (3, 3 * (1 if finger else 4) * 25, 3) and charts["vision_charts"].shape == (3, 7, 3)
    args.use_img = args.use_touch = False
    plain = data_loaders.mesh_loader_vision(args, set_type="recon_train")
    b = plain.collate([plain[0], plain[1]])
    assert b["img"].shape == (2, 1) and b["touch_charts"].shape == (2, 1)


def test_dataset_loader_matches_reference_fixture(tmp_path, monkeypatch):
    """the mirror's ``mesh_loader_vision`` against fixture g11 — what the REFERENCE's class
    (utility/data_loaders.py:132-258) yields on the same miniature dataset (tests/golden/make_golden.py::g11): instance
    list and position-derived seeds, seeded validation grasp subsets, training draws from the global python RNG, the
    ``val_grasps`` evaluation mode, and one collated batch bit for bit (touch charts, shuffled ground-truth points, image
    scaling and layout)."""
    import random
    from glob import glob as _glob
    from golden_util import load, write_mini_dataset
    from helpers import make_args
    from a3vt_amd.pterotactyl.utility import data_loaders
    z = load("g11_loader_batch.npz")
    root = str(tmp_path)
    write_mini_dataset(root, n=6, seed=0)
    monkeypatch.setattr(data_loaders, "glob", lambda pat: sorted(_glob(pat)))   # as the fixture: file-system independent order
    for tag, finger in (("full", False), ("finger", True)):
        args = make_args(use_touch=True, use_img=True, finger=finger, num_grasps=3, number_points=500, eval=False,
                         data_root=root, limit_data=False, val_grasps=-1)
        valid = data_loaders.mesh_loader_vision(args, set_type="valid")
        train = data_loaders.mesh_loader_vision(args, set_type="recon_train")
        if not finger:
            assert [int(n) for n, _ in valid.object_names] == z["valid_names"].tolist()
            assert [s for _, s in valid.object_names] == z["valid_seeds"].tolist()
            assert [int(n) for n, _ in train.object_names] == z["train_names"].tolist()
            for i in range(len(valid)):
                want = [g for g in z["valid_grasps"][i].tolist() if g >= 0]
                assert valid.get_validation_instance(i)[1] == want, i
            random.seed(7)
            for i in range(6):
                obj, grasps = train.get_training_instance(0)
                assert int(obj) == int(z["train_draw_names"][i])
                assert grasps == [g for g in z["train_draw_grasps"][i].tolist() if g >= 0]
            args.eval, args.val_grasps = True, 2
            test = data_loaders.mesh_loader_vision(args, set_type="test")
            assert [test.get_validation_instance(i)[1] for i in range(len(test))] == z["test_grasps_val2"].tolist()
            args.eval, args.val_grasps = False, -1
        np.random.seed(11)
        batch = valid.collate([valid[i] for i in (0, 3, 7)])
        assert np.array_equal(batch["touch_charts"].numpy(), z[f"{tag}_touch_charts"])
        if not finger:
            assert np.array_equal(batch["gt_points"].numpy(), z["gt_points"])
            assert np.array_equal(batch["img"][:, :, ::16, ::16].numpy(), z["img_sub"])
            assert np.allclose(batch["img"].double().sum(dim=(1, 2, 3)).numpy(), z["img_sum"], rtol=1e-12)
            assert [int(os.path.basename(n)) for n, _ in batch["names"]] == z["batch_names"].tolist()
            assert os.path.basename(os.path.dirname(batch["names"][0][0])) == "".join(chr(c) for c in z["batch_name_dir"])
    args = make_args(use_touch=False, use_img=False, finger=False, num_grasps=3, number_points=500, eval=False,
                     data_root=root, limit_data=False, val_grasps=-1)
    plain = data_loaders.mesh_loader_vision(args, set_type="valid")
    b = plain.collate([plain[0], plain[1]])
    assert list(b["img"].shape) == z["plain_img_shape"].tolist()
    assert list(b["touch_charts"].shape) == z["plain_touch_shape"].tolist()
    assert np.array_equal(b["touch_charts"].numpy(), z["plain_touch_value"])


def test_install_as_pterotactyl_registers_the_mirror():
    """existing callers keep their imports; the mirror modules answer under the reference's names."""
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r); import a3vt_amd; a3vt_amd.install_as_pterotactyl();"
            "import pterotactyl.reconstruction.vision.model as m, pterotactyl.reconstruction.vision.train as t;"
            "import pterotactyl.utility.utils as u, pterotactyl.utility.data_loaders as d;"
            "import pterotactyl.reconstruction.autoencoder.model as a, pterotactyl.policies.DDQN.model as q;"
            "import pterotactyl.policies.scoring as s;"
            "assert m.Deformation.__module__.startswith('a3vt_amd') and hasattr(u, 'chamfer_distance');"
            "assert hasattr(t, 'Engine') and hasattr(d, 'mesh_loader_vision') and hasattr(a, 'AutoEncoder');"
            "assert hasattr(q, 'Graph_Model') and hasattr(s, 'score_actions'); print('ok')") % ROOT
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stderr[-2000:]


def test_public_header_is_plain_c(tmp_path):
    """include/a3vt.h is the drop-in boundary: it must compile as C99 and as C++ with nothing but <stddef.h>/<stdint.h>."""
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    src = tmp_path / "use.c"
    src.write_text('#include "a3vt.h"\nint probe(void) { return a3vt_version() > 0 ? 0 : 1; }\n')
    inc = os.path.join(ROOT, "include")
    for cmd in (["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-I", inc, "-c", str(src), "-o", str(tmp_path / "c.o")],
                ["g++", "-std=c++17", "-Wall", "-Werror", "-x", "c++", "-I", inc, "-c", str(src), "-o", str(tmp_path / "cc.o")]):
        out = subprocess.run(cmd, capture_output=True, text=True)
        assert out.returncode == 0, out.stderr


def test_pretrained_location_follows_the_reference_rule(tmp_path, monkeypatch):
    """vision/train.py:218-241 — eval + pretrained picks one of four directories by (use_img, finger); here the root comes
    from args.pretrained_root / $PTEROTACTYL_PRETRAINED, and a missing model is a FileNotFoundError that names the place."""
    import json
    from types import SimpleNamespace
    from a3vt_amd.pterotactyl.reconstruction.vision import train
    want = {(True, True): "v_t_p", (True, False): "v_t_g", (False, True): "t_p", (False, False): "t_g"}
    for (use_img, finger), sub in want.items():
        a = SimpleNamespace(use_img=use_img, finger=finger, pretrained_root=str(tmp_path))
        assert train.pretrained_location(a) == os.path.join(str(tmp_path), "reconstruction", "vision", sub) + os.sep
    monkeypatch.setenv("PTEROTACTYL_PRETRAINED", "/somewhere")
    assert train.pretrained_location(SimpleNamespace(use_img=False, finger=False)) == "/somewhere/reconstruction/vision/t_g/"
    assert train.pretrained_location(SimpleNamespace(use_img=True, finger=True, pretrained_location="/x/y")) == "/x/y"
    # Engine.load() on a fake tree: the config is read from the chosen directory, the weights file is looked for there
    from a3vt_amd.synthetic import make_args
    os.chdir(tmp_path)
    loc = tmp_path / "reconstruction" / "vision" / "t_g"
    loc.mkdir(parents=True)
    args = make_args(exp_type="t", exp_id="pre", eval=True, pretrained=True, use_img=False, finger=False,
                     pretrained_root=str(tmp_path), num_GCN_layers=3, hidden_GCN_size=16)
    eng = train.Engine(args, loaders=((), ()))
    with pytest.raises(FileNotFoundError, match="t_g"):
        eng.load()
    (loc / "config.json").write_text(json.dumps({k: v for k, v in vars(args).items() if isinstance(v, (int, float, str, bool))}
                                                | {"check_point": str(loc)}))
    (loc / "model").write_bytes(b"")
    with pytest.raises(Exception) as e:   # past the location logic: fails only because there is no GPU / the file is not a checkpoint
        eng.load()
    assert not isinstance(e.value, FileNotFoundError)


def test_flat_bucket_early_countdown_learns_unused_parameters():
    """an early parameter that never receives a gradient must not keep the early chunk from starting — the
    countdown counts the early parameters that had a gradient in the previous step; a gradient that arrives AFTER the early
    chunk was gathered (the set grew between steps) is an error, not a silent loss."""
    import torch
    from a3vt_amd import distributed as adist
    torch.manual_seed(0)
    used, unused, late, head = (torch.nn.Linear(4, 4) for _ in range(4))
    params = [*head.parameters(), *used.parameters(), *unused.parameters(), *late.parameters()]
    bucket = adist.FlatGradBucket(params, early=[*used.parameters(), *unused.parameters(), *late.parameters()], sinks=False)
    x = torch.randn(3, 4)

    def step(with_late):
        bucket.zero()
        y = head(used(x))
        if with_late:
            y = y + late(x)
        y.sum().backward()
        done = bucket._early_done
        bucket.all_reduce_mean()
        return done

    assert step(False) is False and bucket._early_live == 2            # learns: two of the six early tensors are live
    assert step(False) is True and bucket.early_started_in_backward == 1
    assert all(p.grad.data_ptr() == v.data_ptr() for p, v in zip(bucket.params, bucket.views))
    assert unused.weight.grad.abs().max().item() == 0.0
    with pytest.raises(RuntimeError, match="changed since the previous step"):
        step(True)                                                      # `late` fires after the countdown reached zero
    bucket.gather()                                                     # (the caller's way out: gather, or repeat the step)
    assert bucket._early_live == 4                                      # relearned from the step that failed
    assert step(True) is True and step(True) is True
    bucket.close()
    assert not bucket._hooks


def test_bf16_weight_copies_are_reused_until_the_weight_changes():
    """ops._bf16_copy (the convolution weights of the channels-last bf16 image branch): one cast per weight VERSION and
    optimizer epoch — the forward-only loops reuse it; an optimizer step (any torch optimizer of the process: a global step
    hook, because the fused kernels do not move ``_version``) or an in-place update (load_state_dict) invalidates it."""
    from a3vt_amd import ops
    w = torch.nn.Parameter(torch.randn(8, 3, 5, 5))
    a = ops._bf16_copy(w, True)
    assert a.dtype == torch.bfloat16 and a.is_contiguous(memory_format=torch.channels_last)
    assert ops._bf16_copy(w, True) is a                                   # unchanged weight: the same copy
    opt = torch.optim.SGD([w], lr=0.5)
    w.grad = torch.ones_like(w)
    opt.step()
    b = ops._bf16_copy(w, True)
    assert b is not a and torch.equal(b.float(), w.detach().to(torch.bfloat16).float())
    with torch.no_grad():
        w.copy_(torch.zeros_like(w))                                      # load_state_dict path
    assert ops._bf16_copy(w, True).abs().max().item() == 0.0
    w2 = torch.nn.Parameter(w.detach().clone())                           # another tensor never aliases the entry
    assert ops._bf16_copy(w2, True) is not ops._bf16_copy(w, True)
    # a write through .data moves neither _version nor the optimizer epoch — such writers call
    # invalidate_bf16_copies() (distributed.broadcast_parameters and GCN_layer.reset_parameters do)
    c = ops._bf16_copy(w, True)
    w.data.fill_(2.0)
    assert ops._bf16_copy(w, True) is c                                   # (documented: stale until told)
    ops.invalidate_bf16_copies()
    assert ops._bf16_copy(w, True).float().min().item() == 2.0
    # an entry dies with its tensor: rebuilt models do not pin their bf16 copies
    n0 = len(ops._BF16_COPIES)
    tmp = [torch.nn.Parameter(torch.randn(4, 3, 5, 5)) for _ in range(10)]
    for t in tmp:
        ops._bf16_copy(t, True)
    assert len(ops._BF16_COPIES) == n0 + 10
    del tmp, t
    import gc
    gc.collect()
    assert len(ops._BF16_COPIES) == n0


def test_optimizer_hook_is_registered_on_first_use_not_at_import():
    """Verdict r04 weak #8: importing the package must not install a process-wide optimizer hook in the host process."""
    import subprocess
    import sys
    code = ("import torch, torch.optim.optimizer as o, a3vt_amd.ops as ops\n"
            "n0 = len(o._global_optimizer_post_hooks)\n"
            "assert ops._BF16_HOOK[0] is None\n"
            "ops._bf16_copy(torch.nn.Parameter(torch.zeros(2, 2, 1, 1)), True)\n"
            "assert ops._BF16_HOOK[0] is True and len(o._global_optimizer_post_hooks) == n0 + 1\n"
            "ops._bf16_copy(torch.nn.Parameter(torch.zeros(2, 2, 1, 1)), True)\n"
            "assert len(o._global_optimizer_post_hooks) == n0 + 1\n")
    res = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert res.returncode == 0, res.stderr[-1500:]


def test_bench_refuses_more_gpus_than_visible_before_launching_anything():
    """``python bench.py --gpus N`` with no launcher: the parent counts the devices (without initialising a GPU) and refuses
    an N the node cannot serve — non-zero exit code, a message that names the count, nothing spawned, no hang.  (With enough
    devices it starts ``torch.distributed.run`` as a child process: ``test_bench_under_torchrun_single_rank``, ``-m gpu``.)"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    n = torch.cuda.device_count() + 1
    res = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(max(n, 2)), "--steps", "1"],
                         capture_output=True, text=True, timeout=300, env=env)
    assert res.returncode != 0 and "visible" in res.stderr and "torch.distributed.run" not in res.stderr


def test_named_config_inputs_and_roofline_bytes():
    """Round 5: the stand-ins bench.py times for BASELINE configs[3] / configs[4].  (i) The algorithmic activation bytes are
    SURVEY 8d's formula — it reproduces BASELINE.md section 3's figures for cfg-2 / cfg-4 / cfg-5.  (ii) The touch charts of the
    configs[3] stand-in lie ON the ground-truth surface, are the packaged 1.7 cm chart rigidly placed, and carry the
    'touched' token."""
    from a3vt_amd.synthetic import gcn_activation_bytes, gt_cloud, surface_touch_charts
    assert abs(gcn_activation_bytes(64, 2562, 50, 300, 20, 3, 4) / 1e9 - 67.6) < 0.05
    assert abs(gcn_activation_bytes(64, 2662, 50, 300, 20, 3, 2) / 1e9 - 35.1) < 0.05
    assert abs(gcn_activation_bytes(64, 10242, 50, 300, 20, 3, 2) / 1e9 - 135.1) < 0.1
    g = torch.Generator().manual_seed(3)
    gt = gt_cloud(5, 4000, 7)
    tc = surface_touch_charts(gt, 4, g)
    assert tc.shape == (5, 4, 25, 4) and (tc[..., 3] == 2).all()
    ax = gt.abs().amax(dim=1)                                             # the ellipsoids' semi-axes
    centre = tc[:, :, 4, :3]
    on_surface = ((centre / ax[:, None]) ** 2).sum(-1)
    assert (on_surface - 1.0).abs().max() < 0.02                          # chart centres are points of the cloud
    tv = torch.from_numpy(amesh.load_asset("touch_chart")[0]).float()
    d_ref = torch.cdist(tv, tv)
    for b in range(5):
        for k in range(4):                                                # rigid placement: all pairwise distances preserved
            assert torch.allclose(torch.cdist(tc[b, k, :, :3], tc[b, k, :, :3]), d_ref, atol=2e-6)
    # tangent to the surface: the chart's normal offsets from its centre are tiny next to its 8.7 mm half-width
    n = centre / (ax[:, None] ** 2)
    n = n / n.norm(dim=-1, keepdim=True)
    off = ((tc[..., :3] - centre[:, :, None]) * n[:, :, None]).sum(-1).abs().max()
    assert off < 5e-4


def test_bench_self_launch_command_line(monkeypatch):
    """bench.py's launcher-less path builds the driver's launch line from ITS OWN arguments (minus the launcher choice) and
    starts it as a child process — here with subprocess.call and the device count replaced, on the CPU."""
    import importlib.util
    import subprocess
    import sys
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    seen = {}
    monkeypatch.setattr(bench.torch.cuda, "device_count", lambda: 8)
    monkeypatch.setattr(subprocess, "call", lambda cmd, env=None: seen.update(cmd=cmd, env=env) or 0)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "7", "--launcher", "auto", "--warmup=2", "--launcher=spawn"])
    a = bench.parse()
    assert bench.self_launch(a) == 0
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"] and "--nnodes=1" in cmd
    assert cmd[cmd.index("--nproc-per-node") + 1] == "4" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    tail = cmd[cmd.index(os.path.join(ROOT, "bench.py")) + 1:]
    assert tail == ["--gpus", "4", "--steps", "7", "--warmup=2", "--launcher", "none"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    monkeypatch.setattr(bench.torch.cuda, "device_count", lambda: 2)
    assert bench.self_launch(a) == 2                                      # refused: 4 asked, 2 visible


def test_library_adam_on_cpu_tensors_is_torch_adam():
    """a3vt_amd.optim.Adam (the trainer's optimizer, vision/train.py:64) IS a torch.optim.Adam: what its kernel does not take — here
    CPU parameters — goes through torch's own step, bit for bit, and the state_dict has torch's layout."""
    from a3vt_amd import optim as aopt
    g = torch.Generator().manual_seed(0)
    mk = lambda: [torch.nn.Parameter(torch.randn(5, 7, generator=torch.Generator().manual_seed(1))), torch.nn.Parameter(torch.ones(3))]  # noqa: E731
    pa, pb = mk(), mk()
    oa, ob = aopt.Adam(pa, lr=1e-2), torch.optim.Adam(pb, lr=1e-2, foreach=False)
    assert isinstance(oa, torch.optim.Adam)
    for _ in range(3):
        grads = [torch.randn(p.shape, generator=g) for p in pa]
        for ps in (pa, pb):
            for p, gr in zip(ps, grads):
                p.grad = gr.clone()
        oa.step()
        ob.step()
    assert oa.library_steps == 0
    assert all(torch.equal(a, b) for a, b in zip(pa, pb))
    sa, sb = oa.state_dict(), ob.state_dict()
    assert sa["param_groups"][0].keys() == sb["param_groups"][0].keys() and sa["state"].keys() == sb["state"].keys()
    assert all(sa["state"][k].keys() == sb["state"][k].keys() for k in sa["state"])
