"""Synthetic inputs of the benchmark shapes: seeded ground-truth clouds and loader-format batches.

The reference's dataset (``download_data.sh``) is not available offline; these batches have the wire format
of ``mesh_loader_vision.collate`` (``utility/data_loaders.py:249-258``): ``gt_points (B,P,3)``,
``img`` (dummy ``(B,1)`` when ``use_img`` is off), ``touch_charts`` and ``names``.
"""
from types import SimpleNamespace

import numpy as np
import torch


def make_args(**kw):
    """Argument namespace with the reference trainer's defaults for this path (``vision/train.py:375-386``: 20 GCN
    layers x 300 hidden, cut 0.33, loss_coeff 9000, Adam lr 3e-4) — what ``Engine`` / ``Deformation`` read from
    ``args``.  Keyword arguments override; ``num_stages`` and ``gemm_precision`` are this package's own knobs."""
    d = dict(use_img=False, use_touch=False, finger=False, num_grasps=1, num_GCN_layers=20, hidden_GCN_size=300,
             cut=0.33, number_points=1000, loss_coeff=9000.0, lr=3e-4, seed=0, num_stages=3)
    d.update(kw)
    return SimpleNamespace(**d)


def gt_cloud(batch, points, seed=0, kind="ellipsoid"):
    """'ellipsoid': points on random ellipsoid surfaces with semi-axes U(0.05,0.16) (headline, surface-like);
    'cube': uniform in [-0.16,0.16]^3 (dataset objects are scaled to ~0.32 extent, utils.py:348-356)."""
    g = np.random.default_rng(seed)
    if kind == "cube":
        return torch.from_numpy(g.uniform(-0.16, 0.16, (batch, points, 3)).astype(np.float32))
    d = g.normal(size=(batch, points, 3))
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    ax = g.uniform(0.05, 0.16, (batch, 1, 3))
    return torch.from_numpy((d * ax).astype(np.float32))


def touch_charts(batch, args, seed=0):
    """Loader-format touch tensor: (B,G,4,25,4) or (B,G,25,4) when ``finger``; masks in {0,1,2} per chart."""
    g = np.random.default_rng(seed + 1)
    shape = (batch, args.num_grasps, 25, 4) if args.finger else (batch, args.num_grasps, 4, 25, 4)
    t = np.zeros(shape, dtype=np.float32)
    centre = g.uniform(-0.12, 0.12, shape[:-2] + (1, 3))
    t[..., :3] = centre + g.normal(scale=0.004, size=shape[:-1] + (3,))
    t[..., 3] = g.integers(0, 3, shape[:-2] + (1,))
    t[..., :3] *= (t[..., 3:] > 0)  # empty slots are all-zero (data_loaders.py:222-228)
    return torch.from_numpy(t)


class SyntheticLoader:
    """Iterable of `steps` identical-shape batches (fresh clouds per step, deterministic in `seed`)."""

    def __init__(self, args, steps, batch_size, seed=0, kind="ellipsoid"):
        self.args, self.steps, self.batch_size, self.seed, self.kind = args, steps, batch_size, seed, kind

    def __len__(self):
        return self.steps

    def __iter__(self):
        for s in range(self.steps):
            b = {"names": [(f"synthetic_{self.seed}_{s}_{i}", []) for i in range(self.batch_size)],
                 "gt_points": gt_cloud(self.batch_size, self.args.number_points, self.seed * 100003 + s, self.kind),
                 "img": torch.zeros(self.batch_size, 1)}
            b["touch_charts"] = touch_charts(self.batch_size, self.args, self.seed * 7 + s) \
                if self.args.use_touch else torch.ones(self.batch_size, 1)
            yield b


def surface_touch_charts(gt, charts, generator):
    """SURVEY 8d, touch inputs of configs[3]: per sample ``charts`` touch charts — the packaged 25-vertex chart (a 5 x 5 grid,
    1.7 cm wide, vertex 4 its centre) laid tangent to the ground-truth surface at a randomly chosen point of the cloud, mask
    token 2 (a successful touch).  ``gt`` (B,P,3) points on an origin-centred ellipsoid.  Returns (B, charts, 25, 4).
    (Until round 4 the configs[3] stand-in drew every chart VERTEX uniformly in a 0.3 cube: 32 faces per chart spanning the
    whole volume, which the area-weighted sampler then covered with points — a surface no touch sensor produces, and one
    whose samples near the centre defeat any nearest-neighbour pruning.)"""
    from . import mesh as amesh
    tv = torch.from_numpy(amesh.load_asset("touch_chart")[0]).float()          # (25,3): x = normal, (y,z) = tangent coordinates
    B, P, _ = gt.shape
    idx = torch.randint(0, P, (B, charts), generator=generator)
    p = torch.gather(gt, 1, idx[..., None].expand(B, charts, 3))               # (B,charts,3) surface points
    ax = gt.abs().amax(dim=1, keepdim=True)                                    # semi-axes of each sample's ellipsoid
    n = p / (ax * ax)
    n = n / n.norm(dim=-1, keepdim=True)
    e = torch.zeros_like(n)
    e.scatter_(-1, n.abs().argmin(dim=-1, keepdim=True), 1.0)                  # the coordinate axis least aligned with n
    t1 = torch.linalg.cross(n, e)
    t1 = t1 / t1.norm(dim=-1, keepdim=True)
    t2 = torch.linalg.cross(n, t1)
    v = p[:, :, None, :] + tv[None, None, :, 0:1] * n[:, :, None, :] + tv[None, None, :, 1:2] * t1[:, :, None, :] \
        + tv[None, None, :, 2:3] * t2[:, :, None, :]
    return torch.cat((v, torch.full((B, charts, 25, 1), 2.0)), dim=-1)


# ---- BASELINE.json configs[3] / configs[4] on one GPU (bench.py `named_configs`, tools/named_configs.py) ----------------
def gcn_activation_bytes(batch, n_vert, in_features, hidden, layers, stages=3, elem=4):
    """SURVEY §8d ``Bytes_act`` for a whole batch: ``S * 3 * sum_i elem * N * (d_i + d_{i+1})`` with the layer widths
    ``[in_features, hidden x (L-1), 3]`` — forward reads X and writes Y, backward reads dY and X and writes dX (about twice the
    forward).  ``elem`` = 4 (fp32) or 2 (bf16 storage).  This is the ALGORITHMIC traffic the HBM roofline of the bf16
    configurations is priced with (35.1 GB for cfg-4 without image, 135 GB for cfg-5 at bs 64)."""
    dims = [in_features] + [hidden] * (layers - 1) + [3]
    return batch * stages * 3 * sum(elem * n_vert * (a + b) for a, b in zip(dims[:-1], dims[1:]))


LIBRARY_ADAM = [True]   # False: torch's fused Adam in the named configurations' step (tools/named_configs.py --torch-adam, A/B)


def named_config(which, dev, precision="bf16s", batch=None):
    """Model and synthetic inputs of BASELINE.json configs[3] (vision + touch: image model with the default CNNs + chart
    atlas with 4 touch charts laid on the ground-truth surface, N = 1924, 25 000-point Chamfer, bs 64) or configs[4]'s per-GPU shard (10 242-vertex
    icosphere-5, 50 000-point Chamfer, bs 8 of the global 64).  Returns a dict; ``NamedStep`` runs training steps on it."""
    from . import mesh as amesh
    from .pterotactyl.reconstruction.vision import model
    from .pterotactyl.utility import utils
    if which == 3:
        B = batch or 64
        g = torch.Generator().manual_seed(0)
        args = make_args(use_img=True, use_touch=True, finger=False, num_grasps=1, number_points=25000,
                         gemm_precision=precision, CNN_ker_size=5, num_CNN_blocks=6, layers_per_block=3)
        info, verts = utils.load_mesh_vision(args, "vision_charts")
        torch.manual_seed(0)
        net = model.Deformation(info, verts, args).to(dev)
        gt = gt_cloud(B, args.number_points, 0)
        tc = surface_touch_charts(gt, 4, g).view(B, 1, 4, 25, 4)
        img = torch.rand(B, 3, 256, 256, generator=g).to(dev)
        charts = model.prepare_mesh({"img": img, "touch_charts": tc}, verts, args)
        n_vert = int(verts.shape[0]) + 100
        name = f"configs[3]: image + 4 touch charts (N={n_vert}), 25k-pt Chamfer, {precision}, bs={B}"
        in_features = 448
    elif which == 4:
        B = batch or 8
        args = make_args(number_points=50000, gemm_precision=precision)
        v, f = amesh.icosphere(5)
        verts, ft = torch.from_numpy(v).to(dev), torch.from_numpy(f).to(dev)
        info = utils.adj_init(verts, ft, args)
        torch.manual_seed(0)
        net = model.Deformation(info, verts, args).to(dev)
        img = torch.zeros(B, 1, device=dev)
        charts = model.prepare_mesh({"img": img}, verts, args)
        n_vert = int(verts.shape[0])
        name = f"configs[4] shard: icosphere-5 (N={n_vert}), 50k-pt Chamfer, {precision}, bs={B}"
        in_features = 50
    else:
        raise ValueError(which)
    gt = gt_cloud(B, args.number_points, 0).to(dev)   # (the same clouds the touch charts of configs[3] were placed on)
    elem = 2 if precision == "bf16s" else 4
    return {"name": name, "net": net, "info": info, "charts": charts, "img": img, "gt": gt, "args": args, "batch": B,
            "n_vert": n_vert, "in_features": in_features,
            "activation_bytes": gcn_activation_bytes(B, n_vert, in_features, args.hidden_GCN_size, args.num_GCN_layers, 3, elem)}


class NamedStep:
    """The trainer's step (``Engine.train_step``: flat bucket, the library's Adam) on a ``named_config``."""

    def __init__(self, cfg):
        from . import distributed as adist
        self.cfg = cfg
        self.params = list(cfg["net"].parameters())
        self.bucket = adist.FlatGradBucket(self.params)
        from . import optim as a3vt_optim
        self.opt = a3vt_optim.make_adam(self.params, cfg["args"].lr, library=LIBRARY_ADAM[0])

    def __call__(self):
        from .pterotactyl.utility import utils
        c = self.cfg
        self.bucket.zero()
        v = c["net"](c["img"], c["charts"])[0]
        loss = c["args"].loss_coeff * utils.chamfer_distance(v, c["info"]["faces_i32"], c["gt"], num=c["args"].number_points).mean()
        loss.backward()
        self.bucket.all_reduce_mean()   # single process: gathers the gradients and re-homes .grad
        self.opt.step()
        self.verts = v.detach()
        return loss.detach()

    def chamfer_forward_ms(self, reps=3):
        """Device time of the loss forward alone (3 surface draws + the exact pruned search both ways + reduce) on the
        step's last predicted vertices: the search's share of a step."""
        from .pterotactyl.utility import utils
        c = self.cfg
        with torch.no_grad():
            utils.chamfer_distance(self.verts, c["info"]["faces_i32"], c["gt"], num=c["args"].number_points)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                utils.chamfer_distance(self.verts, c["info"]["faces_i32"], c["gt"], num=c["args"].number_points)
            e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    def close(self):
        self.bucket.close()


PROFILE_CLASSES = ("product_fwd", "product_dx", "product_dw", "aggregation", "output_layer", "search", "sampling_and_chamfer_bwd",
                   "encoders_and_pooling", "optimizer")


def kernel_classes(step, cfg, precision, step_ms):
    """Where one training step's device time goes, by kernel class, from the library's own HIP events (``a3vt_profile_*``: an
    event pair around every call of the class on its launch stream) over ONE extra step: ms per step, launches, and — for the
    GCN classes whose algorithmic bytes are a closed form — the HBM rate they reach against 8 TB/s.  ``not_this_library`` is
    the rest of the step: MIOpen convolutions / batch-norm, torch element-wise kernels, gaps between kernels."""
    import ctypes
    from . import lib as _lib
    L = _lib.load()
    L.a3vt_profile_enable(1)
    step()
    torch.cuda.synchronize()
    n = len(PROFILE_CLASSES)
    tot, cnt = (ctypes.c_double * n)(), (ctypes.c_int * n)()
    _lib.check(min(L.a3vt_profile_read_classes(tot, cnt, n), 0), "profile_read_classes")
    L.a3vt_profile_enable(0)
    e = 2 if precision == "bf16s" else 4
    a = cfg["args"]
    m, h, c = cfg["batch"] * cfg["n_vert"], a.hidden_GCN_size, round(a.hidden_GCN_size * a.cut)
    hidden_launches = 3 * (a.num_GCN_layers - 2)          # hidden x hidden products per direction and step (3 stages)
    per_launch = {"product_fwd": m * 2 * h * e, "product_dx": m * 2 * h * e, "product_dw": m * 2 * h * e,
                  "aggregation": m * 2 * c * e}           # rows read + rows written (dW: X and dZ read)
    out = {}
    for i, k in enumerate(PROFILE_CLASSES):
        rec = {"ms_per_step": tot[i], "launches": cnt[i]}
        if k in per_launch and tot[i] > 0:
            launches = 2 * 3 * (a.num_GCN_layers - 1) if k == "aggregation" else hidden_launches
            gbs = per_launch[k] * launches / (tot[i] * 1e-3) / 1e9
            rec.update({"algorithmic_GBps": gbs, "frac_of_8TBps": gbs / 8000.0})
        out[k] = rec
    ours = sum(tot[i] for i in range(n))
    out["not_this_library"] = {"ms_per_step": max(step_ms - ours, 0.0),
                               "what": "MIOpen convolutions (image layers 7-12) and fp32 batch-norm, torch element-wise kernels, gaps between kernels"}
    return out


def time_named_config(which, dev, precision="bf16s", batch=None, steps=10, warm=8):
    """Build, warm up (MIOpen's find mode and the allocator's growth take several steps to settle in the image mode), time
    ``steps`` training steps; returns the JSON-able record ``bench.py`` prints under ``named_configs``."""
    import time
    t_build = time.perf_counter()
    cfg = named_config(which, dev, precision, batch)
    step = NamedStep(cfg)
    for _ in range(warm):
        loss = step()
    torch.cuda.synchronize()
    t_build = time.perf_counter() - t_build
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(steps):
        loss = step()
        marks[i + 1].record()
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / steps
    dev_ms = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(steps))
    search = step.chamfer_forward_ms()
    classes = kernel_classes(step, cfg, precision, dev_ms[len(dev_ms) // 2])
    nbytes = cfg["activation_bytes"]
    gbs = nbytes / (ms * 1e-3) / 1e9
    rec = {"config": cfg["name"], "ms_per_step": ms, "device_ms_median": dev_ms[len(dev_ms) // 2], "iters_per_s": 1e3 / ms,
           "steps": steps, "warmup": warm, "setup_s": t_build, "loss": float(loss), "finite": bool(torch.isfinite(loss)),
           "roofline": {"bound": "hbm", "achieved": gbs, "peak": 8000.0, "unit": "GB/s", "frac": gbs / 8000.0,
                        "bytes_per_step": nbytes,
                        "bytes_how": f"SURVEY 8d Bytes_act = B*S*3*sum_i e*N*(d_i+d_{{i+1}}), e={2 if precision == 'bf16s' else 4} B, "
                                     f"N={cfg['n_vert']}, d=[{cfg['in_features']},{cfg['args'].hidden_GCN_size}x"
                                     f"{cfg['args'].num_GCN_layers - 1},3], B={cfg['batch']}, S=3 (GCN activations only: no CNN, "
                                     f"no Chamfer bytes)",
                        "hbm_ms_at_peak": nbytes / 8e12 * 1e3,
                        "chamfer_forward_ms": search, "chamfer_share": search / ms},
           "kernel_classes": classes}
    step.close()
    return rec
