"""The pruned exact nearest-neighbour search (csrc/nn_prune.hip, ``a3vt_chamfer_fwd_ws`` algo 3) against the brute-force
searches and the C oracle: integer / bit-exact bar — every distance and every index identical, ties to the lowest index.
Covers what the domain offers as edge cases: ragged sizes (last block padded), single-point clouds, duplicated points
(exact ties inside a block, across blocks and across the pad), surfaces far apart (pruning degenerates towards brute
force), degenerate extents (all points equal, planar and collinear clouds: grid cells collapse), unequal cloud sizes,
clouds above 20k points (the 32^3 grid), and the trainer's call through ``ChamferFn`` (automatic choice)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

NAMES = ("dist_xy", "idx_xy", "dist_yx", "idx_yx", "cd")


def _surface(gen, n, kind, shift=0.0):
    """n points on a closed surface (what the trainer samples), float32."""
    u = torch.randn(n, 3, generator=gen)
    u = u / u.norm(dim=1, keepdim=True)
    if kind == "sphere":
        p = 0.4 * u
    elif kind == "ellipsoid":
        p = u * torch.tensor([0.5, 0.3, 0.2])
    elif kind == "cube":
        p = 0.35 * u / u.abs().max(dim=1, keepdim=True).values
    elif kind == "volume":
        p = torch.rand(n, 3, generator=gen) - 0.5
    else:
        raise ValueError(kind)
    return (p + shift).float()


def _both(x, y, cuda):
    from a3vt_amd import ops
    xd, yd = x.to(cuda), y.to(cuda)
    return ops.chamfer_nn(xd, yd, algo="pruned"), ops.chamfer_nn(xd, yd, algo="sweep")


def _assert_same(a, b):
    for name, u, v in zip(NAMES, a, b):
        assert torch.equal(u, v), f"{name}: {(u != v).sum().item()} of {u.numel()} differ"


@pytest.mark.parametrize("draws,B,P,Q,kx,ky,shift", [
    (1, 1, 1, 1, "sphere", "sphere", 0.0),
    (1, 2, 63, 65, "sphere", "cube", 0.0),
    (2, 3, 64, 128, "volume", "volume", 0.0),
    (1, 2, 100, 37, "ellipsoid", "sphere", 0.0),
    (3, 2, 1000, 2176, "sphere", "ellipsoid", 0.0),
    (2, 1, 4099, 5000, "cube", "sphere", 0.0),
    (1, 1, 5, 3000, "sphere", "sphere", 0.0),
    (1, 1, 3000, 5, "sphere", "sphere", 0.0),
    (1, 4, 2600, 700, "volume", "sphere", 0.0),
    (2, 2, 2048, 2048, "sphere", "sphere", 3.0),        # surfaces far apart: every block is about equally far
    (1, 2, 3001, 2999, "ellipsoid", "cube", 0.05),
])
def test_pruned_equals_brute_force(cuda, draws, B, P, Q, kx, ky, shift):
    g = torch.Generator().manual_seed(P * 31 + Q)
    x = torch.stack([torch.stack([_surface(g, P, kx, shift) for _ in range(B)]) for _ in range(draws)])
    y = torch.stack([_surface(g, Q, ky) for _ in range(B)])
    _assert_same(*_both(x, y, cuda))


@pytest.mark.parametrize("P,Q", [(777, 1025), (2176, 1000), (4096, 4096)])
def test_pruned_ties_go_to_the_lowest_index(cuda, P, Q):
    """Every candidate exists twice (and some three times, at indices far apart, so the copies land in one sorted block
    next to each other AND — for the copies of the cloud's last points — next to the pad), queries are duplicated too,
    a few coincide with candidates (zero distance)."""
    g = torch.Generator().manual_seed(P + Q)
    x = torch.stack([torch.stack([_surface(g, P, "sphere") for _ in range(3)]) for _ in range(2)])
    y = torch.stack([_surface(g, Q, "ellipsoid") for _ in range(3)])
    y[:, Q // 2:] = y[:, :Q - Q // 2]
    y[:, -5:] = y[:, 10:15]
    x[:, :, P // 2:] = x[:, :, :P - P // 2]
    x[0, :, :3] = y[:, :3]
    pruned, brute = _both(x, y, cuda)
    _assert_same(pruned, brute)
    assert (pruned[1] < Q // 2 + 1).all()          # of identical candidates the first one


def test_pruned_degenerate_extents(cuda):
    """All points equal; all on a plane; all on a line; one cloud a single repeated point: the grid collapses to one
    cell / one layer and the block boxes are flat — results unchanged."""
    g = torch.Generator().manual_seed(5)
    P = Q = 2500
    x = torch.rand(1, 4, P, 3, generator=g) - 0.5
    y = torch.rand(4, Q, 3, generator=g) - 0.5
    x[0, 0] = 0.25                                   # one point, P times
    y[1, :, 2] = 0.125                               # planar
    x[0, 2, :, 1:] = -0.5                            # collinear
    y[3] = y[3, :1]                                  # one point, Q times
    _assert_same(*_both(x.float(), y.float(), cuda))


def test_pruned_against_the_c_oracle(cuda):
    """Bit for bit against oracle/chamfer_nn.c with the device's contraction (fma=True), both directions."""
    from a3vt_amd import ops
    from oracle import chamfer as ochamfer
    g = torch.Generator().manual_seed(11)
    P, Q = 6000, 5000
    x = _surface(g, P, "sphere")
    y = _surface(g, Q, "ellipsoid")
    dxy, ixy, dyx, iyx, _ = ops.chamfer_nn(x[None, None].to(cuda), y[None].to(cuda), algo="pruned")
    d1, i1 = ochamfer.nn_sqdist_c(x.numpy(), y.numpy(), fma=True)
    d2, i2 = ochamfer.nn_sqdist_c(y.numpy(), x.numpy(), fma=True)
    assert np.array_equal(dxy[0, 0].cpu().numpy(), d1) and np.array_equal(ixy[0, 0].cpu().numpy(), i1)
    assert np.array_equal(dyx[0, 0].cpu().numpy(), d2) and np.array_equal(iyx[0, 0].cpu().numpy(), i2)


@pytest.mark.parametrize("P,Q,B", [(25000, 25000, 2), (50000, 50000, 1), (30000, 9000, 2)])
def test_pruned_large_clouds(cuda, P, Q, B):
    """BASELINE configs[3] / configs[4] cloud sizes (the 32^3 grid above 20k points), unequal sizes."""
    g = torch.Generator().manual_seed(P // 1000 + Q)
    x = torch.stack([_surface(g, P, "sphere", 0.02) for _ in range(B)])[None]
    y = torch.stack([_surface(g, Q, "ellipsoid") for _ in range(B)])
    _assert_same(*_both(x, y, cuda))


def test_automatic_choice_and_workspace_checks(cuda):
    """ChamferFn (the trainer's call) lets the library choose: the result equals the brute-force one either side of the
    2048-point threshold.  A workspace too small for the algorithm asked for is an error, not a silent fallback."""
    from a3vt_amd import lib, ops
    g = torch.Generator().manual_seed(3)
    for P, Q in ((1500, 1500), (2500, 2500)):
        x = _surface(g, P, "sphere")[None, None].to(cuda)
        y = _surface(g, Q, "cube")[None].to(cuda)
        auto = ops.chamfer_nn(x, y, algo="auto")
        _assert_same(auto, ops.chamfer_nn(x, y, algo="two_pass"))
        cd = ops.ChamferFn.apply(x, y)
        assert torch.equal(cd, auto[4])
    L = lib.load()
    x = torch.rand(1, 1, 4096, 3, device=cuda)
    y = torch.rand(1, 4096, 3, device=cuda)
    out = [torch.empty(4096, device=cuda) for _ in range(2)] + [torch.empty(4096, dtype=torch.int32, device=cuda) for _ in range(2)]
    cd = torch.empty(1, device=cuda)
    ws = torch.empty(1024, dtype=torch.uint8, device=cuda)
    rc = L.a3vt_chamfer_fwd_ws(lib.ptr(x), lib.ptr(y), 1, 1, 4096, 4096, lib.ptr(out[0]), lib.ptr(out[2]), lib.ptr(out[1]),
                               lib.ptr(out[3]), lib.ptr(cd), lib.ptr(ws), 1024, 3, None)
    assert rc < 0 and b"too small" in L.a3vt_last_error()
    rc = L.a3vt_chamfer_fwd_ws(lib.ptr(x), lib.ptr(y), 1, 1, 4096, 4096, lib.ptr(out[0]), lib.ptr(out[2]), lib.ptr(out[1]),
                               lib.ptr(out[3]), lib.ptr(cd), lib.ptr(ws), 1024, 7, None)
    assert rc < 0 and b"unknown search algorithm" in L.a3vt_last_error()
    torch.cuda.synchronize()


def test_pruned_many_small_clouds(cuda):
    """More cloud pairs than a grid's y dimension holds (65 535): the search kernels run on flattened grids."""
    from a3vt_amd import ops
    g = torch.Generator().manual_seed(21)
    B, P, Q = 70000, 8, 5
    x = torch.rand(1, B, P, 3, generator=g)
    y = torch.rand(B, Q, 3, generator=g)
    dxy, ixy, dyx, iyx, cd = ops.chamfer_nn(x.to(cuda), y.to(cuda), algo="pruned")
    d = ((x[0][:, :, None, :] - y[:, None, :, :]) ** 2).sum(-1)          # (B,P,Q), same fp32 order is not needed: compare loosely
    assert torch.allclose(dxy[0].cpu(), d.min(2).values, rtol=1e-5, atol=1e-7)
    assert torch.allclose(dyx[0].cpu(), d.min(1).values, rtol=1e-5, atol=1e-7)
    assert torch.equal(torch.gather(d, 2, ixy[0].cpu().long()[..., None])[..., 0], d.min(2).values) or \
        torch.allclose(torch.gather(d, 2, ixy[0].cpu().long()[..., None])[..., 0], d.min(2).values, rtol=1e-5, atol=1e-7)
    assert torch.isfinite(cd).all()


def test_pruned_nonfinite_coordinates_stay_in_bounds(cuda):
    """NaN / Inf coordinates (a diverged network): no hang, no out-of-range index — the backward kernel gathers through
    these indices — and the clouds that are finite are unaffected."""
    from a3vt_amd import ops
    g = torch.Generator().manual_seed(13)
    P = Q = 3000
    x = torch.stack([_surface(g, P, "sphere") for _ in range(3)])[None]
    y = torch.stack([_surface(g, Q, "ellipsoid") for _ in range(3)])
    clean = ops.chamfer_nn(x.to(cuda), y.to(cuda), algo="pruned")
    x[0, 1, ::7] = float("nan")
    x[0, 1, 5] = float("inf")
    y[2, 100:200] = float("nan")
    y[2, 7, 1] = -float("inf")
    dxy, ixy, dyx, iyx, cd = ops.chamfer_nn(x.to(cuda), y.to(cuda), algo="pruned")
    torch.cuda.synchronize()
    assert ixy.min() >= 0 and ixy.max() < Q and iyx.min() >= 0 and iyx.max() < P
    assert torch.equal(dxy[0, 0], clean[0][0, 0]) and torch.equal(ixy[0, 0], clean[1][0, 0])
    assert torch.equal(dyx[0, 0], clean[2][0, 0]) and torch.equal(iyx[0, 0], clean[3][0, 0])
    # the gradient kernel runs on these indices without faulting
    xg = x.to(cuda).requires_grad_(True)
    ops.ChamferFn.apply(xg, y.to(cuda)).sum().backward()
    torch.cuda.synchronize()
    assert torch.isfinite(xg.grad[0, 0]).all()


def test_training_is_bit_identical_whichever_search_runs(cuda):
    """Five optimiser steps of the 3-stage network (icosphere-4, bs 4, 4 000-point Chamfer, fresh Philox draws per step)
    with the pruned search and with the brute-force sweep: every loss and every weight identical bit for bit — the
    searches return the same neighbours and every other kernel of the step is deterministic."""
    from a3vt_amd import ops
    from a3vt_amd.pterotactyl.reconstruction.vision import model
    from a3vt_amd.pterotactyl.utility import utils
    from a3vt_amd.synthetic import gt_cloud, make_args
    from helpers import template
    args = make_args(number_points=4000)
    v, f = template("ico4")
    vt, ft = torch.from_numpy(v).to(cuda), torch.from_numpy(f).to(cuda)
    info = utils.adj_init(vt, ft, args)
    B = 4
    charts = model.prepare_mesh({"img": torch.zeros(B, 1)}, vt, args)
    gt = gt_cloud(B, 4000, 7).to(cuda)
    runs = {}
    try:
        for name, algo in (("pruned", ops.NN_ALGOS["pruned"]), ("sweep", ops.NN_ALGOS["sweep"])):
            ops.CHAMFER_ALGO = algo
            torch.manual_seed(0)
            net = model.Deformation(info, vt, args).to(cuda)
            opt = torch.optim.Adam(net.parameters(), lr=1e-3)
            losses = []
            for _ in range(5):
                opt.zero_grad()
                out = net(torch.zeros(B, 1), charts)[0]
                loss = args.loss_coeff * utils.chamfer_distance(out, info["faces"], gt, num=4000).mean()
                loss.backward()
                opt.step()
                losses.append(loss.item())
            runs[name] = (losses, torch.cat([p.detach().reshape(-1) for p in net.parameters()]).clone())
    finally:
        ops.CHAMFER_ALGO = 0
    assert runs["pruned"][0] == runs["sweep"][0], (runs["pruned"][0], runs["sweep"][0])
    assert torch.equal(runs["pruned"][1], runs["sweep"][1])
    assert runs["pruned"][0][-1] < runs["pruned"][0][0]          # and it trains


def test_pruned_repeats_bit_for_bit(cuda):
    """The order of points inside a grid cell depends on LDS atomics; the outputs must not."""
    from a3vt_amd import ops
    g = torch.Generator().manual_seed(9)
    x = torch.stack([_surface(g, 10000, "sphere") for _ in range(4)])[None].to(cuda)
    y = torch.stack([_surface(g, 10000, "ellipsoid") for _ in range(4)]).to(cuda)
    first = ops.chamfer_nn(x, y, algo="pruned")
    for _ in range(3):
        _assert_same(first, ops.chamfer_nn(x, y, algo="pruned"))


@pytest.mark.parametrize("algo", ["two_pass", "sweep", "pruned"])
def test_shared_ground_truth_equals_repeated(cuda, algo):
    """a3vt_chamfer_fwd_shared (policies/environment.py:174-180,252-257: K candidates of an element share its ground truth):
    y with E clouds for a candidate-major batch of K * E meshes gives, bit for bit, what the K-times repeated y gives —
    distances, indices and the per-mesh Chamfer value, in every search algorithm."""
    from a3vt_amd import ops
    K, E, P, Q, draws = 5, 3, 2300, 2600, 2
    g = torch.Generator().manual_seed(3)
    x = (torch.randn(draws, K * E, P, 3, generator=g) * 0.1).to(cuda)
    y = (torch.randn(E, Q, 3, generator=g) * 0.1).to(cuda)
    shared = ops.chamfer_nn(x, y, algo=algo)
    repeated = ops.chamfer_nn(x, y.repeat(K, 1, 1), algo=algo)
    for a, b in zip(shared, repeated):
        assert torch.equal(a, b)
    with torch.no_grad():
        assert torch.equal(ops.ChamferFn.apply(x, y), ops.ChamferFn.apply(x, y.repeat(K, 1, 1)))
    with pytest.raises(RuntimeError, match="forward-only"):
        ops.ChamferFn.apply(x.clone().requires_grad_(True), y).sum().backward()


@pytest.mark.parametrize("P,Q", [(10000, 10000), (4133, 9001), (25000, 2500)])
def test_oriented_boxes_on_concentric_thin_and_tilted_surfaces(cuda, P, Q):
    """Round 5: the pruning bounds are oriented boxes in the principal frame of every block / group (valid through stated
    margins, not through monotone arithmetic).  The geometries that lean on them hardest: concentric surfaces 0.1-0.2 apart
    (the untrained network: a sphere of 0.25 around small ellipsoids — also the configuration in which the pad lanes of a
    cloud's last query block used to ask about the origin), a sphere sampled from its own centre region, tilted thin slabs
    (boxes 1000 x thinner than wide), exactly planar tilted patches and a cloud far from the origin (margins relative to the
    coordinates' magnitude).  Bit for bit the brute force."""
    g = torch.Generator().manual_seed(P + 7 * Q)

    def unit(n):
        u = torch.randn(n, 3, generator=g)
        return u / u.norm(dim=1, keepdim=True)

    rot = torch.linalg.qr(torch.randn(3, 3, generator=g))[0]
    sphere = 0.25 * unit(P)
    ell = unit(Q) * torch.tensor([0.05, 0.16, 0.09])
    tiny = 0.01 * unit(Q)                                             # queries near the centre: everything is about equally far
    slab = (torch.rand(P, 3, generator=g) - 0.5) * torch.tensor([0.6, 0.6, 6e-4]) @ rot.T
    plane = torch.cat(((torch.rand(Q, 2, generator=g) - 0.5) * 0.5, torch.zeros(Q, 1)), dim=1) @ rot.T + 0.07
    far = sphere + torch.tensor([1000.0, -2000.0, 500.0])
    x = torch.stack([torch.stack([sphere, slab, sphere, far])]).float()
    y = torch.stack([ell, plane, tiny, ell + torch.tensor([1000.0, -2000.0, 500.0])]).float()
    _assert_same(*_both(x, y, cuda))
    _assert_same(*_both(y[None], x[0], cuda))


@pytest.mark.parametrize("seed", range(10))
def test_oriented_boxes_fuzz_over_scales_and_shapes(cuda, seed):
    """The oriented boxes prune through stated margins (2^-20 of the coordinates' magnitude, 2^-16 on the bound): random
    clouds under random affine maps — scales from 1e-6 to 1e4, anisotropy up to 1e4 : 1 (needles and sheets), offsets up to
    1e3 times the extent, mixtures of a surface and a volume, sizes that leave 1-63 valid lanes in the last block — must stay
    bit for bit the brute force in both directions."""
    g = torch.Generator().manual_seed(1000 + seed)

    def cloud(n):
        kind = int(torch.randint(0, 4, (1,), generator=g))
        u = torch.randn(n, 3, generator=g)
        if kind == 0:
            p = u / u.norm(dim=1, keepdim=True)                                  # a sphere
        elif kind == 1:
            p = torch.rand(n, 3, generator=g) - 0.5                              # a volume
        elif kind == 2:
            p = torch.cat((u[: n // 2] / u[: n // 2].norm(dim=1, keepdim=True), 0.3 * (torch.rand(n - n // 2, 3, generator=g) - 0.5)))
        else:
            p = torch.cat((torch.rand(n, 2, generator=g) - 0.5, torch.zeros(n, 1)), dim=1)   # a plane
        stretch = torch.diag(10.0 ** (torch.rand(3, generator=g) * 4 - 2))       # up to 1e4 : 1
        rot = torch.linalg.qr(torch.randn(3, 3, generator=g))[0]
        return p @ stretch @ rot.T

    scale = 10.0 ** float(torch.rand(1, generator=g) * 10 - 6)
    shift = scale * (10.0 ** float(torch.rand(1, generator=g) * 3)) * torch.randn(3, generator=g) * float(torch.rand(1, generator=g) < 0.5)
    P = int(torch.randint(2049, 6000, (1,), generator=g))
    Q = int(torch.randint(2049, 6000, (1,), generator=g))
    x = torch.stack([torch.stack([scale * cloud(P) + shift for _ in range(2)])]).float()
    y = torch.stack([scale * cloud(Q) + shift + scale * 0.3 * torch.randn(3, generator=g) for _ in range(2)]).float()
    _assert_same(*_both(x, y, cuda))


def test_search_work_stays_bounded(cuda):
    """Round 6 (verdict r05 #2c): a WORK regression test.  Every test above checks values, and values cannot show a search that
    does too much: in rounds 2-4 the pad lanes of a cloud's last query block asked about the origin instead of repeating a real
    query — harmless for the results, 1.85 ms of tail on a 1.5 ms launch (192 waves evaluating 440-590 of 625 groups).  The
    library's work counters (``a3vt_dbg_nn_work``) on the geometry of an untrained network — predicted sphere of radius 0.25
    around ellipsoidal targets, 10 000 points each, i.e. with the ORIGIN at the centre of the targets — must stay below
    bounds recorded at round 6 (mean 33 groups of 16 per wave, worst wave 112; pad lanes at the origin: worst wave > 400)."""
    from a3vt_amd import ops
    from a3vt_amd.synthetic import gt_cloud
    draws, B, N = 3, 8, 10000
    g = torch.Generator().manual_seed(0)
    u = torch.randn(draws, B, N, 3, generator=g)
    x = (u / u.norm(dim=-1, keepdim=True) * 0.25).to(cuda)
    y = gt_cloud(B, N, 0).to(cuda)
    ops.nn_work(True)
    try:
        d0 = ops.chamfer_nn(x, y, algo="pruned")
        w = ops.nn_work(False)
    finally:
        ops.nn_work(False)
    nblk = (N + 63) // 64
    assert w["waves"] == 2 * draws * B * nblk                          # one wave per block of 64 queries, both directions
    mean_groups, mean_blocks = w["groups"] / w["waves"], w["blocks"] / w["waves"]
    print(f"\n[nn work] {w['waves']} waves: {mean_groups:.1f} groups, {mean_blocks:.1f} blocks, {w['box_tests'] / w['waves']:.1f} box tests per "
          f"wave; worst wave {w['max_groups_per_wave']} groups of {4 * nblk}")
    assert mean_groups <= 48 and mean_blocks <= 26, (mean_groups, mean_blocks)
    assert w["max_groups_per_wave"] <= 200, w
    # the counters change nothing: same bits with them off
    d1 = ops.chamfer_nn(x, y, algo="pruned")
    for a, b in zip(d0, d1):
        assert torch.equal(a, b)
    assert ops.nn_work(False)["waves"] == w["waves"]                   # off: nothing was counted by the second search


@pytest.mark.parametrize("scale", [1e-15, 1e-19, 1e-20, 1e-22])
def test_oriented_boxes_at_denormal_squared_distances(cuda, scale):
    """the oriented-box bound is shrunk by (1 - 2^-16) — a multiplication that no longer shrinks once the squared
    distances are denormal (coordinates below ~1e-19), where only the absolute 1e-37 slack on the extents is left.  Sphere
    against ellipsoid at coordinate scales whose squares are tiny normals (1e-30), at the edge (1e-38), denormal (1e-40) and
    mostly zero (1e-44, where nearly every distance flushes to +0 and ties go to the lowest index): bit for bit the brute
    force in both directions."""
    g = torch.Generator().manual_seed(77)
    u = torch.randn(1, 2, 3000, 3, generator=g)
    v = torch.randn(2, 2500, 3, generator=g)
    x = (scale * 0.25 * u / u.norm(dim=-1, keepdim=True)).float()
    y = (scale * (v / v.norm(dim=-1, keepdim=True)) * torch.tensor([0.15, 0.1, 0.06])).float()
    _assert_same(*_both(x, y, cuda))
