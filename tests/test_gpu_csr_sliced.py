"""-m gpu: the channel-sliced, LDS-resident neighbour aggregation (csrc/gcn_csr.hip ``csrq_kernel``; replaces the dense
adjacency products of reconstruction/vision/model.py:356,360 for the hidden layers of a stack) against (i) the
half-wave-per-vertex kernels it replaces — bit for bit, the summation order per element is the same — and (ii) the fp64
oracle, on shapes that take the new path (>= 12 288 rows, hidden 300): plain icospheres, the reference atlas, and the
fused vision + touch graph whose hub rows (~1150 neighbours) go through ``csrq_heavy_kernel``."""
import pytest
import torch

from helpers import assert_grad_close, make_args, oracle_adj, rel_err, template

pytestmark = pytest.mark.gpu


def _run(cuda, adj, st, feats, gup, L, H, cut_len):
    from a3vt_amd import ops
    ws = [st[f"mesh_deform_1.layers.{i}.weight"].to(cuda).requires_grad_(True) for i in range(L)]
    bs = [st[f"mesh_deform_1.layers.{i}.bias"].to(cuda).requires_grad_(True) for i in range(L)]
    fd = torch.nn.functional.pad(feats, (0, 2)).to(cuda).requires_grad_(True)
    out = ops.gcn_stack(fd, adj, 50, H, cut_len, ws, bs)
    (out * gup.to(cuda)).sum().backward()
    torch.cuda.synchronize()
    return out.detach(), fd.grad, [w.grad for w in ws], [b.grad for b in bs]


@pytest.mark.parametrize("tname,use_touch,L,B,cut", [("ico3", False, 4, 24, 0.33), ("ico4", False, 3, 6, 0.33),
                                                      ("atlas", False, 3, 8, 0.33), ("atlas", True, 3, 8, 0.33),
                                                      ("atlas", 5, 3, 8, 0.33),     # 20 touch charts: regular rows of 26-30 entries
                                                      ("ico3", False, 3, 20, 0.5), ("ico3", False, 3, 20, 0.04),
                                                      ("ico4", False, 20, 6, 0.33)])   # the benchmark's depth and template
def test_channel_sliced_aggregation_equals_row_kernels_and_oracle(cuda, tname, use_touch, L, B, cut):
    from a3vt_amd import mesh as amesh, ops
    from oracle import gcn as og
    H = 300
    grasps = 1 if use_touch in (True, False) else int(use_touch)   # use_touch = 5: the production topology's five grasps
    use_touch = bool(use_touch)
    args = make_args(use_touch=use_touch, num_GCN_layers=L, hidden_GCN_size=H, num_grasps=grasps, cut=cut)
    verts, faces = template(tname)
    adj_o, _ = oracle_adj(verts, faces, args)
    n = adj_o[0].numel() - 1
    assert B * n >= 12288                       # the shape takes the channel-sliced path (capi.hip use_csrq)
    st = og.init_state(50, H, L, seed=5)
    g = torch.Generator().manual_seed(21)
    feats = torch.randn(B, n, 50, generator=g) * 0.5
    gup = torch.randn(B, n, 3, generator=g)
    if use_touch:
        sv, sf = amesh.load_asset("touch_chart")
        r, c, nn_, _ = amesh.fused_pairs(verts, faces, sf, grasps, False)
    else:
        r, c = amesh.vision_pairs(faces, verts.shape[0])
        nn_ = verts.shape[0]
    adj = ops.DeviceCSR(amesh.CSRAdjacency.from_pairs(r, c, nn_), cuda)
    adj.use_split = False      # this test pins the GENERIC kernels (any CSR); the P + bipartite split has its own below
    cut_len = og.cut_length(H, cut)
    try:
        ops.dbg_csr_algo("rows")
        ref = _run(cuda, adj, st, feats, gup, L, H, cut_len)
        # "sliced" forces the channel-sliced path on graphs with rows longer than its eight register slots too (the fused
        # touch graph: CSR continuation of long rows, hub rows through csrq_heavy_kernel) — by default those graphs stay on
        # the half-wave kernels, which are faster for them
        ops.dbg_csr_algo("sliced")
        new = _run(cuda, adj, st, feats, gup, L, H, cut_len)
        again = _run(cuda, adj, st, feats, gup, L, H, cut_len)
        if not use_touch:
            ops.dbg_csr_algo("auto")
            default = _run(cuda, adj, st, feats, gup, L, H, cut_len)     # short rows: the default IS the sliced path
            assert torch.equal(default[0], new[0]) and torch.equal(default[1], new[1])
    finally:
        ops.dbg_csr_algo("auto")
    # (i) same bits as the kernels it replaces: outputs, input gradient, every weight gradient
    assert torch.equal(new[0], ref[0])
    assert torch.equal(new[1], ref[1])
    for i in range(L):
        # weight gradients are sums over all rows too: dw_kernel cuts them into the same row groups on both paths (same bits);
        # round 6's dww_kernel (hidden layers of cut-0.33 stacks on hybrid rows with whole 16-row tiles) cuts them elsewhere
        if torch.equal(new[2][i], ref[2][i]):
            continue
        assert 0 < i < L - 1 and cut == 0.33 and (B * n) % 16 == 0, f"dW layer {i}"
        assert rel_err(new[2][i], ref[2][i]) < 2e-5, f"dW layer {i}"
    for i in range(L):
        # bias gradients are sums over all rows: the partial sums are grouped per mesh here, per workgroup there
        assert rel_err(new[3][i], ref[3][i]) < 2e-5, f"db layer {i}"
        if i < L - 1 and cut_len < H:
            assert new[3][i][cut_len:].abs().max().item() == 0.0               # dead bias channels stay exactly zero
    # repeatable bit for bit
    for a, b in zip([new[0], new[1], *new[2], *new[3]], [again[0], again[1], *again[2], *again[3]]):
        assert torch.equal(a, b)
    # (ii) the fp64 oracle
    st64 = {k: v.double().requires_grad_(True) for k, v in st.items() if k.startswith("mesh_deform_1")}
    f64 = feats.double().requires_grad_(True)
    out_o = og.gcn(f64, st64, "mesh_deform_1", (adj_o[0], adj_o[1], adj_o[2].double()), L, cut)
    (out_o * gup.double()).sum().backward()
    assert rel_err(new[0], out_o) < 1e-4
    assert_grad_close(new[1][..., :50], f64.grad, "grad_feats")
    for i in range(L):
        assert_grad_close(new[2][i], st64[f"mesh_deform_1.layers.{i}.weight"].grad, f"dW layer {i}")
        assert_grad_close(new[3][i], st64[f"mesh_deform_1.layers.{i}.bias"].grad, f"db layer {i}")


def test_forward_only_call_takes_the_sliced_path_without_a_stash(cuda):
    """No saved activations / sign bytes (policy scoring path): the sliced kernels run with a null sign array."""
    from a3vt_amd import mesh as amesh, ops
    from oracle import gcn as og
    L, H, B = 3, 300, 24
    verts, faces = template("ico3")
    args = make_args(num_GCN_layers=L, hidden_GCN_size=H)
    adj_o, _ = oracle_adj(verts, faces, args)
    st = og.init_state(50, H, L, seed=9)
    g = torch.Generator().manual_seed(2)
    feats = torch.randn(B, verts.shape[0], 50, generator=g) * 0.5
    r, c = amesh.vision_pairs(faces, verts.shape[0])
    adj = ops.DeviceCSR(amesh.CSRAdjacency.from_pairs(r, c, verts.shape[0]), cuda)
    ws = [st[f"mesh_deform_1.layers.{i}.weight"].to(cuda) for i in range(L)]
    bs = [st[f"mesh_deform_1.layers.{i}.bias"].to(cuda) for i in range(L)]
    with torch.no_grad():
        out = ops.gcn_stack(torch.nn.functional.pad(feats, (0, 2)).to(cuda), adj, 50, H, 99, ws, bs)
        out_o = og.gcn(feats.double(), {k: v.double() for k, v in st.items()}, "mesh_deform_1",
                       (adj_o[0], adj_o[1], adj_o[2].double()), L, 0.33)
    assert rel_err(out, out_o) < 1e-4


def test_backward_follows_the_layout_the_forward_recorded(cuda):
    """a C-ABI caller may hand different max-degree hints to the forward and the backward call (csr_max_degree /
    csrT_max_degree, or 0 = unknown).  The forward records the layout it left in the stash; the backward adopts it — here
    the hook flips the shape rule between the two calls and the gradients must not change."""
    from a3vt_amd import mesh as amesh, ops
    from oracle import gcn as og
    L, H, B = 3, 300, 24
    verts, faces = template("ico3")
    st = og.init_state(50, H, L, seed=5)
    g = torch.Generator().manual_seed(21)
    feats = torch.randn(B, verts.shape[0], 50, generator=g) * 0.5
    gup = torch.randn(B, verts.shape[0], 3, generator=g)
    r, c = amesh.vision_pairs(faces, verts.shape[0])
    adj = ops.DeviceCSR(amesh.CSRAdjacency.from_pairs(r, c, verts.shape[0]), cuda)
    ref = _run(cuda, adj, st, feats, gup, L, H, 99)

    ws = [st[f"mesh_deform_1.layers.{i}.weight"].to(cuda).requires_grad_(True) for i in range(L)]
    bs = [st[f"mesh_deform_1.layers.{i}.bias"].to(cuda).requires_grad_(True) for i in range(L)]
    fd = torch.nn.functional.pad(feats, (0, 2)).to(cuda).requires_grad_(True)
    try:
        out = ops.gcn_stack(fd, adj, 50, H, 99, ws, bs)          # forward: channel-sliced path, hybrid rows in the stash
        ops.dbg_csr_algo("rows")                                  # the backward's own rule would now say "row-major"
        (out * gup.to(cuda)).sum().backward()
        torch.cuda.synchronize()
    finally:
        ops.dbg_csr_algo("auto")
    assert torch.equal(fd.grad, ref[1])
    for i in range(L):
        assert torch.equal(ws[i].grad, ref[2][i]) and torch.equal(bs[i].grad, ref[3][i])


@pytest.mark.parametrize("grasps,finger,L,B,cut", [(5, False, 3, 8, 0.33),      # t_g: atlas + 20 charts, N = 2324 (6 vertices per thread)
                                                   (1, False, 3, 8, 0.33),      # the configs[3] graph: 4 charts, N = 1924
                                                   (5, True, 4, 8, 0.33),       # t_p: 5 finger charts, N = 1949 (the scoring caller)
                                                   (1, True, 3, 8, 0.04),       # ONE centre: S absorbs its whole chart ring; 12 aggregated channels
                                                   (5, False, 20, 6, 0.33)])    # the production depth
def test_split_aggregation_against_oracle_and_row_kernels(cuda, grasps, finger, L, B, cut):
    """Round 6: the fused vision + touch matrix (utils.py:75-130) as D^-1 (P + J), J = complete bipartite seam x centres
    (``a3vt_adj_split``, csrc/gcn_csrqs.hip): two per-mesh sums instead of the hub and seam rows.  Another association of the
    same sums, so not bit-identical to the row walk over the full CSR — gated by the fp64 oracle at the north-star tolerance,
    by the row kernels at fp32 rounding, and by bit-repeatability / batch invariance."""
    from a3vt_amd import mesh as amesh, ops
    from oracle import gcn as og
    H = 300
    args = make_args(use_touch=True, finger=finger, num_GCN_layers=L, hidden_GCN_size=H, num_grasps=grasps, cut=cut)
    verts, faces = template("atlas")
    adj_o, _ = oracle_adj(verts, faces, args)
    n = adj_o[0].numel() - 1
    assert B * n >= 12288
    st = og.init_state(50, H, L, seed=5)
    g = torch.Generator().manual_seed(21)
    feats = torch.randn(B, n, 50, generator=g) * 0.5
    gup = torch.randn(B, n, 3, generator=g)
    sv, sf = amesh.load_asset("touch_chart")
    r, c, nn_, _ = amesh.fused_pairs(verts, faces, sf, grasps, finger)
    host = amesh.CSRAdjacency.from_pairs(r, c, nn_)
    adj = ops.DeviceCSR(host, cuda)
    assert adj.split is not None and host.split().max_degree <= 10 and host.split().n_centre == (1 if finger else 4) * grasps
    cut_len = og.cut_length(H, cut)
    ops.path_counts(reset=True)
    new = _run(cuda, adj, st, feats, gup, L, H, cut_len)
    counts = ops.path_counts()
    assert counts["stack_split"] == 1 and counts["stack_quad"] == 1, counts      # the split kernels ran, on hybrid rows
    again = _run(cuda, adj, st, feats, gup, L, H, cut_len)
    for a, b in zip([new[0], new[1], *new[2], *new[3]], [again[0], again[1], *again[2], *again[3]]):
        assert torch.equal(a, b)                                                  # fixed summation order
    # the meshes of a batch do not see each other: the first meshes of a LARGER batch (other `parts`, other workgroups) give
    # the same bits
    big = _run(cuda, adj, st, torch.cat([feats, feats.flip(0)]), torch.cat([gup, gup.flip(0)]), L, H, cut_len)
    assert torch.equal(big[0][:B], new[0]) and torch.equal(big[1][:B], new[1])
    # the row walk over the FULL CSR (hub rows through csr_heavy_kernel): same values to fp32 rounding
    adj.use_split = False
    try:
        ops.dbg_csr_algo("rows")
        ref = _run(cuda, adj, st, feats, gup, L, H, cut_len)
    finally:
        ops.dbg_csr_algo("auto")
        adj.use_split = True
    tol = 2e-5 if L <= 4 else 2e-4
    assert rel_err(new[0], ref[0]) < tol
    assert_grad_close(new[1], ref[1], "grad_feats vs rows", tol=1e-4, l2_tol=1e-4 if L <= 4 else 1e-3)
    # the fp64 oracle (dense reference matrix -> CSR)
    st64 = {k: v.double().requires_grad_(True) for k, v in st.items() if k.startswith("mesh_deform_1")}
    f64 = feats.double().requires_grad_(True)
    out_o = og.gcn(f64, st64, "mesh_deform_1", (adj_o[0], adj_o[1], adj_o[2].double()), L, cut)
    (out_o * gup.double()).sum().backward()
    assert rel_err(new[0], out_o) < 1e-4
    assert_grad_close(new[1][..., :50], f64.grad, "grad_feats")
    for i in range(L):
        assert_grad_close(new[2][i], st64[f"mesh_deform_1.layers.{i}.weight"].grad, f"dW layer {i}")
        assert_grad_close(new[3][i], st64[f"mesh_deform_1.layers.{i}.bias"].grad, f"db layer {i}")
        if i < L - 1 and cut_len < H:
            assert new[3][i][cut_len:].abs().max().item() == 0.0
