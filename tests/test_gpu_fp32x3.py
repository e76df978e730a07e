"""-m gpu: gemm mode 3 ("fp32x3", csrc/gcn_gemm3.hip) — the fp32 products of the hidden layers
(torch.matmul(features, self.weight), reconstruction/vision/model.py:352, and autograd's two products) as six bf16 MFMA
passes on operands split exactly into three bf16 pieces.  Checked: the split itself bit for bit; the stack against the
fp64 oracle with the SAME limits as the exact mode (1e-4 on the outputs, helpers.assert_grad_close on every gradient), on
the shapes that take the split-operand kernels (hidden 300, >= 12 288 rows; row-major and hybrid quad-major layouts); the
error of both modes side by side; shapes the kernels do not take run the exact kernels (bit-equal to mode 0)."""
import numpy as np
import pytest
import torch

from helpers import assert_grad_close, make_args, oracle_adj, rel_err, rel_l2, template

pytestmark = pytest.mark.gpu


def _bf16_bits_to_f64(t):
    return (t.to(torch.int32) << 16).view(torch.float32).double()


def test_split_is_exact(cuda):
    """hi + mid + lo == x bit for bit: normal values of every magnitude, FLT_MAX, values around the bf16 overflow
    threshold, tiny normals; fp32 subnormals to within the bf16 subnormal grid (2^-133)."""
    from a3vt_amd import ops
    g = torch.Generator().manual_seed(0)
    parts = [torch.randn(1 << 16, generator=g),
             torch.randn(1 << 14, generator=g) * 1e30, torch.randn(1 << 14, generator=g) * 1e-25,
             torch.tensor([0.0, -0.0, 1.0, -1.0, 3.4028234663852886e38, -3.4028234663852886e38, 3.3895313892515355e38,
                           3.39e38, 3.4e38, 2.0 ** -100, -2.0 ** -109, 1.0 + 2.0 ** -23, 1.0 - 2.0 ** -24, 255.99998]),
             # every exponent, random mantissas
             (torch.randint(0, 1 << 23, (254 * 64,), generator=g, dtype=torch.int32)
              | (torch.arange(1, 255, dtype=torch.int32).repeat_interleave(64) << 23)).view(torch.float32)]
    x = torch.cat(parts)
    x = torch.cat([x, -x]).to(cuda)
    hi, mid, lo = ops.split3(x)
    s = _bf16_bits_to_f64(hi) + _bf16_bits_to_f64(mid) + _bf16_bits_to_f64(lo)
    xd = x.double()
    big = xd.abs() >= 2.0 ** -110
    assert torch.equal(s[big], xd[big])
    assert (s[~big] - xd[~big]).abs().max().item() < 2.0 ** -133
    assert torch.isfinite(_bf16_bits_to_f64(hi)).all()                      # truncation: FLT_MAX does not round to infinity
    # piece magnitudes: mid <= 2^-7 |x|, lo <= 2^-15 |x| (what the six-product truncation bound rests on)
    nz = xd.abs() >= 2.0 ** -100
    assert (_bf16_bits_to_f64(mid)[nz].abs() <= xd[nz].abs() * 2.0 ** -7).all()
    assert (_bf16_bits_to_f64(lo)[nz].abs() <= xd[nz].abs() * 2.0 ** -15).all()
    # subnormals
    sub = (torch.randint(1, 1 << 23, (4096,), generator=g, dtype=torch.int32)).view(torch.float32).to(cuda)
    hi, mid, lo = ops.split3(sub)
    s = _bf16_bits_to_f64(hi) + _bf16_bits_to_f64(mid) + _bf16_bits_to_f64(lo)
    assert (s - sub.double()).abs().max().item() < 2.0 ** -133


def _adjacency(cuda, tname, use_touch):
    from a3vt_amd import mesh as amesh, ops
    verts, faces = template(tname)
    if use_touch:
        sv, sf = amesh.load_asset("touch_chart")
        r, c, nn_, _ = amesh.fused_pairs(verts, faces, sf, 1, False)
    else:
        r, c = amesh.vision_pairs(faces, verts.shape[0])
        nn_ = verts.shape[0]
    return ops.DeviceCSR(amesh.CSRAdjacency.from_pairs(r, c, nn_), cuda), verts, faces


def _run(cuda, adj, st, feats, gup, L, H, cut_len, mode):
    from a3vt_amd import ops
    ws = [st[f"mesh_deform_1.layers.{i}.weight"].to(cuda).requires_grad_(True) for i in range(L)]
    bs = [st[f"mesh_deform_1.layers.{i}.bias"].to(cuda).requires_grad_(True) for i in range(L)]
    fd = torch.nn.functional.pad(feats, (0, 2)).to(cuda).requires_grad_(True)
    out = ops.gcn_stack(fd, adj, 50, H, cut_len, ws, bs, bf16=mode)
    (out * gup.to(cuda)).sum().backward()
    torch.cuda.synchronize()
    return out.detach(), fd.grad, [w.grad for w in ws], [b.grad for b in bs]


def _oracle(adj_o, st, feats, gup, L, cut):
    from oracle import gcn as og
    st64 = {k: v.double().requires_grad_(True) for k, v in st.items() if k.startswith("mesh_deform_1")}
    f64 = feats.double().requires_grad_(True)
    out_o = og.gcn(f64, st64, "mesh_deform_1", (adj_o[0], adj_o[1], adj_o[2].double()), L, cut)
    (out_o * gup.double()).sum().backward()
    return out_o.detach(), f64.grad, [st64[f"mesh_deform_1.layers.{i}.weight"].grad for i in range(L)], \
        [st64[f"mesh_deform_1.layers.{i}.bias"].grad for i in range(L)]


@pytest.mark.parametrize("tname,use_touch,L,B,cut", [("ico3", False, 4, 24, 0.33), ("ico4", False, 3, 6, 0.33),
                                                      ("atlas", False, 3, 8, 0.33), ("atlas", True, 3, 8, 0.33),
                                                      ("ico3", False, 3, 20, 0.5), ("ico3", False, 3, 20, 0.04),
                                                      ("ico4", False, 20, 6, 0.33), ("ico5", False, 4, 4, 0.33),
                                                      # 2071 / 2082 tiles: 23 / 34 leftover tiles go to the launches' tails
                                                      # (row-major and hybrid layouts)
                                                      ("atlas", True, 3, 17, 0.33), ("ico4", False, 3, 13, 0.33)])
def test_stack_fp32x3_vs_fp64_oracle(cuda, tname, use_touch, L, B, cut):
    """Hidden 300, >= 12 288 rows: the three products of layers 1 .. L-2 run on the split-operand kernels — with the
    hybrid quad-major rows on the plain templates (channel-sliced aggregation), row-major on the touch graph / ico5.
    Same limits as the exact mode's tests; the exact mode is run beside it and its errors printed."""
    from oracle import gcn as og
    H = 300
    args = make_args(use_touch=use_touch, num_GCN_layers=L, hidden_GCN_size=H, num_grasps=1, cut=cut)
    adj, verts, faces = _adjacency(cuda, tname, use_touch)
    adj_o, _ = oracle_adj(verts, faces, args)
    n = adj_o[0].numel() - 1
    assert B * n >= 12288
    st = og.init_state(50, H, L, seed=5)
    g = torch.Generator().manual_seed(21)
    feats = torch.randn(B, n, 50, generator=g) * 0.5
    gup = torch.randn(B, n, 3, generator=g)
    cut_len = og.cut_length(H, cut)
    exact = _run(cuda, adj, st, feats, gup, L, H, cut_len, "fp32")
    split = _run(cuda, adj, st, feats, gup, L, H, cut_len, "fp32x3")
    again = _run(cuda, adj, st, feats, gup, L, H, cut_len, "fp32x3")
    ref = _oracle(adj_o, st, feats, gup, L, cut)
    if L > 2:
        assert not torch.equal(split[0], exact[0])                          # the split-operand kernels DID run
    for a, b in zip([split[0], split[1], *split[2], *split[3]], [again[0], again[1], *again[2], *again[3]]):
        assert torch.equal(a, b)                                            # repeatable bit for bit
    e_x, e_s = rel_err(exact[0], ref[0]), rel_err(split[0], ref[0])
    gw_x = max(rel_l2(exact[2][i], ref[2][i]) for i in range(L))
    gw_s = max(rel_l2(split[2][i], ref[2][i]) for i in range(L))
    print(f"\n[{tname} touch={use_touch} L={L} B={B} cut={cut}] outputs rel-max exact {e_x:.2e} fp32x3 {e_s:.2e}; "
          f"dW rel-L2 (worst layer) exact {gw_x:.2e} fp32x3 {gw_s:.2e}; "
          f"grad_feats rel-L2 exact {rel_l2(exact[1][..., :50], ref[1]):.2e} fp32x3 {rel_l2(split[1][..., :50], ref[1]):.2e}")
    assert e_s < 1e-4
    assert e_s < max(3.0 * e_x, 2e-6)                                       # fp32-level, not bf16-level
    assert_grad_close(split[1][..., :50], ref[1], "grad_feats")
    assert split[1][..., 50:].abs().max().item() == 0.0
    # (ico5: the element tolerance of the exact mode's test at that size, tests/test_gpu_named_sizes.py — sums over 41 k rows
    # see a few ReLU arguments within rounding of zero take the other branch than in float64)
    tol = 5e-3 if tname == "ico5" else 1e-3
    for i in range(L):
        assert_grad_close(split[2][i], ref[2][i], f"dW layer {i}", tol=tol)
        assert_grad_close(split[3][i], ref[3][i], f"db layer {i}", tol=tol)
        if i < L - 1 and cut_len < H:
            assert split[3][i][cut_len:].abs().max().item() == 0.0


@pytest.mark.parametrize("tname,L,H,B", [("ico2", 3, 32, 3), ("ico3", 4, 300, 2), ("ico3", 3, 256, 24), ("ico2", 1, 300, 2)])
def test_shapes_outside_the_split_kernels_run_exact(cuda, tname, L, H, B):
    """Mode 3 is a per-launch choice: narrow, short or few-row stacks run the mode-0 kernels — bit-equal results."""
    from oracle import gcn as og
    args = make_args(num_GCN_layers=L, hidden_GCN_size=H)
    adj, verts, faces = _adjacency(cuda, tname, False)
    st = og.init_state(50, H, L, seed=3)
    g = torch.Generator().manual_seed(11)
    feats = torch.randn(B, verts.shape[0], 50, generator=g) * 0.5
    gup = torch.randn(B, verts.shape[0], 3, generator=g)
    a = _run(cuda, adj, st, feats, gup, L, H, round(H * 0.33), "fp32")
    b = _run(cuda, adj, st, feats, gup, L, H, round(H * 0.33), "fp32x3")
    for u, v in zip([a[0], a[1], *a[2], *a[3]], [b[0], b[1], *b[2], *b[3]]):
        assert torch.equal(u, v)


def test_forward_only_fp32x3(cuda):
    """No stash (policy scoring path): ping-pong outputs, null sign bytes."""
    from a3vt_amd import ops
    from oracle import gcn as og
    L, H, B = 3, 300, 24
    args = make_args(num_GCN_layers=L, hidden_GCN_size=H)
    adj, verts, faces = _adjacency(cuda, "ico3", False)
    adj_o, _ = oracle_adj(verts, faces, args)
    st = og.init_state(50, H, L, seed=9)
    g = torch.Generator().manual_seed(2)
    feats = torch.randn(B, verts.shape[0], 50, generator=g) * 0.5
    ws = [st[f"mesh_deform_1.layers.{i}.weight"].to(cuda) for i in range(L)]
    bs = [st[f"mesh_deform_1.layers.{i}.bias"].to(cuda) for i in range(L)]
    with torch.no_grad():
        out = ops.gcn_stack(torch.nn.functional.pad(feats, (0, 2)).to(cuda), adj, 50, H, 99, ws, bs, bf16="fp32x3")
        out_o = og.gcn(feats.double(), {k: v.double() for k, v in st.items()}, "mesh_deform_1",
                       (adj_o[0], adj_o[1], adj_o[2].double()), L, 0.33)
    assert rel_err(out, out_o) < 1e-4


def test_wide_cut_runs_exact_in_mode3_and_scratch_covers_every_mode(cuda):
    """(i) hidden 300 with cut_len > 212 makes the ReLU-sign rows longer than the 128 bytes the split-operand dX
    epilogue parks per row — the forward used to run on those kernels and every hidden dX then failed ('rowgemm3: unsupported
    call'); the whole stack now runs the exact kernels (bit-equal to mode 0).  (ii) a3vt_gcn_stack_scratch_bytes (no mode
    argument) is enough for every mode (mode 3's three bf16 images per layer are a larger weight slot than mode 0's).
    (iii) gemm modes outside 0..3 are refused."""
    import ctypes
    from a3vt_amd import lib, ops
    from oracle import gcn as og
    L, H, B = 3, 300, 24
    adj, verts, faces = _adjacency(cuda, "ico3", False)
    n = verts.shape[0]
    st = og.init_state(50, H, L, seed=4)
    g = torch.Generator().manual_seed(8)
    feats = torch.randn(B, n, 50, generator=g) * 0.5
    gup = torch.randn(B, n, 3, generator=g)
    ws = [st[f"mesh_deform_1.layers.{i}.weight"].to(cuda) for i in range(L)]
    bs = [st[f"mesh_deform_1.layers.{i}.bias"].to(cuda) for i in range(L)]
    fd = torch.nn.functional.pad(feats, (0, 2)).to(cuda)
    for cut_len in (216, 300):
        # (forward: the exact mode's own backward does not take cuts this wide at hidden 300 — "dw: rows too wide" — so the
        # forward-only callers are the ones that can meet the shape)
        with torch.no_grad():
            a = ops.gcn_stack(fd, adj, 50, H, cut_len, ws, bs, bf16="fp32")
            b = ops.gcn_stack(fd, adj, 50, H, cut_len, ws, bs, bf16="fp32x3")
        assert torch.equal(a, b)
    # the widest cut whose sign rows still fit: trains in both modes (the split-operand kernels run: other low bits)
    a = _run(cuda, adj, st, feats, gup, L, H, 132, "fp32")
    b = _run(cuda, adj, st, feats, gup, L, H, 132, "fp32x3")
    assert not torch.equal(a[0], b[0]) and rel_err(b[0], a[0]) < 1e-5
    Lb = lib.load()
    for (b_, n_, nl) in ((64, 2562, 20), (B, n, L), (2, 162, 3), (8, 10242, 20)):
        for bwd in (0, 1):
            every = Lb.a3vt_gcn_stack_scratch_bytes(b_, n_, 50, H, nl, 99, bwd)
            for mode in range(4):
                assert every >= Lb.a3vt_gcn_stack_scratch_bytes_mode(b_, n_, 50, H, nl, 99, bwd, mode)
    assert Lb.a3vt_gcn_stack_scratch_bytes(64, 2562, 50, H, 20, 99, 1) == \
        max(Lb.a3vt_gcn_stack_scratch_bytes_mode(64, 2562, 50, H, 20, 99, 1, m) for m in range(4))
    with pytest.raises(RuntimeError):
        with torch.no_grad():
            ops.gcn_stack(fd, adj, 50, H, 99, ws, bs, bf16=7)
