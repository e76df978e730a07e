"""Training BatchNorm2d + ReLU of the image pyramid as one operator (csrc/bnrelu.hip, a3vt_bnrelu_fwd / _bwd) against
torch's own fp32 batch_norm + relu on the same bf16-rounded inputs (reference: CNN_layer, vision/model.py:15-23)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _reference(x, gamma, beta, rm, rv, eps, momentum, gy):
    """fp32 BatchNorm2d (training) + ReLU by torch ops on the bf16 VALUES of x; gradients for the bf16 values of gy."""
    xf = x.float().detach().requires_grad_(True)
    g = gamma.detach().clone().requires_grad_(True)
    b = beta.detach().clone().requires_grad_(True)
    rm, rv = rm.clone(), rv.clone()
    y = torch.relu(torch.nn.functional.batch_norm(xf, rm, rv, g, b, True, momentum, eps))
    y.backward(gy.float())
    return y.detach(), rm, rv, xf.grad, g.grad, b.grad


# the pyramid's channel counts (3, 16 ... 256) on odd map sizes; 3 x 17 x 19 x 3 elements is not a multiple of 8 (ragged last piece)
@pytest.mark.parametrize("shape", [(3, 3, 17, 19), (64, 3, 62, 62), (5, 16, 31, 29), (64, 16, 60, 60), (7, 32, 13, 11),
                                   (4, 64, 27, 27), (2, 128, 9, 9), (64, 256, 3, 3), (1, 48, 5, 7)])
def test_bnrelu_against_torch(shape):
    from a3vt_amd import ops
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(sum(shape))
    B, C, H, W = shape
    x = (torch.randn(shape, generator=g) * 1.7 + torch.randn(1, C, 1, 1, generator=g)).to(dev).to(torch.bfloat16)
    x = x.contiguous(memory_format=torch.channels_last)
    gamma = (torch.rand(C, generator=g) + 0.5).to(dev)
    beta = (torch.randn(C, generator=g) * 0.3).to(dev)
    rm0 = (torch.randn(C, generator=g) * 0.1).to(dev)
    rv0 = (torch.rand(C, generator=g) + 0.5).to(dev)
    gy = torch.randn(shape, generator=g).to(dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    eps, momentum = 1e-5, 0.1
    y_ref, rm_ref, rv_ref, gx_ref, gg_ref, gb_ref = _reference(x, gamma, beta, rm0, rv0, eps, momentum, gy)

    outs = []
    for _ in range(2):   # twice: the second call finds the arrival counter as the first left it, and must repeat bit for bit
        xx = x.clone().requires_grad_(True)
        gm, bt = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
        rm, rv, nb = rm0.clone(), rv0.clone(), torch.tensor(5, dtype=torch.int64, device=dev)
        y = ops.BNReLUFn.apply(xx, gm, bt, rm, rv, nb, eps, momentum)
        y.backward(gy)
        outs.append((y.detach(), rm, rv, nb, xx.grad, gm.grad, bt.grad))
    for a, b in zip(outs[0], outs[1]):
        assert torch.equal(a, b)
    y, rm, rv, nb, gx, gg, gb = outs[0]
    assert y.dtype == torch.bfloat16 and y.is_contiguous(memory_format=torch.channels_last) and gx.dtype == torch.bfloat16
    assert int(nb) == 6
    # forward: the fp32 value rounded to bf16 (one rounding: 2^-8 relative), a different but equivalent fp32 expression inside
    err = (y.float() - y_ref).abs()
    assert bool((err <= y_ref.abs() * 2.0 ** -7 + 2e-5).all()), float(err.max())
    torch.testing.assert_close(rm, rm_ref, rtol=2e-5, atol=2e-6)
    torch.testing.assert_close(rv, rv_ref, rtol=2e-5, atol=2e-6)
    # backward: fp32 sums in another order; ReLU arguments within rounding of zero may take the other branch
    rel = lambda a, b: float((a.float() - b).norm() / b.norm().clamp_min(1e-20))  # noqa: E731
    assert rel(gg, gg_ref) < 2e-3 and rel(gb, gb_ref) < 2e-3, (rel(gg, gg_ref), rel(gb, gb_ref))
    assert rel(gx, gx_ref) < 6e-3, rel(gx, gx_ref)


def test_bnrelu_propagates_nan_and_checks_arguments():
    from a3vt_amd import ops
    dev = torch.device("cuda", 0)
    x = torch.randn(4, 16, 6, 6, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    x[1, 3, 2, 2] = float("nan")
    one, zero = torch.ones(16, device=dev), torch.zeros(16, device=dev)
    y = ops.BNReLUFn.apply(x, one, zero, zero.clone(), one.clone(), None, 1e-5, 0.1)
    assert bool(torch.isnan(y[:, 3]).all()) and bool(torch.isfinite(y[:, 2]).all())   # the channel's statistics are NaN
    with pytest.raises(RuntimeError):
        ops.BNReLUFn.apply(x.float(), one, zero, None, None, None, 1e-5, 0.1)          # fp32 maps stay on torch's kernels
    with pytest.raises(RuntimeError):
        ops.BNReLUFn.apply(x, one, zero, zero.clone(), None, None, 1e-5, 0.1)           # running_mean without running_var


def test_image_encoder_fused_matches_miopen_branch():
    """The bf16 channels-last image pyramid with the fused operator against the same pyramid on MIOpen's BatchNorm + torch's
    ReLU, both measured against the fp32 pyramid: the two bf16 pipelines differ from each other by what each differs from fp32
    (bf16 activations through 13 normalised layers); running statistics and counters alike."""
    from types import SimpleNamespace
    from a3vt_amd.pterotactyl.reconstruction.vision import model
    dev = torch.device("cuda", 0)
    mk = lambda **kw: SimpleNamespace(CNN_ker_size=5, num_CNN_blocks=6, layers_per_block=3, gemm_precision="bf16s", **kw)  # noqa: E731
    torch.manual_seed(3)
    enc_a = model.Image_Encoder(mk()).to(dev).train()
    enc_b = model.Image_Encoder(mk()).to(dev).train()
    enc_f = model.Image_Encoder(mk(cnn_precision="fp32")).to(dev).train()
    enc_b.load_state_dict(enc_a.state_dict())
    enc_f.load_state_dict(enc_a.state_dict())
    img = torch.rand(6, 3, 256, 256, device=dev)

    def run(enc, fused):
        model.Image_Encoder.fused_bn_relu = fused
        try:
            maps = enc(img)
            loss = sum((m.float() * m.float()).mean() for m in maps)
            loss.backward()
        finally:
            model.Image_Encoder.fused_bn_relu = True
        return [m.detach().float() for m in maps]

    from a3vt_amd import ops
    ops.STATS["bias_grad_from_bnrelu"] = 0
    ma = run(enc_a, True)
    # 13 convolutions feed a BatchNorm; the outputs of layers 9 and 12 are also pooled (their gradient is a sum of two): 11 of the
    # 13 bias gradients come out of the BatchNorm backward's dx launch, the others (and the last layer's) from bias_grad_nhwc
    assert ops.STATS["bias_grad_from_bnrelu"] == 11, ops.STATS
    mb, mf = run(enc_b, False), run(enc_f, False)
    rel = lambda u, v: float((u - v).norm() / v.norm().clamp_min(1e-20))  # noqa: E731
    ea, eb = [rel(a, f) for a, f in zip(ma, mf)], [rel(b, f) for b, f in zip(mb, mf)]
    print(f"[image pyramid, bf16 vs fp32 maps] fused {ea}  MIOpen BatchNorm + ReLU {eb}")
    for x, y in zip(ea, eb):
        assert x < 1.5 * y + 5e-3 and x < 0.1, (ea, eb)
    sa, sb = enc_a.state_dict(), enc_b.state_dict()
    for k in sa:
        if "running" in k:
            torch.testing.assert_close(sa[k], sb[k], rtol=5e-2, atol=5e-3)
        if "num_batches" in k:   # (the pyramid stops at its 14th layer on a 256-pixel image: the layers behind it stay at 0)
            assert int(sa[k]) == int(sb[k]) == (1 if int(k.split(".")[1]) < 14 else 0), k
    # parameter gradients against the fp32 pyramid's.  The bias of a convolution that feeds a BatchNorm has a gradient of zero
    # in exact arithmetic (the normalisation removes the shift): those tensors hold rounding noise in all three pyramids and
    # are left out (norm below 1e-3 of the largest); on the rest the fused branch is as close to fp32 as MIOpen's
    norms = {n: float(p.grad.norm()) for n, p in enc_f.named_parameters() if p.grad is not None}
    floor = 1e-3 * max(norms.values())
    ga, gb = [], []
    for (n, pa), (_, pb), (_, pf) in zip(enc_a.named_parameters(), enc_b.named_parameters(), enc_f.named_parameters()):
        if int(n.split(".")[1]) >= 14:
            assert pa.grad is None and pb.grad is None
            continue
        assert pa.grad is not None and pb.grad is not None and pf.grad is not None, n
        if norms[n] < floor:
            continue
        ga.append(rel(pa.grad, pf.grad))
        gb.append(rel(pb.grad, pf.grad))
    ma_, mb_ = sorted(ga)[len(ga) // 2], sorted(gb)[len(gb) // 2]
    print(f"[image pyramid, bf16 vs fp32 parameter gradients, {len(ga)} tensors] worst fused {max(ga):.3f}  MIOpen {max(gb):.3f}; median {ma_:.3f} / {mb_:.3f}")
    assert ma_ < 1.5 * mb_ + 0.02 and max(ga) < 2.0 * max(gb) + 0.05, (ma_, mb_, max(ga), max(gb))

def test_batched_weight_cast_matches_torch_copies():
    """a3vt_cast_weights_bf16 (one launch for all conv weights and biases) against torch's own bf16 channels-last copies."""
    from a3vt_amd import ops
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(5)
    shapes = [(3, 3, 5, 5), (16, 3, 5, 5), (32, 16, 5, 5), (7, 5, 3, 2), (256, 128, 5, 5)]
    ws = [torch.randn(s, generator=g).to(dev) for s in shapes]
    bs = [torch.randn(s[0], generator=g).to(dev) for s in shapes]
    ops.invalidate_bf16_copies()
    ops.prefetch_bf16_copies([(w, True) for w in ws] + [(b, False) for b in bs])
    for w in ws:
        c = ops._bf16_copy(w, True)
        assert c.dtype == torch.bfloat16 and c.is_contiguous(memory_format=torch.channels_last)
        assert torch.equal(c, w.to(torch.bfloat16).contiguous(memory_format=torch.channels_last))
    for b in bs:
        assert torch.equal(ops._bf16_copy(b, False), b.to(torch.bfloat16))
    # a stale entry (in-place update) is refreshed by the next prefetch, fresh ones are left alone
    ws[1].mul_(2.0)
    keep = ops._bf16_copy(ws[0], True)
    ops.prefetch_bf16_copies([(w, True) for w in ws])
    assert ops._bf16_copy(ws[0], True) is keep
    assert torch.equal(ops._bf16_copy(ws[1], True), ws[1].to(torch.bfloat16).contiguous(memory_format=torch.channels_last))


def test_conv_bias_gradient_from_the_batchnorm_backward():
    """Conv2d -> BatchNorm2d -> ReLU: the convolution's bias gradient is the column sum of the BatchNorm's input gradient, which
    a3vt_bnrelu_bwd forms while writing it; against bias_grad_nhwc reading the same bf16 tensor again."""
    from a3vt_amd import ops
    dev = torch.device("cuda", 0)
    torch.manual_seed(11)
    for cin, cout, hw, B in ((3, 16, 40, 5), (16, 32, 21, 8), (3, 3, 33, 3)):
        x = torch.randn(B, cin, hw, hw, device=dev)
        w = (torch.randn(cout, cin, 5, 5, device=dev) * 0.1).requires_grad_(True)
        b = (torch.randn(cout, device=dev) * 0.1).requires_grad_(True)
        gamma, beta = (torch.rand(cout, device=dev) + 0.5).requires_grad_(True), torch.zeros(cout, device=dev, requires_grad=True)
        grads = []
        for fast in (True, False):
            ops.STATS["bias_grad_from_bnrelu"] = 0
            w.grad = b.grad = None
            h = ops.ConvNHWCFn.apply(x, w, b, [1, 1], [1, 1])
            h2 = h if fast else h * 1.0          # (a second consumer-side node: the gradient that reaches the convolution is a new tensor)
            y = ops.BNReLUFn.apply(h2, gamma, beta, None, None, None, 1e-5, 0.1)
            (y.float() * torch.linspace(-1, 1, y.numel(), device=dev).reshape(y.shape)).sum().backward()
            assert ops.STATS["bias_grad_from_bnrelu"] == (1 if fast else 0)
            grads.append((b.grad.clone(), w.grad.clone()))
        torch.testing.assert_close(grads[0][0], grads[1][0], rtol=1e-3, atol=5e-3)
        torch.testing.assert_close(grads[0][1], grads[1][1], rtol=2e-2, atol=2e-2)   # (MIOpen's weight gradient: fp32 atomics, bf16 result)


def test_bnrelu_pre_bias_moves_only_the_running_mean():
    """A Conv2d bias left to the BatchNorm behind it (``pre_bias``): batch statistics remove a per-channel shift, so the output
    and the gradients are those without it, bit for bit, and ``running_mean`` is the one of ``x + bias``."""
    from a3vt_amd import ops
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(9)
    x = torch.randn(6, 32, 19, 17, generator=g).to(dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    gamma, beta = (torch.rand(32, generator=g) + 0.5).to(dev), (torch.randn(32, generator=g) * 0.2).to(dev)
    pb = (torch.randn(32, generator=g) * 0.7).to(dev)
    gy = torch.randn(x.shape, generator=g).to(dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    outs = []
    for bias in (None, pb):
        xx = x.clone().requires_grad_(True)
        rm, rv = torch.zeros(32, device=dev), torch.ones(32, device=dev)
        y = ops.BNReLUFn.apply(xx, gamma, beta, rm, rv, None, 1e-5, 0.1, bias)
        y.backward(gy)
        outs.append((y.detach(), xx.grad, rm, rv))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1]) and torch.equal(outs[0][3], outs[1][3])
    torch.testing.assert_close(outs[1][2], outs[0][2] + 0.1 * pb, rtol=1e-6, atol=1e-7)
    ref_mean = (x.float() + pb.view(1, -1, 1, 1)).mean(dim=(0, 2, 3))
    torch.testing.assert_close(outs[1][2], 0.1 * ref_mean, rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("shape,cout,stride", [((2, 16, 20, 20), 16, 1), ((3, 16, 37, 29), 16, 1), ((64, 16, 126, 126), 16, 1),
                                               ((1, 16, 3, 5), 16, 1), ((3, 32, 23, 31), 32, 1), ((64, 32, 60, 60), 32, 1),
                                               ((2, 16, 37, 30), 32, 2), ((64, 16, 122, 122), 32, 2),
                                               ((2, 3, 40, 33), 3, 1), ((64, 3, 256, 256), 3, 1), ((3, 3, 41, 38), 16, 2), ((2, 3, 40, 34), 16, 2), ((64, 3, 254, 254), 16, 2)])
def test_conv5_against_torch(shape, cout, stride):
    """a3vt_conv5_nhwc (csrc/conv5.hip: layers 2-6 of the pyramid — 16 -> 16 and 32 -> 32 at stride 1, 16 -> 32 at stride 2; 5 x 5,
    padding 1, channels-last bf16 maps) against torch's fp32 convolution of the same bf16 values: forward with and without the
    bias, and for the stride-1 shapes the input gradient (the same kernel on the output gradient with flipped weights, padding
    3), and the weight gradient (a3vt_conv5_weight_grad)."""
    from a3vt_amd import ops
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(sum(shape) + cout)
    cin = shape[1]
    x = torch.randn(shape, generator=g).to(dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(cout, cin, 5, 5, generator=g) * 0.08).to(dev)
    b = (torch.randn(cout, generator=g) * 0.3).to(dev)
    wb = w.to(torch.bfloat16).float()          # the values the kernel multiplies
    ref = torch.nn.functional.conv2d(x.float(), wb, b, stride=stride, padding=1)
    gy = torch.randn(ref.shape, generator=torch.Generator().manual_seed(5)).to(dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    tol = lambda r: r.abs() * 2.0 ** -7 + 1e-2 * float(r.abs().mean())     # noqa: E731  one bf16 rounding of an fp32 sum of 400-800 terms
    img_f = ops._conv5_image(w, 0)
    y = ops.conv5_nhwc(x, img_f, b, cout, stride, 1)
    y0 = ops.conv5_nhwc(x, img_f, None, cout, stride, 1)
    assert y.shape == ref.shape and y.is_contiguous(memory_format=torch.channels_last)
    assert bool(((y.float() - ref).abs() <= tol(ref)).all()), float((y.float() - ref).abs().max())
    assert bool(((y0.float() - (ref - b.view(1, -1, 1, 1))).abs() <= tol(ref)).all())
    if stride == 1 and cin != 3:
        ref_gx = torch.nn.grad.conv2d_input(x.shape, wb, gy.float(), padding=1)
        gx = ops.conv5_nhwc(gy, ops._conv5_image(w, 1), None, cin, 1, 3)
        assert gx.shape == x.shape
        assert bool(((gx.float() - ref_gx).abs() <= tol(ref_gx)).all()), float((gx.float() - ref_gx).abs().max())
    if True:
        # the weight gradient (a3vt_conv5_weight_grad): fp32 sums of exact bf16 products in a fixed order — against torch's fp32
        # gradient of the same values, and bit-repeatable
        ref_gw = torch.nn.grad.conv2d_weight(x.float(), w.shape, gy.float(), stride=stride, padding=1)
        gw = ops._conv5_weight_grad(x, gy, w, [stride, stride], [1, 1], None)
        gw2 = ops._conv5_weight_grad(x, gy, w, [stride, stride], [1, 1], None)
        assert gw.shape == w.shape and gw.dtype == torch.float32 and torch.equal(gw, gw2)
        err = float((gw - ref_gw).norm() / ref_gw.norm())
        assert err < 2e-4, err
        assert float((gw - ref_gw).abs().max()) < 2e-3 * float(ref_gw.abs().max()) + 1e-4 * float(ref_gw.norm()) / ref_gw.numel() ** 0.5
    # through the autograd function: same forward, same input gradient, the weight gradient against MIOpen's own
    res = []
    for own in (True, False):
        ops.LIBRARY_CONV5[0] = own
        try:
            xx = x.clone().requires_grad_(True)
            ww, bb = w.clone().requires_grad_(True), b.clone().requires_grad_(True)
            out = ops.ConvNHWCFn.apply(xx, ww, bb, [stride, stride], [1, 1])
            out.backward(gy)
            res.append((out.detach().float(), xx.grad.float(), ww.grad, bb.grad))
        finally:
            ops.LIBRARY_CONV5[0] = True
    rel = lambda a, c: float((a - c).norm() / c.norm().clamp_min(1e-20))  # noqa: E731
    # (two bf16 roundings of fp32 sums formed in two orders)
    assert rel(res[0][0], res[1][0]) < 8e-3 and rel(res[0][1], res[1][1]) < 8e-3, (rel(res[0][0], res[1][0]), rel(res[0][1], res[1][1]))
    assert rel(res[0][2], res[1][2]) < 2e-2 and rel(res[0][3], res[1][3]) < 1e-3
