"""The trainer's optimizer step as one launch (a3vt_amd/optim.py -> a3vt_adam_step, csrc/adam.hip) against torch.optim.Adam — the
reference's ``optim.Adam(params, lr, weight_decay=0)`` + ``optimizer.step()``, pterotactyl/reconstruction/vision/train.py:64,148."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu

SHAPES = [(1,), (3,), (4097,), (300, 300), (16, 3, 5, 5), (4096,), (8191,), (2, 4096), (300,), (448, 300)]


def _params(dev, seed, offset_views=False):
    g = torch.Generator().manual_seed(seed)
    out = []
    for i, sh in enumerate(SHAPES):
        n = 1
        for d in sh:
            n *= d
        if offset_views and i % 2 == 1:      # storage that does not start on a 16-byte boundary: the scalar path
            base = torch.randn(n + 1, generator=g).to(dev)
            p = base[1:].view(sh)
        else:
            p = torch.randn(sh, generator=g).to(dev)
        out.append(torch.nn.Parameter(p))
    return out


def _grads(params, seed):
    g = torch.Generator().manual_seed(seed)
    for p in params:
        p.grad = (torch.randn(p.shape, generator=g) * 0.1).to(p.device)


@pytest.mark.parametrize("weight_decay", [0.0, 0.01])
@pytest.mark.parametrize("offset_views", [False, True])
def test_library_adam_follows_torch_adam(weight_decay, offset_views):
    from a3vt_amd import optim as aopt
    dev = torch.device("cuda", 0)
    pa, pb = _params(dev, 0, offset_views), _params(dev, 0, offset_views)
    oa = aopt.Adam(pa, lr=3e-4, weight_decay=weight_decay)
    ob = torch.optim.Adam(pb, lr=3e-4, weight_decay=weight_decay, foreach=False, fused=False)
    for step in range(6):
        _grads(pa, 10 + step)
        _grads(pb, 10 + step)
        oa.step()
        ob.step()
    assert oa.library_steps == 6
    for a, b in zip(pa, pb):
        # the same operations in the same order; what is left is the compilers' choice of fused multiply-adds: a few ulps of the update
        torch.testing.assert_close(a.detach(), b.detach(), rtol=5e-7, atol=1e-8)     # (an ulp or two of the parameter itself)
        torch.testing.assert_close(oa.state[a]["exp_avg"], ob.state[b]["exp_avg"], rtol=2e-6, atol=1e-8)
        torch.testing.assert_close(oa.state[a]["exp_avg_sq"], ob.state[b]["exp_avg_sq"], rtol=2e-6, atol=1e-11)
        assert float(oa.state[a]["step"]) == 6.0


def test_library_adam_is_bit_repeatable_and_moves_the_parameters():
    from a3vt_amd import optim as aopt
    dev = torch.device("cuda", 0)
    runs = []
    for _ in range(2):
        ps = _params(dev, 3)
        before = [p.detach().clone() for p in ps]
        opt = aopt.Adam(ps, lr=1e-3)
        for step in range(3):
            _grads(ps, 20 + step)
            opt.step()
        runs.append([p.detach().clone() for p in ps])
        # Adam's first steps move an element by at most lr each, and by lr exactly in the very first one
        for p, b in zip(ps, before):
            d = (p.detach() - b).abs()
            assert float(d.max()) <= 3.01e-3 and float(d.mean()) > 5e-4
    assert all(torch.equal(x, y) for x, y in zip(*runs))


def test_state_dict_is_interchangeable_with_torch_adam():
    """A checkpoint of torch's Adam (plain or fused: the reference's optim file, train.py:213,262) continues in the library's and back."""
    from a3vt_amd import optim as aopt
    dev = torch.device("cuda", 0)
    for fused in (False, True):
        pt = _params(dev, 5)
        ot = torch.optim.Adam(pt, lr=3e-4, fused=fused) if fused else torch.optim.Adam(pt, lr=3e-4, foreach=False)
        for step in range(2):
            _grads(pt, 30 + step)
            ot.step()
        sd = copy.deepcopy(ot.state_dict())
        pl = [torch.nn.Parameter(p.detach().clone()) for p in pt]
        ol = aopt.Adam(pl, lr=3e-4)
        ol.load_state_dict(sd)
        _grads(pt, 40)
        _grads(pl, 40)
        ot.step()
        ol.step()
        assert ol.library_steps == 1
        for a, b in zip(pl, pt):
            torch.testing.assert_close(a.detach(), b.detach(), rtol=5e-7, atol=1e-8)
            assert float(ol.state[a]["step"]) == 3.0
        # and back: the library's state into torch's plain Adam
        pb = [torch.nn.Parameter(p.detach().clone()) for p in pl]
        ob = torch.optim.Adam(pb, lr=3e-4, foreach=False)
        ob.load_state_dict(copy.deepcopy(ol.state_dict()))
        _grads(pl, 41)
        _grads(pb, 41)
        ol.step()
        ob.step()
        for a, b in zip(pl, pb):
            torch.testing.assert_close(a.detach(), b.detach(), rtol=5e-7, atol=1e-8)


def test_parameters_without_gradients_and_late_joiners():
    from a3vt_amd import optim as aopt
    dev = torch.device("cuda", 0)
    pa, pb = _params(dev, 7), _params(dev, 7)
    oa = aopt.Adam(pa, lr=3e-4)
    ob = torch.optim.Adam(pb, lr=3e-4, foreach=False)
    for step in range(4):
        _grads(pa, 50 + step)
        _grads(pb, 50 + step)
        if step < 2:       # the last three tensors get their first gradient at step 2: their own step count from then on
            for ps in (pa, pb):
                for p in ps[-3:]:
                    p.grad = None
        oa.step()
        ob.step()
    for a, b in zip(pa, pb):
        torch.testing.assert_close(a.detach(), b.detach(), rtol=5e-7, atol=1e-8)
    assert float(oa.state[pa[-1]]["step"]) == 2.0 and float(oa.state[pa[0]]["step"]) == 4.0
