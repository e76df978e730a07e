#!/usr/bin/env python
"""Round-4 check: grad_verts of the rewritten posenc_bwd against the round-3 kernel, bit for bit (run once with
A3VT_LIB=<library with the old posenc.hip> and once with the shipped library; the second run compares)."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from a3vt_amd import lib as _lib, ops  # noqa: E402

dev = torch.device("cuda", 0)
out = os.path.join(ROOT, "gpurun_out", "pe_bits.pt")
res = {}
for B, N in ((3, 517), (16, 2563), (64, 2562)):
    g = torch.Generator().manual_seed(11)
    verts = ((torch.rand(B, N, 3, generator=g) - 0.5) * 1.2).to(dev).requires_grad_(True)
    mask = torch.randint(0, 4, (B, N, 1), generator=g).float().to(dev)
    packed = (torch.randn(_lib.load().a3vt_posenc_param_count(50), generator=g) * 0.3).to(dev).requires_grad_(True)
    gout = torch.randn(B, N, 52, generator=g).to(dev)
    gout[..., 50:] = 0
    f = ops.PosEncMaskFn.apply(verts, mask, packed, 50, 52)
    gv, gp = torch.autograd.grad(f, (verts, packed), gout)
    res[(B, N)] = (f.detach().cpu(), gv.cpu(), gp.cpu())
if os.path.exists(out):
    old = torch.load(out)
    for k in res:
        f0, gv0, gp0 = old[k]
        f1, gv1, gp1 = res[k]
        print(k, "forward equal", torch.equal(f0, f1), "| grad_verts equal", torch.equal(gv0, gv1), "max abs diff",
              (gv0 - gv1).abs().max().item(), "| grad_params rel", ((gp0 - gp1).abs().max() / gp0.abs().max()).item())
else:
    torch.save(res, out)
    print("saved", out)
