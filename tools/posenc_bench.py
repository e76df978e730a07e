#!/usr/bin/env python
"""Developer aid (GPU box): the fused vertex-feature encoder (csrc/posenc.hip, I = 50) at the benchmark's 64 x 2562 rows:
microseconds per forward and backward call (HIP events around 50 calls each).  A3VT_LIB selects a variant build."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def main():
    from a3vt_amd import lib as _lib, ops
    L = _lib.load()
    dev = torch.device("cuda", 0)
    B, N, I, ld = 64, 2562, 50, int(sys.argv[1]) if len(sys.argv) > 1 else 56
    g = torch.Generator().manual_seed(0)
    verts = ((torch.rand(B, N, 3, generator=g) - 0.5)).to(dev).requires_grad_(True)
    mask = torch.full((B, N, 1), 3.0, device=dev)
    packed = (torch.randn(L.a3vt_posenc_param_count(I), generator=g) * 0.1).to(dev).requires_grad_(True)
    gout = torch.randn(B, N, ld, generator=g).to(dev)
    gout[..., I:] = 0
    out = ops.PosEncMaskFn.apply(verts, mask, packed, I, ld)
    ctx_run = lambda: torch.autograd.grad(out, (verts, packed), gout, retain_graph=True)
    for name, fn in (("fwd", lambda: ops.PosEncMaskFn.apply(verts, mask, packed, I, ld)), ("bwd", ctx_run)):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            fn()
        e1.record()
        torch.cuda.synchronize()
        print(f"posenc {name}: {e0.elapsed_time(e1) / 50 * 1e3:8.1f} us per call (bwd includes its slab reduce)")


if __name__ == "__main__":
    main()
