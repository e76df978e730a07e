import sys, torch
sys.path.insert(0, '/root/repo')
from a3vt_amd import lib as _lib, ops
cuda = torch.device('cuda', 0)
g = torch.Generator().manual_seed(7)
B, N, ld = 3, 517, 52
verts = ((torch.rand(B, N, 3, generator=g) - 0.5) * 1.2).to(cuda)
mask = torch.randint(0, 4, (B, N, 1), generator=g).float().to(cuda)
packed = (torch.randn(_lib.load().a3vt_posenc_param_count(50), generator=g) * 0.3).to(cuda)
ops.dbg_posenc_fwd_algo("threads")
ref = ops.PosEncMaskFn.apply(verts, mask, packed, 50, ld).clone()
ops.dbg_posenc_fwd_algo("auto")
new = ops.PosEncMaskFn.apply(verts, mask, packed, 50, ld).clone()
d = (new - ref).abs()
print("max abs diff", d.max().item(), "ref max", ref.abs().max().item(), "frac differing", (d > 0).float().mean().item())
rel = d / ref.abs().clamp_min(1e-30)
print("max rel", rel[ref.abs() > 1e-3].max().item())
