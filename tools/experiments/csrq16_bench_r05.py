#!/usr/bin/env python
"""EXPERIMENT: the bf16 channel-sliced aggregation forward (gpurun_variants/liba3vt_CSRQ16.so, `build_variants.sh csrq16`) against
the half-wave row walk it would replace — same outputs bit for bit? how many microseconds per launch at configs[1] sizes?
A3VT_LIB=gpurun_variants/liba3vt_CSRQ16.so python tools/csrq16_bench.py [--batch 64] [--level 4]"""
import argparse
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from a3vt_amd import lib, mesh as amesh  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--level", type=int, default=4)
ap.add_argument("--reps", type=int, default=20)
a = ap.parse_args()
L = lib.load()
vp, i32 = ctypes.c_void_p, ctypes.c_int
L.a3vt_dbg_csrq16_fwd.argtypes = [vp, vp, i32, vp, vp, vp, i32, i32, vp, i32, vp, i32, i32, vp, vp]
L.a3vt_dbg_csr16_fwd.argtypes = [vp, i32, vp, i32, vp, vp, vp, i32, i32, vp, i32, vp, i32, i32, vp]
dev = torch.device("cuda", 0)
verts, faces = amesh.icosphere(a.level)
r, c = amesh.vision_pairs(faces, verts.shape[0])
adj = amesh.CSRAdjacency.from_pairs(r, c, verts.shape[0])
N, B, C, H = verts.shape[0], a.batch, 99, 300
cpad, ldy = 104, 304
mld = (cpad // 4 + (H + 3) // 4 + 1) & ~1
rowptr = torch.from_numpy(adj.rowptr).to(dev)
col = torch.from_numpy(adj.col).to(dev)
val = torch.from_numpy(adj.val).to(dev)
torch.manual_seed(0)
za = (torch.randn(B, N, cpad, device=dev) * 0.5).to(torch.bfloat16)
za[..., C:] = 0
zq = za.view(B, N, cpad // 8, 8).permute(0, 2, 1, 3).contiguous()
bias = torch.randn(H, device=dev) * 0.1
st = torch.cuda.current_stream().cuda_stream
ell = torch.empty(N * 17 + 64, dtype=torch.int32, device=dev)
outs = []
for which in ("rows", "sliced"):
    y = torch.zeros(B * N, ldy, dtype=torch.bfloat16, device=dev)
    mk = torch.zeros(((B * N + 31) // 32 * 32) * mld, dtype=torch.uint8, device=dev)

    def run():
        if which == "rows":
            rc = L.a3vt_dbg_csr16_fwd(za.data_ptr(), cpad, bias.data_ptr(), C, rowptr.data_ptr(), col.data_ptr(), val.data_ptr(), N, B,
                                      y.data_ptr(), ldy, mk.data_ptr(), mld, 1, st)
        else:
            rc = L.a3vt_dbg_csrq16_fwd(zq.data_ptr(), bias.data_ptr(), C, rowptr.data_ptr(), col.data_ptr(), val.data_ptr(), N, B,
                                       y.data_ptr(), ldy, mk.data_ptr(), mld, 1, ell.data_ptr(), st)
        assert rc == 0, L.a3vt_last_error()
    run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.reps):
        run()
    e1.record()
    torch.cuda.synchronize()
    print(f"{which:7s} {1e3 * e0.elapsed_time(e1) / a.reps:8.1f} us per call (B={B}, N={N}; the sliced call includes the 2 us index-image build)")
    outs.append((y.clone(), mk.clone()))
print("outputs identical:", torch.equal(outs[0][0][:, :cpad].view(torch.int16), outs[1][0][:, :cpad].view(torch.int16)),
      " sign bytes identical:", torch.equal(outs[0][1].view(-1, mld)[:B * N, :cpad // 4], outs[1][1].view(-1, mld)[:B * N, :cpad // 4]))
