#!/usr/bin/env python
"""Micro-benchmark of one GCN stack (fwd+bwd) at the headline shape; prints the mean device time per MFMA launch
class (HIP events via a3vt_profile_*).  Development aid:  python tools/stack_bench.py [--layers 8] [--reps 5]"""
import argparse
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))  # repo root
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

p = argparse.ArgumentParser()
p.add_argument("--layers", type=int, default=8)
p.add_argument("--hidden", type=int, default=300)
p.add_argument("--batch", type=int, default=64)
p.add_argument("--level", type=int, default=4)
p.add_argument("--reps", type=int, default=5)
p.add_argument("--precision", default="fp32", choices=["fp32", "bf16", "bf16s", "fp32x3"])
p.add_argument("--no-profile", action="store_true", help="no per-launch HIP events (they cost ~1 us per launch): only the time of a whole forward + backward call")
p.add_argument("--csr-algo", default="auto", choices=["auto", "rows", "sliced"], help="ops.dbg_csr_algo: which aggregation kernels")
p.add_argument("--subdivision-order", action="store_true", help="icosphere vertices in subdivision order (poor locality)")
a = p.parse_args()

from a3vt_amd import lib, mesh as amesh, ops  # noqa: E402

dev = torch.device("cuda", 0)
verts, faces = amesh.icosphere(a.level, spatial_order=not a.subdivision_order)
r, c = amesh.vision_pairs(faces, verts.shape[0])
adj = ops.DeviceCSR(amesh.CSRAdjacency.from_pairs(r, c, verts.shape[0]), dev)
from types import SimpleNamespace  # noqa: E402
from a3vt_amd.pterotactyl.reconstruction.vision import model  # noqa: E402
torch.manual_seed(0)
gcn = model.GCN(50, SimpleNamespace(num_GCN_layers=a.layers, hidden_GCN_size=a.hidden, cut=0.33)).to(dev)   # reference init
ws = [l.weight for l in gcn.layers]
bs = [l.bias for l in gcn.layers]
feats = torch.zeros(a.batch, verts.shape[0], 52, device=dev)
feats[..., :50] = torch.randn(a.batch, verts.shape[0], 50, device=dev) * 0.5
feats.requires_grad_(True)
gup = torch.randn(a.batch, verts.shape[0], 3, device=dev)
L = lib.load()
ops.dbg_csr_algo(a.csr_algo)


def run():
    out = ops.gcn_stack(feats, adj, 50, a.hidden, round(a.hidden * 0.33), ws, bs, bf16=a.precision)
    out.backward(gup)


run()
torch.cuda.synchronize()
L.a3vt_profile_enable(0 if a.no_profile else 1)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(a.reps):
    run()
e1.record()
torch.cuda.synchronize()
if a.no_profile:
    print(f"stack fwd+bwd: {e0.elapsed_time(e1) / a.reps:.3f} ms per call (no per-launch events)")
    raise SystemExit(0)
tot = (ctypes.c_double * 3)()
cnt = (ctypes.c_int * 3)()
lib.check(L.a3vt_profile_read(tot, cnt), "profile_read")
M = a.batch * verts.shape[0]
flop = 2.0 * M * a.hidden * a.hidden
for name, i in (("fwd Z=XW", 0), ("bwd dX", 1), ("bwd dW", 2)):
    ms = tot[i] / max(cnt[i], 1)
    print(f"{name:10s} launches {cnt[i]:4d}  mean {ms * 1e3:8.1f} us   {flop / (ms * 1e-3) / 1e12 if ms else 0:6.1f} TFLOP/s (hidden x hidden launches dominate)")
print(f"stack fwd+bwd WITH a HIP event pair around every launch (not a step time: --no-profile): {e0.elapsed_time(e1) / a.reps:.2f} ms per call  (MFMA classes: {sum(tot) / a.reps:.2f} ms)")
