#!/usr/bin/env python
"""Per-wave timeline of nn_query_kernel (library liba3vt_NN_TRACE.so built by `tools/build_variants.sh nn`): when every wave started and ended
(s_memrealtime, 100 MHz), how many groups it evaluated — is the launch bound by its throughput or by its slowest waves?
A3VT_LIB=gpurun_variants/liba3vt_NN_TRACE.so python tools/nn_trace.py [--shape 3x64x10000] [--geometry bench]"""
import argparse
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from a3vt_amd import lib, ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--shape", default="3x64x10000")
ap.add_argument("--geometry", default="bench", choices=["synthetic", "bench"])
args = ap.parse_args()
L = lib.load()
dev = torch.device("cuda", 0)
torch.manual_seed(0)
draws, B, N = (int(v) for v in args.shape.split("x"))


def surface(*shape, radii):
    u = torch.randn(*shape, 3, device=dev)
    return u / u.norm(dim=-1, keepdim=True) * torch.tensor(radii, device=dev)


if args.geometry == "bench":
    from a3vt_amd.synthetic import gt_cloud
    x, y = surface(draws, B, N, radii=(0.25, 0.25, 0.25)), gt_cloud(B, N, 0).to(dev)
else:
    x, y = surface(draws, B, N, radii=(0.4, 0.4, 0.4)) + 0.05, surface(B, N, radii=(0.5, 0.3, 0.2))
buf = np.zeros((1 << 16, 4), dtype=np.uint64)
n = ctypes.c_uint(0)
ops.chamfer_nn(x, y, algo="pruned")
torch.cuda.synchronize()
L.a3vt_dbg_nn_trace(buf.ctypes.data_as(ctypes.c_void_p), ctypes.byref(n))
ops.chamfer_nn(x, y, algo="pruned")
torch.cuda.synchronize()
L.a3vt_dbg_nn_trace(buf.ctypes.data_as(ctypes.c_void_p), ctypes.byref(n))
k = min(n.value, 1 << 16)
t0, t1 = buf[:k, 0].astype(np.int64), buf[:k, 1].astype(np.int64)
grp = (buf[:k, 3] >> np.uint64(32)).astype(np.int64)
base = t0.min()
span = (t1.max() - base) / 100.0          # microseconds
dur = (t1 - t0) / 100.0
print(f"{args.shape} {args.geometry}: {n.value} waves ({k} recorded), launch span {span:.0f} us; wave duration mean {dur.mean():.1f} "
      f"median {np.median(dur):.1f} p90 {np.percentile(dur, 90):.1f} p99 {np.percentile(dur, 99):.1f} max {dur.max():.1f} us; "
      f"groups mean {grp.mean():.1f} p99 {np.percentile(grp, 99):.0f} max {grp.max()}")
# waves in flight over time (20 bins)
edges = np.linspace(0, span, 21)
for i in range(20):
    a, b = edges[i], edges[i + 1]
    s, e = (t0 - base) / 100.0, (t1 - base) / 100.0
    ov = np.clip(np.minimum(e, b) - np.maximum(s, a), 0, None).sum() / (b - a)
    started = ((s >= a) & (s < b)).sum()
    print(f"  {a:7.0f}-{b:7.0f} us: {ov:7.0f} waves in flight ({ov / 1024:.2f} per SIMD), {started} started")
heavy = np.argsort(-dur)[:5]
for h in heavy:
    print(f"  slow wave: start {(t0[h] - base) / 100.0:.0f} us, {dur[h]:.0f} us, {grp[h]} groups, pair {int(buf[h, 2]) >> 16} block {int(buf[h, 2]) & 0xffff}")
print(f"  correlation(duration, groups) = {np.corrcoef(dur, grp)[0, 1]:.3f}; us per group (fit) = {np.polyfit(grp, dur, 1)[0]:.3f}, intercept {np.polyfit(grp, dur, 1)[1]:.1f} us")
