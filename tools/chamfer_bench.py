#!/usr/bin/env python
"""Time the Chamfer forward (two nearest-neighbour launches + reduce) at the headline shape: 3 draws x 64 clouds x
10,000 points against 64 x 10,000.  Development aid:  python tools/chamfer_bench.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from a3vt_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)
torch.manual_seed(0)
x = torch.randn(3, 64, 10000, 3, device=dev) * 0.1
y = torch.randn(64, 10000, 3, device=dev) * 0.1
single = os.environ.get("SINGLE", "1") == "1"   # 0: the two-pass search
out = ops.chamfer_nn(x, y, single_pass=single)
torch.cuda.synchronize()
reps = int(os.environ.get("REPS", "10"))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    ops.chamfer_nn(x, y, single_pass=single)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
pairs = 2.0 * 3 * 64 * 10000 * 10000
print(f"chamfer fwd {ms:.3f} ms per call  ({pairs / ms / 1e9:.1f} T pair evaluations/s)  checksum {out[4].sum().item():.6f} "
      f"{out[1].sum().item()} {out[3].sum().item()}")
