#!/usr/bin/env python
"""Where a csrq launch (channel-sliced aggregation, csrc/gcn_csrq.hip) spends its time, per quad and per workgroup: reads
the s_memrealtime / s_memtime stamps of the diagnostic build (tools/build_variants.sh stampsq;
A3VT_LIB=gpurun_variants/liba3vt_CSRQ_STAMPS.so).  Runs a 4-layer stack forward + backward at the bench shape; the stamps
are those of the LAST forward / backward aggregation launch.  Development aid.

Stamps per quad: 0 loop top (next slice requested right after), 1 gathers + stores issued, 2 next slice parked (its loads
have arrived), 3 behind the barrier."""
import ctypes
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from types import SimpleNamespace  # noqa: E402
from a3vt_amd import lib, mesh as amesh, ops  # noqa: E402
from a3vt_amd.pterotactyl.reconstruction.vision import model  # noqa: E402

dev = torch.device("cuda", 0)
B = int(os.environ.get("BATCH", 64))
verts, faces = amesh.icosphere(4)
r, c = amesh.vision_pairs(faces, verts.shape[0])
adj = ops.DeviceCSR(amesh.CSRAdjacency.from_pairs(r, c, verts.shape[0]), dev)
torch.manual_seed(0)
gcn = model.GCN(50, SimpleNamespace(num_GCN_layers=4, hidden_GCN_size=300, cut=0.33)).to(dev)
ws, bs = [l.weight for l in gcn.layers], [l.bias for l in gcn.layers]
feats = torch.zeros(B, verts.shape[0], 52, device=dev)
feats[..., :50] = torch.randn(B, verts.shape[0], 50, device=dev) * 0.5
feats.requires_grad_(True)
gup = torch.randn(B, verts.shape[0], 3, device=dev)
for _ in range(4):
    ops.gcn_stack(feats, adj, 50, 300, 99, ws, bs).backward(gup)
torch.cuda.synchronize()
raw = ctypes.CDLL(lib.LIB_PATH)
n = 2 * 256 * 8 * 4
buf = np.zeros(2 * n, dtype=np.uint64)
assert raw.a3vt_dbg_csrq_stamps(buf.ctypes.data_as(ctypes.c_void_p)) == 0
for e, tag in enumerate(("forward (csrq<0>)", "backward (csrq<1>)")):
    st = buf[:n].reshape(2, 256, 8, 4)[e].astype(np.float64) * 0.01   # us
    cyc = buf[n:].reshape(2, 256, 8, 4)[e].astype(np.float64)
    t00 = st[:, 0, 0][st[:, 0, 0] > 0].min()
    print(f"== {tag}")
    ent = st[:, 7, 0]
    print(f"prologue (entry -> first quad: first slice + index entries): median {np.median(st[:, 0, 0] - ent):.2f}  "
          f"p90 {np.percentile(st[:, 0, 0] - ent, 90):.2f} us; entries spread over {ent.max() - ent.min():.2f} us")
    for qi in range(7):
        s = st[:, qi, :]
        ok = s[:, 3] > 0
        if not ok.any():
            continue
        s = s[ok]
        d = [s[:, 1] - s[:, 0], s[:, 2] - s[:, 1], s[:, 3] - s[:, 2], s[:, 3] - s[:, 0]]
        kc = cyc[ok, qi, 3] - cyc[ok, qi, 0]
        print(f"quad {qi} ({ok.sum():3d} workgroups): starts {np.median(s[:, 0]) - t00:6.2f} | gather+stores {np.median(d[0]):5.2f}"
              f"  wait+park {np.median(d[1]):5.2f}  barrier {np.median(d[2]):5.2f}  total {np.median(d[3]):5.2f} us"
              f"  ({np.median(kc):.0f} cycles, {np.median(kc / np.maximum(d[3], 1e-9)) / 1e3:.2f} GHz)")
    last = np.where(st[:, :, 3] > 0, st[:, :, 3], 0).max(axis=1)
    print(f"first start -> last workgroup done: median {np.median(last) - t00:.2f}  max {last.max() - t00:.2f} us")
