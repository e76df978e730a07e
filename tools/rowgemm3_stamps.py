#!/usr/bin/env python
"""Where a rowgemm3 launch (gemm mode 3, csrc/gcn_gemm3.hip) spends its time, per round and per workgroup: reads the
s_memrealtime / s_memtime stamps of the diagnostic build (tools/build_variants.sh stamps3;
A3VT_LIB=gpurun_variants/liba3vt_RG3_STAMPS.so).  Runs one 4-layer stack forward + backward at the bench shape; the stamps
are those of the LAST forward (Z = X W) and the last dX launch.  Development aid.

Stamps per round: 0 round start, 1 first chunk landed (after the first barrier of the K loop), 5 second chunk,
2 K loop done (ring idle), 3 epilogue stores issued, 4 barrier behind the epilogue."""
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
    ops.gcn_stack(feats, adj, 50, 300, 99, ws, bs, bf16="fp32x3").backward(gup)
torch.cuda.synchronize()
raw = ctypes.CDLL(lib.LIB_PATH)
n = 2 * 256 * 4 * 8
buf = np.zeros(2 * n, dtype=np.uint64)
assert raw.a3vt_dbg_rg3_stamps(buf.ctypes.data_as(ctypes.c_void_p)) == 0
names = ["first chunk wait (0->1)", "second chunk (1->5)", "K loop (1->2)", "epilogue (2->3)", "tail barrier (3->4)", "round total (0->4)"]
for e, tag in enumerate(("forward Z = X W", "backward dX = dZ W^T")):
    st = buf[:n].reshape(2, 256, 4, 8)[e].astype(np.float64) * 0.01   # us
    cyc = buf[n:].reshape(2, 256, 4, 8)[e].astype(np.float64)
    t00 = st[:, 0, 0].min()
    print(f"== {tag}: rows {B * verts.shape[0]}")
    for rd in range(3):
        s = st[:, rd, :]
        d = [s[:, 1] - s[:, 0], s[:, 5] - s[:, 1], s[:, 2] - s[:, 1], s[:, 3] - s[:, 2], s[:, 4] - s[:, 3], s[:, 4] - s[:, 0]]
        print(f"round {rd}: starts at {np.median(s[:, 0]) - t00:7.2f}")
        for nm, x in zip(names, d):
            print(f"    {nm:26s} median {np.median(x):7.2f}  p10 {np.percentile(x, 10):7.2f}  p90 {np.percentile(x, 90):7.2f}")
        kc, kt = cyc[:, rd, 2] - cyc[:, rd, 1], s[:, 2] - s[:, 1]
        print(f"    K loop: {np.median(kc):.0f} shader cycles -> clock {np.median(kc / kt) / 1e3:.3f} GHz")
    print(f"last barrier: median {np.median(st[:, 2, 4]) - t00:.2f} us after the first start")
