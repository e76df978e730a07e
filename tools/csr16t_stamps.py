#!/usr/bin/env python
"""Wall-clock phase stamps of csr16t_fwd_kernel (gpurun_variants/liba3vt_T16_STAMPS.so, tools/build_variants.sh t16): per
persistent workgroup: entry -> first unit requested -> all its units gathered and stored.
Run:  A3VT_LIB=$GRAFT_REPO_ROOT/gpurun_variants/liba3vt_T16_STAMPS.so python tools/csr16t_stamps.py"""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from a3vt_amd import lib, mesh as amesh, ops  # noqa: E402

dev = torch.device("cuda", 0)
L, H, B = 3, 300, 64
verts, faces = amesh.icosphere(4)
n = verts.shape[0]
adj = ops.DeviceCSR(amesh.CSRAdjacency.from_pairs(*amesh.vision_pairs(faces, n), n), dev)
g = torch.Generator().manual_seed(0)
ws = [((torch.rand(1, 50 if i == 0 else H, H if i < L - 1 else 3, generator=g) - 0.5) * 0.2).to(dev) for i in range(L)]
bs = [((torch.rand(H if i < L - 1 else 3, generator=g) - 0.5) * 0.2).to(dev) for i in range(L)]
feats = torch.nn.functional.pad(torch.randn(B, n, 50, generator=g) * 0.5, (0, 2)).to(dev)
with torch.no_grad():
    for _ in range(3):
        ops.gcn_stack(feats, adj, 50, H, 99, ws, bs, bf16="bf16s")
torch.cuda.synchronize()
dll = ctypes.CDLL(lib.LIB_PATH)
if not hasattr(dll, "a3vt_dbg_t16_stamps"):
    raise SystemExit("this library has no stamps: build the T16_STAMPS variant and select it with A3VT_LIB")
buf = np.zeros(4096 * 4, dtype=np.uint64)
dll.a3vt_dbg_t16_stamps(ctypes.c_void_p(buf.ctypes.data))
s = buf.reshape(4096, 4).astype(np.int64)
s = s[s[:, 0] > 0]
t0 = s[:, 0].min()
us = lambda a: "p10 %6.2f  median %6.2f  p90 %6.2f  max %6.2f us" % tuple(np.percentile(a, [10, 50, 90, 100]) / 100.0)  # noqa: E731
print(f"csr16t_kernel<forward>, {s.shape[0]} persistent workgroups of the last launch (64 x 2562 rows), wall clock per workgroup:")
print("  entry -> first unit requested (header round trip, DMA issue) :", us(s[:, 1] - s[:, 0]))
print("  -> all units gathered and stored                             :", us(s[:, 3] - s[:, 1]))
print("  entry after the launch's first                               :", us(s[:, 0] - t0))
print("  launch (first entry -> last end)                             : %.2f us" % ((s[:, 3].max() - t0) / 100.0))
