#!/usr/bin/env python
"""Where a rowgemm launch spends its time, per round and per workgroup: reads the s_memrealtime stamps of the
diagnostic build (tools/build_variants.sh stamps; A3VT_LIB=gpurun_variants/liba3vt_RG_STAMPS.so).  Development aid.

Stamps per round: 0 round start, 1 first chunk landed (after the first barrier of the K loop), 5 second chunk,
2 K loop done (ring idle), 3 epilogue stores issued, 4 barrier behind the epilogue.
"""
import ctypes
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from a3vt_amd import lib  # noqa: E402

M, K, N = int(os.environ.get("ROWS", 64 * 2562)), 300, 300
dev = torch.device("cuda", 0)
a = torch.randn(M, K, device=dev)
w = torch.randn(K, N, device=dev) * 0.05
L = lib.load()
wt = torch.empty((L.a3vt_wt_rows(N), L.a3vt_wt_ld(K)), device=dev)
lib.check(L.a3vt_transpose_weight(lib.ptr(w), K, N, lib.ptr(wt), None), "t")
c = torch.empty(M, N, device=dev)
for _ in range(5):
    lib.check(L.a3vt_rowgemm(lib.ptr(a), K, M, K, lib.ptr(wt), N, 0, lib.ptr(c), N, None), "g")
torch.cuda.synchronize()
raw = ctypes.CDLL(lib.LIB_PATH)
buf = np.zeros(2 * 256 * 64, dtype=np.uint64)
rc = raw.a3vt_dbg_rg_stamps(buf.ctypes.data_as(ctypes.c_void_p))
assert rc == 0, rc
st = buf[:256 * 64].reshape(256, 8, 8).astype(np.float64) * 0.01   # us
cyc = buf[256 * 64:].reshape(256, 8, 8).astype(np.float64)     # shader cycles (s_memtime)
t00 = st[:, 0, 0].min()
rounds = int(os.environ.get("ROUNDS", 3))
print(f"rows {M}: launch spread of round-0 starts {st[:, 0, 0].max() - t00:.2f} us")
names = ["first chunk wait (0->1)", "second chunk (1->5)", "K loop (1->2)", "epilogue (2->3)", "tail barrier (3->4)", "round total (0->4)"]
for r in range(rounds):
    s = st[:, r, :]
    d = [s[:, 1] - s[:, 0], s[:, 5] - s[:, 1], s[:, 2] - s[:, 1], s[:, 3] - s[:, 2], s[:, 4] - s[:, 3], s[:, 4] - s[:, 0]]
    print(f"round {r}: starts at {np.median(s[:, 0]) - t00:7.2f} (p10 {np.percentile(s[:, 0], 10) - t00:.2f}, p90 {np.percentile(s[:, 0], 90) - t00:.2f})")
    for n, x in zip(names, d):
        print(f"    {n:26s} median {np.median(x):7.2f}  p10 {np.percentile(x, 10):7.2f}  p90 {np.percentile(x, 90):7.2f}  max {x.max():7.2f}")
    kc = cyc[:, r, 2] - cyc[:, r, 1]
    kt = s[:, 2] - s[:, 1]
    print(f"    K loop: {np.median(kc):.0f} shader cycles -> clock {np.median(kc / kt) / 1e3:.3f} GHz (p10 {np.percentile(kc / kt, 10) / 1e3:.3f}, p90 {np.percentile(kc / kt, 90) / 1e3:.3f})")
end = st[:, rounds - 1, 4]
print(f"last barrier: median {np.median(end) - t00:.2f}, max {end.max() - t00:.2f} us after the first start")
