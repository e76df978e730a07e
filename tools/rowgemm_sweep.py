#!/usr/bin/env python
"""rowgemm (plain epilogue) time against the number of rows, i.e. against the bytes one launch touches.

Question (round 3): does the 256 MiB Infinity Cache decide what the epilogue stores of a launch cost?  Rows are
multiples of 2048 tiles (every wave gets whole tiles) so that the tile quantisation does not mix in.  Two access
patterns: the same (a, c) pair every launch, and a ring of NBUF distinct buffers c_i = a_i W (a layer chain).
Development aid; prints one line per size.
"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from a3vt_amd import lib  # noqa: E402

K = N = 300
dev = torch.device("cuda", 0)
L = lib.load()
w = torch.randn(K, N, device=dev) * 0.05
wt = torch.empty((L.a3vt_wt_rows(N), L.a3vt_wt_ld(K)), device=dev)
lib.check(L.a3vt_transpose_weight(lib.ptr(w), K, N, lib.ptr(wt), None), "t")
reps = int(os.environ.get("REPS", "30"))
ks = [int(x) for x in os.environ.get("KS", "1,2,3,4,5,6,8,10").split(",")]
nbuf = int(os.environ.get("NBUF", "4"))


def timed(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for k in ks:
    M = 2048 * 16 * k
    bufs = [torch.randn(M, K, device=dev) for _ in range(nbuf)]

    def same():
        lib.check(L.a3vt_rowgemm(lib.ptr(bufs[0]), K, M, K, lib.ptr(wt), N, 0, lib.ptr(bufs[1]), N, None), "g")

    def chain():
        for i in range(nbuf):
            lib.check(L.a3vt_rowgemm(lib.ptr(bufs[i]), K, M, K, lib.ptr(wt), N, 0,
                                     lib.ptr(bufs[(i + 1) % nbuf]), N, None), "g")
        # values stay bounded: W is small, and the timing does not depend on them

    t_same = timed(same)
    t_chain = timed(chain) / nbuf
    mb = M * K * 4 / 1e6
    print(f"tiles/wave={k} rows={M} array={mb:.0f} MB  same-pair {t_same:.1f} us ({t_same / k:.1f}/tile-round, "
          f"{2.0 * M * K * N / t_same / 1e6:.1f} TF)  chain{nbuf} {t_chain:.1f} us ({t_chain / k:.1f}/tile-round)",
          flush=True)
    del bufs
