#!/usr/bin/env python
"""Time a3vt_rowgemm (plain epilogue) at the headline shape M=B*N x 300 x 300.  Development aid."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from a3vt_amd import lib, ops  # noqa: E402

M, K, N = int(os.environ.get('ROWS', 64 * 2562)), 300, 300
dev = torch.device("cuda", 0)
zero = os.environ.get("ZERO", "0") == "1"
BF16 = 1 if os.environ.get("BF16", "0") == "1" else 0   # operand mode (a3vt.h: gemm_bf16)
a = torch.zeros(M, K, device=dev) if zero else torch.randn(M, K, device=dev)
w = torch.zeros(K, N, device=dev) if zero else torch.randn(K, N, device=dev)
L = lib.load()
wt = torch.empty((L.a3vt_wt_rows(N), L.a3vt_wt_ld(K)), device=dev)
lib.check(L.a3vt_transpose_weight(lib.ptr(w), K, N, lib.ptr(wt), None), "t")
c = torch.empty(M, N, device=dev)
for _ in range(3):
    lib.check(L.a3vt_rowgemm(lib.ptr(a), K, M, K, lib.ptr(wt), N, BF16, lib.ptr(c), N, None), "g")
torch.cuda.synchronize()
reps = int(os.environ.get('REPS', '20'))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    L.a3vt_rowgemm(lib.ptr(a), K, M, K, lib.ptr(wt), N, BF16, lib.ptr(c), N, None)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
print(f"{os.environ.get('A3VT_LIB', 'default')} zero={zero} reps={reps}: rowgemm {ms * 1e3:.1f} us  {2.0 * M * K * N / ms / 1e9:.1f} TFLOP/s")
