"""Developer aid (GPU box): the image pyramid's 5 x 5 weight gradients at the configs[3] map shapes — a3vt_conv5_weight_grad
against MIOpen's split-K kernel (+ its fill and cast launches); run under rocprofv3 --kernel-trace --stats for per-kernel time,
or read the wall-clock per call printed here (launch overhead included).  A3VT_WRW_WGS caps the workgroups (partial images)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from a3vt_amd import ops  # noqa: E402

SHAPES = [((64, 3, 256, 256), 3, 1), ((64, 3, 254, 254), 16, 2), ((64, 16, 126, 126), 16, 1), ((64, 16, 124, 124), 16, 1), ((64, 16, 122, 122), 32, 2), ((64, 32, 60, 60), 32, 1), ((64, 32, 58, 58), 32, 1)]


def main():
    dev = torch.device("cuda", 0)
    reps = int(os.environ.get("REPS", "30"))
    for shape, cout, stride in SHAPES:
        g = torch.Generator().manual_seed(1)
        x = torch.randn(shape, generator=g).to(dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        w = (torch.randn(cout, shape[1], 5, 5, generator=g) * 0.05).to(dev)
        wb = w.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        ho, wo = (shape[2] + 2 - 5) // stride + 1, (shape[3] + 2 - 5) // stride + 1
        gy = torch.randn((shape[0], cout, ho, wo), generator=g).to(dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        res = {}
        for own in (True, False):
            ops.LIBRARY_CONV5_WRW[0] = own
            for _ in range(5):
                gw = ops._conv5_weight_grad(x, gy, w, [stride, stride], [1, 1], wb)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                gw = ops._conv5_weight_grad(x, gy, w, [stride, stride], [1, 1], wb)
            torch.cuda.synchronize()
            res[own] = ((time.perf_counter() - t0) / reps * 1e6, gw.float())
        ops.LIBRARY_CONV5_WRW[0] = True
        rel = float((res[True][1] - res[False][1]).norm() / res[False][1].norm())
        print(f"{shape} -> {cout} stride {stride}: library {res[True][0]:7.1f} us/call   MIOpen {res[False][0]:7.1f} us/call   rel diff {rel:.2e}")


if __name__ == "__main__":
    main()
