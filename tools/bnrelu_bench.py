#!/usr/bin/env python
"""Training BatchNorm2d + ReLU on the image pyramid's maps (bs 64, channels-last bf16): the fused operator (csrc/bnrelu.hip,
two launches each way) against MIOpen's BatchNorm + torch's in-place ReLU, forward + backward per layer, HIP events.
Run on the GPU box:  python tools/bnrelu_bench.py [--batch 64]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--layers", default="", help="comma-separated layer indices (default: all 13); with tools/prof_stats.sh: per-kernel times of one shape")
a = ap.parse_args()
from a3vt_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)
# (channels, map size) of the 13 normalised layers of one encoder (k = 5, pad 1, stride 2 at block starts; vision/model.py:28-47)
LAYERS = [(3, 254), (16, 126), (16, 124), (16, 122), (32, 60), (32, 58), (32, 56), (64, 27), (64, 25), (64, 23), (128, 11), (128, 9), (128, 7)]


def timed(fn):
    """Device time per call: `reps` calls captured in one HIP graph (the Python side of a call costs more than these kernels)."""
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        for _ in range(3):
            fn()
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        for _ in range(a.reps):
            fn()
    graph.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    graph.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / a.reps * 1e3


tot = [0.0, 0.0]
if a.layers:
    LAYERS = [LAYERS[int(i)] for i in a.layers.split(",")]
for c, hw in LAYERS:
    x = torch.randn(a.batch, c, hw, hw, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    gy = torch.randn_like(x)
    bn = torch.nn.BatchNorm2d(c).to(dev).train()
    relu = torch.nn.ReLU(inplace=True)

    def fused():
        y = ops.BNReLUFn.apply(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.num_batches_tracked, bn.eps, bn.momentum)
        y.backward(gy)

    def miopen():
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = relu(bn(x))
        y.backward(gy)

    tf, tm = timed(fused), timed(miopen)
    tot[0] += tf
    tot[1] += tm
    mb = x.numel() * 2 / 1e6
    print(f"C={c:4d} {hw:3d}x{hw:<3d} {mb:7.1f} MB   fused {tf:7.1f} us ({7 * mb / tf:5.2f} TB/s on its 7 passes)   MIOpen + ReLU {tm:7.1f} us")
print(f"one encoder, fwd + bwd: fused {tot[0] / 1e3:.2f} ms, MIOpen + ReLU {tot[1] / 1e3:.2f} ms")
