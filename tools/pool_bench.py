#!/usr/bin/env python
"""Per-vertex image-feature pooling (csrc/pooling.hip: a3vt_image_pool_fwd / _bwd) at the configs[3] sizes — 64 meshes x 1 924
vertices, the default pyramid's maps 64 x 23 x 23, 128 x 7 x 7, 256 x 3 x 3 (448 channels) — device time per call through a
captured graph.  Run on the GPU box:  python tools/pool_bench.py   (kernel times: tools/prof_stats.sh pool pool_bench.py)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from a3vt_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)
B, N = 64, 1924
g = torch.Generator().manual_seed(0)
maps = [torch.randn(B, c, h, h, generator=g).to(dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
        for c, h in ((64, 23), (128, 7), (256, 3))]
verts = ((torch.rand(B, N, 3, generator=g) - 0.5) * 0.4).to(dev).requires_grad_(True)
K = [221.7025, 0, 128.0, 0, 221.7025, 128.0, 0, 0, 1]
proj = [-9.7e-6, -221.7, -90.5, 54.3, -156.8, 1.7e-5, -247.3, 54.3, 0.7071, 0.0, -0.7071, 0.4243]   # ~ K.RT of vision/model.py:50-64
gy = torch.randn(B, N, 448, generator=g).to(dev)


def step():
    f = ops.image_pool(verts, proj, maps)
    f.backward(gy)


for _ in range(3):
    step()
torch.cuda.synchronize()
side = torch.cuda.Stream()
with torch.cuda.stream(side):
    step()
torch.cuda.synchronize()
graph = torch.cuda.CUDAGraph()
reps = 10
with torch.cuda.graph(graph, stream=side):
    for _ in range(reps):
        step()
graph.replay()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
graph.replay()
e1.record()
torch.cuda.synchronize()
print(f"image_pool forward + backward (+ autograd's gradient accumulation), {B} x {N} vertices x 448 channels: {e0.elapsed_time(e1) / reps * 1e3:.1f} us per call")
