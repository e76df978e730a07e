#!/usr/bin/env python
"""The image pyramid's first seven 5 x 5 convolutions at the configs[3] map shapes (bs 64, channels-last bf16): the library's
direct convolution (csrc/conv5.hip) against MIOpen — forward, input gradient, weight gradient per layer; device time per call
through a captured HIP graph (plain HIP events around the calls where a capture is refused).
Run on the GPU box:  python tools/conv5_bench.py [--batch 64]      (under rocprofv3 --kernel-trace --stats: per-kernel times)"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--reps", type=int, default=20)
a = ap.parse_args()
from a3vt_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)
# (cin, input map, cout, stride) of layers 0-6 of one encoder (k = 5, padding 1; vision/model.py:28-47)
LAYERS = [(3, 256, 3, 1), (3, 254, 16, 2), (16, 126, 16, 1), (16, 124, 16, 1), (16, 122, 32, 2), (32, 60, 32, 1), (32, 58, 32, 1)]


def timed(fn):
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        for _ in range(3):
            fn()
    torch.cuda.synchronize()
    try:
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            for _ in range(a.reps):
                fn()
        run = graph.replay
    except Exception:  # noqa: BLE001  (a capture MIOpen refuses)
        torch.cuda.synchronize()

        def run():
            for _ in range(a.reps):
                fn()
    run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    run()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / a.reps * 1e3


def main():
    tot = {"lib": 0.0, "miopen": 0.0}
    print("layer                      bytes in+out |  forward: library / MIOpen  | input gradient: library / MIOpen | weight gradient: library / MIOpen   (us per call)")
    for cin, hw, cout, stride in LAYERS:
        g = torch.Generator().manual_seed(1)
        x = torch.randn((a.batch, cin, hw, hw), generator=g).to(dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        w = (torch.randn(cout, cin, 5, 5, generator=g) * 0.05).to(dev)
        b = torch.zeros(cout, device=dev)
        wb = w.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        ho = (hw + 2 - 5) // stride + 1
        gy = torch.randn((a.batch, cout, ho, ho), generator=g).to(dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        st, pd = [stride, stride], [1, 1]
        img_f = ops._conv5_image(w, 0)
        img_b = ops._conv5_image(w, 1) if (stride == 1 and cin != 3) or (cin, cout) == (3, 16) else None
        L = ops._lib.load()
        gx_buf = torch.empty_like(x, memory_format=torch.channels_last)

        def conv_bwd(mask):
            return torch.ops.aten.convolution_backward(gy, x, wb, None, st, pd, [1, 1], False, [0, 0], 1, mask)

        fwd = (lambda: ops.conv5_nhwc(x, img_f, b, cout, stride, 1),
               lambda: torch.ops.aten.convolution(x, wb, None, st, pd, [1, 1], False, [0, 0], 1))
        if stride == 1 and cin != 3:
            dgrad = (lambda: ops.conv5_nhwc(gy, img_b, None, cin, 1, 3), lambda: conv_bwd([True, False, False]))
        elif (cin, cout) == (3, 16):
            dgrad = (lambda: ops._lib.check(L.a3vt_conv5_input_grad_3x16s2(ops._lib.ptr(gy), a.batch, ho, ho, ops._lib.ptr(img_b), ops._lib.ptr(gx_buf),
                                                                            ops._stream()), "up3"), lambda: conv_bwd([True, False, False]))
        else:
            dgrad = None      # 3 -> 3: the image needs no gradient; 16 -> 32 stride 2: MIOpen's in both columns
        wgrad = (lambda: ops._conv5_weight_grad(x, gy, w, st, pd, wb), lambda: conv_bwd([False, True, False])[1].float())
        row = []
        for pair in (fwd, dgrad, wgrad):
            if pair is None:
                row.append("        -        ")
                continue
            tl, tm = timed(pair[0]), timed(pair[1])
            tot["lib"] += tl
            tot["miopen"] += tm
            row.append(f"{tl:7.1f} / {tm:7.1f}")
        mb = (x.numel() + gy.numel()) * 2 / 1e6
        print(f"{cin:3d} -> {cout:3d} s{stride} on {hw:3d}^2    {mb:7.1f} MB    |  {row[0]}        |  {row[1]}             |  {row[2]}")
    print(f"sum of the three products over the seven layers of one encoder: library {tot['lib'] / 1e3:.2f} ms, MIOpen {tot['miopen'] / 1e3:.2f} ms "
          "(MIOpen's weight gradient includes its fill and cast launches and the cast to fp32 the optimizer needs)")


if __name__ == "__main__":
    main()
