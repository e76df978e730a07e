#!/usr/bin/env python
"""Phase stamps of rowgemmw_kernel (gpurun_variants/liba3vt_RGW_STAMPS.so, tools/build_variants.sh rgw): per 16-row tile the
shader-clock time of the K loop (with the side work dealt into it), of the s_waitcnt vmcnt(0) behind it and of the workgroup
barrier, per wave.  Run:  A3VT_LIB=$GRAFT_REPO_ROOT/gpurun_variants/liba3vt_RGW_STAMPS.so python tools/rowgemmw_stamps.py"""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def main():
    from a3vt_amd import lib, mesh as amesh, ops
    dev = torch.device("cuda", 0)
    L, H, B = 4, 300, 64
    verts, faces = amesh.icosphere(4)
    n = verts.shape[0]
    adj = ops.DeviceCSR(amesh.CSRAdjacency.from_pairs(*amesh.vision_pairs(faces, n), n), dev)
    g = torch.Generator().manual_seed(0)
    ws = [((torch.rand(1, 50 if i == 0 else H, H if i < L - 1 else 3, generator=g) - 0.5) * 0.2).to(dev).requires_grad_(True) for i in range(L)]
    bs = [((torch.rand(H if i < L - 1 else 3, generator=g) - 0.5) * 0.2).to(dev).requires_grad_(True) for i in range(L)]
    feats = torch.nn.functional.pad(torch.randn(B, n, 50, generator=g) * 0.5, (0, 2)).to(dev).requires_grad_(True)
    for _ in range(2):
        out = ops.gcn_stack(feats, adj, 50, H, 99, ws, bs)
        out.sum().backward()
    torch.cuda.synchronize()
    buf = np.zeros(256 * 4 * 48 * 4, dtype=np.uint64)
    fn = lib.load().a3vt_dbg_rgw_stamps if hasattr(ctypes.CDLL(lib.LIB_PATH), "a3vt_dbg_rgw_stamps") else None
    if fn is None:
        raise SystemExit("this library has no stamps: build the RGW_STAMPS variant and select it with A3VT_LIB")
    dll = ctypes.CDLL(lib.LIB_PATH)
    dll.a3vt_dbg_rgw_stamps(ctypes.c_void_p(buf.ctypes.data))
    s = buf.reshape(256, 4, 48, 4).astype(np.int64)   # the LAST launch that ran (a dX product)
    tiles = 40
    loop = (s[:, :, 1:tiles, 1] - s[:, :, 1:tiles, 0])
    wait = (s[:, :, 1:tiles, 2] - s[:, :, 1:tiles, 1])
    barr = (s[:, :, 1:tiles, 3] - s[:, :, 1:tiles, 2])
    whole = (s[:, :, 2:tiles, 0] - s[:, :, 1:tiles - 1, 0])
    pc = lambda a: "p10 %6d  median %6d  p90 %6d  max %7d" % tuple(np.percentile(a, [10, 50, 90, 100]).astype(int))  # noqa: E731
    print("shader-clock ticks (s_memtime) per 16-row tile; 380 MFMAs x 32 cycles = 12 160")
    print("K loop + side work :", pc(loop))
    print("s_waitcnt vmcnt(0) :", pc(wait))
    print("barrier            :", pc(barr))
    print("tile to tile       :", pc(whole))
    for w in range(4):
        print(f"  wave {w}: loop {np.median(loop[:, w]):7.0f}  wait {np.median(wait[:, w]):6.0f}  barrier {np.median(barr[:, w]):6.0f}")
    print("whole kernel per workgroup (first stamp to last):", pc((s[:, :, tiles - 1, 3] - s[:, :, 0, 0]).ravel()))
    if hasattr(dll, "a3vt_dbg_rgw_rt"):
        realtime(dll.a3vt_dbg_rgw_rt, "rowgemmw_kernel", s[:, 0, tiles - 1, 3] - s[:, 0, 0, 0])


def realtime(fn, name, loop_ticks=None):
    """Wall-clock stamps (s_memrealtime, 100 MHz) of wave 0 of every workgroup: entry, first tile, behind the last, end."""
    rt = np.zeros(256 * 4, dtype=np.uint64)
    fn(ctypes.c_void_p(rt.ctypes.data))
    r = rt.reshape(256, 4).astype(np.int64)
    r = r[r[:, 0] > 0]
    if r.shape[0] == 0:
        return
    t0 = r[:, 0].min()
    us = lambda a: "min %6.2f  median %6.2f  max %6.2f us" % (a.min() / 100.0, np.median(a) / 100.0, a.max() / 100.0)  # noqa: E731
    print(f"{name}, wall clock of wave 0 per workgroup (s_memrealtime, 10 ns ticks), {r.shape[0]} workgroups:")
    print("  entry after the first workgroup's :", us(r[:, 0] - t0))
    print("  entry -> first tile (weights etc.) :", us(r[:, 1] - r[:, 0]))
    print("  the tile loop                      :", us(r[:, 2] - r[:, 1]))
    print("  behind the loop -> kernel end      :", us(r[:, 3] - r[:, 2]))
    print("  first entry -> last end            : %.2f us" % ((r[:, 3].max() - t0) / 100.0))
    if loop_ticks is not None:
        lt = loop_ticks[: r.shape[0]]
        print("  shader clock during the loop       : %.3f GHz (s_memtime ticks / wall clock)" % (np.median(lt / ((r[:, 2] - r[:, 1]) * 10.0))))


if __name__ == "__main__":
    main()


def dww():
    """The same for dww_kernel (gcn_dww.hip): per 16-row stage, per wave, stage start -> in front of the barrier."""
    from a3vt_amd import lib
    dll = ctypes.CDLL(lib.LIB_PATH)
    if not hasattr(dll, "a3vt_dbg_dww_stamps"):
        return
    buf = np.zeros(256 * 4 * 96 * 2, dtype=np.uint64)
    dll.a3vt_dbg_dww_stamps(ctypes.c_void_p(buf.ctypes.data))
    s = buf.reshape(256, 4, 96, 2).astype(np.int64)
    n = 78
    work = s[:, :, 1:n, 1] - s[:, :, 1:n, 0]
    whole = s[:, :, 2:n, 0] - s[:, :, 1:n - 1, 0]
    print("\ndww_kernel, shader-clock ticks per 16-row stage; 50 MFMAs x 4 steps x 32 cycles = 6 400 (half B: 45 -> 5 760; wave 3: 40 / 36)")
    for half in (0, 1):
        blk = [b for b in range(256) if ((b >> 3) & 1) == half]
        print(f"  half {'A' if half == 0 else 'B'}: stage to stage median {np.median(whole[blk]):6.0f}; per wave work (start -> barrier): "
              + "  ".join(f"w{w} {np.median(work[blk][:, w]):6.0f}" for w in range(4)))
    if hasattr(dll, "a3vt_dbg_dww_rt"):
        realtime(dll.a3vt_dbg_dww_rt, "dww_kernel")


if __name__ == "__main__":
    dww()
