#!/usr/bin/env python
"""Where a rowgemm16 launch (bf16 storage mode, gcn_gemm16.hip) spends its time, per 48-row block: reads the s_memrealtime
stamps of the diagnostic build (tools/build_variants.sh stamps16; A3VT_LIB=gpurun_variants/liba3vt_R16_STAMPS.so) after
a few bf16s stack forward + backward calls.  Development aid.
Stamps per block (wave 0 = a DMA wave, wave 4 = a store wave): 0 top, 1 block landed (counted wait), 2 after barrier 1 + DMA
issue, 3 K phase done, 4 output tile written, 5 after barrier 2, 6 (wave 4) store phase starts, 7 (wave 4) store phase done."""
import ctypes
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import a3vt_amd  # noqa: E402,F401
from a3vt_amd import lib, mesh as amesh  # noqa: E402
from a3vt_amd.pterotactyl.reconstruction.vision import model  # noqa: E402
from a3vt_amd.pterotactyl.utility import utils  # noqa: E402
from a3vt_amd.synthetic import make_args  # noqa: E402

dev = torch.device("cuda", 0)
B = int(os.environ.get("BATCH", 64))
args = make_args(gemm_precision="bf16s")
v, f = amesh.icosphere(4)
vt, ft = torch.from_numpy(v).to(dev), torch.from_numpy(f).to(dev)
info = utils.adj_init(vt, ft, args)
torch.manual_seed(0)
net = model.Deformation(info, vt, args).to(dev)
charts = model.prepare_mesh({"img": torch.zeros(B, 1)}, vt, args)
for _ in range(3):
    out = net(torch.zeros(B, 1, device=dev), charts)[0]
    out.square().sum().backward()
torch.cuda.synchronize()
raw = ctypes.CDLL(lib.LIB_PATH)
buf = np.zeros(2 * 256 * 16 * 8, dtype=np.uint64)
assert raw.a3vt_dbg_r16_stamps(buf.ctypes.data_as(ctypes.c_void_p)) == 0
st = buf.reshape(2, 256, 16, 8).astype(np.float64) * 0.01   # us
names = ["counted wait (0->1)", "barrier 1 + issue (1->2)", "K phase (2->3)", "tile write (3->4)", "barrier 2 (4->5)",
         "store phase, wave 4 (6->7)", "block total (0 -> next 0)"]
for e, label in enumerate(("forward (EPI_FWD_HIDDEN)", "backward (EPI_DX_MASK)")):
    s = st[e]
    entry = s[:, 15, 0].copy()
    s = s.copy()
    s[:, 15, :] = 0
    nb = int((s[:, :, 0] > 0).sum(1).min())
    print(f"{label}: prologue (kernel entry -> top of block 0, weights in registers) median {np.median(s[:, 0, 0] - entry):.2f} us; "
          f"entry spread over workgroups {entry.max() - entry.min():.2f} us; last store end - first entry {s[:, nb - 1, 7].max() - entry.min():.1f} us")
    print(f"{label}: {nb} blocks per workgroup stamped; launch = {np.median(s[:, nb - 1, 7] - s[:, 0, 0]):.1f} us first top -> last store phase end (median)")
    for b in (0, 1, 2, nb // 2, nb - 1):
        x = s[:, b, :]
        d = [x[:, 1] - x[:, 0], x[:, 2] - x[:, 1], x[:, 3] - x[:, 2], x[:, 4] - x[:, 3], x[:, 5] - x[:, 4], x[:, 7] - x[:, 6]]
        d.append(s[:, b + 1, 0] - x[:, 0] if b + 1 < nb else x[:, 7] - x[:, 0])
        print(f"  block {b}: " + "  ".join(f"{n.split(' (')[0]} {np.median(v_):.2f}" for n, v_ in zip(names, d)))
