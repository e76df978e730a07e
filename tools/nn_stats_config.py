#!/usr/bin/env python
"""Counters of the pruned search inside a training step of BASELINE configs[3] / configs[4] (their own geometry, not a synthetic
sphere): A3VT_LIB=gpurun_variants/liba3vt_NN_STATS.so python tools/nn_stats_config.py [--which 3]"""
import argparse
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from a3vt_amd import lib  # noqa: E402
from a3vt_amd.synthetic import NamedStep, named_config  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--which", type=int, default=3)
ap.add_argument("--batch", type=int, default=0)
args = ap.parse_args()
L = lib.load()
L.a3vt_dbg_nn_stats.argtypes = [ctypes.c_void_p]
dev = torch.device("cuda", 0)
cfg = named_config(args.which, dev, "bf16s", args.batch or None)
step = NamedStep(cfg)
out = (ctypes.c_ulonglong * 16)()
for k in range(6):
    step()
    torch.cuda.synchronize()
    L.a3vt_dbg_nn_stats(out)
    w = max(out[0], 1)
    print(f"step {k}: {out[0]} waves, {out[1] / w:.1f} blocks ({out[5] / w:.1f} groups of 16) evaluated, {out[2] / w:.1f} point-box tests per wave; "
          f"{out[3] / w:.1f} blocks needed by some lane, {out[4] / w / 64:.1f} by a lane on average; worst wave {out[6]} groups")
