#!/usr/bin/env python
"""One training step timing for BASELINE.json configs[3] and configs[4] on ONE MI355X (synthetic inputs):
  configs[3]  vision + touch: image model (default CNNs) + chart atlas with 4 touch charts (N = 1924), 25 000-point
              Chamfer, bf16 mode (--precision bf16s = bf16 storage, default; bf16 = operands only)
  configs[4]  10 242-vertex icosphere, 50 000-point Chamfer, same mode (per-GPU shard of the 8-GPU config)
Prints one JSON object per config.  These are capability / parity-case configurations, not the headline metric.
--only 3|4 runs one of them (e.g. under rocprofv3)."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--batch3", type=int, default=64)
    p.add_argument("--batch4", type=int, default=8)
    p.add_argument("--precision", default="bf16s", choices=["fp32", "bf16", "bf16s", "fp32x3"])
    p.add_argument("--only", type=int, default=0, choices=[0, 3, 4])
    p.add_argument("--steps", type=int, default=10)
    p.add_argument("--torch-adam", action="store_true", help="A/B: torch's fused Adam instead of the library's one-launch step")
    a = p.parse_args()
    from a3vt_amd import synthetic
    from a3vt_amd.synthetic import time_named_config   # the same leg bench.py prints under `named_configs`
    synthetic.LIBRARY_ADAM[0] = not a.torch_adam
    dev = torch.device("cuda", 0)
    for which, batch in ((3, a.batch3), (4, a.batch4)):
        if a.only in (0, which):
            print(json.dumps(time_named_config(which, dev, a.precision, batch, a.steps)))
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
