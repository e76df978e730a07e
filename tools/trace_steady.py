#!/usr/bin/env python
"""Steady-state per-kernel table from a rocprofv3 `*kernel_trace.csv`: the run is cut into steps at every launch of a marker
kernel that runs once per step (default: the Chamfer reduce), the first `--skip` steps (MIOpen find mode, allocator
growth) are dropped, and the remaining launches are summed per kernel.
usage: python tools/trace_steady.py <kernel_trace.csv> [--marker chamfer_reduce] [--skip 4] [--top 30]"""
import argparse
import collections
import csv

ap = argparse.ArgumentParser()
ap.add_argument("trace")
ap.add_argument("--marker", default="chamfer_reduce")
ap.add_argument("--skip", type=int, default=4)
ap.add_argument("--top", type=int, default=30)
a = ap.parse_args()
rows = sorted(csv.DictReader(open(a.trace)), key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if a.marker in r["Kernel_Name"]]
if len(marks) < a.skip + 2:
    raise SystemExit(f"only {len(marks)} marker launches")
lo, hi = marks[a.skip], marks[-1]          # whole steps between two marker launches
steps = len(marks) - 1 - a.skip
tot = collections.defaultdict(lambda: [0, 0.0])
for r in rows[lo:hi]:
    t = tot[r["Kernel_Name"]]
    t[0] += 1
    t[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
busy = sum(v[1] for v in tot.values())
span = int(rows[hi]["Start_Timestamp"]) - int(rows[lo]["Start_Timestamp"])
print(f"{steps} steady steps: {span / steps / 1e6:.2f} ms per step wall between markers, {busy / steps / 1e6:.2f} ms of kernels")
for name, (n, ns) in sorted(tot.items(), key=lambda kv: -kv[1][1])[: a.top]:
    print(f"{name[:90]:90s} {n / steps:7.1f}/step  avg {ns / n / 1e3:8.1f} us {ns / steps / 1e6:7.2f} ms/step")
