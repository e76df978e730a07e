#!/usr/bin/env python
"""Summarise a rocprofv3 `*kernel_stats.csv`: per-step launch counts, mean duration and ms per step of the top kernels.
usage: python tools/kstats_summary.py <kernel_stats.csv> <steps in the run> [rows]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2])
top = int(sys.argv[3]) if len(sys.argv) > 3 else 24
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"all kernels: {tot / steps / 1e6:.2f} ms per step over {steps:.0f} steps")
for r in rows[:top]:
    print(f'{r["Name"][:86]:86s} {float(r["Calls"]) / steps:7.1f}/step  avg {float(r["AverageNs"]) / 1e3:8.1f} us '
          f'{float(r["TotalDurationNs"]) / steps / 1e6:7.2f} ms/step')
