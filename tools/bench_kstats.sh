#!/bin/bash
# Developer aid (GPU box): per-kernel average durations of a short `python bench.py` run (all kernels of the training step)
# from rocprofv3 kernel stats.  Usage: tools/bench_kstats.sh [grep pattern] [bench.py options]
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
PAT=${1:-.}; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/bkstats; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/bkstats -- python $ROOT/bench.py --steps 6 --warmup 2 --alt-steps 0 --no-cpu-baseline --no-traffic "$@" > /tmp/bkstats.log 2>&1
tail -1 /tmp/bkstats.log | cut -c1-200
python - "$(find /tmp/bkstats -name '*kernel_stats.csv' | head -1)" "$PAT" <<'PY'
import csv, re, sys
for row in csv.DictReader(open(sys.argv[1])):
    if re.search(sys.argv[2], row["Name"]) and float(row["Percentage"]) > 0.02:
        print(f'{row["Name"][:70]:70s} calls {row["Calls"]:>5s} avg {float(row["AverageNs"]) / 1e3:8.1f} us  {row["Percentage"]}%')
PY
