#!/bin/bash
# Developer aid (GPU box): per-kernel average durations of `python tools/stack_bench.py "$@"` from rocprofv3 kernel stats.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kstats; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kstats -- python $ROOT/tools/stack_bench.py "$@" > /dev/null 2>&1
python - "$(find /tmp/kstats -name '*kernel_stats.csv' | head -1)" <<'PY'
import csv, sys
for row in csv.DictReader(open(sys.argv[1])):
    if float(row["Percentage"]) > 0.3:
        print(f'{row["Name"][:70]:70s} calls {row["Calls"]:>5s} avg {float(row["AverageNs"]) / 1e3:8.1f} us  {row["Percentage"]}%')
PY
