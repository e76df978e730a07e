#!/bin/bash
# Developer aid (GPU box): per-kernel average durations (rocprofv3 kernel stats) of any tool of this directory.
# Usage: tools/prof_stats.sh <name pattern> <tool.py> [its options]
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
PAT=$1; TOOL=$2; shift 2
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pstats; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pstats -- python $ROOT/tools/$TOOL "$@" > /tmp/pstats.log 2>&1
python - "$(find /tmp/pstats -name '*kernel_stats.csv' | head -1)" "$PAT" <<'PY'
import csv, re, sys
for row in csv.DictReader(open(sys.argv[1])):
    if re.search(sys.argv[2], row["Name"]):
        print(f'{row["Name"][:70]:70s} calls {row["Calls"]:>5s} avg {float(row["AverageNs"]) / 1e3:8.1f} us  {row["Percentage"]}%')
PY
