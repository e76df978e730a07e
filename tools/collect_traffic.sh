#!/bin/bash
# HBM traffic of the MFMA kernels from PMC counters (run on the GPU box):  tools/collect_traffic.sh [--precision fp32x3]
# Two separate --pmc passes (FETCH_SIZE and WRITE_SIZE do not fit one pass on gfx950, MI355X_MICROARCH.md
# "rocprofv3 PMC slots"); kernel-trace only, no other trace domains.  Writes gpurun_out/traffic/*.csv and a summary.
set -e
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/traffic
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/$c -- python $ROOT/tools/stack_bench.py --layers 4 --reps 2 "$@" > /dev/null 2>&1
done
python - <<PY
import csv, glob, collections, json
res = collections.defaultdict(dict)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob("$OUT/%s/*/*counter_collection.csv" % c)[0]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == c:
            agg[r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        if "rowgemm" in k or "dw_kernel" in k or "dww_kernel" in k or "dw16" in k or "dw3" in k or "csr" in k or "slab" in k or "thin" in k:
            res[k][c + "_KiB_max"] = max(v)       # hidden x hidden launches are the largest
            res[k][c + "_KiB_mean"] = sum(v) / len(v)
            res[k]["launches"] = len(v)
json.dump(res, open("$OUT/summary.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
