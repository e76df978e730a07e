#!/bin/bash
# Scalar-cache (SQC) counters of nn_query_kernel (run on the GPU box): tools/collect_nn_sqc.sh [geometry]
# Two --pmc passes (kernel-trace only) over tools/chamfer_bench.py; prints per-launch averages and writes
# gpurun_out/nn_sqc/summary.json.  What it answers: is the wave-uniform candidate stream (s_load_dwordx16) served by the
# scalar data cache or by its miss path to L2?
set -e
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
GEO=${1:-bench}
OUT=$ROOT/gpurun_out/nn_sqc
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_DCACHE_MISSES_DUPLICATE SQC_DCACHE_BUSY_CYCLES SQC_DCACHE_INPUT_VALID_READYB GRBM_GUI_ACTIVE \
  --output-format csv -d $OUT/p1 -- python $ROOT/tools/chamfer_bench.py --geometry $GEO --algos pruned --reps 2 --shapes ${SHAPES:-3x64x10000} > $OUT/run1.log 2>&1 || { tail -5 $OUT/run1.log; exit 1; }
rocprofv3 --kernel-trace --pmc SQC_TC_DATA_READ_REQ SQC_TC_STALL SQ_INST_CYCLES_SMEM SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY \
  --output-format csv -d $OUT/p2 -- python $ROOT/tools/chamfer_bench.py --geometry $GEO --algos pruned --reps 2 --shapes ${SHAPES:-3x64x10000} > $OUT/run2.log 2>&1 || { tail -5 $OUT/run2.log; exit 1; }
python - <<PY
import csv, glob, collections, json
res = collections.defaultdict(dict)
for f in glob.glob("$OUT/p*/*/*counter_collection.csv"):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if "nn_query" in k:
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in agg.items():
        for c, v in d.items():
            res[k][c] = sum(v) / len(v)
json.dump(res, open("$OUT/summary.json", "w"), indent=1)
for k, m in res.items():
    print(k, json.dumps({c: round(v, 1) for c, v in sorted(m.items())}, indent=1))
PY
