#!/bin/bash
# SQ-level PMC counters of the nearest-neighbour kernel (run on the GPU box):  tools/collect_nn.sh
# One --pmc pass (kernel-trace only) over tools/chamfer_bench.py; writes gpurun_out/nn/summary.json (copy to profiles/).
set -e
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/nn
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY \
  SQ_INSTS_SALU SQ_INSTS_SMEM GRBM_GUI_ACTIVE --output-format csv -d $OUT/pass \
  -- python $ROOT/tools/chamfer_bench.py --reps 2 --shapes ${SHAPES:-3x64x10000} > $OUT/run.log 2>&1 || { tail -5 $OUT/run.log; exit 1; }
python - <<PY
import csv, glob, collections, json
f = glob.glob("$OUT/pass/*/*counter_collection.csv")[0]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0]
    if "nn_kernel" in k or "nn2_kernel" in k or "nn_query" in k or "nn_sort" in k or "nn_boxes" in k:
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {}
for k, d in agg.items():
    m = {c: sum(v) / len(v) for c, v in d.items()}
    m["launches"] = len(next(iter(d.values())))
    ga = m.get("GRBM_GUI_ACTIVE", 0)
    cyc = ga / 8 if ga else 0           # GRBM_GUI_ACTIVE is summed over the 8 XCDs
    m["kernel_cycles"] = cyc
    # wave-instructions issued per SIMD and cycle (1024 SIMDs); a wave64 fp32 VALU op occupies its SIMD for 2 cycles on gfx950
    m["valu_insts_per_simd_cycle"] = m.get("SQ_INSTS_VALU", 0) / (cyc * 1024) if cyc else 0.0
    m["valu_issue_frac_at_2cyc"] = 2.0 * m["valu_insts_per_simd_cycle"]
    wc = m.get("SQ_WAVE_CYCLES", 0) or 1
    m["wait_any_frac"] = m.get("SQ_WAIT_ANY", 0) / wc
    m["wait_inst_frac"] = m.get("SQ_WAIT_INST_ANY", 0) / wc
    res[k] = m
json.dump(res, open("$OUT/summary.json", "w"), indent=1)
for k, m in res.items():
    print(k[:60], {c: (round(v, 4) if isinstance(v, float) else v) for c, v in m.items()})
PY
