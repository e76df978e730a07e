#!/bin/bash
# SQ-level PMC counters of the hot kernels (run on the GPU box):  tools/collect_sq.sh [--precision bf16s]
# One --pmc pass with the 8 SQ slots + GRBM_GUI_ACTIVE (MI355X_MICROARCH.md "rocprofv3 PMC slots"); kernel-trace only.
# Writes gpurun_out/sq/*.csv and gpurun_out/sq/summary.json (copy to profiles/ to keep).
set -e
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/sq
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY \
  SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/pass \
  -- python $ROOT/tools/stack_bench.py --layers 4 --reps 2 "$@" > $OUT/run.log 2>&1 || { tail -5 $OUT/run.log; exit 1; }
python - <<PY
import csv, glob, collections, json
f = glob.glob("$OUT/pass/*/*counter_collection.csv")[0]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]
    if any(t in k for t in ("rowgemm", "dw_kernel", "dww_kernel", "dw16", "dw3", "csr", "slab", "thin")):
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {}
for k, d in agg.items():
    m = {c: max(v) for c, v in d.items()}          # hidden x hidden launches are the largest
    m["launches"] = len(next(iter(d.values())))
    wc = m.get("SQ_WAVE_CYCLES", 0) or 1
    m["wait_any_frac"] = m.get("SQ_WAIT_ANY", 0) / wc
    m["wait_inst_frac"] = m.get("SQ_WAIT_INST_ANY", 0) / wc
    m["active_inst_frac"] = m.get("SQ_ACTIVE_INST_ANY", 0) / wc
    ia = m.get("SQ_LDS_IDX_ACTIVE", 0)
    m["lds_conflict_frac"] = m.get("SQ_LDS_BANK_CONFLICT", 0) / ia if ia else 0.0
    ga = m.get("GRBM_GUI_ACTIVE", 0)
    # MFMA-pipe busy share: busy cycles summed over the chip's 1024 SIMDs / (kernel cycles x 1024);
    # GRBM_GUI_ACTIVE is reported summed over the 8 XCDs, so kernel cycles = GRBM_GUI_ACTIVE / 8
    m["mfma_busy_frac"] = m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (ga / 8 * 1024) if ga else 0.0
    res[k] = m
json.dump(res, open("$OUT/summary.json", "w"), indent=1)
for k, m in res.items():
    print(k[:60], {c: round(v, 3) for c, v in m.items() if c.endswith("frac")})
PY
