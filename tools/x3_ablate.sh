#!/bin/bash
# Developer aid (GPU box): times the gemm-mode-3 ablation builds of `tools/build_variants.sh x3` on one stack
# (per-class MFMA launch times from the library's own HIP events).  Timing only: the variants' outputs are wrong by design.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
echo "== shipped"; python tools/stack_bench.py --precision fp32x3 "$@" 2>&1 | grep "launches\|stack"
for v in X3_NOMFMA X3_NOSPLIT RG3_NOA RG3_NOB RG3_NOEPI RG3_NODMA_NOEPI DW3_NODMA DW3_NOSPLITPHASE DW3_MFMAONLY; do
  f=$ROOT/gpurun_variants/liba3vt_$v.so
  [ -f "$f" ] || continue
  echo "== $v"; A3VT_LIB=$f python tools/stack_bench.py --precision fp32x3 "$@" 2>&1 | grep "launches"
done
