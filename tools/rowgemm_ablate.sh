#!/bin/bash
# Developer aid (GPU box): time the rowgemm ablation builds of tools/build_variants.sh rowgemm on one stack fwd+bwd.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for v in ${VARIANTS:-"" RG_NOEPI RG_NOSTORE RG_NOA RG_NOB RG_NODMA RG_NOMFMA RG_NODMA_NOEPI}; do
  if [ -n "$v" ]; then export A3VT_LIB=$ROOT/gpurun_variants/liba3vt_$v.so; else unset A3VT_LIB; fi
  echo "== ${v:-shipped}"
  python $ROOT/tools/stack_bench.py --layers 6 --reps 3 "$@" 2>/dev/null | grep -E "fwd Z|bwd dX"
done
