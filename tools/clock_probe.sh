#!/bin/bash
# Developer aid (GPU box): sample the shader clock / power while rowgemm_bench.py loops, for the shipped library and for
# timing-only ablation builds (tools/build_variants.sh rowgemm).  Usage: tools/clock_probe.sh [variant ...]
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for v in "" "$@"; do
  if [ -n "$v" ]; then export A3VT_LIB=$ROOT/gpurun_variants/liba3vt_$v.so; else unset A3VT_LIB; fi
  echo "== ${v:-shipped}"
  REPS=${REPS:-6000} python $ROOT/tools/rowgemm_bench.py 2>/dev/null &
  pid=$!
  sleep ${WARM:-14}
  for i in 1 2 3; do
    rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | tr -s ' ' | head -4
    sleep 0.3
  done
  wait $pid
done
