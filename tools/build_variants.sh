#!/bin/bash
# Developer aid: build ablation variants of the library (timing only; outputs are wrong by design).
set -e
cd "$(dirname "$0")/../active-3d-vision-and-touch_amd/csrc"
mkdir -p ../../gpurun_variants
build() { # name flags...
  local name=$1; shift
  hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -shared "$@" capi.hip gcn_gemm.hip gcn_csr.hip gcn_bf16s.hip posenc.hip sample.hip -fno-slp-vectorize chamfer.hip nn_prune.hip pooling.hip -o ../../gpurun_variants/liba3vt_$name.so
}
if [ "$1" = "csr" ]; then   # csr_fwd / csr16_fwd without their stores: tools/kstats.sh with A3VT_LIB=...
  build CSR_NOSTORE -DA3VT_DBG_CSR_NOSTORE
elif [ "$1" = "nn" ]; then   # pruned nearest-neighbour search with its counters (tools/nn_stats.py)
  build NN_STATS -DA3VT_DBG_NN_STATS
elif [ "$1" = "prefetch" ]; then   # rowgemm prefetch-depth experiments: tools/rowgemm_bench.py / stack_bench.py with A3VT_LIB=...
  build RG_NOSPREAD -DA3VT_DBG_RG_NOSPREAD &
  build RG_NSTAGE4 -DA3VT_DBG_RG_NSTAGE4 &
  build RG_NSTAGE4_NOSPREAD -DA3VT_DBG_RG_NSTAGE4 -DA3VT_DBG_RG_NOSPREAD &
  wait
elif [ "$1" = "rowgemm" ]; then   # rowgemm_kernel ablations (all gemm modes): python tools/stack_bench.py with A3VT_LIB=...
  build RG_NOEPI -DA3VT_DBG_RG_NOEPI &
  build RG_NOA -DA3VT_DBG_RG_NOA &
  build RG_NOB -DA3VT_DBG_RG_NOB &
  build RG_NODMA -DA3VT_DBG_RG_NOA -DA3VT_DBG_RG_NOB &
  wait
  build RG_NOMFMA -DA3VT_DBG_RG_NOMFMA &
  build RG_NOSTORE -DA3VT_DBG_RG_NOSTORE &
  build RG_NODMA_NOEPI -DA3VT_DBG_RG_NOA -DA3VT_DBG_RG_NOB -DA3VT_DBG_RG_NOEPI &
  wait
else
build V1 -DA3VT_DBG_NODMA &
build V2 -DA3VT_DBG_NOLDSREAD &
build V3 -DA3VT_DBG_NOBARRIER &
build V4 -DA3VT_DBG_NODMA -DA3VT_DBG_NOEPI -DA3VT_DBG_NOLDSREAD -DA3VT_DBG_NOBARRIER &
wait
fi
ls ../../gpurun_variants
