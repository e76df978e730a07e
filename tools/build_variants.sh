#!/bin/bash
# Developer aid: build ablation variants of the library (timing only; outputs are wrong by design).
set -e
cd "$(dirname "$0")/../active-3d-vision-and-touch_amd/csrc"
mkdir -p ../../gpurun_variants
build() { # name flags...
  local name=$1; shift
  hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -shared "$@" capi.hip gcn_gemm.hip gcn_csr.hip posenc.hip sample.hip chamfer.hip pooling.hip -o ../../gpurun_variants/liba3vt_$name.so
}
build V1 -DA3VT_DBG_NODMA &
build V2 -DA3VT_DBG_NOLDSREAD &
build V3 -DA3VT_DBG_NOBARRIER &
build V4 -DA3VT_DBG_NODMA -DA3VT_DBG_NOEPI -DA3VT_DBG_NOLDSREAD -DA3VT_DBG_NOBARRIER &
wait
ls ../../gpurun_variants
