#!/bin/bash
# Developer aid: build ablation / diagnostic variants of the library into gpurun_variants/ (timing only; the outputs of
# most of them are wrong by design).  Every variant is built by lib.build itself — per file, with the shipped library's
# flags (chamfer.hip / nn_prune.hip alone get -fno-slp-vectorize) plus the -D switches — so a variant differs from the
# shipped build by its switches only.  Select one at run time with A3VT_LIB=gpurun_variants/liba3vt_<name>.so.
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
mkdir -p "$ROOT/gpurun_variants"
build() { # name defines...
  local name=$1; shift
  local defs=""
  for d in "$@"; do defs="$defs'${d#-D}',"; done
  (cd "$ROOT" && python -c "from a3vt_amd import lib; lib.build(defines=[$defs], out='$ROOT/gpurun_variants/liba3vt_$name.so')")
}
if [ "$1" = "rgw" ]; then   # rowgemmw_kernel (round 6) ablations + the round-5 kernel for A/B: python tools/stack_bench.py with A3VT_LIB=...
  build RGW_OFF -DA3VT_DBG_RGW_OFF -DA3VT_DBG_DWW_OFF   # the round-5 product kernels (rowgemm_kernel<ADIRECT>, dw_kernel<fast, hybrid>)
  build RGW_NODMA -DA3VT_DBG_RGW_NODMA
  build RGW_NOEPI -DA3VT_DBG_RGW_NOEPI
  build RGW_MFMAONLY -DA3VT_DBG_RGW_NODMA -DA3VT_DBG_RGW_NOEPI
  build RGW_STAMPS -DA3VT_DBG_RGW_STAMPS   # per-tile phase stamps (tools/rowgemmw_stamps.py)
elif [ "$1" = "t16" ]; then   # csr16t_fwd_kernel (bf16 tiled aggregation) with wall-clock stamps at its phase boundaries (tools/csr16t_stamps.py)
  build T16_STAMPS -DA3VT_DBG_T16_STAMPS
  build T16_NO_NT -DA3VT_T16_NO_NT    # A/B: plain instead of streaming (nontemporal) stores of the results
  build T16_2BUF -DA3VT_T16_BUFS=2    # A/B: two staging buffers, two workgroups per CU
elif [ "$1" = "csr" ]; then   # csr_fwd / csr16_fwd without their stores: tools/kstats.sh with A3VT_LIB=...
  build CSR_NOSTORE -DA3VT_DBG_CSR_NOSTORE
elif [ "$1" = "csrq" ]; then   # channel-sliced aggregation without its LDS gathers (what the gathers cost)
  build CSRQ_NOGATHER -DA3VT_DBG_CSRQ_NOGATHER
  build DW_NOWRAP -DA3VT_DBG_DW_NOWRAP   # dw without its mesh-boundary bookkeeping
  build DW_NOHYB -DA3VT_DBG_DW_NOHYB     # the plain dw kernel on the same buffers
elif [ "$1" = "epi" ]; then   # rowgemm / rowgemm3 epilogues with their lane arithmetic hoisted (and spilled) as before round 4: A/B timing
  build RG_HOISTED_EPI -DA3VT_DBG_RG_HOISTED_EPI
elif [ "$1" = "stamps3" ]; then   # rowgemm3 (gemm mode 3) with stamps at its phase boundaries (tools/rowgemm3_stamps.py)
  build RG3_STAMPS -DA3VT_DBG_RG3_STAMPS
elif [ "$1" = "x3" ]; then   # gemm mode 3 (gcn_gemm3.hip) ablations: python tools/stack_bench.py --precision fp32x3 with A3VT_LIB=...
  build X3_NOMFMA -DA3VT_DBG_X3_NOMFMA
  build X3_NOSPLIT -DA3VT_DBG_X3_NOSPLIT
  build RG3_NOA -DA3VT_DBG_RG3_NOA
  build RG3_NOB -DA3VT_DBG_RG3_NOB
  build RG3_NOEPI -DA3VT_DBG_RG3_NOEPI
  build RG3_NODMA_NOEPI -DA3VT_DBG_RG3_NOA -DA3VT_DBG_RG3_NOB -DA3VT_DBG_RG3_NOEPI
  build DW3_NODMA -DA3VT_DBG_DW3_NODMA
  build DW3_NOSPLITPHASE -DA3VT_DBG_DW3_NOSPLITPHASE
  build DW3_MFMAONLY -DA3VT_DBG_DW3_NODMA -DA3VT_DBG_DW3_NOSPLITPHASE
elif [ "$1" = "stampsq" ]; then   # csrq_kernel with stamps per quad (tools/csrq_stamps.py)
  build CSRQ_STAMPS -DA3VT_DBG_CSRQ_STAMPS
  build CSRQ_NOINDEX -DA3VT_DBG_CSRQ_NOINDEX   # timing-only: conflict-free gathers (fixed offsets instead of the index image)
elif [ "$1" = "posenc" ]; then   # posenc_bwd ablations: python tools/posenc_bench.py with A3VT_LIB=...
  build PE_NOSINCOS -DA3VT_DBG_PE_NOSINCOS
  build PE_NOOUTER -DA3VT_DBG_PE_NOOUTER
  build PE_NOLOOP -DA3VT_DBG_PE_NOLOOP
elif [ "$1" = "env" ]; then   # the developer environment overrides (A3VT_RG_MAXWG, A3VT_ROWTILE_SEPARATE, A3VT_NN_R, A3VT_NN_ALGO): compiled out of the shipped library
  build ENV -DA3VT_DBG_ENV
elif [ "$1" = "nn" ]; then   # pruned nearest-neighbour search with its counters (tools/nn_stats.py)
  build NN_STATS -DA3VT_DBG_NN_STATS
  build NN_AABB -DA3VT_DBG_NN_AABB       # A/B: the boxes in the coordinate frame (axis-aligned, rounds 2-4) instead of the principal frame
  build NN_TRACE -DA3VT_DBG_NN_TRACE     # per-wave timeline only (tools/nn_trace.py): start / end / groups evaluated
elif [ "$1" = "stamps" ]; then   # rowgemm with s_memrealtime stamps at its phase boundaries (tools/rowgemm_stamps.py)
  build RG_STAMPS -DA3VT_DBG_RG_STAMPS
elif [ "$1" = "adirect" ]; then   # A/B builds for the round-3 product kernels (A3VT_LIB=$GRAFT_REPO_ROOT/gpurun_variants/...)
  build RG_ADIRECT_OFF -DA3VT_DBG_RG_ADIRECT_OFF     # fp32 rowgemm with the A operand through the LDS ring
  build ROWGEMM16_OFF -DA3VT_DBG_ROWGEMM16_OFF       # bf16 storage mode on rowgemm_kernel<..., 2> instead of rowgemm16
elif [ "$1" = "stamps16" ]; then   # rowgemm16 (bf16 storage mode) with stamps at its phase boundaries (tools/rowgemm16_stamps.py)
  build R16_STAMPS -DA3VT_DBG_R16_STAMPS
  build R16_STAMPS_NOB -DA3VT_DBG_R16_STAMPS -DA3VT_DBG_R16_NOB   # ... without the weight loads of the prologue
elif [ "$1" = "prefetch" ]; then   # rowgemm prefetch-depth experiments: tools/rowgemm_bench.py / stack_bench.py with A3VT_LIB=...
  build RG_NOSPREAD -DA3VT_DBG_RG_NOSPREAD
  build RG_NSTAGE4 -DA3VT_DBG_RG_NSTAGE4
  build RG_NSTAGE4_NOSPREAD -DA3VT_DBG_RG_NSTAGE4 -DA3VT_DBG_RG_NOSPREAD
elif [ "$1" = "rowgemm" ]; then   # rowgemm_kernel ablations (all gemm modes): python tools/stack_bench.py with A3VT_LIB=...
  build RG_NOEPI -DA3VT_DBG_RG_NOEPI
  build RG_NOA -DA3VT_DBG_RG_NOA
  build RG_NOB -DA3VT_DBG_RG_NOB
  build RG_NODMA -DA3VT_DBG_RG_NOA -DA3VT_DBG_RG_NOB
  build RG_NOMFMA -DA3VT_DBG_RG_NOMFMA
  build RG_NOSTORE -DA3VT_DBG_RG_NOSTORE
  build RG_NODMA_NOEPI -DA3VT_DBG_RG_NOA -DA3VT_DBG_RG_NOB -DA3VT_DBG_RG_NOEPI
else
  build V1 -DA3VT_DBG_NODMA
  build V2 -DA3VT_DBG_NOLDSREAD
  build V3 -DA3VT_DBG_NOBARRIER
  build V4 -DA3VT_DBG_NODMA -DA3VT_DBG_NOEPI -DA3VT_DBG_NOLDSREAD -DA3VT_DBG_NOBARRIER
fi
ls "$ROOT/gpurun_variants"
