#!/bin/bash
# Developer aid (GPU box): csrq launch time of the shipped library against the gather ablations (tools/build_variants.sh csrq, stampsq).
O=gpurun_out/refresh; mkdir -p $O
(for v in "" CSRQ_NOGATHER CSRQ_NOINDEX; do echo "== ${v:-shipped}"; if [ -n "$v" ]; then export A3VT_LIB=$GRAFT_REPO_ROOT/gpurun_variants/liba3vt_$v.so; else unset A3VT_LIB; fi; bash tools/kstats.sh 2>/dev/null | grep csrq; done; unset A3VT_LIB) > $O/r04_csrq_ablation.txt
cat $O/r04_csrq_ablation.txt
