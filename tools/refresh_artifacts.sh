#!/bin/bash
# One pass over everything under profiles/ that a kernel change can move (run on the GPU box from the repo root; ~6 min):
#   tools/refresh_artifacts.sh r06      -> gpurun_out/refresh/r06_*   (copy what should be kept into profiles/)
set -x
R=${1:-r06}
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/refresh
mkdir -p $O
python bench.py > $O/${R}_bench_default.json 2> $O/bench.err
tail -c 700 $O/${R}_bench_default.json
python bench.py --gemm-precision bf16s --no-cpu-baseline > $O/${R}_bench_bf16s.json 2>/dev/null; tail -c 400 $O/${R}_bench_bf16s.json
python bench.py --gemm-precision bf16 --no-cpu-baseline > $O/${R}_bench_bf16_operand_mode.json 2>/dev/null; tail -c 400 $O/${R}_bench_bf16_operand_mode.json
python bench.py --gemm-precision fp32x3 --no-cpu-baseline > $O/${R}_bench_fp32x3.json 2>/dev/null; tail -c 400 $O/${R}_bench_fp32x3.json
python tools/measure_protocol.py > $O/${R}_timing_protocol.json 2> $O/protocol.err; tail -c 400 $O/${R}_timing_protocol.json
python tools/named_configs.py > $O/${R}_named_configs.jsonl 2> $O/named.err; cut -c 1-300 $O/${R}_named_configs.jsonl
python tools/named_configs.py --precision bf16 >> $O/${R}_named_configs.jsonl 2>> $O/named.err
python tools/named_configs.py --precision fp32 >> $O/${R}_named_configs.jsonl 2>> $O/named.err
python tools/named_configs.py --precision fp32x3 >> $O/${R}_named_configs.jsonl 2>> $O/named.err
python tools/touch_bench.py > $O/${R}_touch_topology_step.log 2>&1; tail -3 $O/${R}_touch_topology_step.log
python tools/touch_bench.py --precision fp32x3 2>/dev/null | tail -1 >> $O/${R}_touch_topology_step.log
(cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/bstats && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/bstats -- python $GRAFT_REPO_ROOT/bench.py --steps 15 --warmup 5 --no-cpu-baseline --no-traffic --alt-steps 0 > /tmp/bstats.log 2>&1; cp $(find /tmp/bstats -name '*kernel_stats.csv' | head -1) $O/${R}_bench_kernel_stats.csv; tail -c 300 /tmp/bstats.log)
(cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/bstats2 && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/bstats2 -- python $GRAFT_REPO_ROOT/bench.py --steps 15 --warmup 5 --no-cpu-baseline --no-traffic --gemm-precision bf16s > /tmp/bstats2.log 2>&1; cp $(find /tmp/bstats2 -name '*kernel_stats.csv' | head -1) $O/${R}_bench_bf16s_kernel_stats.csv)
(cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/bstats3 && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/bstats3 -- python $GRAFT_REPO_ROOT/bench.py --steps 15 --warmup 5 --no-cpu-baseline --no-traffic --alt-steps 0 --gemm-precision fp32x3 > /tmp/bstats3.log 2>&1; cp $(find /tmp/bstats3 -name '*kernel_stats.csv' | head -1) $O/${R}_bench_fp32x3_kernel_stats.csv)
# gemm mode 3: ablation builds (tools/build_variants.sh x3), phase stamps (stamps3), error table against the fp64 oracle
[ -f gpurun_variants/liba3vt_X3_NOMFMA.so ] && bash tools/x3_ablate.sh > $O/${R}_x3_ablation.txt 2>&1
[ -f gpurun_variants/liba3vt_RG3_STAMPS.so ] && A3VT_LIB=$GRAFT_REPO_ROOT/gpurun_variants/liba3vt_RG3_STAMPS.so python tools/rowgemm3_stamps.py > $O/${R}_rowgemm3_phase_stamps.txt 2>/dev/null
# round 6 product kernels: per-tile / per-stage phase stamps of rowgemmw_kernel and dww_kernel (tools/build_variants.sh rgw)
[ -f gpurun_variants/liba3vt_RGW_STAMPS.so ] && A3VT_LIB=$GRAFT_REPO_ROOT/gpurun_variants/liba3vt_RGW_STAMPS.so python tools/rowgemmw_stamps.py > $O/${R}_rowgemmw_dww_stamps.txt 2>/dev/null
[ -f gpurun_variants/liba3vt_RGW_OFF.so ] && (echo '== round-5 product kernels (liba3vt_RGW_OFF.so)'; A3VT_LIB=$GRAFT_REPO_ROOT/gpurun_variants/liba3vt_RGW_OFF.so python tools/stack_bench.py | tail -4 | head -3; A3VT_LIB=$GRAFT_REPO_ROOT/gpurun_variants/liba3vt_RGW_OFF.so python tools/stack_bench.py --reps 20 --no-profile | tail -1; echo '== shipped'; python tools/stack_bench.py | tail -4 | head -3; python tools/stack_bench.py --reps 20 --no-profile | tail -1) > $O/${R}_product_kernels_ab.txt 2>/dev/null
# round 6, bf16 configurations: the fused BatchNorm + ReLU operator against MIOpen's per map shape; the tiled aggregation's stamps and its A/B builds
python tools/bnrelu_bench.py > $O/${R}_bnrelu_vs_miopen.txt 2>/dev/null
python tools/conv5_bench.py > $O/${R}_conv5_vs_miopen.txt 2>/dev/null; tail -2 $O/${R}_conv5_vs_miopen.txt
[ -f gpurun_variants/liba3vt_T16_STAMPS.so ] && A3VT_LIB=$GRAFT_REPO_ROOT/gpurun_variants/liba3vt_T16_STAMPS.so python tools/csr16t_stamps.py > $O/${R}_csr16t_stamps.txt 2>/dev/null
# (whole forward + backward calls of one 20-layer stack without per-launch events, the builds interleaved, three rounds behind a warm-up process:
#  a launch's isolated rocprofv3 average overstates what the tiles gain in situ — 35 / 33 -> 28 / 24 us isolated, ~3 us per launch in the stack)
t16ab() { if [ "$1" = shipped ]; then python tools/stack_bench.py --precision bf16s --layers 20 --reps 30 --no-profile $2 | tail -1; else A3VT_LIB=$GRAFT_REPO_ROOT/gpurun_variants/liba3vt_$1.so python tools/stack_bench.py --precision bf16s --layers 20 --reps 30 --no-profile | tail -1; fi; }
[ -f gpurun_variants/liba3vt_T16_2BUF.so ] && (t16ab shipped > /dev/null; for round in 1 2 3; do
  echo "shipped (tiles of 64, one staging buffer, four workgroups per CU, streaming stores): $(t16ab shipped)"
  echo "T16_NO_NT (plain stores):                                                            $(t16ab T16_NO_NT)"
  echo "T16_2BUF (two staging buffers, two workgroups per CU):                               $(t16ab T16_2BUF)"
  echo "the row walk (a3vt_dbg_csr_algo = 1):                                                $(t16ab shipped '--csr-algo rows')"
done) > $O/${R}_csr16t_ab.txt 2>/dev/null
(./tools/ubench/mfma_plus_valu; ./tools/ubench/mfma_gap_budget; ./tools/ubench/mfma_one_wave) > $O/${R}_fp32_pipe_ubench.txt 2>&1
# channel-sliced aggregation: phase stamps per quad (tools/build_variants.sh stampsq) and the gather ablations (csrq)
[ -f gpurun_variants/liba3vt_CSRQ_STAMPS.so ] && A3VT_LIB=$GRAFT_REPO_ROOT/gpurun_variants/liba3vt_CSRQ_STAMPS.so python tools/csrq_stamps.py > $O/${R}_csrq_stamps.txt 2>/dev/null
[ -f gpurun_variants/liba3vt_CSRQ_NOINDEX.so ] && bash tools/csrq_ablate.sh > /dev/null 2>&1
python -m pytest tests/test_gpu_fullsize.py -q -s -k benchmark_configuration 2>&1 | grep "^\[configs\|passed\|failed" > $O/${R}_mode_error_table.txt; cat $O/${R}_mode_error_table.txt
python -m pytest tests/test_gpu_fp32x3.py -q -s -k vs_fp64 2>&1 | grep "^\[\|passed\|failed" >> $O/${R}_mode_error_table.txt
# configs[3]: the first steps run MIOpen's find mode, so the table is cut from the kernel TRACE after 5 steps (tools/trace_steady.py)
(cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/c3 && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/c3 -- python $GRAFT_REPO_ROOT/tools/named_configs.py --only 3 --steps 10 > /tmp/c3.log 2>&1; python $GRAFT_REPO_ROOT/tools/trace_steady.py $(find /tmp/c3 -name "*kernel_trace.csv" | head -1) --marker chamfer_bwd --skip 5 --top 70 > $O/${R}_config3_bf16s_steady_kernels.txt; head -8 $O/${R}_config3_bf16s_steady_kernels.txt)
# the exact search: both geometries ("bench" = the untrained network's concentric sphere / ellipsoids), counters, per-wave timeline
(python tools/chamfer_bench.py --geometry bench; python tools/chamfer_bench.py) > $O/${R}_chamfer_search.txt 2>/dev/null; cat $O/${R}_chamfer_search.txt
[ -f gpurun_variants/liba3vt_NN_STATS.so ] && (A3VT_LIB=gpurun_variants/liba3vt_NN_STATS.so python tools/nn_stats.py --geometry bench; A3VT_LIB=gpurun_variants/liba3vt_NN_STATS.so python tools/nn_stats.py; A3VT_LIB=gpurun_variants/liba3vt_NN_STATS.so python tools/nn_stats_config.py --which 3 --batch 16 | tail -1; A3VT_LIB=gpurun_variants/liba3vt_NN_STATS.so python tools/nn_stats_config.py --which 4 | tail -1) > $O/${R}_nn_pruning_stats.txt 2>/dev/null
[ -f gpurun_variants/liba3vt_NN_TRACE.so ] && A3VT_LIB=gpurun_variants/liba3vt_NN_TRACE.so python tools/nn_trace.py > $O/${R}_nn_wave_timeline.txt 2>/dev/null
[ -f gpurun_variants/liba3vt_NN_AABB.so ] && (A3VT_LIB=gpurun_variants/liba3vt_NN_AABB.so python tools/chamfer_bench.py --geometry bench --algos pruned; A3VT_LIB=gpurun_variants/liba3vt_NN_AABB.so python tools/chamfer_bench.py --algos pruned) > $O/${R}_chamfer_search_axis_aligned_boxes.txt 2>/dev/null
python tools/score_bench.py 2>/dev/null | tail -1 > $O/${R}_scoring_batched_vs_sequential.json; cut -c 1-400 $O/${R}_scoring_batched_vs_sequential.json
python tools/host_bound.py 2>/dev/null | tail -2 > $O/${R}_named_configs_host_time.txt
bash tools/collect_nn.sh > $O/nn.log 2>&1; cp gpurun_out/nn/summary.json $O/${R}_pmc_nn_summary.json
bash tools/collect_traffic.sh > $O/traffic.log 2>&1; cp gpurun_out/traffic/summary.json $O/${R}_pmc_traffic_summary.json
bash tools/collect_sq.sh > $O/sq.log 2>&1; cp gpurun_out/sq/summary.json $O/${R}_pmc_sq_summary.json
bash tools/collect_sq.sh --precision bf16s > $O/sq16.log 2>&1; cp gpurun_out/sq/summary.json $O/${R}_pmc_sq_summary_bf16s.json
bash tools/collect_traffic.sh --precision fp32x3 > $O/traffic3.log 2>&1; cp gpurun_out/traffic/summary.json $O/${R}_pmc_traffic_summary_fp32x3.json
bash tools/collect_sq.sh --precision fp32x3 > $O/sq3.log 2>&1; cp gpurun_out/sq/summary.json $O/${R}_pmc_sq_summary_fp32x3.json
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline --no-traffic 2>/dev/null | tail -c 400
python tools/kstats_summary.py $O/${R}_bench_kernel_stats.csv 22 16
python tools/kstats_summary.py $O/${R}_bench_bf16s_kernel_stats.csv 22 12
python tools/kstats_summary.py $O/${R}_bench_fp32x3_kernel_stats.csv 22 14
