cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out/quick; mkdir -p $O
(cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/bstats && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/bstats -- python $GRAFT_REPO_ROOT/bench.py --steps 15 --warmup 5 --no-cpu-baseline --no-traffic --alt-steps 0 --no-named-configs > /tmp/bstats.log 2>&1; cp $(find /tmp/bstats -name '*kernel_stats.csv' | head -1) $O/bench_kernel_stats.csv)
python tools/kstats_summary.py $O/bench_kernel_stats.csv 22 16
(cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/c3 && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/c3 -- python $GRAFT_REPO_ROOT/tools/named_configs.py --only 3 --steps 10 > /tmp/c3.log 2>&1; python $GRAFT_REPO_ROOT/tools/trace_steady.py $(find /tmp/c3 -name '*kernel_trace.csv' | head -1) --skip 5 --top 40 > $O/config3_steady.txt; head -42 $O/config3_steady.txt)
python tools/score_bench.py 2>&1 | tail -3
