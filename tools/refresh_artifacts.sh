set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/refresh
python bench.py > gpurun_out/refresh/bench.json 2> gpurun_out/refresh/bench.err
tail -c 600 gpurun_out/refresh/bench.json
python tools/measure_protocol.py > gpurun_out/refresh/protocol.log 2>&1; tail -3 gpurun_out/refresh/protocol.log
(cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/bstats && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/bstats -- python $GRAFT_REPO_ROOT/bench.py --steps 15 > /tmp/bstats.log 2>&1; cp $(find /tmp/bstats -name '*kernel_stats.csv' | head -1) $GRAFT_REPO_ROOT/gpurun_out/refresh/kernel_stats.csv; tail -c 300 /tmp/bstats.log)
python bench.py --gemm-precision bf16 --no-cpu-baseline > gpurun_out/refresh/bench_bf16.json 2>/dev/null; tail -c 400 gpurun_out/refresh/bench_bf16.json
python tools/named_configs.py > gpurun_out/refresh/named.jsonl 2> gpurun_out/refresh/named.err; cat gpurun_out/refresh/named.jsonl | cut -c 1-300
python tools/touch_bench.py > gpurun_out/refresh/touch.log 2>&1; tail -3 gpurun_out/refresh/touch.log
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 5 --warmup 2 2>/dev/null | tail -c 400
