cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out/wrw; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_bnrelu.py -x -q -m gpu -k conv5 2>&1 | tail -15
(cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/cb && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/cb -- python $GRAFT_REPO_ROOT/tools/conv5_bench.py > /tmp/cb.log 2>&1; grep "stride" /tmp/cb.log; python - <<'P'
import csv,glob
f=glob.glob('/tmp/cb/**/*kernel_stats.csv',recursive=True)[0]
for r in csv.DictReader(open(f)):
    n=r['Name']
    if 'wrw' in n or 'SubTensor' in n:
        print(f"{n[:90]:90s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:8.1f} us")
P
)
timeout 300 python tools/named_configs.py --only 3 2>/dev/null | grep -o '"ms_per_step": [0-9.]*, "device_ms_median"'
