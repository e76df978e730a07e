cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_adam.py tests/test_gpu_trainer.py -x -q -m gpu 2>&1 | tail -8
for r in 1 2; do
timeout 300 python tools/named_configs.py --only 3 2>/dev/null | grep -o '"ms_per_step": [0-9.]*, "device_ms_median"'
timeout 300 python tools/named_configs.py --only 3 --torch-adam 2>/dev/null | grep -o '"ms_per_step": [0-9.]*, "device_ms_median"' | sed 's/^/torch-adam /'
done
timeout 300 python bench.py --no-cpu-baseline --no-traffic --alt-steps 0 --no-named-configs 2>/dev/null | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' | head -2
timeout 300 python bench.py --no-cpu-baseline --no-traffic --alt-steps 0 --no-named-configs --torch-adam 2>/dev/null | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' | head -2 | sed 's/^/torch-adam /'
