cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_adam.py -q -m gpu 2>&1 | grep -v "^$" | grep -B5 -A25 "Error\|assert\|passed" | head -60
