cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out/wrw; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 | tee $O/gpu_tests.txt
