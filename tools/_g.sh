cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
timeout 900 python -m pytest tests/test_gpu_model.py -x -q -k "direct or hamming or rectwin" > $O/r5l_test.log 2>&1; tail -15 $O/r5l_test.log | cut -c1-300
