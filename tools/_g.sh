cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -k "2405 or 2407 or 7001" > $O/r5c_slabtest.log 2>&1; tail -30 $O/r5c_slabtest.log | cut -c1-300
