cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
( time timeout 1500 python -m pytest tests -x -q -m gpu --durations=25 ) > $O/r5p_tests.log 2>&1; tail -34 $O/r5p_tests.log | cut -c1-160
