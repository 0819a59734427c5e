cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
timeout 300 python tools/optable.py --json $O/r3b_optable.json > $O/r3b_optable.log 2>&1; tail -45 $O/r3b_optable.log
timeout 600 python -m pytest tests -x -q -m gpu -k "pmsqe" > $O/r3b_tests.log 2>&1; tail -3 $O/r3b_tests.log
