cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "4001 or 2403" > $O/r3e_tests.log 2>&1; tail -3 $O/r3e_tests.log
timeout 300 python tools/optable.py > $O/r3e_optable.log 2>&1; grep "RUNGEMM" $O/r3e_optable.log | awk '$7+0<=64 && $5+0 > 100000' | head -40; tail -30 $O/r3e_optable.log | head -8
