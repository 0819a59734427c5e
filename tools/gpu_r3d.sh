cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "4001 or 2403" > $O/r3d_tests.log 2>&1; tail -12 $O/r3d_tests.log
timeout 300 python tools/optable.py > $O/r3d_optable.log 2>&1; grep "RUNGEMM" $O/r3d_optable.log | awk '$7+0<=64' | head -40; tail -30 $O/r3d_optable.log | head -8
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > $O/r3d_bench.log 2>&1; tail -1 $O/r3d_bench.log | cut -c1-200
