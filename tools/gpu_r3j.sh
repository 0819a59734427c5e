cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
timeout 1500 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "DCCRN and (4001 or 2403 or 2401 or C-kn3 or R-kn2)" > $O/r3j_tests.log 2>&1; tail -4 $O/r3j_tests.log
timeout 600 python bench.py --steps 20 --warmup 5 > $O/r3j_bench.log 2>&1; tail -1 $O/r3j_bench.log | cut -c1-2500
