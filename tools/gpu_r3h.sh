cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
timeout 1500 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "DCCRN" > $O/r3h_tests.log 2>&1; tail -4 $O/r3h_tests.log
timeout 300 python tools/optable.py > $O/r3h_optable.log 2>&1; tail -32 $O/r3h_optable.log | head -14
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > $O/r3h_bench.log 2>&1; tail -1 $O/r3h_bench.log | cut -c1-200
SEFD_BN_FUSE=0 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > $O/r3h_bench0.log 2>&1; tail -1 $O/r3h_bench0.log | cut -c1-200
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > $O/r3h_bench1.log 2>&1; tail -1 $O/r3h_bench1.log | cut -c1-200
