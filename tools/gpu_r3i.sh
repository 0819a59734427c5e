cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
timeout 1500 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "DCCRN and (4001 or 2403 or 2401 or E-kn0 or C-kn3)" > $O/r3i_tests.log 2>&1; tail -4 $O/r3i_tests.log
timeout 300 python tools/optable.py > $O/r3i_optable.log 2>&1; tail -32 $O/r3i_optable.log | head -14
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > $O/r3i_bench.log 2>&1; tail -1 $O/r3i_bench.log | cut -c1-200
SEFD_WG_SWAP=0 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > $O/r3i_bench0.log 2>&1; tail -1 $O/r3i_bench0.log | cut -c1-200
