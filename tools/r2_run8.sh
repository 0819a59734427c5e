cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 1700 python -m pytest tests -q -m gpu --durations=12 > $O/r2_run8_tests.log 2>&1; echo "rc=$?" >> $O/r2_run8_tests.log
tail -22 $O/r2_run8_tests.log
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/r2_run8_bench.log 2>&1
tail -1 $O/r2_run8_bench.log | cut -c1-300
