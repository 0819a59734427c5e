cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/run15_bench_bf16.log 2>&1
tail -1 $O/run15_bench_bf16.log | cut -c1-200
timeout 1500 python -m pytest tests -x -q -m gpu -k "kn2 or kn0 or golden or CRN" > $O/run15_tests.log 2>&1; echo "rc=$?" >> $O/run15_tests.log
tail -3 $O/run15_tests.log
