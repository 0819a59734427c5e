cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 600 python -m pytest tests/test_gpu_pmsqe.py -q -x > $O/r2_run29_tests.log 2>&1; tail -15 $O/r2_run29_tests.log
for p in PMSQE LMS; do timeout 300 python bench.py --perceptual $p --steps 30 --warmup 5 --no-roofline > $O/r2_bench_$p.log 2>&1; tail -1 $O/r2_bench_$p.log | cut -c1-400; done
