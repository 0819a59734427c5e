cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/run13_tests.log 2>&1; echo "rc=$?" >> gpurun_out/run13_tests.log
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/run13_bench_bf16.log 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_r1_bf16_v6 -o r1 -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/run13_prof.log 2>&1
cd $GRAFT_REPO_ROOT
tail -3 gpurun_out/run13_tests.log; tail -1 gpurun_out/run13_bench_bf16.log | cut -c1-300
