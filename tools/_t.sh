cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 740 python -m pytest tests -x -q -m gpu > gpurun_out/last_tests.log 2>&1; echo "rc=$?" >> gpurun_out/last_tests.log; tail -4 gpurun_out/last_tests.log | cut -c1-160
