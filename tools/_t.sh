cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -m gpu --durations=25 > gpurun_out/dur_tests.log 2>&1; tail -40 gpurun_out/dur_tests.log | cut -c1-160
