# whole GPU suite + smoke + default bench (what the driver runs at round end)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
timeout 2400 python -m pytest tests -x -q -m gpu > $O/full_tests.log 2>&1; echo "rc=$?" >> $O/full_tests.log; tail -6 $O/full_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/full_smoke.log 2>&1; tail -1 $O/full_smoke.log
timeout 600 python bench.py > $O/full_bench.log 2>&1; tail -1 $O/full_bench.log | cut -c1-300
