cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 2400 python -m pytest tests -q -m gpu --durations=15 > $O/r2_run19_tests.log 2>&1; echo "rc=$?" >> $O/r2_run19_tests.log
tail -28 $O/r2_run19_tests.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/r2_run19_smoke.log 2>&1; tail -1 $O/r2_run19_smoke.log
