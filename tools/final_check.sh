# Round-end validation on the GPU box: bench (default flags), smoke, profile, PMC traffic passes, then the whole -m gpu suite.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 600 python bench.py > $O/final_bench.log 2>&1; echo "bench rc=$?" >> $O/final_bench.log
tail -2 $O/final_bench.log | cut -c1-400
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/final_smoke.log 2>&1; tail -1 $O/final_smoke.log
cd /tmp && export TMPDIR=/tmp
B="python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-roofline"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_final -o r1 -- $B > $O/final_prof.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/pmc_final_c -o c -- $B > $O/final_pmc_c.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc TCC_EA0_WRREQ_sum --output-format csv -d $O/pmc_final_d -o d -- $B > $O/final_pmc_d.log 2>&1
find $O/pmc_final_c $O/pmc_final_d -name "*kernel_trace*" -delete 2>/dev/null
cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -x -q -m gpu > $O/final_tests.log 2>&1; echo "rc=$?" >> $O/final_tests.log
tail -3 $O/final_tests.log
