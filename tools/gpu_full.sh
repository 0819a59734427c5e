# whole GPU suite + smoke + default bench + kernel stats + PMC traffic passes (what the driver runs at round end, plus the profiles)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
timeout 2400 python -m pytest tests -x -q -m gpu > $O/full_tests.log 2>&1; echo "rc=$?" >> $O/full_tests.log; tail -4 $O/full_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/full_smoke.log 2>&1; tail -1 $O/full_smoke.log
timeout 600 python bench.py --steps 20 --warmup 5 > $O/full_bench.log 2>&1; tail -1 $O/full_bench.log | cut -c1-300
cd /tmp && export TMPDIR=/tmp
B="python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-roofline"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_r3 -o r3 -- $B > $O/full_prof.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/pmc_r3_c -o c -- $B > $O/full_pmc_c.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc TCC_EA0_WRREQ_sum --output-format csv -d $O/pmc_r3_d -o d -- $B > $O/full_pmc_d.log 2>&1
find $O/pmc_r3_c $O/pmc_r3_d -name "*kernel_trace*" -delete 2>/dev/null
ls $O/prof_r3 $O/pmc_r3_c $O/pmc_r3_d
