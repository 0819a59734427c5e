# Round-2 validation on the GPU box: the -m gpu suite, smoke, the three bench lines, kernel-trace profile and PMC traffic passes
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 2400 python -m pytest tests -q -m gpu --durations=8 > $O/r2_final_tests.log 2>&1; echo "rc=$?" >> $O/r2_final_tests.log
tail -14 $O/r2_final_tests.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/r2_final_smoke.log 2>&1; tail -1 $O/r2_final_smoke.log
timeout 700 python bench.py > $O/r2_final_bench.log 2>&1; echo "rc=$?" >> $O/r2_final_bench.log
tail -2 $O/r2_final_bench.log | cut -c1-300
timeout 400 python bench.py --batch 64 --steps 30 --warmup 5 --no-cpu-baseline > $O/r2_final_bench_b64.log 2>&1
tail -1 $O/r2_final_bench_b64.log | cut -c1-200
for m in dccrn_large fullsubnet; do
  timeout 400 python bench.py --model $m --steps 20 --warmup 5 --no-cpu-baseline > $O/r2_final_bench_$m.log 2>&1
  tail -1 $O/r2_final_bench_$m.log | cut -c1-200
done
cd /tmp && export TMPDIR=/tmp
B="python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-roofline"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_r2f -o r2 -- $B > $O/r2_final_prof.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/pmc_r2f_c -o c -- $B > $O/r2_final_pmc_c.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc TCC_EA0_WRREQ_sum --output-format csv -d $O/pmc_r2f_d -o d -- $B > $O/r2_final_pmc_d.log 2>&1
for m in dccrn_large fullsubnet; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_r2f_$m -o r2 -- python $GRAFT_REPO_ROOT/bench.py --model $m --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > $O/r2_final_prof_$m.log 2>&1
done
find $O/pmc_r2f_c $O/pmc_r2f_d $O/prof_r2f $O/prof_r2f_dccrn_large $O/prof_r2f_fullsubnet -name "*kernel_trace*" -delete 2>/dev/null
ls $O/prof_r2f $O/pmc_r2f_c | head
