# Round-6 final artefacts on one GPU box: whole -m gpu suite, smoke, bench lines (default with roofline + CPU baseline, B64, large, fullsubnet, PMSQE),
# kernel stats + timeline + PMC traffic (separate --pmc passes) for the default step and for the two other models.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
timeout 2400 python -m pytest tests -x -q -m gpu > $O/fin_tests.log 2>&1; echo "rc=$?" >> $O/fin_tests.log; tail -4 $O/fin_tests.log | cut -c1-200
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/fin_smoke.log 2>&1; tail -1 $O/fin_smoke.log
pmc() {   # $1 = tag, $2.. = bench args ; two counter passes -> $O/fin_pmc_$1.json
  local tag=$1; shift
  ( cd /tmp && export TMPDIR=/tmp
    B="python $GRAFT_REPO_ROOT/bench.py $* --steps 4 --warmup 2 --no-cpu-baseline --no-roofline --no-extra"
    timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/fin_prof_$tag -o k -- $B > $O/fin_prof_$tag.log 2>&1
    timeout 900 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/fin_pmc_c_$tag -o c -- $B > $O/fin_pmc_c_$tag.log 2>&1
    timeout 900 rocprofv3 --kernel-trace --pmc TCC_EA0_WRREQ_sum --output-format csv -d $O/fin_pmc_d_$tag -o d -- $B > $O/fin_pmc_d_$tag.log 2>&1 )
  python tools/pmc_traffic.py $O/fin_pmc_$tag.json 6 $(find $O/fin_pmc_c_$tag $O/fin_pmc_d_$tag -name "*counter_collection.csv") > $O/fin_pmc_$tag.log 2>&1
  find $O/fin_pmc_c_$tag $O/fin_pmc_d_$tag -name "*kernel_trace*" -delete 2>/dev/null
  find $O/fin_pmc_c_$tag $O/fin_pmc_d_$tag -name "*counter_collection.csv" -delete 2>/dev/null
  cp $O/fin_prof_$tag/k_kernel_stats.csv $O/fin_kernel_stats_$tag.csv 2>/dev/null
}
pmc default
cp $O/fin_pmc_default.json profiles/r06_pmc_traffic.json 2>/dev/null     # the bench line below reads the traffic of THIS build's kernels
python tools/timeline.py $O/fin_prof_default/k_kernel_trace.csv 1 v > $O/fin_timeline_default.txt 2>&1
pmc dccrn_large --model dccrn_large
cp $O/fin_pmc_dccrn_large.json profiles/r06_pmc_traffic_dccrn_large.json 2>/dev/null
pmc fullsubnet --model fullsubnet
cp $O/fin_pmc_fullsubnet.json profiles/r06_pmc_traffic_fullsubnet.json 2>/dev/null
TIMELINE_MARK=fsn_in_kernel:1 python tools/timeline.py $O/fin_prof_fullsubnet/k_kernel_trace.csv 1 v > $O/fin_timeline_fullsubnet.txt 2>&1
find $O -name "k_kernel_trace.csv" -size +20M -delete 2>/dev/null
timeout 900 python bench.py > $O/fin_bench_default.log 2>&1; tail -1 $O/fin_bench_default.log | cut -c1-250
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-extra > $O/fin_bench_driver.log 2>&1; tail -1 $O/fin_bench_driver.log | cut -c60-180
timeout 600 python bench.py --batch 64 --no-cpu-baseline > $O/fin_bench_B64.log 2>&1; tail -1 $O/fin_bench_B64.log | cut -c60-180
timeout 600 python bench.py --model dccrn_large --no-cpu-baseline > $O/fin_bench_dccrn_large.log 2>&1; tail -1 $O/fin_bench_dccrn_large.log | cut -c60-180
timeout 600 python bench.py --model fullsubnet --no-cpu-baseline > $O/fin_bench_fullsubnet.log 2>&1; tail -1 $O/fin_bench_fullsubnet.log | cut -c60-180
timeout 600 python bench.py --perceptual PMSQE --no-cpu-baseline > $O/fin_bench_pmsqe.log 2>&1; tail -1 $O/fin_bench_pmsqe.log | cut -c60-180
timeout 600 python bench.py --perceptual LMS --no-cpu-baseline > $O/fin_bench_lms.log 2>&1; tail -1 $O/fin_bench_lms.log | cut -c60-180
