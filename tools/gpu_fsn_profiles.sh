cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
pmc() {
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
pmc fullsubnet --model fullsubnet
cp $O/fin_pmc_fullsubnet.json profiles/r05_pmc_traffic_fullsubnet.json 2>/dev/null
TIMELINE_MARK=fsn_in_kernel:1 python tools/timeline.py $O/fin_prof_fullsubnet/k_kernel_trace.csv 1 v > $O/fin_timeline_fullsubnet.txt 2>&1
find $O -name "k_kernel_trace.csv" -size +20M -delete 2>/dev/null
timeout 600 python bench.py --model fullsubnet --no-cpu-baseline > $O/fin_bench_fullsubnet.log 2>&1; tail -1 $O/fin_bench_fullsubnet.log | cut -c60-180
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-extra 2>&1 | tail -1 | cut -c60-180
