cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
timeout 600 python tools/probes/fsn_loss_debug.py > $O/g2_dbg.log 2>&1; tail -40 $O/g2_dbg.log
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-extra > $O/g2_bench_driver.log 2>&1; tail -1 $O/g2_bench_driver.log | cut -c1-200
( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/g2_prof -o k -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-roofline --no-extra > $O/g2_prof.log 2>&1 )
head -40 $O/g2_prof/k_kernel_stats.csv | cut -c1-200
python tools/timeline.py $O/g2_prof/k_kernel_trace.csv 1 v > $O/g2_timeline.txt 2>&1
rm -f $O/g2_prof/k_kernel_trace.csv
timeout 2400 python -m pytest tests -q -m gpu > $O/g2_all.log 2>&1; tail -8 $O/g2_all.log
