cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out
( cd /tmp && export TMPDIR=/tmp
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/x_prof_large -o k -- python $GRAFT_REPO_ROOT/bench.py --model dccrn_large --steps 4 --warmup 2 --no-cpu-baseline --no-roofline --no-extra > $O/x_prof_large.log 2>&1 )
python tools/timeline.py $O/x_prof_large/k_kernel_trace.csv 1 v > $O/x_timeline_large.txt 2>&1
find $O -name "k_kernel_trace.csv" -delete 2>/dev/null
head -30 $O/x_timeline_large.txt
