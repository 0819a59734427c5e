cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out
( cd /tmp && export TMPDIR=/tmp
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/x_prof_fullsubnet -o k -- python $GRAFT_REPO_ROOT/bench.py --model fullsubnet --steps 4 --warmup 2 --no-cpu-baseline --no-roofline --no-extra > $O/x_prof_fullsubnet.log 2>&1 )
TIMELINE_MARK=fsn_in_kernel:1 python tools/timeline.py $O/x_prof_fullsubnet/k_kernel_trace.csv 1 v > $O/x_timeline_fullsubnet.txt 2>&1
find $O -name "k_kernel_trace.csv" -delete 2>/dev/null
grep -E "sbbuild|splitsum|sbbwd|sbsum" $O/x_timeline_fullsubnet.txt
