cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/prof_tl -o tl -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-roofline > $O/r2_run24_prof.log 2>&1
f=$(find $O/prof_tl -name "*kernel_trace.csv" | head -1)
head -1 $f | cut -c1-400
python $GRAFT_REPO_ROOT/tools/timeline.py $f
rm -f $f
