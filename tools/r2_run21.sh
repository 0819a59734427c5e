cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_fsn3 -o fsn -- python $GRAFT_REPO_ROOT/bench.py --model fullsubnet --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > $O/r2_run21_prof.log 2>&1
find $O/prof_fsn3 -name "*kernel_trace*" -delete
head -14 $O/prof_fsn3/fsn_kernel_stats.csv | cut -c1-170
