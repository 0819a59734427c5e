cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_fsn -o fsn -- python $GRAFT_REPO_ROOT/bench.py --model fullsubnet --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > $O/r2_run11_prof.log 2>&1
find $O/prof_fsn -name "*kernel_trace*" -delete
head -25 $O/prof_fsn/fsn_kernel_stats.csv | cut -c1-200
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_large -o large -- python $GRAFT_REPO_ROOT/bench.py --model dccrn_large --steps 4 --warmup 2 --no-cpu-baseline --no-roofline > $O/r2_run11_prof2.log 2>&1
find $O/prof_large -name "*kernel_trace*" -delete
head -25 $O/prof_large/large_kernel_stats.csv | cut -c1-200
