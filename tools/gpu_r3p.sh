cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/p_bench.log 2>&1; tail -1 $O/p_bench.log | cut -c1-200
timeout 600 python tools/optable.py > $O/p_optable.log 2>&1; tail -5 $O/p_optable.log
cd /tmp && export TMPDIR=/tmp
B="python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-roofline"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_p -o p -- $B > $O/p_prof.log 2>&1
cd $GRAFT_REPO_ROOT; python tools/timeline.py $O/prof_p/p_kernel_trace.csv 1 v > $O/p_timeline.txt 2>&1; head -3 $O/p_timeline.txt
