cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
runf() { echo "== $1"; env $1 timeout 600 python bench.py --model fullsubnet --steps 6 --warmup 2 --no-cpu-baseline --no-roofline 2>&1 | tail -1 | cut -c60-175; }
runf A=1
runf SEFD_FSN_HOLD=0
runf SEFD_FSN_LANES=0
runf A=2
echo "== dccrn"; timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>&1 | tail -1 | cut -c60-175
echo "== large"; timeout 600 python bench.py --model dccrn_large --steps 10 --warmup 3 --no-cpu-baseline --no-roofline 2>&1 | tail -1 | cut -c60-175
timeout 900 python -m pytest tests/test_gpu_model.py -x -q -m gpu > $O/v_tests.log 2>&1; echo "rc=$?" >> $O/v_tests.log; tail -3 $O/v_tests.log | cut -c1-200
cd /tmp && export TMPDIR=/tmp
B="python $GRAFT_REPO_ROOT/bench.py --model fullsubnet --steps 3 --warmup 2 --no-cpu-baseline --no-roofline"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_v -o v -- $B > $O/v_prof.log 2>&1
cd $GRAFT_REPO_ROOT; TIMELINE_MARK=fsn_in_kernel:1 python tools/timeline.py $O/prof_v/v_kernel_trace.csv 1 v > $O/v_timeline.txt 2>&1; head -3 $O/v_timeline.txt
