cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
runf() { echo "== $1"; env $1 timeout 600 python bench.py --model fullsubnet --steps 6 --warmup 2 --no-cpu-baseline --no-roofline 2>&1 | tail -1 | cut -c60-175; }
runf A=1
runf SEFD_FSN_WG_ROUNDS=1
runf SEFD_FSN_WG_ROUNDS=2
runf SEFD_FSN_WG_ROUNDS=6
runf "SEFD_FSN_WG_ROUNDS=3 SEFD_FSN_HOLD=0"
runf SEFD_ROWS_FWD=54
runf A=2
cd /tmp && export TMPDIR=/tmp
B="python $GRAFT_REPO_ROOT/bench.py --model fullsubnet --steps 3 --warmup 2 --no-cpu-baseline --no-roofline"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_w -o w -- $B > $O/w_prof.log 2>&1
cd $GRAFT_REPO_ROOT; TIMELINE_MARK=fsn_in_kernel:1 python tools/timeline.py $O/prof_w/w_kernel_trace.csv 1 v > $O/w_timeline.txt 2>&1; head -3 $O/w_timeline.txt
