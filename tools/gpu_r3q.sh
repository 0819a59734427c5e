cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -x -q -m gpu -k "DCCRN or dccrn" > $O/q_tests.log 2>&1; echo "rc=$?" >> $O/q_tests.log; tail -3 $O/q_tests.log
run() { echo "== $1"; env $1 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>&1 | tail -1 | cut -c60-175; }
run A=1
run SEFD_LSTM_RPW_BWD=16
run SEFD_SPLITSUM_MULTI=0
run SEFD_SPECPAD_FUSE=0
run SEFD_LSTM_CHUNKS=6
run SEFD_LSTM_CHUNKS=8
run A=2
cd /tmp && export TMPDIR=/tmp
B="python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-roofline"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_q -o q -- $B > $O/q_prof.log 2>&1
cd $GRAFT_REPO_ROOT; python tools/timeline.py $O/prof_q/q_kernel_trace.csv 1 v > $O/q_timeline.txt 2>&1; head -3 $O/q_timeline.txt
