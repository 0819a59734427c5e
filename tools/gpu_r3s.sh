cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
timeout 900 python -m pytest tests/test_gpu_model.py -x -q -m gpu -k "DCCRN or dccrn" > $O/s_tests.log 2>&1; echo "rc=$?" >> $O/s_tests.log; tail -3 $O/s_tests.log | cut -c1-200
run() { echo "== $1"; env $1 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>&1 | tail -1 | cut -c60-175; }
run A=1
run SEFD_SPLITSUM_MID=0
run SEFD_UNPACK_MID=0
run "SEFD_UNPACK_MID=0 SEFD_SPLITSUM_MID=0"
run A=2
echo "== B64"; timeout 600 python bench.py --steps 20 --warmup 5 --batch 64 --no-cpu-baseline --no-roofline 2>&1 | tail -1 | cut -c60-175
cd /tmp && export TMPDIR=/tmp
B="python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-roofline"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_s -o s -- $B > $O/s_prof.log 2>&1
cd $GRAFT_REPO_ROOT; python tools/timeline.py $O/prof_s/s_kernel_trace.csv 1 v > $O/s_timeline.txt 2>&1; head -3 $O/s_timeline.txt
