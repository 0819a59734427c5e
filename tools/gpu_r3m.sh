cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
timeout 1200 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -x -q -m gpu > $O/m_tests.log 2>&1; echo "rc=$?" >> $O/m_tests.log; tail -4 $O/m_tests.log
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/m_bench.log 2>&1; tail -1 $O/m_bench.log | cut -c1-260
SEFD_LSTM_RPW=16 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > $O/m_bench_rpw16.log 2>&1; tail -1 $O/m_bench_rpw16.log | cut -c1-200
SEFD_PHASE_MERGE_MAXN=0 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > $O/m_bench_nomerge.log 2>&1; tail -1 $O/m_bench_nomerge.log | cut -c1-200
SEFD_PACK_EARLY=0 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > $O/m_bench_nopack.log 2>&1; tail -1 $O/m_bench_nopack.log | cut -c1-200
cd /tmp && export TMPDIR=/tmp
B="python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-roofline"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_m -o m -- $B > $O/m_prof.log 2>&1
cd $GRAFT_REPO_ROOT; python tools/timeline.py $O/prof_m/m_kernel_trace.csv 1 v > $O/m_timeline.txt 2>&1; head -32 $O/m_timeline.txt
