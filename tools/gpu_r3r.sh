cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
timeout 1500 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -q -m gpu > $O/r_tests.log 2>&1; echo "rc=$?" >> $O/r_tests.log; tail -8 $O/r_tests.log | cut -c1-300
run() { echo "== $1"; env $1 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>&1 | tail -1 | cut -c60-175; }
run A=1
run A=2
timeout 600 python tools/optable.py > $O/r_optable.log 2>&1; grep -E "STFT|SPLITSUM|UNPACK|LSTM_BWD" $O/r_optable.log | head -20
