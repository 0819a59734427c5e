# cluster LSTM: per-op parity, model-level bf16 parity on the large golden, DCCRN-large bench
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -k "512 or 1024" > $O/r2_run10_ops.log 2>&1; echo "rc=$?" >> $O/r2_run10_ops.log
tail -15 $O/r2_run10_ops.log
timeout 600 python -m pytest tests/test_gpu_model.py -q -k "large" > $O/r2_run10_model.log 2>&1; echo "rc=$?" >> $O/r2_run10_model.log
tail -8 $O/r2_run10_model.log
timeout 400 python bench.py --model dccrn_large --steps 20 --warmup 5 --no-cpu-baseline > $O/r2_run10_bench_large.log 2>&1; echo "rc=$?" >> $O/r2_run10_bench_large.log
tail -2 $O/r2_run10_bench_large.log | cut -c1-1500
