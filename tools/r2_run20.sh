cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -k "FullSubNet and 11" > $O/r2_run20_ops.log 2>&1; echo "rc=$?" >> $O/r2_run20_ops.log
tail -8 $O/r2_run20_ops.log
timeout 300 python -m pytest tests/test_gpu_model.py -q -k "fullsubnet or fsn" > $O/r2_run20_model.log 2>&1; tail -3 $O/r2_run20_model.log
timeout 400 python bench.py --model fullsubnet --steps 10 --warmup 3 --no-cpu-baseline > $O/r2_run20_bench_fsn.log 2>&1; echo "rc=$?" >> $O/r2_run20_bench_fsn.log
tail -2 $O/r2_run20_bench_fsn.log | cut -c1-1500
