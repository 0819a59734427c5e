cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 1200 python -m pytest tests/test_gpu_ops.py -q -x -k "FullSubNet or 512 or 1024" > $O/r2_run12_ops.log 2>&1; echo "rc=$?" >> $O/r2_run12_ops.log
tail -15 $O/r2_run12_ops.log
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_validate.py -q -k "fullsubnet or fsn or large" > $O/r2_run12_model.log 2>&1; echo "rc=$?" >> $O/r2_run12_model.log
tail -8 $O/r2_run12_model.log
timeout 400 python bench.py --model fullsubnet --steps 10 --warmup 3 --no-cpu-baseline > $O/r2_run12_bench_fsn.log 2>&1; echo "rc=$?" >> $O/r2_run12_bench_fsn.log
tail -2 $O/r2_run12_bench_fsn.log | cut -c1-1800
timeout 400 python bench.py --model dccrn_large --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > $O/r2_run12_bench_large.log 2>&1
tail -1 $O/r2_run12_bench_large.log | cut -c1-300
